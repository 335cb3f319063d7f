"""RoadmapBuilder.knn_edges_gpu / Engine.knn_edges / Engine.knn repeated at config 3's size: steady-state wall time per call."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
st = W.random_states(robot, n, seed=3)
for name, fn in (("rb.knn_edges_gpu", lambda: rb.knn_edges_gpu(st, 11)), ("eng.knn_edges", lambda: chk.engine.knn_edges(st, 11)),
                 ("eng.knn", lambda: chk.engine.knn(st, 11))):
    ts = []
    for _ in range(6):
        chk.engine.profile_begin()
        t0 = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t0))
        p = chk.engine.profile_read(); chk.engine.profile_end()
    print(name, ["%.1f" % t for t in ts], {k: round(v["total_ms"], 2) for k, v in p.items() if v["launches"]}, flush=True)
