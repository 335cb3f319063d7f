import importlib, sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3(); vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
chk.engine.reserve(1 << 20)
states, tips = rb.sample_valid_vertices(100000, batch=1 << 17)
edges = rb.knn_edges_gpu(states, 11)
a, b = states[edges[:, 0]], states[edges[:, 1]]
eng = chk.engine
eng.validate_edges(a[:1000], b[:1000])
for rep in range(3):
    eng.profile_begin()
    t0 = time.perf_counter(); out = eng.validate_edges(a, b); dt = time.perf_counter() - t0
    pr = eng.profile_read(); eng.profile_end()
    gpu = sum(v["total_ms"] for v in pr.values())
    print("edges", len(a), "wall ms", round(dt * 1e3, 1), "gpu kernels ms", round(gpu, 1), {k: (v["launches"], round(v["total_ms"], 1)) for k, v in pr.items()}, "samples", int(out["n_fk"].sum()))
