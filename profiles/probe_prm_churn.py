import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
mode = sys.argv[1]
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    t0 = time.perf_counter()
    states, tips = rb.sample_valid_vertices(100000)
    cand = rb.knn_edges_gpu(states, 11)
    rb.engine.reserve_edges(len(cand))
    edges, ec = rb.connect(states, cand, device=True)
    vc = rb.vertex_caches(states, device=True)
    if mode.startswith("prm"):
        prm = irt.VoxelCachedLazyPRM(chk, states, edges)
        if "nocaches" not in mode:
            prm.set_caches(vc, ec)
        if "noprepare" not in mode:
            prm.prepare(16)
    dt = time.perf_counter() - t0
    print(mode, "call %d: %.1f ms" % (i, 1e3 * dt), {k: round(1e3 * v["seconds"], 2) for k, v in rb.timing.items() if "seconds" in v and k in ("vertices", "connect", "vertex_caches")}, flush=True)
    if mode.startswith("prm"):
        del prm
    del ec, vc
