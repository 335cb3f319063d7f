"""How many edge samples of config 3 (10^5 vertices, 10-NN edges) take the fallback pass (exact pairwise self-collision sweep)?
One lane, TENDON_HIP_EDGE_TIMING=1: level sizes and fallback counts on stderr.  Also the same roadmap with tau up to 30 N (tighter curls)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TENDON_HIP_EDGE_TIMING"] = "1"
os.environ["TENDON_HIP_EDGE_LANES"] = "1"
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
for V in (100000,):
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
    states, _ = rb.sample_valid_vertices(V, batch=1 << 17)
    edges = rb.knn_edges_gpu(states, 11)
    chk.engine.reserve_edges(len(edges))
    t0 = time.perf_counter(); v, nf = rb.validate_edges(states, edges); dt = time.perf_counter() - t0
    print("%d vertices, %d edges, %d valid, %d FK samples, %.2f ms" % (V, len(edges), int(v.sum()), int(nf.sum()), 1e3 * dt), flush=True)
    h = np.bincount(nf)
    print("FK samples per edge histogram:", {int(i): int(c) for i, c in enumerate(h) if c}, flush=True)
