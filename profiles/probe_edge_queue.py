"""The indexed edge check as one persistent launch over a device work queue (TENDON_HIP_EDGE_QUEUE=1, the default) against the
level-synchronous lanes (=0): same roadmap, verdicts and FK counts compared edge by edge, best of five calls each.
usage: probe_edge_queue.py [vertices ...]   (default 20000 100000)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [20000, 100000]
rot = "rot" in sys.argv[1:]
for V in sizes:
    res = {}
    for mode in ("0", "1"):
        os.environ["TENDON_HIP_EDGE_QUEUE"] = mode
        robot = W.robot_config3()
        robot.enable_rotation = rot
        vox, _ = W.reach_environment(seed=7, n_spheres=64)
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
        states, _ = rb.sample_valid_vertices(V, batch=1 << 17)
        edges = rb.knn_edges_gpu(states, 11)
        chk.engine.reserve_edges(len(edges))
        t0 = time.perf_counter(); v, nf = rb.validate_edges(states, edges); first = time.perf_counter() - t0
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); v2, nf2 = rb.validate_edges(states, edges); best = min(best, time.perf_counter() - t0)
            assert np.array_equal(v, v2) and np.array_equal(nf, nf2), "results differ between two calls"
        res[mode] = (first, best, v, nf, len(edges))
        print("queue=%s: %d vertices, %d edges, %d valid, %d FK samples: first call %.2f ms, best %.2f ms = %.3g edges/s, %.3g samples/s"
              % (mode, V, len(edges), int(v.sum()), int(nf.sum()), 1e3 * first, 1e3 * best, len(edges) / best, (int(nf.sum()) - 2 * len(edges) + V) / best), flush=True)
    same_v = np.array_equal(res["0"][2], res["1"][2]); same_n = np.array_equal(res["0"][3], res["1"][3])
    print("   verdicts equal: %s, FK counts equal: %s" % (same_v, same_n), flush=True)
    if not (same_v and same_n):
        bad = np.flatnonzero((res["0"][2] != res["1"][2]) | (res["0"][3] != res["1"][3]))
        print("   %d edges differ, first: %s" % (len(bad), [(int(b), int(res["0"][2][b]), int(res["1"][2][b]), int(res["0"][3][b]), int(res["1"][3][b])) for b in bad[:10]]))
        sys.exit(1)
