"""tr_validate_edges_indexed with one to four lanes (TENDON_HIP_EDGE_LANES, read when a context is created): config 3's robot,
roadmaps of 25 k / 50 k / 100 k (tension-only: also 600 k) vertices, 10-NN edges, best of five calls each."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
QUICK = "quick" in sys.argv[1:]
for rot in ((False,) if QUICK else (False, True, "rotation + retraction")):
    for V in ((100000, 600000) if QUICK else (25000, 50000, 100000) + ((600000,) if rot is False else ())):
        res = {}
        for lanes in ("1", "2", "3", "4", "auto"):
            os.environ.pop("TENDON_HIP_EDGE_LANES", None)
            if lanes != "auto":
                os.environ["TENDON_HIP_EDGE_LANES"] = lanes
            robot = W.robot_config3()
            robot.enable_rotation = bool(rot)
            robot.enable_retraction = rot == "rotation + retraction"
            vox, _ = W.reach_environment(seed=7, n_spheres=64)
            chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
            rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
            states, _ = rb.sample_valid_vertices(V, batch=1 << 17)
            edges = rb.knn_edges_gpu(states, 11)
            chk.engine.reserve_edges(len(edges))
            rb.validate_edges(states, edges)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); v, nf = rb.validate_edges(states, edges); best = min(best, time.perf_counter() - t0)
            res[lanes] = (best, int(np.flatnonzero(v).sum()), int(nf.sum()), len(edges))
        assert all(res[l][1:] == res["1"][1:] for l in res), res
        print("rotation %s, %d vertices, %d edges: lanes 1 / 2 / 3 / 4 / by the edge count: %s ms; best %.3g edges/s, %.3g FK samples/s"
              % (rot, V, res["1"][3], " / ".join("%.2f" % (1e3 * res[l][0]) for l in ("1", "2", "3", "4", "auto")),
                 res["1"][3] / min(r[0] for r in res.values()), res["1"][2] / min(r[0] for r in res.values())), flush=True)
