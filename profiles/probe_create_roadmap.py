"""createRoadmap as one call (RoadmapBuilder.create_roadmap) for config 3's robot, plain and with rotation + retraction enabled
(the planner's full state space): wall time per phase."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
V = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for full in (False, True):
    robot = W.robot_config3()
    robot.enable_rotation = full
    robot.enable_retraction = full
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    best = None
    for rep in range(5):                                    # the first two also warm the process up (allocations, clocks)
        rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
        t0 = time.perf_counter()
        prm, rm = rb.create_roadmap(V, k=10)
        dt = time.perf_counter() - t0
        print("  rep", rep, "%.1f ms" % (1e3 * dt), {k: round(1e3 * v["seconds"], 1) for k, v in rb.timing.items() if "seconds" in v}, flush=True)
        if best is None or dt < best[0]:
            best = (dt, {k: round(1e3 * v["seconds"], 1) for k, v in rb.timing.items() if "seconds" in v}, len(rm["edges"]))
    knn = []
    for _ in range(3):
        t0 = time.perf_counter(); rb.knn_edges_gpu(rm["states"], 11); knn.append(round(1e3 * (time.perf_counter() - t0), 1))
    print("  knn_edges_gpu alone, three more calls (ms):", knn)
    print("rotation + retraction" if full else "tensions only", "vertices", V, "edges kept", best[2], "create_roadmap %.1f ms" % (1e3 * best[0]), best[1], flush=True)
