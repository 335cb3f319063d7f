"""Where connect's time goes (HIP-event totals per kernel slot), config 3's robot plain and with rotation + retraction, 100 k vertices."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
for full in (False, True):
    robot = W.robot_config3()
    robot.enable_rotation = full
    robot.enable_retraction = full
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
    states, _ = rb.sample_valid_vertices(100000, batch=1 << 17)
    edges = rb.knn_edges_gpu(states, 11)
    chk.engine.reserve_edges(len(edges))
    for _ in range(2):
        rb.connect(states, edges, device=True)
    chk.engine.profile_begin()
    t0 = time.perf_counter(); kept, ec = rb.connect(states, edges, device=True); dt = time.perf_counter() - t0
    pr = chk.engine.profile_read(); chk.engine.profile_end()
    v, nf = rb.validate_edges(states, edges)
    print("rotation + retraction" if full else "tensions only", "connect %.1f ms" % (1e3 * dt), "candidate edges", len(edges), "kept", len(kept),
          "FK samples (reference count)", int(nf.sum()), {k: (v["launches"], round(v["total_ms"], 2)) for k, v in pr.items() if v["launches"]}, flush=True)
