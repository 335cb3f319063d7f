"""Edge validation, connect (checkMotion + voxelizeEdge, stored points) and state validation for the config-3 robot with and
without retraction enabled.  TENDON_HIP_FUSED (read when a context is created): 2 (default) = verdict-only kernels (retraction:
fk_verdict_retract, samples ordered by backbone length, bisection on tip-aligned cell signatures), 1 = stored points (retraction:
K1r -> K2, point-reading interval test).  `numbered by length`: the roadmap's vertices numbered by retraction
(RoadmapBuilder.sample_valid_vertices does that), so that every wave of every edge launch holds backbones of one length."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
for ret, fused, smax in ((False, "2", 0.0), (True, "1", 0.2), (True, "2", 0.2)):
    os.environ["TENDON_HIP_FUSED"] = fused
    robot = W.robot_config3()
    robot.enable_retraction = ret
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
    states0, _ = rb.sample_valid_vertices(50000, batch=1 << 16)
    for numbered in ((False,) if not ret else (False, True)):
        states = states0
        if ret:
            states = states0[np.random.default_rng(1).permutation(len(states0))]          # arrival order
            if numbered:
                states = states[np.argsort(states[:, -1], kind="stable")]
        edges = rb.knn_edges_gpu(states, 11)
        chk.engine.reserve_edges(len(edges))
        rb.validate_edges(states, edges)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); v, nf = rb.validate_edges(states, edges); best = min(best, time.perf_counter() - t0)
        rb.connect(states, edges, device=True)
        bc = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); kept, ec = rb.connect(states, edges, device=True); bc = min(bc, time.perf_counter() - t0)
        print("retraction %s TENDON_HIP_FUSED=%s numbered by length %s: %d vertices, %d edges validated in %.1f ms = %.3g edges/s (%.3g FK samples/s, "
              "valid %.3f, FK samples %d); connect %.1f ms (%d kept)"
              % (ret, fused, numbered, len(states), len(edges), 1e3 * best, len(edges) / best, nf.sum() / best, v.mean(), int(nf.sum()), 1e3 * bc, len(kept)), flush=True)
