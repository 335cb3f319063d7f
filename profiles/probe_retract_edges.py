"""Edge validation and state validation for the config-3 robot with and without retraction enabled.  TENDON_HIP_FUSED (read
when a context is created): 2 (default) = verdict-only kernels (retraction: fk_verdict_retract, samples ordered by backbone
length, bisection on tip-aligned cell signatures), 1 = stored points (retraction: K1r -> K2, point-reading interval test)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
for ret, fused, smax in ((False, "2", 0.0), (True, "1", 0.05), (True, "2", 0.05), (True, "1", 0.2), (True, "2", 0.2)):
    os.environ["TENDON_HIP_FUSED"] = fused
    robot = W.robot_config3()
    robot.enable_retraction = ret
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
    states, _ = rb.sample_valid_vertices(50000, batch=1 << 16)
    if ret:
        states[:, -1] = np.random.default_rng(1).uniform(0, smax, len(states))
        states = states[chk.is_valid(states)]
    edges = rb.knn_edges_gpu(states, 11)
    chk.engine.reserve_edges(len(edges))
    rb.validate_edges(states, edges)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); v, nf = rb.validate_edges(states, edges); best = min(best, time.perf_counter() - t0)
    n = 1 << 18
    st = W.random_states(robot, n, seed=3, tau_max=20.0)
    if ret:
        st[:, -1] = np.random.default_rng(2).uniform(0, smax, n)
    chk.is_valid(st)
    t0 = time.perf_counter(); chk.is_valid(st); tv = time.perf_counter() - t0
    print("retraction %s s_start ~ U[0, %.2f) TENDON_HIP_FUSED=%s: %d vertices, %d edges in %.1f ms = %.3g edges/s (%.3g FK samples/s, valid %.3f, "
          "verdict checksum %d, FK samples %d);  is_valid 2^18: %.3g checks/s"
          % (ret, smax, fused, len(states), len(edges), 1e3 * best, len(edges) / best, nf.sum() / best, v.mean(), int(np.flatnonzero(v).sum()), int(nf.sum()), n / tv), flush=True)
