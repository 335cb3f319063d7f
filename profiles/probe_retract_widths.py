"""Retraction kernels per tendon count (state validation of 2^18 configurations, s_start ~ U[0, L)): which widths run faster with
two waves per SIMD.  A/B: TENDON_HIP_LIB=<alt build> (see _lib.py)."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
vox, _ = W.reach_environment(seed=7, n_spheres=64)
RET = os.environ.get("PROBE_RETRACTION", "1") != "0"      # PROBE_RETRACTION=0: the same robots without retraction (shared-grid kernels)
for nt in (3, 4, 5, 6, 7, 8):
    rng = np.random.default_rng(40 + nt)
    tendons = [irt.TendonSpecs(C=[2 * np.pi * k / nt, float(rng.uniform(-6, 6)), float(rng.uniform(-10, 10))],
                               D=[0.01, float(rng.uniform(-0.01, 0.01))], max_tension=12.0) for k in range(nt)]
    robot = irt.TendonRobot(tendons=tendons, specs=irt.BackboneSpecs(dL=0.2 / 128), enable_rotation=True, enable_retraction=RET)
    for fused in ("1", "2"):
        os.environ["TENDON_HIP_FUSED"] = fused
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        n = 1 << 18
        st = W.random_states(robot, n, seed=1, tau_max=12.0 / np.sqrt(nt))
        d = torch.from_numpy(st).cuda()
        bits = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        chk.engine.reserve(n)
        for _ in range(2):
            chk.engine.validate_batch_dev(d, n, bits)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            chk.engine.validate_batch_dev(d, n, bits)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        print("tendons", nt, "retraction", RET, "TENDON_HIP_FUSED", fused, "ms per 2^18 %.2f" % (dt * 1e3), "checks/s %.3g" % (n / dt), "valid %.3f" % float(irt.unpack_bits(bits.cpu().numpy().view(np.uint64), n).mean()), flush=True)
