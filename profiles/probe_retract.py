"""K1r (retraction kernel, per-lane arc-length grid) against K1 + K2 on the same robot: device-resident rates."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
vox, _ = W.reach_environment(seed=7, n_spheres=64)
# TENDON_HIP_FUSED is read when a context is created: 2 (default) = verdict-only kernels (fk_verdict / fk_verdict_retract),
# 1 = stored points (retraction: K1r -> K2)
# TENDON_HIP_RETRACT_SORT=0: the verdict-only kernel takes the batch in arrival order instead of ordered by backbone length
for ret, fused, order, smax in ((False, "2", "1", 0.0), (True, "1", "1", 0.1), (True, "2", "0", 0.1), (True, "2", "8192", 0.1),
                               (True, "1", "1", 0.2), (True, "2", "0", 0.2), (True, "2", "8192", 0.2), (True, "2", "8192/wave start off", 0.2)):
    for mk in (W.robot_config2, W.robot_config3):
        robot = mk()
        robot.enable_retraction = ret
        os.environ["TENDON_HIP_FUSED"] = fused
        os.environ["TENDON_HIP_RETRACT_SORT"] = order.split("/")[0]
        os.environ.pop("TENDON_HIP_RETRACT_KBEGIN_OFF", None)
        if "off" in order:
            os.environ["TENDON_HIP_RETRACT_KBEGIN_OFF"] = "1"
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        n = 1 << 19
        st = W.random_states(robot, n, seed=1, tau_max=10.0)
        if ret:
            st[:, -1] = np.random.default_rng(2).uniform(0, smax, n)
        d = torch.from_numpy(st).cuda()
        bits = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        chk.engine.reserve(n)
        for _ in range(2):
            chk.engine.validate_batch_dev(d, n, bits)
        torch.cuda.synchronize()
        chk.engine.profile_begin()
        t0 = time.perf_counter()
        for _ in range(5):
            chk.engine.validate_batch_dev(d, n, bits)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        pr = chk.engine.profile_read(); chk.engine.profile_end()
        print("retract", ret, "s_start ~ U[0, %.1f)" % smax, "TENDON_HIP_FUSED", fused, "sorted", order, "tendons", len(robot.tendons), "ms per 2^19", round(dt * 1e3, 2), "checks/s %.3g" % (n / dt),
              {k: round(v["total_ms"] / max(1, v["launches"]), 2) for k, v in pr.items() if v["launches"]})
