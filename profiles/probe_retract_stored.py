"""fk_rk4_batch_retract (stored backbone points of a retraction robot) in arrival order against the order of the batch's backbone
lengths (the default from 8 192 configurations on): same planes, device-resident rates; TENDON_HIP_RETRACT_SORT is read per context."""
import importlib, json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
irt = importlib.import_module("interactive-rate-tendons_amd")
import bench
W = irt.workloads
vox, _ = W.reach_environment(seed=7, n_spheres=64)
out = {}
for mk, name in ((W.robot_config2, "3 tendons"), (W.robot_config3, "4 tendons")):
    for order in ("0", "8192"):
        robot = mk()
        robot.enable_retraction = True
        os.environ["TENDON_HIP_RETRACT_SORT"] = order
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        e = chk.engine
        n = 1 << 19
        st = W.random_states(robot, n, seed=1, tau_max=10.0)
        st[:, -1] = np.random.default_rng(2).uniform(0, 0.2, n)
        P, N = e.num_points, e.n_tendons
        d = torch.from_numpy(st).cuda()
        px, py, pz = (torch.empty(P * n, dtype=torch.float64, device="cuda") for _ in range(3))
        Li, hl = torch.empty(N * n, dtype=torch.float64, device="cuda"), torch.empty(N * n, dtype=torch.float64, device="cuda")
        conv, npts = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda")
        e.reserve(n)
        for _ in range(2):
            e.fk_batch_retraction_dev(d, n, n, px, py, pz, Li, conv, npts, hl)
        torch.cuda.synchronize()
        e.profile_begin()
        t0 = time.perf_counter()
        for _ in range(5):
            e.fk_batch_retraction_dev(d, n, n, px, py, pz, Li, conv, npts, hl)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        pr = e.profile_read()["fk_rk4_batch"]; e.profile_end()
        steps = float((npts.double() - 1).clamp(min=0).sum())                      # RK4 steps the backbones have
        fl = bench.algorithmic_flops_per_rk4_step(N) * steps
        kms = pr["total_ms"] / max(1, pr["launches"])
        key = "%s, %s" % (name, "arrival order" if order == "0" else "ordered by backbone length")
        out[key] = {"call_ms_incl_ordering": round(dt * 1e3, 3), "kernel_ms": round(kms, 3), "fk_per_s": n / dt,
                    "fp64_frac_of_78.6_TF_counting_the_steps_the_backbones_have": fl / (kms * 1e-3) / 78.6e12,
                    "checksum": float(px.nan_to_num().sum() + Li.sum() + npts.sum())}
        print(key, json.dumps(out[key]), flush=True)
print(json.dumps(out))
