"""Do two fk_verdict launches on two streams fill each other's partial rounds?  Two contexts (own workspaces), n configurations each
(a non-integer number of resident rounds), launched back to back on ONE stream against on TWO streams."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chks = [irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox) for _ in range(2)]
round_ = 1 << 17
for frac in (0.3, 0.6, 1.0, 1.3, 2.5, 3.26):
    n = int(frac * round_) // 64 * 64
    sts = [torch.from_numpy(W.random_states(robot, n, seed=3 + i, tau_max=20.0)).cuda() for i in range(2)]
    bits = [torch.zeros(n // 64 + 1, dtype=torch.int64, device="cuda") for _ in range(2)]
    for c in chks:
        c.engine.reserve(n)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def run(two):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(6):
            for i in range(2):
                s = streams[i if two else 0]
                chks[i].engine.validate_batch_dev(sts[i], n, bits[i], stream=s)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 6
    run(False); run(True)
    a, b = min(run(False) for _ in range(3)), min(run(True) for _ in range(3))
    print("n = %.2f rounds each: one stream %.3f ms, two streams %.3f ms (%.1f %%)" % (frac, 1e3 * a, 1e3 * b, 100 * (a - b) / a), flush=True)
