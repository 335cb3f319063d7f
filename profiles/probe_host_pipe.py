#!/usr/bin/env python3
"""Where the host-buffer entry point (tr_validate_batch: pageable arrays in, bits + tips + flags out) spends its time at
the headline size, next to the device-resident call: per pipeline chunk size, with fresh and with reused output arrays."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    irt = importlib.import_module("interactive-rate-tendons_amd")
    W = irt.workloads
    robot = W.robot_config2()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    n = 1 << 20
    states = W.random_states(robot, n, seed=3, tau_max=10.0)
    out = {}
    for log2 in (None, 14, 15, 16, 17, 18):
        if log2 is None:
            os.environ.pop("TENDON_HIP_PIPE_LOG2", None)
        else:
            os.environ["TENDON_HIP_PIPE_LOG2"] = str(log2)
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        eng = chk.engine
        for _ in range(2):
            eng.validate_batch(states)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter()
            eng.validate_batch(states)
            ts.append(time.perf_counter() - t0)
        ts2 = []
        for _ in range(8):
            t0 = time.perf_counter()
            eng.validate_batch(states, want_tips=False, want_flags=False)
            ts2.append(time.perf_counter() - t0)
        d = torch.from_numpy(states).cuda()
        bits = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        tips = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
        eng.validate_batch_dev(d, n, bits, tips)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.validate_batch_dev(d, n, bits, tips)
        torch.cuda.synchronize()
        t_dev = (time.perf_counter() - t0) / 5
        out[str(log2)] = {"host_ms_min": 1e3 * min(ts), "host_ms_median": 1e3 * float(np.median(ts)),
                          "bits_only_ms_min": 1e3 * min(ts2), "device_resident_ms": 1e3 * t_dev}
        print(log2, out[str(log2)], flush=True)
        del chk, eng
    # raw host costs on this box
    a = np.empty_like(states)
    t0 = time.perf_counter(); a[:] = states; t1 = time.perf_counter(); a[:] = states; t2 = time.perf_counter()
    out["memcpy_32MiB_ms_first_touch_then_warm"] = [1e3 * (t1 - t0), 1e3 * (t2 - t1)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
