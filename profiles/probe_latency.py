#!/usr/bin/env python3
"""Latency of the single-state / small-batch entry points through the C ABI (host buffers in, verdict out), next to the
oracle's CPU port on ONE core -- the numbers behind INTEGRATION.md's advice on isValid / checkMotion from serial planners
(the OMPL virtuals call them one state at a time: motion-planning/AbstractValidityChecker.cpp:124-133).

    python profiles/probe_latency.py > gpurun_out/<tag>/latency.json
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def med(f, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    irt = importlib.import_module("interactive-rate-tendons_amd")
    from oracle import oracle as orc
    W = irt.workloads
    out = {}
    for name, robot, tau in (("config2_3tendon", W.robot_config2(), 10.0), ("config3_4tendon", W.robot_config3(), 20.0)):
        vox, _ = W.reach_environment(seed=7, n_spheres=64)
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        mv = irt.VoxelBackboneMotionValidator(chk)
        eng = chk.engine
        st = W.random_states(robot, 1 << 14, seed=5, tau_max=tau)
        eng.validate_batch(st[:8], False, False)                       # creates streams / pinned buffers
        rows = []
        for k in range(0, 15):
            n = 1 << k
            reps = 200 if n <= 256 else 40
            t = med(lambda: eng.validate_batch(st[:n], False, False), reps)
            rows.append({"n": n, "ms": 1e3 * t, "checks_per_s": n / t})
        s = robot.specs
        orb = orc.Robot([t_.C for t_ in robot.tendons], [t_.D for t_ in robot.tendons], r=robot.r, L=s.L, dL=s.dL, ro=s.ro, ri=s.ri,
                        E=s.E, nu=s.nu, max_tension=[t_.max_tension for t_ in robot.tendons],
                        min_length=[t_.min_length for t_ in robot.tendons], max_length=[t_.max_length for t_ in robot.tendons], lib="omp")
        og = orc.Grid(vox.Nx(), vox.limits(), lib="omp")
        og.blocks()[...] = vox.blocks
        t0 = time.perf_counter()
        orc.validate_batch(orb, og, st[:2000], nthreads=1, lib=orc.omp_lib())
        cpu1 = (time.perf_counter() - t0) / 2000
        # edges: checkMotion one at a time and in small batches (step 1.0 in tension space: ~50 FK samples at most, ~8 typical)
        rng = np.random.default_rng(3)
        d = rng.normal(size=st.shape)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        b = np.clip(st + d, 0.0, None)
        mv.check_motion(st[:8], b[:8])
        erows = []
        for n in (1, 4, 16, 64, 256, 1024):
            t = med(lambda: mv.check_motion(st[:n], b[:n]), 30 if n <= 64 else 10)
            erows.append({"n": n, "ms": 1e3 * t, "edges_per_s": n / t})
        t0 = time.perf_counter()
        _, nfk, _ = orc.check_motion_batch(orb, og, st[:200], b[:200], nthreads=1, lib=orc.omp_lib())
        cpu_e = (time.perf_counter() - t0) / 200
        be = next((r["n"] for r in rows if r["ms"] * 1e-3 / r["n"] < cpu1), None)
        out[name] = {"validate_batch": rows, "cpu_one_core_ms_per_check": 1e3 * cpu1, "break_even_batch_vs_one_core": be,
                     "validate_edges": erows, "cpu_one_core_ms_per_edge": 1e3 * cpu_e, "fk_samples_per_edge_mean": float(nfk.mean())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
