#!/usr/bin/env python3
"""Soak run (opt-in): contexts created and destroyed in a loop with every family of call in between; device memory must
return to its level, results must not drift.

    python tests/soak.py [iterations]
"""
import gc
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    irt = importlib.import_module("interactive-rate-tendons_amd")
    W = irt.workloads
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    ref = None
    free0 = None
    t0 = time.perf_counter()
    for it in range(iters):
        robot = W.robot_config3() if it % 2 else W.robot_config2()
        robot.enable_rotation = (it % 3 == 0)
        chk = (irt.VoxelValidityChecker if it % 5 == 0 else irt.VoxelBackboneValidityChecker)(robot, irt.VoxelEnvironment(), vox)
        mv = irt.VoxelBackboneMotionValidator(chk)
        rb = irt.RoadmapBuilder(chk, mv, seed=3)
        st = W.random_states(robot, 3000, seed=5, tau_max=12.0)
        v = chk.is_valid(st)
        e = rb.knn_edges_gpu(st, 6)[:4000]
        ev, _ = rb.validate_edges(st, e)
        lv, lt = mv.check_motion_last_valid(st[e[:300, 0]], st[e[:300, 1]])
        vc = rb.vertex_caches(st[:500], device=bool(it % 2))
        e_ok, ec = rb.connect(st, e[:1500], device=bool(it % 2))
        prm = irt.VoxelCachedLazyPRM(chk, st, e_ok)
        full_vc = rb.vertex_caches(st, device=bool(it % 2))
        prm.set_caches(full_vc, ec)
        out = prm.solveWithRoadmap(np.arange(0, 200), np.arange(200, 400))
        if it % 4 == 0:
            # a round large enough for the device searches (their tables come from and go back to the library's buffer cache; every other time
            # they are handed back explicitly): the same answers as the host threads' for the queries both have seen
            os.environ["TENDON_HIP_SEARCH"] = "device"          # (whatever the number of queries left to search)
            big = prm.solveWithRoadmap(np.arange(0, 800) % 3000, (np.arange(0, 800) * 7 + 200) % 3000)
            del os.environ["TENDON_HIP_SEARCH"]
            assert prm.search_state_bytes() > 0, dict(prm.search_stats)
            assert np.array_equal(big["status"][:1], out["status"][:1]) and (big["status"][0] != 0 or big["cost"][0] == out["cost"][0])
            if it % 8 == 0:
                assert prm.release_search_state() > 0
        sig = (int(v.sum()), int(ev.sum()), int(lv.sum()), float(lt.sum()), int(len(e_ok)), int((out["status"] == 0).sum()), float(np.nansum(out["cost"][out["status"] == 0])))
        key = (it % 2, it % 3 == 0, it % 5 == 0)
        if ref is None:
            ref = {}
        if key in ref:
            assert ref[key] == sig, (it, key, ref[key], sig)
        else:
            ref[key] = sig
        prm.close()
        del prm, rb, mv, chk, vc, ec, full_vc
        robot.close() if hasattr(robot, "close") else None
        gc.collect()
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        if it == 12:
            free0 = free
        if it % 20 == 0 or it == iters - 1:
            print("iteration %d: %.1f s, free device memory %.2f GiB%s" % (it, time.perf_counter() - t0, free / 2 ** 30,
                                                                          "" if free0 is None else " (%.1f MiB below iteration 12)" % ((free0 - free) / 2 ** 20)), flush=True)
    assert free0 is not None and free0 - free < 256 * 2 ** 20, "device memory keeps shrinking"
    print("soak ok:", iters, "iterations,", len(ref), "distinct configurations, results stable")


if __name__ == "__main__":
    main()
