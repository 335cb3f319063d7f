#!/usr/bin/env python3
"""Config 5's graph searches on the host threads against the device kernel (search_kernel.hpp): 10 000 queries on the 100 k-vertex
roadmap, eager form (validity known: the searches alone) and lazy form (the whole loop), same call, paths compared."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    irt = importlib.import_module("interactive-rate-tendons_amd")
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    new_vox, _ = W.reach_environment(seed=7, n_spheres=72)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
    nv = int(os.environ.get("PROBE_VERTICES", "100000"))
    states, _ = rb.sample_valid_vertices(nv, batch=1 << 17)
    edges = rb.knn_edges_gpu(states, 11)
    valid, _ = rb.validate_edges(states, edges)
    e_ok = edges[valid]
    vc, ec = rb.vertex_caches(states), rb.edge_caches(states, e_ok)
    prm = irt.VoxelCachedLazyPRM(chk, states, e_ok)
    prm.set_caches(vc, ec)
    prm.set_obstacles(new_vox)
    nq = int(os.environ.get("PROBE_QUERIES", "10000"))
    pairs = np.random.default_rng(17).integers(0, len(states), size=(nq, 2))
    out = {}
    for nl in (16, 0):
        prm.prepare(nl)
        res = {}
        for form in ("eager", "lazy"):
            for mode in ("host", "device", "auto", "host", "device", "auto"):
                if mode == "auto":
                    os.environ.pop("TENDON_HIP_SEARCH", None)      # the default: large rounds shared between the kernel and the host threads
                else:
                    os.environ["TENDON_HIP_SEARCH"] = mode
                prm.clearValidity()
                if form == "eager":
                    prm.revalidate()
                t0 = time.perf_counter()
                r = prm.solveWithRoadmap(pairs[:, 0], pairs[:, 1])
                dt = time.perf_counter() - t0
                key = "%s_%s" % (form, mode)
                res.setdefault(key, []).append(dt)
                res[key + "_expanded"] = prm.stats["expanded"]
                res[key + "_rounds"] = prm.stats["rounds"]
                res[key + "_where"] = dict(prm.search_stats)
                if mode == "host":
                    ref = r
                else:
                    res[form + "_" + mode + "_same"] = bool(np.array_equal(ref["status"], r["status"]) and np.array_equal(ref["cost"], r["cost"])
                                                            and np.array_equal(ref["path_vertices"], r["path_vertices"]))
        for form in ("eager", "lazy"):
            h, d, a = min(res[form + "_host"]), min(res[form + "_device"]), min(res[form + "_auto"])
            res[form + "_queries_per_s"] = {"host": nq / h, "device": nq / d, "auto": nq / a, "device_ratio": h / d, "auto_ratio": h / a}
        out[str(nl)] = res
        print(nl, json.dumps(res), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
