#!/usr/bin/env python3
"""Config 5's searches, the kernel alone and the shared schedule, validity known: a quick A/B loop for search_kernel.hpp
(PROBE_VERTICES, PROBE_QUERIES; TENDON_HIP_LIB selects an A/B build, TENDON_HIP_SEARCH_STATS=1 prints where the time went)."""
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
    prm.prepare(int(os.environ.get("PROBE_LANDMARKS", "16")))
    out = {"vertices": nv, "edges": int(len(e_ok)), "queries": nq}
    for mode in os.environ.get("PROBE_MODES", "device,auto,device,auto,auto").split(","):
        if mode == "auto":
            os.environ.pop("TENDON_HIP_SEARCH", None)
        elif mode == "host":
            os.environ["TENDON_HIP_SEARCH"] = "host"
        else:
            os.environ["TENDON_HIP_SEARCH"] = mode
        for form in ("eager", "lazy"):
            prm.clearValidity()
            if form == "eager":
                prm.revalidate()
            t0 = time.perf_counter()
            prm.solveWithRoadmap(pairs[:, 0], pairs[:, 1])
            dt = time.perf_counter() - t0
            out.setdefault("%s_%s_ms" % (form, mode), []).append(round(dt * 1e3, 2))
            out["%s_%s_where" % (form, mode)] = dict(prm.search_stats, rounds=prm.stats["rounds"], expanded=prm.stats["expanded"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
