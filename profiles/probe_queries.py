#!/usr/bin/env python3
"""Config 5's query loop per landmark count: expansions and wall time of 10 000 queries on the 100 k-vertex roadmap
(eager form: validity known, so the time is the host searches alone)."""
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
    states, _ = rb.sample_valid_vertices(100000, batch=1 << 17)
    edges = rb.knn_edges_gpu(states, 11)
    valid, _ = rb.validate_edges(states, edges)
    e_ok = edges[valid]
    vc, ec = rb.vertex_caches(states), rb.edge_caches(states, e_ok)
    prm = irt.VoxelCachedLazyPRM(chk, states, e_ok)
    prm.set_caches(vc, ec)
    prm.set_obstacles(new_vox)
    pairs = np.random.default_rng(17).integers(0, len(states), size=(10000, 2))
    out = {}
    ref = None
    for nl in (0, 4, 8, 16, 24, 32, 48, 64):
        t0 = time.perf_counter()
        prm.prepare(nl)
        tp = time.perf_counter() - t0
        prm.clearValidity(); prm.revalidate()
        prm.solveWithRoadmap(pairs[:64, 0], pairs[:64, 1])
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            r = prm.solveWithRoadmap(pairs[:, 0], pairs[:, 1])
            best = min(best, time.perf_counter() - t0)
        if ref is None:
            ref = r
        same = bool(np.array_equal(ref["path_vertices"], r["path_vertices"]) and np.array_equal(ref["cost"], r["cost"]))
        for nt in (1, 4):
            t0 = time.perf_counter()
            prm.solveWithRoadmap(pairs[:2000, 0], pairs[:2000, 1], n_threads=nt)
            out.setdefault(str(nl), {})["queries_per_s_%d_threads" % nt] = 2000 / (time.perf_counter() - t0)
        out[str(nl)].update(prepare_s=tp, seconds=best, queries_per_s=10000 / best, expanded=prm.stats["expanded"], same_paths=same)
        print(nl, out[str(nl)], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
