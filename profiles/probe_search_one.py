#!/usr/bin/env python3
"""One configuration of probe_search.py (16 landmarks, default search mode), eager and lazy, with the library's own round timing
(TENDON_HIP_SEARCH_STATS=1) -- for tuning TENDON_HIP_SEARCH_HOST_SHARE / TENDON_HIP_SEARCH_BUDGET."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
new_vox, _ = W.reach_environment(seed=7, n_spheres=int(os.environ.get("PROBE_SPHERES", "72")))   # (64 when the roadmap was built)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
states, _ = rb.sample_valid_vertices(100000, batch=1 << 17)
edges = rb.knn_edges_gpu(states, 11)
valid, _ = rb.validate_edges(states, edges)
e_ok = edges[valid]
prm = irt.VoxelCachedLazyPRM(chk, states, e_ok)
prm.set_caches(rb.vertex_caches(states), rb.edge_caches(states, e_ok))
prm.set_obstacles(new_vox)
pairs = np.random.default_rng(17).integers(0, len(states), size=(int(os.environ.get("PROBE_QUERIES", "10000")), 2))
prm.prepare(16)
settings = [s for s in os.environ.get("PROBE_SETTINGS", "3:6000").split(",")]
for st in settings:
    share, budget = st.split(":")
    os.environ["TENDON_HIP_SEARCH_HOST_SHARE"] = share
    os.environ["TENDON_HIP_SEARCH_BUDGET"] = budget
    for form in ("eager", "lazy"):
        best = 1e9
        for rep in range(3):
            prm.clearValidity()
            if form == "eager":
                prm.revalidate()
            sys.stderr.flush()
            t0 = time.perf_counter()
            out = prm.solveWithRoadmap(pairs[:, 0], pairs[:, 1])
            best = min(best, time.perf_counter() - t0)
            prm.last_status = out["status"]
        print("share %s%% budget %s %s: %.2f ms, %.0f queries/s   (searches: %s; statuses %s)" % (share, budget, form, best * 1e3, len(pairs) / best, prm.search_stats,
              np.bincount(prm.last_status, minlength=4).tolist() if hasattr(prm, "last_status") else ""), flush=True)
