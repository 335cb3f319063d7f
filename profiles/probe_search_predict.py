#!/usr/bin/env python3
"""What predicts a search's length?  Expansions per search of config 5's 10 000 queries (host searches, validity known) against the
end points' state-space distance, the landmark lower bound and the landmark upper bound (TENDON_HIP_SEARCH_HIST=<file>)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = "/tmp/search_hist.txt"
if os.path.exists(path):
    os.remove(path)
os.environ["TENDON_HIP_SEARCH"] = "host"
os.environ["TENDON_HIP_SEARCH_HIST"] = path
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
new_vox, _ = W.reach_environment(seed=7, n_spheres=72)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
states, _ = rb.sample_valid_vertices(int(os.environ.get("PROBE_VERTICES", "100000")), batch=1 << 17)
edges = rb.knn_edges_gpu(states, 11)
valid, _ = rb.validate_edges(states, edges)
e_ok = edges[valid]
prm = irt.VoxelCachedLazyPRM(chk, states, e_ok)
prm.set_caches(rb.vertex_caches(states), rb.edge_caches(states, e_ok))
prm.set_obstacles(new_vox)
pairs = np.random.default_rng(17).integers(0, len(states), size=(10000, 2))
prm.prepare(16)
prm.clearValidity(); prm.revalidate()
out = prm.solveWithRoadmap(pairs[:, 0], pairs[:, 1])
d = np.loadtxt(path)
d = d[d[:, 0] == d[:, 0].min()]
q = d[:, 1].astype(int)
ex, dist, lb, ub = d[:, 3], d[:, 4], d[:, 5], d[:, 6]
cost = out["cost"][q]
print("searches", len(ex), "expansions", int(ex.sum()), "above 4000 / 5000 / 6500 / 8000:", [(int((ex > t).sum()), int(ex[ex > t].sum())) for t in (4000, 5000, 6500, 8000)])
cands = {"distance": dist, "landmark lower bound": lb, "landmark upper bound": ub, "upper - lower": ub - lb, "lower - distance": lb - dist,
         "upper - distance": ub - dist, "(upper - lower) * lower": (ub - lb) * lb, "true cost": cost, "cost - lower": cost - lb}
for top in (100, 200, 400):
    print("-- taking the", top, "searches a predictor ranks highest: how many of the searches above 5000 / 6500 expansions are among them, and the expansions taken")
    for name, p in cands.items():
        idx = np.argsort(-p)[:top]
        print("   %-26s %4d of %4d   %4d of %4d   %9d expansions" % (name, int((ex[idx] > 5000).sum()), int((ex > 5000).sum()), int((ex[idx] > 6500).sum()),
                                                                    int((ex > 6500).sum()), int(ex[idx].sum())))
for name, p in cands.items():
    print("rank correlation with expansions: %-26s %.3f" % (name, np.corrcoef(np.argsort(np.argsort(p)), np.argsort(np.argsort(ex)))[0, 1]))
