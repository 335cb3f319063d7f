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
# ... and what is known DURING a search: the key at the head of the list after 2 000 / 3 000 / 4 000 expansions against h(start) and the landmark upper bound
if d.shape[1] >= 13:
    h0, f2, f3, f4 = d[:, 9], d[:, 10], d[:, 11], d[:, 12]
    for n_at, f_at in ((2000, f2), (3000, f3), (4000, f4)):
        alive = ex > n_at
        long_ = ex > 6500
        print("-- searches still running after %d expansions: %d, of them above 6 500 in the end: %d" % (n_at, int(alive.sum()), int((alive & long_).sum())))
        feats = {"progress (f - h0) / (upper - h0)": (f_at - h0) / np.maximum(ub - h0, 1e-12), "progress (f - h0) / (cost - h0) [not known to the search]": (f_at - h0) / np.maximum(cost - h0, 1e-12),
                 "f - h0": f_at - h0, "upper - f": ub - f_at, "h0": h0, "distance": dist}
        for name, p in feats.items():
            p = np.where(alive, p, np.nan)
            pa, la = p[alive], long_[alive]
            order = np.argsort(pa)          # small progress first
            for frac in (0.1, 0.2, 0.3):
                n_take = int(frac * len(pa))
                lo = la[order[:n_take]].sum(); hi = la[order[::-1][:n_take]].sum()
                print("   %-58s lowest %2d %%: %3d of %3d long ones; highest %2d %%: %3d" % (name, int(100 * frac), int(lo), int(la.sum()), int(100 * frac), int(hi)))
