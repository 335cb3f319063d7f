import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = "/tmp/search_hist_tail.txt"
os.environ["TENDON_HIP_SEARCH_HIST"] = path
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
new_vox, _ = W.reach_environment(seed=7, n_spheres=72)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
states, _ = rb.sample_valid_vertices(600000, batch=1 << 17)
edges = rb.knn_edges_gpu(states, 11)
valid, _ = rb.validate_edges(states, edges)
e_ok = edges[valid]
prm = irt.VoxelCachedLazyPRM(chk, states, e_ok)
prm.set_caches(rb.vertex_caches(states), rb.edge_caches(states, e_ok))
prm.set_obstacles(new_vox)
pairs = np.random.default_rng(17).integers(0, len(states), size=(10000, 2))
prm.prepare(64)
for form in ("eager", "lazy", "eager", "lazy"):
    if os.path.exists(path):
        os.remove(path)
    prm.clearValidity()
    if form == "eager":
        prm.revalidate()
    t0 = time.perf_counter()
    out = prm.solveWithRoadmap(pairs[:, 0], pairs[:, 1])
    dt = time.perf_counter() - t0
    d = np.loadtxt(path)
    for rnd in np.unique(d[:, 0]):
        x = d[d[:, 0] == rnd]
        ex = x[:, 3]
        h = ex > 0
        print("%s %.1f ms, round %d: %d searches, on the host %d: expansions sum %.3g, median %d, max %d; found among them %d; the ten longest: %s" %
              (form, 1e3 * dt, rnd, len(x), int(h.sum()), ex[h].sum(), int(np.median(ex[h])) if h.any() else 0, int(ex.max()), int(x[h, 2].sum()),
               np.sort(ex)[-10:].astype(int).tolist()), flush=True)
