"""Wall time of the exact k-NN table (engine.knn: upload, cell sort, search, download) at the two roadmap sizes, k = 11."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
eng = robot.engine()
for n in (100000, 600000):
    st = W.random_states(robot, n, seed=5)
    for k in (11,):
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); idx, dist = eng.knn(st, k); best = min(best, time.perf_counter() - t0)
        print("n=%d k=%d: %.2f ms  checksum %d %.12g" % (n, k, 1e3 * best, int(idx.astype(np.int64).sum()), float(dist[np.isfinite(dist)].sum())), flush=True)
