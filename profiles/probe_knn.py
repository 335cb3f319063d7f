"""tr_knn alone (exact k-NN in the compound state-space metric): wall time per call for the seeding-window divisor (a tuning
switch of knn_impl, read per call)."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
eng = robot.engine(0)
sizes = [int(a) for a in sys.argv[1:]] or [100000, 400000, 1000000]
for n in sizes:
    st = W.random_states(robot, n, seed=3)
    eng.knn(st[:5000], 11)
    ref = None
    for div in (8, 16, 32, 64, 128, 256):
        for pre in (False,):
            os.environ["TENDON_HIP_KNN_HW_DIV"] = str(div)
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter(); idx, dist = eng.knn(st, 11); best = min(best, time.perf_counter() - t0)
            if ref is None:
                ref = (idx, dist)
            same = np.array_equal(idx, ref[0]) and np.array_equal(dist, ref[1])
            print("knn n %d k 11  window n/%d: %.1f ms  (%.3g pair distances/s)  same %s" % (n, div, 1e3 * best, n * float(n) / best, same), flush=True)
