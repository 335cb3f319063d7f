"""tr_knn alone (brute-force exact k-NN in the compound state-space metric) against the Python post-processing."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
eng = robot.engine(0)
for n in (100000, 400000):
    st = W.random_states(robot, n, seed=3)
    eng.knn(st[:1000], 11)
    t0 = time.perf_counter(); idx, dist = eng.knn(st, 11); t1 = time.perf_counter()
    print("knn n", n, "k 11: %.1f ms  (%.3g pair distances/s)" % (1e3 * (t1 - t0), n * n / (t1 - t0)))
