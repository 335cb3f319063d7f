import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
eng = robot.engine()
st = W.random_states(robot, 600000, seed=5)
for k in (2, 6, 11, 21, 41):
    for _ in range(2):
        t0 = time.perf_counter(); eng.knn(st, k); dt = time.perf_counter() - t0
    print("k=%d: %.1f ms" % (k, 1e3 * dt), flush=True)
