"""Durations of the neighbour search's launches (seeding pass, main pass, merge) under rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 profiles/probe_knn_launches.py run [n]
    python3 profiles/probe_knn_launches.py summarize <dir>"""
import csv, glob, importlib, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import numpy as np
    irt = importlib.import_module("interactive-rate-tendons_amd")
    W = irt.workloads
    robot = W.robot_config3()
    eng = robot.engine()
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 600000
    st = W.random_states(robot, n, seed=5)
    for _ in range(3):
        t0 = time.perf_counter(); idx, dist = eng.knn(st, 11); dt = time.perf_counter() - t0
    print("knn n=%d: %.1f ms wall" % (n, 1e3 * dt))
else:
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    for s, e, n in rows[-14:]:
        print("%9.3f ms  %s" % ((e - s) / 1e6, re.sub(r"\(.*", "", n)[:90]))
