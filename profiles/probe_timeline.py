#!/usr/bin/env python3
"""GPU timeline of one call of the roadmap build's phases (config 3: 100 k vertices, 10-NN) from a rocprofv3 --kernel-trace run:
which kernels ran when, how much of the call's span the GPU was busy, how much of that were FK kernels, the longest idle gaps.
The measured call sits between two marker launches (candidate_states_kernel with 7 777 candidates).

    rocprofv3 --kernel-trace --output-format csv -d <dir>/trace -- python3 profiles/probe_timeline.py run <what> <dir>/marks.json
    python3 profiles/probe_timeline.py summarize <dir> <out.json>           what: edges | connect | create_roadmap
"""
import csv
import glob
import importlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(what, marks_path):
    import torch
    irt = importlib.import_module("interactive-rate-tendons_amd")
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
    states, _ = rb.sample_valid_vertices(100000)
    edges = rb.knn_edges_gpu(states, 11)
    chk.engine.reserve_edges(len(edges))
    mark = torch.empty(7777 * chk.engine.state_size, dtype=torch.float64, device="cuda")

    def call():
        if what == "edges":
            v, nf = rb.validate_edges(states, edges)
            return dict(edges=int(len(edges)), fk_samples=int(nf.sum()), valid=int(v.sum()))
        if what == "connect":
            e, ec = rb.connect(states, edges, device=True)
            return dict(edges=int(len(edges)), accepted=int(len(e)), blocks=int(ec["offsets"][-1]))
        prm, rm = rb.create_roadmap(100000, k=10, device=True)
        return dict(vertices=int(len(rm["states"])), edges=int(len(rm["edges"])), timing={k: v.get("seconds") for k, v in rb.timing.items()})

    for _ in range(2):
        call()
    torch.cuda.synchronize()
    chk.engine.candidate_states_dev(999, 0, 7777, mark)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = call()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    chk.engine.candidate_states_dev(999, 0, 7777, mark)
    torch.cuda.synchronize()
    json.dump(dict(what=what, wall_ms=1e3 * wall, **info), open(marks_path, "w"))


def union(iv):
    tot, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if ce is None or s > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ((ce - cs) if ce is not None else 0)


def summarize(src, out):
    marks = json.load(open(glob.glob(src + "/**/marks.json", recursive=True)[0]))
    f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        m = re.search(r"trk::([A-Za-z0-9_]+)", kn)
        name = m.group(1) if m else ("rocprim" if "rocprim" in kn else re.sub(r"[^A-Za-z0-9_:]", "", kn)[:36])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    mk = [i for i, r in enumerate(rows) if r[2] == "candidate_states_kernel"]
    call = rows[mk[-2] + 1: mk[-1]]
    t0, t1 = rows[mk[-2]][1], rows[mk[-1]][0]                 # end of the first marker .. start of the second = the call incl. host work
    k0, k1 = call[0][0], max(r[1] for r in call)
    per = {}
    for s, e, n in call:
        d = per.setdefault(n, {"launches": 0, "sum_ms": 0.0})
        d["launches"] += 1
        d["sum_ms"] += (e - s) / 1e6
    busy = sorted((s, e) for s, e, _ in call)
    gaps, ce = [], None
    for s, e in busy:
        if ce is not None and s > ce:
            gaps.append(((s - ce) / 1e6, (ce - t0) / 1e6))
        ce = e if ce is None else max(ce, e)
    gaps.sort(reverse=True)
    res = dict(marks=marks, between_markers_ms=(t1 - t0) / 1e6, first_to_last_kernel_ms=(k1 - k0) / 1e6,
               idle_before_first_kernel_ms=(k0 - t0) / 1e6, idle_after_last_kernel_ms=(t1 - k1) / 1e6,
               any_kernel_busy_ms=union([(s, e) for s, e, _ in call]) / 1e6,
               fk_kernels_busy_ms=union([(s, e) for s, e, n in call if n.startswith("fk_")]) / 1e6,
               largest_gaps_ms_at_ms=gaps[:12], per_kernel=per)
    json.dump(res, open(out, "w"), indent=1)
    print("%s: host wall %.2f ms; between markers %.2f ms; kernels %.2f ms first-to-last, GPU busy %.2f ms (FK kernels %.2f ms); idle before %.2f, after %.2f" %
          (marks["what"], marks["wall_ms"], res["between_markers_ms"], res["first_to_last_kernel_ms"], res["any_kernel_busy_ms"],
           res["fk_kernels_busy_ms"], res["idle_before_first_kernel_ms"], res["idle_after_last_kernel_ms"]))
    print("  largest idle gaps (ms @ ms into the call):", ", ".join("%.2f@%.1f" % g for g in gaps[:10]))
    for s_, e_, n in call:
        if n.startswith("fk_") and (e_ - s_) > 2e5:              # every FK launch longer than 0.2 ms: start within the call, duration
            print("  %-22s t = %7.2f ms   %7.3f ms" % (n, (s_ - t0) / 1e6, (e_ - s_) / 1e6))
    for n, d in sorted(per.items(), key=lambda kv: -kv[1]["sum_ms"])[:18]:
        print("  %-36s %4d launches %8.3f ms" % (n, d["launches"], d["sum_ms"]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3])
    else:
        summarize(sys.argv[2], sys.argv[3])
