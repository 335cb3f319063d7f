#!/usr/bin/env python3
"""Where do the fk_verdict kernels' L2-miss bytes go?  Launches the verdict-only kernels in labelled phases with distinct
batch sizes, for two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; profiles/collect_traffic_split.sh), and -- as
`python profiles/probe_traffic.py --summarize <dir> <out.json>` -- condenses the per-dispatch counter CSVs of those passes.

    python profiles/probe_traffic.py <out_dir>
"""
import csv
import glob
import importlib
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def run(out_dir):
    import torch
    irt = importlib.import_module("interactive-rate-tendons_amd")
    W = irt.workloads
    os.makedirs(out_dir, exist_ok=True)
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    phases = []

    def batch(label, robot, n, tau, flags=False, env=None, reps=2):
        chk = with_env(env or {}, lambda: irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox))
        st = torch.from_numpy(W.random_states(robot, n, seed=3, tau_max=tau)).cuda()
        bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        tips = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
        fl = torch.zeros(n, dtype=torch.uint8, device="cuda") if flags else None
        for _ in range(reps):
            chk.engine.validate_batch_dev(st, n, bits, tips, fl)
        torch.cuda.synchronize()
        S = robot.state_size()
        phases.append(dict(label=label, n=n, grid=(n + 63) // 64 * 64, reps=reps, algorithmic_bytes_per_check=8 * S + 24 + 0.125 + (1 if flags else 0)))
        del chk

    r3, r4 = W.robot_config2(), W.robot_config3()
    batch("verdict3_plain", r3, (1 << 19) + 64 * 1, 10.0)
    batch("verdict4_plain", r4, (1 << 19) + 64 * 2, 20.0)
    batch("verdict4_flags", r4, (1 << 19) + 64 * 3, 20.0, flags=True)
    for k, (mk, tau) in enumerate(((W.robot_config2, 10.0), (W.robot_config3, 20.0))):
        rr = mk()
        rr.enable_retraction = True
        batch("retract%d_sorted" % (3 + k), rr, (1 << 19) + 64 * (4 + 2 * k), tau)
        batch("retract%d_arrival" % (3 + k), rr, (1 << 19) + 64 * (5 + 2 * k), tau, env={"TENDON_HIP_RETRACT_SORT": "0"})
    # edge samples (signature rows written from the point hook): config 3's roadmap at 100 k vertices
    chk = irt.VoxelBackboneValidityChecker(r4, irt.VoxelEnvironment(), vox)
    mv = irt.VoxelBackboneMotionValidator(chk)
    rb = irt.RoadmapBuilder(chk, mv, seed=1)
    states, _ = rb.sample_valid_vertices(100000)
    edges = rb.knn_edges_gpu(states, 11)
    for _ in range(2):
        valid, nfk = rb.validate_edges(states, edges)
    torch.cuda.synchronize()
    phases.append(dict(label="verdict4_edges", edges=int(len(edges)), fk_samples=int(nfk.sum()), reps=2,
                       note="every fk_verdict<4> launch whose grid is none of the batch phases' grids; own samples per call = fk_samples - 2 * edges + vertices"))
    json.dump(phases, open(os.path.join(out_dir, "phases.json"), "w"), indent=1)


def summarize(src, out):
    phases = json.load(open(glob.glob(src + "/**/phases.json", recursive=True)[0]))
    by_grid = {p["grid"]: p for p in phases if "grid" in p}
    res = {}
    for counter, sub, scale in (("FETCH_SIZE", "pmc_fetch", 2 * 1024.0), ("WRITE_SIZE", "pmc_write", 1024.0)):     # KiB; FETCH doubled on gfx950
        for f in glob.glob(os.path.join(src, sub) + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter:
                    continue
                m = re.search(r"trk::(fk_verdict\w*)<(\d+), (true|false), (true|false)", r["Kernel_Name"])
                if not m:
                    # the retraction robots' prologue kernel belongs to the same check (its hand-over planes are part of the path's traffic)
                    m = re.search(r"trk::(fk_retract_prologue)<(\d+), (true|false)()", r["Kernel_Name"])
                if not m:
                    continue
                grid = int(r["Grid_Size"])
                p = by_grid.get(grid)
                label = p["label"] if p else "verdict4_edges" if m.group(1) == "fk_verdict" and m.group(2) == "4" else "other"
                e = res.setdefault(label, {"kernel": "%s<%s>" % (m.group(1).replace("fk_retract_prologue", "fk_verdict_retract"), m.group(2)),
                                           "launches": {"FETCH_SIZE": 0, "WRITE_SIZE": 0},
                                           "bytes": {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}, "lanes": {"FETCH_SIZE": 0, "WRITE_SIZE": 0}})
                e["bytes"][counter] += float(r["Counter_Value"]) * scale
                if m.group(1) == "fk_retract_prologue":
                    e.setdefault("prologue_bytes", {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})[counter] += float(r["Counter_Value"]) * scale
                    continue                                   # same lanes as the main kernel's launch: counted once
                e["launches"][counter] += 1
                e["lanes"][counter] += grid
    for label, e in res.items():
        e["fetch_bytes_per_lane"] = e["bytes"]["FETCH_SIZE"] / max(1, e["lanes"]["FETCH_SIZE"])
        e["write_bytes_per_lane"] = e["bytes"]["WRITE_SIZE"] / max(1, e["lanes"]["WRITE_SIZE"])
        p = next((q for q in phases if q["label"] == label), {})
        if "algorithmic_bytes_per_check" in p:
            e["algorithmic_bytes_per_check"] = p["algorithmic_bytes_per_check"]
            e["traffic_over_algorithmic"] = (e["fetch_bytes_per_lane"] + e["write_bytes_per_lane"]) / p["algorithmic_bytes_per_check"]
    json.dump({"phases": phases, "per_phase": res}, open(out, "w"), indent=1)
    for label, e in sorted(res.items()):
        print("%-18s %-22s fetch %8.1f B/lane  write %8.1f B/lane  x%s" % (label, e["kernel"], e["fetch_bytes_per_lane"], e["write_bytes_per_lane"],
                                                                          ("%.2f" % e["traffic_over_algorithmic"]) if "traffic_over_algorithmic" in e else "-"))


if __name__ == "__main__":
    if sys.argv[1] == "--summarize":
        summarize(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1])
