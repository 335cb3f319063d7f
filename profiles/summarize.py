#!/usr/bin/env python3
"""Condenses gpurun_out/<tag> (written by profiles/collect.sh) into the small files kept under profiles/:
    python profiles/summarize.py gpurun_out/<tag> profiles/r01 <tag>
writes  <dst>/bench_<tag>.json, bench_<tag>_under_rocprof.json, bench_<tag>_kernel_stats.csv,
        traffic_<tag>.json (+ profiles/traffic_latest.json), sq_counters_<tag>.json."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

src, dst, tag = sys.argv[1:4]
os.makedirs(dst, exist_ok=True)
here = os.path.dirname(os.path.abspath(__file__))
shutil.copy(glob.glob(src + "/stats/**/*_kernel_stats.csv", recursive=True)[0], f"{dst}/bench_{tag}_kernel_stats.csv")
for name, out in (("bench_default.log", f"bench_{tag}.json"), ("bench_stats.log", f"bench_{tag}_under_rocprof.json")):
    line = [l for l in open(f"{src}/{name}") if l.startswith("{")][-1]
    open(f"{dst}/{out}", "w").write(line)
subprocess.check_call([sys.executable, f"{here}/collect_traffic.py", f"{src}/pmc_fetch", f"{src}/pmc_write", f"{dst}/traffic_{tag}.json"],
                      stdout=subprocess.DEVNULL)
sys.path.insert(0, os.path.dirname(here))
import importlib
tj = json.load(open(f"{dst}/traffic_{tag}.json"))
tj["source"] = os.path.relpath(f"{dst}/traffic_{tag}.json", os.path.dirname(here))
tj["source_hash"] = importlib.import_module("interactive-rate-tendons_amd._lib").source_hash()   # run on the tree the profile was taken on
json.dump(tj, open(f"{dst}/traffic_{tag}.json", "w"), indent=1)
shutil.copy(f"{dst}/traffic_{tag}.json", f"{here}/traffic_latest.json")
out = {}
for d in ("sq", "grbm"):
    f = glob.glob(f"{src}/{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg, dur = collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        k = "fk_verdict" if "fk_verdict" in kn else "fk_sweep_fused" if "fk_sweep_fused" in kn else ("fk_rk4_batch" if "fk_rk4" in kn else ("backbone_voxel_sweep" if "voxel_sweep" in kn else None))
        if not k:
            continue
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[(k, d)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (k, c), v in agg.items():
        out.setdefault(k, {})[c] = sum(v) / len(v)
    for (k, dd), v in dur.items():
        out.setdefault(k, {})["avg_ms_" + dd] = sum(v) / len(v) / 1e6
for k, d in out.items():
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ cycle counters are quad-cycles over 1024 SIMDs
    d["clock_GHz"] = d["GRBM_GUI_ACTIVE"] / 8 / (d["avg_ms_grbm"] * 1e-3) / 1e9
    d["valu_issue_utilisation"] = d["SQ_INSTS_VALU"] * 4 / 1024 / (d["clock_GHz"] * 1e9 * d["avg_ms_sq"] * 1e-3)
    d["wait_any_frac"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
json.dump(out, open(f"{dst}/sq_counters_{tag}.json", "w"), indent=1)
b = json.load(open(f"{dst}/bench_{tag}.json"))
print(json.dumps({"value": b["value"], "kernels": b["kernels"], "cpu_baseline": b.get("cpu_baseline"),
                  "pcie": b.get("pcie_inclusive_checks_per_s"),
                  "clock": {k: round(v["clock_GHz"], 3) for k, v in out.items()},
                  "valu_util": {k: round(v["valu_issue_utilisation"], 3) for k, v in out.items()},
                  "wait": {k: round(v["wait_any_frac"], 3) for k, v in out.items()}}, indent=1))
