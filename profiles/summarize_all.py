#!/usr/bin/env python3
"""Condenses gpurun_out/<tag> (profiles/collect_all.sh) into the per-kernel table kept under profiles/r02/:
    python profiles/summarize_all.py gpurun_out/<tag> profiles/r02 <tag>
writes <dst>/all_<tag>_kernel_stats.csv (rocprofv3 --stats, as produced), all_<tag>_kernels.json and all_<tag>_kernels.md.

Per kernel: launches and total time (kernel-trace stats); algorithmic bytes / flops of those launches (units.json of
profiles/workload_all.py) -> achieved GB/s against 8 TB/s and TFLOP/s against the 78.6 TFLOP/s fp64 vector peak; HBM
traffic from the FETCH_SIZE / WRITE_SIZE passes (KiB; FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md) and its ratio
to the algorithmic bytes; VALU issue utilisation, wait fraction and clock from the SQ / GRBM passes."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

HBM_PEAK, VALU_PEAK = 8000.0, 78.6


def key_of(name):
    """'void trk::fk_verdict<3, false>(...)' -> 'fk_verdict<3>'; rocPRIM kernels -> 'cache merge (rocPRIM sort + reduce)'."""
    if "rocprim" in name:
        return "cache merge (rocPRIM sort + reduce)"
    name = name.replace("(anonymous namespace)::", "")
    name = name.split("(")[0]                                    # the function's own name, not its argument types
    m = re.search(r"trk::(fk_verdict(?:_retract)?)<(\d+), (?:true|false), (true|false)(?:, (true|false))?>", name)
    if m:                                                         # SPH = true: the sphere-swept checker; SIG = true: edge samples (signature rows)
        return "%s<%s>%s%s" % (m.group(1), m.group(2), " spheres" if m.group(3) == "true" else "", " +sig" if m.group(4) == "true" else "")
    m = re.search(r"trk::([A-Za-z0-9_]+)(?:<(\d+))?", name)
    if not m:
        return None
    base, n = m.group(1), m.group(2)
    if base in ("merge_counts", "merge_keys", "merge_finish"):
        return "cache merge (rocPRIM sort + reduce)"
    if base in ("grid_add_spheres", "grid_add_capsules", "grid_dilate_step", "grid_remove_interior", "dilate2_blocks", "grid_embed", "grid_extract"):
        return "environment edits (grid_add_spheres / grid_add_capsules / grid_dilate_step / grid_remove_interior / dilate2_blocks)"
    if base.startswith("fk_") or base in ("knn_bruteforce", "knn_wave_query"):
        return "%s<%s>" % (base, n) if n else base
    return base


def counters(d, names):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    calls = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = key_of(r["Kernel_Name"])
            if not k or r["Counter_Name"] not in names:
                continue
            out[k][r["Counter_Name"]] += float(r["Counter_Value"])
            did = (r["Dispatch_Id"], k)
            if did not in seen:
                seen.add(did)
                dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                calls[k] += 1
    return out, dur, calls


def main():
    src, dst, tag = sys.argv[1:4]
    os.makedirs(dst, exist_ok=True)
    stats_csv = glob.glob(src + "/stats/**/*_kernel_stats.csv", recursive=True)[0]
    shutil.copy(stats_csv, f"{dst}/all_{tag}_kernel_stats.csv")
    units = json.load(open(f"{src}/u_stats/units.json"))
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(stats_csv)):
        k = key_of(r["Name"])
        if not k:
            continue
        e = rows.setdefault(k, {"calls": 0, "total_ms": 0.0})
        e["calls"] += int(r["Calls"]); e["total_ms"] += float(r["TotalDurationNs"]) / 1e6
    # the first launch of a kernel in a process pays code-object loading and cold caches: its duration is kept apart
    # (first_launch.json, condensed from the kernel trace on the GPU box before the trace is deleted) and the averages below
    # are over the remaining, warm launches
    first = {}
    traces = glob.glob(src + "/stats/**/*kernel_trace.csv", recursive=True)
    if traces:
        seen_at = {}
        for r in csv.DictReader(open(traces[0])):
            k = key_of(r["Kernel_Name"])
            if not k:
                continue
            t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if k not in seen_at or t0 < seen_at[k]:
                seen_at[k] = t0
                first[k] = (t1 - t0) / 1e6
        json.dump(first, open(f"{src}/first_launch.json", "w"))
    elif os.path.exists(f"{src}/first_launch.json"):
        first = json.load(open(f"{src}/first_launch.json"))
    for k, e in rows.items():
        if k in first and e["calls"] > 1:
            e["first_launch_ms"] = first[k]
            e["warm_calls"] = e["calls"] - 1
            e["warm_total_ms"] = e["total_ms"] - first[k]
    fetch, _, _ = counters(src + "/pmc_fetch", {"FETCH_SIZE"})
    write, _, _ = counters(src + "/pmc_write", {"WRITE_SIZE"})
    sq, sq_dur, _ = counters(src + "/sq", {"SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY",
                                           "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU"})
    grbm, grbm_dur, _ = counters(src + "/grbm", {"GRBM_GUI_ACTIVE"})
    for k, e in rows.items():
        u = units.get(k)
        # rates over the warm launches: the work of all launches scaled by (warm calls / calls) -- every launch of a kernel in
        # profiles/workload_all.py does the same amount of work per call within a row, to first order
        warm = "warm_total_ms" in e
        sec = (e["warm_total_ms"] if warm else e["total_ms"]) * 1e-3
        share = (e["warm_calls"] / e["calls"]) if warm else 1.0
        e["avg_ms"] = (e["warm_total_ms"] / e["warm_calls"]) if warm else e["total_ms"] / max(1, e["calls"])
        if u:
            e["unit"] = u.get("unit", "")
            if u.get("units"):
                e["units_per_s"] = share * u["units"] / sec
            if u.get("bytes"):
                e["algorithmic_GBps"] = share * u["bytes"] / sec / 1e9
                e["hbm_frac"] = e["algorithmic_GBps"] / HBM_PEAK
            if u.get("flops"):
                e["fp64_TFLOPs"] = share * u["flops"] / sec / 1e12
                e["fp64_valu_frac"] = e["fp64_TFLOPs"] / VALU_PEAK
            for extra in ("working_set_MiB", "fk_samples"):
                if extra in u:
                    e[extra] = u[extra]
        if k in fetch or k in write:
            e["traffic_bytes"] = (2 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"]) * 1024
            e["fetch_bytes"] = 2 * fetch[k]["FETCH_SIZE"] * 1024
            e["write_bytes"] = write[k]["WRITE_SIZE"] * 1024
            e["traffic_GBps"] = e["traffic_bytes"] / (e["total_ms"] * 1e-3) / 1e9
            if u and u.get("bytes"):
                e["traffic_over_algorithmic"] = e["traffic_bytes"] / u["bytes"]
        if k in grbm and grbm_dur[k] > 0:
            e["clock_GHz"] = grbm[k]["GRBM_GUI_ACTIVE"] / 8 / (grbm_dur[k] * 1e-9) / 1e9     # summed over the 8 XCDs
        if k in sq and sq_dur[k] > 0 and "clock_GHz" in e:
            c = sq[k]
            # SQ cycle counters are quad-cycles summed over the 1024 SIMDs
            e["valu_issue_utilisation"] = c["SQ_INSTS_VALU"] * 4 / 1024 / (e["clock_GHz"] * 1e9 * sq_dur[k] * 1e-9)
            if c["SQ_WAVE_CYCLES"] > 0:
                e["wait_any_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
            e["valu_insts"] = c["SQ_INSTS_VALU"]
    out = {"kernels": rows, "meta": units.get("_meta"), "edge_path_profile_slots": units.get("_edge_path_profile_slots"),
           "units": {k: v for k, v in units.items() if not k.startswith("_")}}
    json.dump(out, open(f"{dst}/all_{tag}_kernels.json", "w"), indent=1)
    cols = [("calls", "%d"), ("total_ms", "%.3f"), ("first_launch_ms", "%.3f"), ("avg_ms", "%.4f"), ("units_per_s", "%.3g"), ("algorithmic_GBps", "%.0f"), ("hbm_frac", "%.3f"),
            ("traffic_over_algorithmic", "%.2f"), ("fp64_TFLOPs", "%.1f"), ("fp64_valu_frac", "%.3f"), ("valu_issue_utilisation", "%.2f"),
            ("wait_any_frac", "%.2f"), ("clock_GHz", "%.2f")]
    with open(f"{dst}/all_{tag}_kernels.md", "w") as f:
        f.write("| kernel | " + " | ".join(c for c, _ in cols) + " | unit |\n|---|" + "---|" * (len(cols) + 1) + "\n")
        for k, e in sorted(rows.items(), key=lambda kv: -kv[1]["total_ms"]):
            f.write("| %s | " % k + " | ".join((fmt % e[c]) if c in e else "" for c, fmt in cols) + " | %s |\n" % e.get("unit", ""))
    print(open(f"{dst}/all_{tag}_kernels.md").read())


if __name__ == "__main__":
    main()
