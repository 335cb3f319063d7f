#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --output-format csv)
into per-launch HBM bytes per kernel.  rocprofv3 reports both counters in KiB.  On gfx950 FETCH_SIZE
reports half of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM: "double it before
comparing with a byte count").  Calibrated on this path: backbone_voxel_sweep reads every backbone
point exactly once (2^20 x 3121 B = 3.27 GB per launch) and rocprofv3 reports FETCH_SIZE = 1.60 GB,
so the doubled figure is the one used (`hbm_bytes_per_launch`); the raw sum is kept next to it.

    python profiles/collect_traffic.py <fetch_dir> <write_dir> <out.json>
"""
import collections
import csv
import glob
import json
import sys

NAMES = {"fk_verdict": "fk_verdict", "fk_sweep_fused": "fk_sweep_fused", "fk_rk4_batch": "fk_rk4_batch", "backbone_voxel_sweep": "backbone_voxel_sweep",
         "cached_blocks_vs_grid": "cached_blocks_vs_grid"}


def avg_counter(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                for key, name in NAMES.items():
                    if key in r["Kernel_Name"]:
                        out[name].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


def main():
    fetch, write, dst = sys.argv[1:4]
    f, w = avg_counter(fetch, "FETCH_SIZE"), avg_counter(write, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, (0.0, 0))[0], w.get(k, (0.0, 0))[0]
        out[k] = {"FETCH_SIZE_KiB_avg": fk, "WRITE_SIZE_KiB_avg": wk, "launches": f.get(k, (0, 0))[1],
                  "hbm_bytes_per_launch": (2 * fk + wk) * 1024, "hbm_bytes_per_launch_raw_counters": (fk + wk) * 1024}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
