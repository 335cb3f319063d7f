#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_sqc
mkdir -p $O
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_sqc/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "fk_sweep_fused" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, sum(v) / len(v))
PY
