#!/bin/bash
# instruction mix of roadmap_astar on config 5's 10 000 queries (every search on the device): per-kernel SQ counters
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_search
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
PROBE_MODES=device rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O -- python3 profiles/probe_search_quick.py > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_search/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "roadmap_astar" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, len(v), max(v))
PY
tail -2 $O/log.txt | cut -c1-400
