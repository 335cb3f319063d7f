#!/bin/bash
# rocprofv3 evidence for EVERY kernel (profiles/workload_all.py), run on the GPU box through gpurun from the repo root:
#   bash profiles/collect_all.sh <tag>      -> gpurun_out/<tag>/{stats,pmc_fetch,pmc_write,sq,grbm}/...
# then, in the dev container:  python profiles/summarize_all.py gpurun_out/<tag> profiles/r02 <tag>
# Counters are collected in their own passes (never together with trace domains other than kernel-trace); the program
# itself follows `--` (no env / bash -c hop).
set -e
TAG=${1:-all}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
O=$R/gpurun_out/$TAG
mkdir -p "$O"
python3 profiles/workload_all.py "$O/plain" > "$O/plain.log" 2>&1
echo "plain done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 profiles/workload_all.py "$O/u_stats" > "$O/stats.log" 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 profiles/workload_all.py "$O/u_fetch" > "$O/fetch.log" 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 profiles/workload_all.py "$O/u_write" > "$O/write.log" 2>&1
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU \
  --output-format csv -d "$O/sq" -- python3 profiles/workload_all.py "$O/u_sq" > "$O/sq.log" 2>&1
echo "sq done"
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$O/grbm" -- python3 profiles/workload_all.py "$O/u_grbm" > "$O/grbm.log" 2>&1
echo "grbm done"
# keep what travels back small: the per-dispatch CSVs are condensed here
python3 profiles/summarize_all.py "$O" "$O/summary" "$TAG" > "$O/summary.log" 2>&1 || true   # (also condenses the trace into first_launch.json)
rm -rf "$O"/pmc_fetch "$O"/pmc_write "$O"/sq "$O"/grbm
find "$O/stats" -name "*kernel_trace.csv" -delete
tail -3 "$O/summary.log"
