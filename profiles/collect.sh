#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh <tag>        -> gpurun_out/<tag>/...
# Afterwards, in the dev container:  python profiles/summarize.py gpurun_out/<tag> profiles/r01 <tag>
# Counters are collected in their own passes (never together with trace domains other than kernel-trace).
set -e
TAG=${1:-prof}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
O=$R/gpurun_out/$TAG
mkdir -p "$O"
python3 bench.py > "$O/bench_default.log" 2>&1
# the kernel-trace run skips the CPU / PCIe legs, so every launch of the hot kernel it averages is one of the timed region's shape
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --no-cpu-baseline > "$O/bench_stats.log" 2>&1
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- $B > "$O/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- $B > "$O/write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU \
  --output-format csv -d "$O/sq" -- $B > "$O/sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$O/grbm" -- $B > "$O/grbm.log" 2>&1
tail -1 "$O/bench_default.log" | cut -c1-200
