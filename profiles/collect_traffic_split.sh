#!/bin/bash
# FETCH / WRITE split of the verdict-only kernels per phase (profiles/probe_traffic.py), on the GPU box through gpurun:
#   bash profiles/collect_traffic_split.sh <tag>   -> gpurun_out/<tag>/traffic_split.json
# Counters in their own passes, the program itself after `--`.
set -e
TAG=${1:-traffic}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
O=$R/gpurun_out/$TAG
mkdir -p "$O"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 profiles/probe_traffic.py "$O/u_fetch" > "$O/fetch.log" 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 profiles/probe_traffic.py "$O/u_write" > "$O/write.log" 2>&1
echo "write done"
python3 profiles/probe_traffic.py --summarize "$O" "$O/traffic_split.json" | tee "$O/traffic_split.txt"
rm -rf "$O"/pmc_fetch "$O"/pmc_write
