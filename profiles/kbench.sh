#!/bin/bash
# Compile and time variants of the RK4 loop on the GPU box (profiles/kbench.hip).  Usage: bash profiles/kbench.sh <out.txt> "<flags 1>" "<flags 2>" ...
# Every variant is compiled for N = 3 and N = 4 in parallel, then run one after the other.
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
mkdir -p /tmp/kb
i=0
for V in "$@"; do
  for N in 3 4; do
    ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I interactive-rate-tendons_amd/csrc -DKB_N=$N -DKB_LOG2=$((23 - N)) $V profiles/kbench.hip -o /tmp/kb/v${i}_$N > /tmp/kb/v${i}_$N.log 2>&1 || echo "compile failed: $V N=$N" ) &
  done
  i=$((i + 1))
  if (( i % 6 == 0 )); then wait; fi
done
wait
i=0
: > "$OUT"
for V in "$@"; do
  for N in 3 4; do
    echo "[$V]" | tee -a "$OUT"
    if [ -x /tmp/kb/v${i}_$N ]; then timeout -k 5 120 /tmp/kb/v${i}_$N 2>&1 | tee -a "$OUT"; else tail -3 /tmp/kb/v${i}_$N.log | tee -a "$OUT"; fi
  done
  i=$((i + 1))
done
