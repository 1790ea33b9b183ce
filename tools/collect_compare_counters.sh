#!/bin/bash
# Runs on the GPU box (under gpurun). BASELINE configs 2 and 5 with counter evidence: per-kernel durations
# (rocprofv3 --kernel-trace --stats) and L2-to-fabric bytes (separate --pmc passes: FETCH_SIZE, WRITE_SIZE,
# TCC_HIT/MISS) of `python3 tools/compare_operators.py <grid>` for every operator, then a per-kernel table of
# traffic / algorithmic bytes (tools/summarise_compare_counters.py).
# usage: tools/collect_compare_counters.sh <tag> <grid> [<grid> ...]
set -u
TAG=${1:-r03}; shift
export TMPDIR=/tmp
for GRID in "$@"; do
  OUT=gpurun_out/compare_${TAG}_${GRID}
  rm -rf "$OUT"; mkdir -p "$OUT"
  python3 tools/compare_operators.py "$GRID" > "$OUT/compare.txt" 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/compare_operators.py "$GRID" > "$OUT/stats.log" 2>&1
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo "$C" | tr ' ' '_')
    rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 tools/compare_operators.py "$GRID" > "$OUT/pmc_$N.log" 2>&1
  done
  python3 tools/summarise_compare_counters.py "$OUT" "$GRID" > "$OUT/summary.txt" 2>&1
  cat "$OUT/summary.txt"
done
