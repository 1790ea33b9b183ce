#!/bin/bash
# Runs on the GPU box: the P = 8 middle slab of the headline grid (rank 3 of 8, two neighbours) through the full multi-rank
# pipeline with the rank as its own neighbour, once with ncclAllReduce and once with the peer mailbox, each under
# rocprofv3 --kernel-trace; tools/trace_gaps.py turns each trace into the per-kernel busy / idle table of the last solve,
# and the solver's own stage timeline (spmv_amd_cg_slab_set_timeline) is printed beside it.
# usage: tools/collect_slab_timeline.sh <tag> [P=8] [rank=3]
set -u
TAG=${1:-r03}; P=${2:-8}; R=${3:-3}
OUT=gpurun_out/slab_timeline_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
for AR in rccl mailbox; do
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$AR" -- python3 tools/probe_slab.py $P $R 5 $AR > "$OUT/probe_$AR.txt" 2>&1
  F=$(find "$OUT/trace_$AR" -name "*kernel_trace.csv" | head -1)
  { echo "== slab $R of $P, all-reduce: $AR"; grep -E "^slab|^stage timeline" "$OUT/probe_$AR.txt"; python3 tools/trace_gaps.py "$F" 14 -2; echo "-- the extra solve with stage-boundary events (a barrier packet in front of every stage):"; python3 tools/trace_gaps.py "$F" 14 -1 | tail -8; } > "$OUT/timeline_$AR.txt" 2>&1
  cat "$OUT/timeline_$AR.txt"
done
