set -u
export TMPDIR=/tmp
O=gpurun_out/r05b; rm -rf $O; mkdir -p $O
for ROLE in "8 3" "8 0" "4 1" "2 1" "1 0"; do
  python3 tools/ab_loop_options.py reduce_one_launch 20000 $ROLE 8 > "$O/ab_$(echo $ROLE | tr ' ' '_').txt" 2>&1
  cat "$O/ab_$(echo $ROLE | tr ' ' '_').txt" | grep -v "^\[" 
done
for V in 1 0; do
  SPMV_AMD_REDUCE_ONE_LAUNCH=$V rocprofv3 --kernel-trace --output-format csv -d "$O/trace_$V" -- python3 tools/probe_slab.py 8 3 5 rccl > "$O/probe_$V.txt" 2>&1
  F=$(find "$O/trace_$V" -name "*kernel_trace.csv" | head -1)
  echo "== slab 3 of 8, SPMV_AMD_REDUCE_ONE_LAUNCH=$V"; grep -E "^slab" "$O/probe_$V.txt"; python3 tools/trace_gaps.py "$F" 14
done
