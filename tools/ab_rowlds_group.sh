#!/bin/bash
# validation of the rule  group = ceil((n + 1100) / (8 * columns per logical block))  (run of 8 * group blocks = one grid row + ~1100 columns)
for rep in 1 2; do
for spec in "20000 4 21 22" "10000 4 11 12" "12000 4 12 13 14" "17000 4 17 18 19" "8192 4 9 10" "4096 4 5 6"; do
  set -- $spec; n=$1; shift
  for g in "$@"; do
    echo -n "rep $rep row-lds grid $n group $g: "; SPMV_AMD_ROWLDS_GROUP=$g python3 tools/compare_operators.py $n stencil5-csr 2>/dev/null | grep -E "^stencil5" | awk '{print $3, $4}'
  done
done
for spec in "20000 8 10 11 12" "10000 4 5 6 8" "15000 7 8 9"; do
  set -- $spec; n=$1; shift
  for g in "$@"; do
    echo -n "rep $rep ell grid $n group $g: "; SPMV_AMD_XCD_GROUP=$g python3 tools/compare_operators.py $n stencil5-ellpack ellpack 2>/dev/null | grep -E "^(stencil5|ellpack)" | awk '{printf "%s %s  ", $1, $3}'; echo
  done
done
done
for g in 4 21; do echo -n "CG 20000 row-lds group $g: "; SPMV_AMD_ROWLDS_GROUP=$g python3 bench.py --no-cpu-baseline --no-spmv --no-scaling-probe --no-ceiling --steps 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
