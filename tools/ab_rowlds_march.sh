#!/bin/bash
# Row-lds march (SPMV_AMD_ROWLDS_ROWS = 1 / 2 / 4 grid rows per wave, x rotating in registers): standalone operator
# launches and the in-loop SpMV of the CG solver at one grid, same box, same session. One bounded attempt at the
# x[row +- n] re-fetch; adopt at >= 2 % in the loop with bit-identical results.
# usage: tools/ab_rowlds_march.sh [grid=20000] [groups...]   (on the GPU box)
GRID=${1:-20000}; shift
GROUPS_=${@:-0}
for G in $GROUPS_; do
for R in 1 2 4; do
  echo "== SPMV_AMD_ROWLDS_ROWS=$R SPMV_AMD_ROWLDS_GROUP=$G"
  SPMV_AMD_ROWLDS_ROWS=$R SPMV_AMD_ROWLDS_GROUP=$G python3 tools/profile_spmv.py stencil5-csr $GRID 12 2>/dev/null | grep "^stencil5"
  SPMV_AMD_ROWLDS_ROWS=$R SPMV_AMD_ROWLDS_GROUP=$G python3 bench.py --grid $GRID --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-probe --no-spmv --no-ceiling 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   in loop: %.4f ms per SpMV launch (%.3f of 8 TB/s), solve %.3f ms, %d iterations, parity max rel err %.2e, history tail %s' % (d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step'], d['config']['iterations_per_solve'], d['parity_vs_golden'].get('max_rel_err', float('nan')), d['config']['residual_history'][-1].hex()))"
done
done
