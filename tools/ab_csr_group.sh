#!/bin/bash
# CSR-stream: consecutive blocks per XCD (SPMV_AMD_XCD_GROUP) x rows per block, at the sizes of BASELINE configs 2 / 5.
# Round 2 swept runs of 2-16 blocks (all slower than dispatch order); the STENCIL5 finding -- the best run is one grid
# row plus ~1100 columns -- puts the interesting run at 57-64 blocks of 176 rows for a 10 000-column grid.
# usage: tools/ab_csr_group.sh <grid> [groups...]
GRID=${1:-10000}; shift
GROUPS_=${@:-1 8 16 32 48 57 60 63 64}
for G in $GROUPS_; do
  echo -n "xcd_group=$G  "
  SPMV_AMD_XCD_GROUP=$G python3 tools/compare_operators.py $GRID cusparse-csr 2>/dev/null | grep "^cusparse-csr"
done
