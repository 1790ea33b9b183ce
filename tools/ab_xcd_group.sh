#!/bin/bash
# ELLPACK kernels: logical-block relabelling (SPMV_AMD_XCD_GROUP = consecutive blocks per XCD) x workgroup shape
# (SPMV_AMD_ELL_SHAPE bit 0: one-wave workgroups, bit 1: nontemporal planes). usage: tools/ab_xcd_group.sh  (on the GPU box)
for shape in 2 3; do
for g in 1 4 6 8 12 16 24 32 64; do
  echo "== SPMV_AMD_ELL_SHAPE=$shape SPMV_AMD_XCD_GROUP=$g"
  SPMV_AMD_ELL_SHAPE=$shape SPMV_AMD_XCD_GROUP=$g python3 tools/compare_operators.py 15000 ellpack stencil5-ellpack 2>/dev/null | grep -E "^(cusparse|ellpack|stencil5)"
done
done
echo "== 20000, shape 2"
for g in 1 8 16; do SPMV_AMD_XCD_GROUP=$g python3 tools/compare_operators.py 20000 stencil5-ellpack stencil5-csr 2>/dev/null | grep -E "^(cusparse|ellpack|stencil5)"; done
