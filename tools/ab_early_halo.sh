#!/bin/bash
# Early halo exchange (SPMV_AMD_EARLY_HALO=1, the default since round 3) against round 2's order (=0) on the stand-in
# slabs of an 8-GPU and a 4-GPU run of the headline grid, one GPU, RCCL send / recv with the rank as its own neighbour.
# usage: tools/ab_early_halo.sh   (on the GPU box)
for CASE in "8 3" "8 0" "4 1"; do
  for E in 0 1 0 1; do
    echo -n "P r = $CASE  SPMV_AMD_EARLY_HALO=$E  "
    SPMV_AMD_EARLY_HALO=$E python3 tools/probe_slab.py $CASE 10 rccl 2>/dev/null | grep -E "^slab|^stage" | sed -e 's/^stage timeline of one extra solve (us per iteration unless named otherwise): /   timeline: /' | cut -c1-420 | tr '\n' ' '
    echo
  done
done
