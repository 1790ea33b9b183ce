#!/bin/bash
# Runs on the GPU box (under gpurun): per-kernel stats and HBM-traffic counters for one operator.
# usage: tools/gpu_profile.sh <tag> <mode> <grid> [reps]
# Counter passes are separate rocprofv3 runs with --pmc only (no tracing domains), as the pool requires.
set -u
TAG=${1:-r01}; MODE=${2:-stencil5-csr}; GRID=${3:-20000}; REPS=${4:-5}
OUT=gpurun_out/prof_${TAG}_${MODE}_${GRID}
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/profile_spmv.py "$MODE" "$GRID" "$REPS" > "$OUT/stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
  N=$(echo "$C" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 tools/profile_spmv.py "$MODE" "$GRID" "$REPS" > "$OUT/pmc_$N.log" 2>&1
done
find "$OUT" -name "*.csv" | head -40
