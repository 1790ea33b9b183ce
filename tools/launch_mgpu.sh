#!/bin/bash
# Starts P ranks of cg_solver_mgpu_stencil on one node, one per GPU (no MPI needed).
# usage: tools/launch_mgpu.sh <P> <matrix.mtx | --stencil=N> [more args]
set -eu
P=$1; shift
HERE=$(cd "$(dirname "$0")/.." && pwd)
ID=$(mktemp -u /tmp/spmv_amd_id.XXXXXX)
export HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for ((r = 0; r < P; r++)); do
  RANK=$r WORLD_SIZE=$P LOCAL_RANK=$r SPMV_AMD_ID_FILE=$ID "$HERE/cuda-spmv-benchmark_amd/bin/cg_solver_mgpu_stencil" "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
rm -f "$ID"
exit $rc
