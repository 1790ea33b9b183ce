#!/bin/bash
# Runs on the GPU box (under gpurun): per-kernel evidence for the reference's single-GPU CG entry points.
#   1. rocprofv3 --kernel-trace --stats of tools/time_cg_entry_points.py <big grid> (cg_solve_device per operator beside
#      the slab solver; synthetic matrices generated in HBM)
#   2. the harness binaries on a real .mtx file: this repo's bin/cg_solver and the reference's own, unmodified
#      src/main/cg_solver.cu built against this library (oracle/_ref/ref_cg_solver), same file, under the same profiler
#   3. bin/cg_solver --stencil=<mid grid> (in-memory generator, host CSR build) for the three device operators
# usage: tools/collect_cg_entry_profiles.sh <tag> [big=20000] [file_grid=3000] [mid=10000]
set -u
TAG=${1:-r03}; BIG=${2:-20000}; FILEGRID=${3:-3000}; MID=${4:-10000}
OUT=gpurun_out/cg_entry_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_py" -- python3 tools/time_cg_entry_points.py $BIG stencil5-csr cusparse-csr ellpack stencil5-ellpack > "$OUT/time_cg_entry_points_$BIG.txt" 2> "$OUT/time_cg_entry_points_$BIG.err"
grep -v "^{" "$OUT/time_cg_entry_points_$BIG.txt"
BIN=cuda-spmv-benchmark_amd/bin
$BIN/generate_matrix $FILEGRID /tmp/stencil_$FILEGRID.mtx > "$OUT/generate.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_bin" -- $BIN/cg_solver /tmp/stencil_$FILEGRID.mtx --mode=stencil5-csr,cusparse-csr,ellpack --device > "$OUT/cg_solver_file_$FILEGRID.txt" 2>&1
if [ -x oracle/_ref/ref_cg_solver ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_ref" -- oracle/_ref/ref_cg_solver /tmp/stencil_$FILEGRID.mtx --mode=stencil5-csr --device > "$OUT/ref_cg_solver_file_$FILEGRID.txt" 2>&1
fi
grep -E "Results for|Converged|Time \(median\)|Sum\(x\)|Norm2" "$OUT"/cg_solver_file_$FILEGRID.txt "$OUT"/ref_cg_solver_file_$FILEGRID.txt 2>/dev/null
$BIN/cg_solver --stencil=$MID --mode=stencil5-csr,cusparse-csr,ellpack --device > "$OUT/cg_solver_stencil_$MID.txt" 2>&1
grep -E "Results for|Converged|Time \(median\)|Sum\(x\)" "$OUT/cg_solver_stencil_$MID.txt"
rm -f /tmp/stencil_$FILEGRID.mtx
