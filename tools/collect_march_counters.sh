#!/bin/bash
# Runs on the GPU box: FETCH_SIZE / WRITE_SIZE / TCC hit+miss of the STENCIL5 operator launch at one grid for the row-lds
# kernel and its march variants (SPMV_AMD_ROWLDS_ROWS = 1 / 2 / 4), separate --pmc passes, the program directly after `--`.
# usage: tools/collect_march_counters.sh [grid=20000]
set -u
GRID=${1:-20000}
OUT=gpurun_out/march_counters_$GRID
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
for R in 1 2 4; do
  export SPMV_AMD_ROWLDS_ROWS=$R
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo "$C" | tr ' ' '_')
    rocprofv3 --pmc $C --output-format csv -d "$OUT/r${R}_$N" -- python3 tools/profile_spmv.py stencil5-csr $GRID 5 > "$OUT/r${R}_$N.log" 2>&1
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r${R}_stats" -- python3 tools/profile_spmv.py stencil5-csr $GRID 8 > "$OUT/r${R}_stats.log" 2>&1
done
unset SPMV_AMD_ROWLDS_ROWS
python3 - "$OUT" "$GRID" <<'PY'
import csv, glob, sys
out, n = sys.argv[1], int(sys.argv[2])
alg = 8 * (5 * n * n - 4 * n) + 16 * n * n
print(f"grid {n}: STENCIL5 operator launch, algorithmic {alg / 1e9:.2f} GB; FETCH_SIZE x 1024 x 2 (gfx950), WRITE_SIZE x 1024, per launch")
print(f"{'rows/wave':>9s} {'avg ms':>8s} {'fetch GB':>9s} {'write GB':>9s} {'traffic/alg':>11s} {'L2 hit':>7s}  kernel")
for R in (1, 2, 4):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
        for f in glob.glob(f"{out}/r{R}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "stencil5_rowlds" in r["Kernel_Name"]:
                    vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                    kname = r["Kernel_Name"]
    ms = None
    for f in glob.glob(f"{out}/r{R}_stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stencil5_rowlds" in r["Name"]:
                ms = float(r["AverageNs"]) / 1e6
    avg = {k: sum(v) / len(v) for k, v in vals.items()}
    fetch, write = avg.get("FETCH_SIZE", float("nan")) * 2048, avg.get("WRITE_SIZE", float("nan")) * 1024
    hit, miss = avg.get("TCC_HIT_sum", 0.0), avg.get("TCC_MISS_sum", 0.0)
    short = kname[kname.find("stencil5_rowlds"):].split("(")[0]
    print(f"{R:9d} {ms or 0:8.3f} {fetch / 1e9:9.2f} {write / 1e9:9.2f} {(fetch + write) / alg:11.3f} {hit / (hit + miss) if hit + miss else 0:7.3f}  {short}")
PY
