#!/bin/bash
# Runs on the GPU box. Round 5, VERDICT item 7: this library's CSR-stream kernel at 10 000^2 / 15 000^2 / 20 000^2 under rocprofv3 --
# kernel durations and separate --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC hit / miss) -- and rocSPARSE's csr-adaptive at 20 000^2,
# printed PER ROW so that the sizes can be compared: is the 20 000^2 launch short of its 10 000^2 rate because it moves more bytes
# per row (x[row +- n] gathered through the fabric) or because the same bytes take longer?
# usage: tools/csr_counters.sh [grid ...]
set -u
export TMPDIR=/tmp
OUT=gpurun_out/csr_counters
rm -rf $OUT; mkdir -p $OUT
GRIDS=${@:-10000 15000 20000}
for G in $GRIDS; do
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo "$C" | tr ' ' '_')
    rocprofv3 --pmc $C --output-format csv -d $OUT/ours_${G}_$N -- python3 tools/profile_spmv.py cusparse-csr $G 5 > $OUT/ours_${G}_$N.log 2>&1
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ours_${G}_stats -- python3 tools/profile_spmv.py cusparse-csr $G 8 > $OUT/ours_${G}_stats.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo "$C" | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/rocsparse_20000_$N -- tools/bin/rocsparse_compare 20000 > $OUT/rocsparse_20000_$N.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocsparse_20000_stats -- tools/bin/rocsparse_compare 20000 > $OUT/rocsparse_20000_stats.log 2>&1
python3 - $OUT $GRIDS <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
grids = [int(g) for g in sys.argv[2:]]
print("CSR SpMV per row: algorithmic 80.0 B (60 B values + column indices, 4 B row pointer, 8 B x, 8 B y); FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64 B)")
for who, gs in (("ours", grids), ("rocsparse", [20000])):
    for g in gs:
        rows = g * g
        vals = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"{out}/{who}_{g}_*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                vals[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur = {}
        for f in glob.glob(f"{out}/{who}_{g}_stats/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r["Name"][:70]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
        for k, v in vals.items():
            a = {c: sorted(x)[len(x) // 2] for c, x in v.items()}
            fetch = a.get("FETCH_SIZE", 0) * 2048
            if fetch < 40 * rows:
                continue
            hit, miss = a.get("TCC_HIT_sum", 0), a.get("TCC_MISS_sum", 0)
            ms = dur.get(k, (0, 0.0))[1]
            wr = a.get("WRITE_SIZE", 0) * 1024
            print(f"   {who:9s} {g:6d}^2: {ms:7.3f} ms = {ms * 1e9 / rows:6.2f} ps/row | fetched {fetch / rows:6.2f} B/row, written {wr / rows:5.2f} B/row, total {(fetch + wr) / rows:6.2f} "
                  f"= {(fetch + wr) / rows / 80.0:.3f} x algorithmic | L2 hit share {hit / (hit + miss) if hit + miss else 0:.3f} | fabric rate {(fetch + wr) / ms / 1e9 if ms else 0:5.2f} TB/s | {k[:60]}")
PY
