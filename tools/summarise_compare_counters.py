"""Per-kernel table for one tools/collect_compare_counters.sh output directory: average duration (kernel-trace stats),
L2-to-fabric bytes per launch (FETCH_SIZE x 1024 x 2 on gfx950 -- 128-byte requests are tallied at 64 B,
MI355X_MICROARCH.md -- plus WRITE_SIZE x 1024), the format's algorithmic bytes and their ratio, TCC hit share.
   python tools/summarise_compare_counters.py gpurun_out/compare_r03_10000 10000"""
import collections
import csv
import glob
import json
import os
import sys

out, n = sys.argv[1], int(sys.argv[2])
rows, nnz = n * n, 5 * n * n - 4 * n
# kernel-name fragment -> (operator, algorithmic bytes per launch)
ALG = {
    "stencil5_rowlds_kernel": ("stencil5-csr", 8 * nnz + 16 * rows),
    "stencil5_rowdirect_kernel": ("stencil5-csr", 8 * nnz + 16 * rows),
    "csr_stream_kernel": ("cusparse-csr", 12 * nnz + 4 * (rows + 1) + 16 * rows),
    "csr_row_scalar_kernel": ("cusparse-csr", 12 * nnz + 4 * (rows + 1) + 16 * rows),
    "ell_spmv_kernel": ("ellpack", rows * 5 * 12 + 16 * rows),
    "ell_stencil5_kernel": ("stencil5-ellpack", rows * 5 * 8 + 16 * rows),
}


def newest(pattern):
    best, best_t = None, -1.0
    for f in glob.glob(pattern, recursive=True):
        t = os.path.getmtime(f)
        if t > best_t and any("spmv_amd" in line for line in open(f)):
            best, best_t = f, t
    return best


dur = {}
f = newest(os.path.join(out, "stats", "**", "*kernel_stats.csv"))
if f:
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(out, "pmc_*")):
    f = newest(os.path.join(d, "**", "*counter_collection.csv"))
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))

table = []
for name, counters in sorted(pmc.items()):
    key = next((k for k in ALG if k in name), None)
    if key is None:
        continue
    op, alg = ALG[key]
    avg = {c: sum(v) / len(v) for c, v in counters.items()}
    calls, ms = next(((c, m) for k, (c, m) in dur.items() if name in k or k in name), (None, None))
    fetch = avg.get("FETCH_SIZE", float("nan")) * 1024 * 2
    write = avg.get("WRITE_SIZE", float("nan")) * 1024
    hit, miss = avg.get("TCC_HIT_sum"), avg.get("TCC_MISS_sum")
    table.append({"operator": op, "kernel": name[name.find(key):].split("(")[0], "grid": n, "launches": calls, "avg_ms": ms, "algorithmic_bytes": alg,
                  "algorithmic_gbs": alg / ms / 1e6 if ms else None, "frac_of_8TBs": alg / ms / 1e6 / 8000 if ms else None,
                  "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write, "traffic_over_algorithmic": (fetch + write) / alg,
                  "tcc_hit_share": hit / (hit + miss) if hit is not None and miss else None})
print(f"grid {n}: kernel-trace durations + separate --pmc passes (FETCH_SIZE x2 corrected, WRITE_SIZE)")
print(f"{'operator':18s} {'avg ms':>8s} {'alg GB':>8s} {'TB/s':>6s} {'of 8':>6s} {'fetch GB':>9s} {'write GB':>9s} {'traffic/alg':>11s} {'L2 hit':>7s}  kernel")
for t in table:
    print(f"{t['operator']:18s} {t['avg_ms'] or 0:8.3f} {t['algorithmic_bytes'] / 1e9:8.2f} {(t['algorithmic_gbs'] or 0) / 1e3:6.2f} {t['frac_of_8TBs'] or 0:6.3f} "
          f"{t['fetch_bytes'] / 1e9:9.2f} {t['write_bytes'] / 1e9:9.2f} {t['traffic_over_algorithmic']:11.3f} {t['tcc_hit_share'] or 0:7.3f}  {t['kernel']}")
print(json.dumps(table))
