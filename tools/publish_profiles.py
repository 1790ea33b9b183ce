#!/usr/bin/env python3
"""Copies what tools/collect_profiles.sh left under gpurun_out/profiles_<tag>/ into profiles/ (tracked)
and derives profiles/hbm_traffic.json for the dominant kernel of bench.py.
   python tools/publish_profiles.py r01"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}")
dst = os.path.join(ROOT, "profiles")


def newest(pattern):
    # gpurun merges every collection into the same local directory (one file set per process id): take the most
    # recently written file that holds our kernels
    best, best_time = None, -1.0
    for f in glob.glob(pattern):
        if not any("spmv_amd" in line for line in open(f)):
            continue
        t = os.path.getmtime(f)
        if t > best_time:
            best, best_time = f, t
    return best


stats = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = newest(os.path.join(d, "*", "*counter_collection.csv"))
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_bench_{os.path.basename(d)}.csv"))
shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, f"{tag}_bench_summary.json"))
if os.path.exists(os.path.join(src, "stream_probe.txt")):
    shutil.copy(os.path.join(src, "stream_probe.txt"), os.path.join(dst, f"{tag}_stream_probe.txt"))
for line in open(os.path.join(src, "bench_default.json")):
    if line.startswith("{"):
        open(os.path.join(dst, f"{tag}_bench_default.json"), "w").write(line)
for line in open(os.path.join(src, "bench_stats_run.log")):
    if line.startswith("{"):
        open(os.path.join(dst, f"{tag}_bench_under_rocprof_stdout.txt"), "w").write(line)

summary = json.load(open(os.path.join(src, "summary.json")))
bench = json.load(open(os.path.join(dst, f"{tag}_bench_default.json")))
kernel = bench["roofline"]["kernel"].split(" (")[0]
if "<true" in kernel:  # line written before the kernel's template head changed from <bool kDot, .> to <int kMode, .>
    kernel = kernel.replace("<true, ", "<1, ")
# the in-loop instantiation (kDot = true): its PMC rows are the SpMV launches of the CG loop
pmc = next(v for k, v in summary["pmc_avg_per_launch"].items() if kernel in k)
source = os.path.join(ROOT, "cuda-spmv-benchmark_amd", "csrc", "spmv_kernels.hip")
fetch = pmc["FETCH_SIZE"] * 1024 * 2
write = pmc["WRITE_SIZE"] * 1024
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
rec = {
    "grid": bench["config"]["grid"], "n_gpus": 1, "kernel": kernel,
    # bench.py reports `traffic` only while this hash equals the hash of the kernel source it runs
    "kernel_source_sha256": hashlib.sha256(open(source, "rb").read()).hexdigest(), "kernel_source": "cuda-spmv-benchmark_amd/csrc/spmv_kernels.hip",
    "bytes_per_launch": fetch + write,
    "fetch_bytes_corrected": fetch, "write_bytes": write, "FETCH_SIZE_KiB": pmc["FETCH_SIZE"], "WRITE_SIZE_KiB": pmc["WRITE_SIZE"],
    "TCC_HIT_sum": pmc.get("TCC_HIT_sum"), "TCC_MISS_sum": pmc.get("TCC_MISS_sum"), "algorithmic_bytes_per_launch": alg,
    "ratio_to_algorithmic": (fetch + write) / alg,
    "source": f"profiles/{tag}_bench_pmc_FETCH_SIZE.csv, {tag}_bench_pmc_WRITE_SIZE.csv (separate rocprofv3 --pmc passes over "
              "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-probe --no-spmv --no-ceiling`, averages over the in-loop launches)",
    "note": "FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-B fabric requests at 64 B; the factor is confirmed in the same passes by "
            "cg_update_r_kernel, whose FETCH_SIZE x 2 equals its 16 B/row of reads) + WRITE_SIZE x 1024. These are L2-to-fabric "
            "bytes: Infinity-Cache hits are counted. The excess over the algorithmic bytes is x[row-n] / x[row+n] missing the "
            "XCD-private L2 (tiles of adjacent grid rows land on different XCDs) and being served by the 256 MiB Infinity Cache, "
            "not by HBM; mappings that pin columns to an XCD remove it and run slower (DESIGN.md 3.1).",
}
json.dump(rec, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
