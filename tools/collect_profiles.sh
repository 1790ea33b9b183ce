#!/bin/bash
# Runs on the GPU box (under gpurun). Collects what profiles/ keeps for a round:
#   1. rocprofv3 --kernel-trace --stats of the default command minus its scaling-probe leg,
#      `python3 bench.py --no-scaling-probe` (the probe launches the same kernels on smaller grids and would mix
#      their durations into the per-kernel averages)
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE, TCC hit/miss, SQ) of `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-probe --no-spmv --no-compare --no-ceiling`, --pmc only (separate passes, no trace domains)
#   3. the streaming probe (practical HBM ceilings of the box)
# usage: tools/collect_profiles.sh <tag>
set -u
TAG=${1:-r01}
OUT=gpurun_out/profiles_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-probe --no-spmv --no-compare --no-ceiling"
# the exact default command for the per-kernel stats; the counter passes use a shorter run of the same workload
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-scaling-probe --no-compare --cpu-sample-grid 10000 > "$OUT/bench_stats_run.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  N=$(echo "$C" | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- $BENCH > "$OUT/pmc_$N.log" 2>&1
done
[ -x tools/bin/stream_probe ] && tools/bin/stream_probe > "$OUT/stream_probe.txt" 2>&1
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
summary = {}
f = sorted(glob.glob(out + "/stats/*/*kernel_stats.csv"))
if f:
    rows = list(csv.DictReader(open(f[-1])))
    summary["kernel_stats"] = [{"name": r["Name"], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])} for r in rows]
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary["pmc_avg_per_launch"] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items() if "spmv_amd" in k}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
print("summary written")
PY
