#!/bin/bash
# Runs on the GPU box. Round 5, VERDICT item 1: the per-rank stand-in slabs in fresh processes and in one process
# (tools/slab_attribution.py), then the two slabs the verdict names (rank 1 of 2, rank 1 of 4) and the full problem under
# rocprofv3: kernel trace + separate --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC hit / miss) of the in-loop SpMV kernel.
# usage: tools/collect_slab_attribution.sh <tag>
set -u
TAG=${1:-r05}
OUT=gpurun_out/slab_attribution_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
python3 tools/slab_attribution.py 20000 5 > "$OUT/attribution.txt" 2>"$OUT/attribution.err"
cat "$OUT/attribution.txt"
for ROLE in 1:0 2:1 4:1; do
  N=$(echo $ROLE | tr ':' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$N" -- python3 tools/slab_attribution.py --child 20000 3 $ROLE > "$OUT/trace_$N.json" 2>"$OUT/trace_$N.err"
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    CN=$(echo "$C" | tr ' ' '_')
    # Counter passes SERIALISE dispatches: a kernel that spins for a flag another stream's kernel raises (the boundary waves'
    # wait for the halo's arrival flag, the side stream's wait for the edge rows) would spin alone until its bound and end the
    # run. The PLAIN loop shape has no such wait and gives the same bits and the same SpMV kernel: SPMV_AMD_NO_OVERLAP=1, read at
    # creation (the concurrency requirement of the pipeline is stated in include/spmv_amd/api.h). Without the
    # switch the library's creation check would notice by itself -- its pipeline solve's waits give up after 2 s under a
    # serialising tool -- and fall back to the plain order with a line on stderr.
    SPMV_AMD_NO_OVERLAP=1 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${N}_$CN" -- python3 tools/slab_attribution.py --child 20000 2 $ROLE > "$OUT/pmc_${N}_$CN.json" 2>"$OUT/pmc_${N}_$CN.err"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for role in ("1_0", "2_1", "4_1"):
    print(f"== slab {role.replace('_', ' of ')[::-1] if False else role}: rocprofv3 per-kernel averages and counters (per launch)")
    for f in sorted(glob.glob(f"{out}/trace_{role}/*/*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "spmv_amd" in r["Name"] and float(r["Percentage"]) > 0.5:
                print(f"   {r['Name'][:110]:110s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:10.1f} us  {float(r['Percentage']):5.1f} %")
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/pmc_{role}_*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "rowlds" in r["Kernel_Name"]:
                pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in pmc.items():
        print(f"   {k[:100]}: " + ", ".join(f"{c} median {sorted(v)[len(v) // 2]:.4g} (n={len(v)})" for c, v in sorted(d.items())))
PY
