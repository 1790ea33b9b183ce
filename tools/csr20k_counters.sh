set -u
export TMPDIR=/tmp
OUT=gpurun_out/csr20k
rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo "$C" | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/ours_$N -- python3 tools/profile_spmv.py cusparse-csr 20000 5 > $OUT/ours_$N.log 2>&1
  rocprofv3 --pmc $C --output-format csv -d $OUT/rocsparse_$N -- tools/bin/rocsparse_compare 20000 > $OUT/rocsparse_$N.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ours_stats -- python3 tools/profile_spmv.py cusparse-csr 20000 8 > $OUT/ours_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocsparse_stats -- tools/bin/rocsparse_compare 20000 > $OUT/rocsparse_stats.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
for who in ("ours","rocsparse"):
    vals=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{who}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            vals[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur={}
    for f in glob.glob(f"{out}/{who}_stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Name"][:95]]=(int(r["Calls"]), float(r["AverageNs"])/1e6)
    print("==", who)
    for k,(c,ms) in sorted(dur.items(), key=lambda kv:-kv[1][1]*kv[1][0])[:6]:
        print(f"   {c:4d} x {ms:8.3f} ms  {k}")
    for k,v in vals.items():
        a={c:sum(x)/len(x) for c,x in v.items()}
        if a.get("FETCH_SIZE",0)*2048 < 1e9: continue
        hit,miss=a.get("TCC_HIT_sum",0),a.get("TCC_MISS_sum",0)
        print(f"   fetch {a.get('FETCH_SIZE',0)*2048/1e9:7.2f} GB  write {a.get('WRITE_SIZE',0)*1024/1e9:6.2f} GB  L2 hit {hit/(hit+miss) if hit+miss else 0:.3f}  launches {len(v.get('FETCH_SIZE',[]))}  {k}")
PY
