set -u
export TMPDIR=/tmp
OUT=gpurun_out/march_sq
rm -rf $OUT; mkdir -p $OUT
for R in 1 2 4; do
  SPMV_AMD_ROWLDS_ROWS=$R rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/r$R -- python3 tools/profile_spmv.py stencil5-csr 20000 5 > $OUT/r$R.log 2>&1
  SPMV_AMD_ROWLDS_ROWS=$R rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/l$R -- python3 tools/profile_spmv.py stencil5-csr 20000 5 > $OUT/l$R.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
print("STENCIL5 operator launch at 20 000^2, SQ counters per launch (averages over 5 launches), row-lds (1 row per wave) vs the march (2 / 4)")
names=None
for R in (1,2,4):
    v=collections.defaultdict(list)
    for f in glob.glob(f"{out}/r{R}/**/*counter_collection.csv", recursive=True)+glob.glob(f"{out}/l{R}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stencil5_rowlds" in r["Kernel_Name"]:
                v[r["Counter_Name"]].append(float(r["Counter_Value"]))
                vg=r["VGPR_Count"]; lds=r["LDS_Block_Size"]
    a={k:sum(x)/len(x) for k,x in v.items()}
    if names is None:
        names=sorted(a)
        print(f"{'rows/wave':>9s} {'VGPRs':>6s} {'LDS B':>6s} " + " ".join(f"{n[3:] if n.startswith('SQ_') else n:>18s}" for n in names))
    print(f"{R:9d} {vg:>6s} {lds:>6s} " + " ".join(f"{a.get(n,0):18.4g}" for n in names))
PY
