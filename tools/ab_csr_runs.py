"""CSR-stream / ELLPACK launch geometry on ONE initialised operator (the same allocations for every setting: two
initialisations of one process differ by placement alone, profiles/r04_placement_probe.txt): rows per block x consecutive
blocks per XCD, each setting measured `reps` times in alternation so that drift cannot pass for a gain.
   python tools/ab_csr_runs.py [grid=10000] [mode=cusparse-csr] [reps=4] [rows per block, comma separated: dispatch order only]
Round 4's question: does a block size that DIVIDES the grid row (200 rows: 10 000 = 50 blocks, 15 000 = 75,
20 000 = 100; 1000 of the strip's 1024 entries), with an XCD run of one grid row + ~1100 columns as the row-lds kernel uses,
keep x[row +- n] in the XCD's own L2?"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("spmv_amd_binding", os.path.join(ROOT, "cuda-spmv-benchmark_amd", "binding.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
mode = sys.argv[2] if len(sys.argv) > 2 else "cusparse-csr"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows_only = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else None  # e.g. 176,192,200,204: dispatch order only
rows = n * n
if mode == "cusparse-csr":
    settings = [(176, 1)]  # the default: strip filled to ~90 %, dispatch order
    for rpb in (176, 200):
        per_row = n / rpb
        for k in (1, 2):
            settings.append((rpb, max(2, round((k * n + 1100) / (8 * rpb)))))
        settings.append((rpb, 8))
    settings.append((200, 1))
    settings.append((200, int(np.ceil(n / 200 / 8))))  # a run of 8 * g blocks = exactly one grid row band per XCD pass
else:
    settings = [(256, 0)] + [(256, g) for g in (1, 4, 8, 16, max(2, round((n + 1100) / (8 * 256))))]
if rows_only:
    settings = [(r, 1) for r in rows_only]
if os.environ.get("AB_CSR_SETTINGS"):  # "rows:group,rows:group,..."
    settings = [(176, 1)] + [tuple(int(v) for v in item.split(":")) for item in os.environ["AB_CSR_SETTINGS"].split(",")]
settings = list(dict.fromkeys(settings))
dx, dy = B.DeviceVector(rows, fill=1.0), B.DeviceVector(rows, fill=0.0)
op = B.Operator(mode)
assert op.init_synthetic(n) == 0
want = None
results = {s: [] for s in settings}
same = True
for rep in range(reps):
    for s in (settings if rep % 2 == 0 else settings[::-1]):
        rpb, g = s
        if mode == "cusparse-csr":
            os.environ["SPMV_AMD_CSR_STREAM_ROWS"] = str(rpb)
        if g > 0:
            os.environ["SPMV_AMD_XCD_GROUP"] = str(g)
        else:
            os.environ.pop("SPMV_AMD_XCD_GROUP", None)
        op.select_variant(None)
        op.time_device(dx, dy, 2)
        results[s].append(float(np.median(op.time_device(dx, dy, 10))))
        if rep == 0:
            y = dy.to_host()
            if want is None:
                want = y
            same = same and bool(np.array_equal(y, want))
print(f"grid {n}, {mode} ({op.variant()}): median ms of 10 launches, {reps} alternating repetitions per setting, one operator instance")
base = float(np.median(results[settings[0]]))
for s in settings:
    rpb, g = s
    span = 8 * max(g, 1) * rpb
    med = float(np.median(results[s]))
    print(f"   rows/block {rpb:4d}  xcd_group {g:4d}  (run = {span:8d} rows = {span / n:6.3f} grid rows)   " + "  ".join(f"{v:.3f}" for v in results[s])
          + f"   median {med:.3f}  ({100.0 * (med / base - 1.0):+.2f} % vs the default)")
print(f"   results bit-identical across settings: {same}")
op.free()
dx.free(), dy.free()
