#!/usr/bin/env python3
"""Row-lds against its march variants with everything else held still: ONE process, the variants alternated.
  standalone: one operator, spmv_amd_operator_select_variant between row-lds / row-lds-march2 / row-lds-march4, five rounds of
              (5 warm-ups + 10 launches) each, x = 1;
  in the loop: two slab solvers alive at once (SPMV_AMD_ROWLDS_ROWS read at creation), solves alternated A B A B ..., the SpMV
              average from the HIP events around every in-loop launch and the solve time. Pairs (1, 2) and (1, 4).
The separate-process comparison of profiles/r03_rowlds_march.txt differs by less than two processes of the SAME variant do
(+-1.3 %, r03_cpu_baseline_full_size_and_variance.txt); this is the measurement that can tell.
   python tools/ab_march_rigorous.py [grid=20000] [ring slots for the in-loop pairs=4]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ring = sys.argv[2] if len(sys.argv) > 2 else "4"
B = load_binding()
B.lib()
B.require_gpu()
rows = n * n

op = B.Operator("stencil5-csr")
assert op.init_synthetic(n) == 0
dx, dy = B.DeviceVector(rows, fill=1.0), B.DeviceVector(rows, fill=0.0)
names = ("row-lds", "row-lds-march2", "row-lds-march4")
res = {v: [] for v in names}
for rnd in range(5):
    for v in (names if rnd % 2 == 0 else names[::-1]):
        op.select_variant(v)
        op.time_device(dx, dy, 5)
        res[v].append(float(np.median(op.time_device(dx, dy, 10))))
print(f"grid {n}, standalone operator launches (median of 10, five alternating rounds):")
for v in names:
    print(f"   {v:16s} " + "  ".join(f"{t:.4f}" for t in res[v]) + f"   mean {np.mean(res[v]):.4f} ms")
op.select_variant(None)
dx.free(), dy.free(), op.free()

os.environ["SPMV_AMD_P_RING"] = ring  # two slabs must fit side by side: a short direction ring (same SpMV, same loop)
for other in (2, 4):
    slabs = {}
    for r in (1, other):
        os.environ["SPMV_AMD_ROWLDS_ROWS"] = str(r)
        slabs[r] = B.CgSlab.stencil5(n)
    for s in slabs.values():
        s.solve()
    spmv = {r: [] for r in slabs}
    total = {r: [] for r in slabs}
    hist = {}
    for rnd in range(6):
        for r in ((1, other) if rnd % 2 == 0 else (other, 1)):
            st = slabs[r].solve()
            spmv[r].append(st.time_spmv_ms / st.iterations)
            total[r].append(st.time_total_ms)
            hist[r] = slabs[r].history()
    print(f"grid {n}, in the CG loop, ring of {ring} direction buffers, rows per wave 1 vs {other} (six alternating solves each); histories bit-identical: {bool(np.array_equal(hist[1], hist[other]))}")
    for r in (1, other):
        print(f"   rows/wave {r}: SpMV ms per launch " + " ".join(f"{t:.4f}" for t in spmv[r]) + f"  mean {np.mean(spmv[r]):.4f};  solve ms " + " ".join(f"{t:.2f}" for t in total[r]) + f"  mean {np.mean(total[r]):.2f}")
    for s in slabs.values():
        s.destroy()
