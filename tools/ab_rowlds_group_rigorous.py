#!/usr/bin/env python3
"""Row-lds tile -> XCD run length (SPMV_AMD_ROWLDS_GROUP = consecutive 128-column tiles per XCD of every run of 8 * group) swept in
ONE process with the settings alternated (the operator re-plans on spmv_amd_operator_select_variant; the matrix stays), standalone
operator launches with x = 1. The rule in kernels.hpp (xcd_run_group: one grid row + ~1100 columns per run) came out of round 2's
separate-process sweeps; this checks it with everything else held still.
   python tools/ab_rowlds_group_rigorous.py [grid=15000] [groups ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
default = max(1, min(64, (n + 1100 + 8 * 128 - 1) // (8 * 128))) if n >= 8000 else 4
groups = [int(a) for a in sys.argv[2:]] or sorted({4, 8, default - 4, default - 2, default - 1, default, default + 1, default + 2, default + 4, 2 * default, 32} - {0, -1, -2, -3})
groups = [g for g in groups if 1 <= g <= 64]
B = load_binding()
B.lib()
B.require_gpu()
rows = n * n
op = B.Operator("stencil5-csr")
assert op.init_synthetic(n) == 0
dx, dy = B.DeviceVector(rows, fill=1.0), B.DeviceVector(rows, fill=0.0)
res = {g: [] for g in groups}
for rnd in range(4):
    for g in (groups if rnd % 2 == 0 else groups[::-1]):
        os.environ["SPMV_AMD_ROWLDS_GROUP"] = str(g)
        op.select_variant("row-lds")  # re-plans with the new group
        op.time_device(dx, dy, 4)
        res[g].append(float(np.median(op.time_device(dx, dy, 10))))
print(f"grid {n}: row-lds, tiles per XCD and run (rule: {default}); median ms of 10 launches, four alternating rounds")
best = min(groups, key=lambda g: np.mean(res[g]))
for g in groups:
    cols = 8 * g * 128
    print(f"   group {g:3d}  (run = {cols:6d} columns = {cols / n:5.2f} grid rows)  " + "  ".join(f"{t:.4f}" for t in res[g]) + f"   mean {np.mean(res[g]):.4f}"
          + ("   <- rule" if g == default else "") + ("   <- best" if g == best else ""))
dx.free(), dy.free(), op.free()
