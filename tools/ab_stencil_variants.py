#!/usr/bin/env python3
"""One command for DESIGN.md section 3.1's table: every STENCIL5 kernel variant built into the library, same matrix,
same x = 1, the reference's benchmark rule (5 warm-ups, 10 timed launches, >2 sigma dropped, median), bit-identical
results checked against the default variant.
   python tools/ab_stencil_variants.py [grid=20000]
Variants: row-lds (default; coefficients through a wave-private LDS strip, W/E from LDS, XCD runs of one grid row),
row-lds-march2 / -march4 (round 3: the same tile, a wave walking 2 / 4 consecutive grid rows with x rotating in registers),
row-direct (one thread per row on a 2-D index, strided coefficient loads), column-march (north_star's sketch taken
literally: x north / centre / south rows kept across a march down the grid rows, coefficients through LDS),
wave-tile (128 consecutive rows per wave), row-generic (the reference's own thread-per-row shape with div/mod)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = load_binding()
B.lib()
B.require_gpu()
rows, nnz = n * n, 5 * n * n - 4 * n
alg = 8 * nnz + 16 * rows
op = B.Operator("stencil5-csr")
assert op.init_synthetic(n) == 0
x = np.random.default_rng(1).standard_normal(rows)
dx1, dxr, dy = B.DeviceVector(rows, fill=1.0), B.DeviceVector.from_host(x), B.DeviceVector(rows, fill=0.0)
out, ref = [], None
for variant in ("row-lds", "row-lds-march2", "row-lds-march4", "row-direct", "column-march", "wave-tile", "row-generic"):
    op.select_variant(variant)
    op.time_device(dx1, dy, 5)
    ms = op.time_device(dx1, dy, 10)
    keep = ms[np.abs(ms - ms.mean()) <= 2.0 * ms.std()]
    med = float(np.median(keep))
    assert op.run_device(dxr, dy) == 0
    y = dy.to_host()
    if ref is None:
        ref = y
    same = bool(np.array_equal(y, ref))
    out.append({"variant": op.variant(), "median_ms": med, "algorithmic_gbs": alg / med / 1e6, "frac_of_8TBs": alg / med / 1e6 / 8000.0,
                "bit_identical_to_default": same})
    print(f"{op.variant():24s} {med:8.3f} ms  {alg / med / 1e6:8.1f} GB/s  ({alg / med / 1e6 / 8000.0:.3f} of 8 TB/s)  bit-identical to row-lds: {same}")
op.select_variant(None)
print(json.dumps({"grid": n, "variants": out}))
