"""SURVEY 8f-2, the generic-matrix path at size: synthetic NON-stencil matrices of >= 1e7 rows through the CSR operator
(every variant) and ELLPACK, with GB/s by the format's algorithmic bytes (x counted once) -- the perf counterpart of the
correctness tests on tests/matrices.py, after the ideas of reference tests/helpers/matrix_fixtures.cpp:234-370
(banded, unbalanced rows, seeded random).
   banded9   : 9 diagonals (offsets -4..4), random values: a banded matrix without any grid structure (grid_size = -1)
   skewed    : row lengths 1-8 except one row in 1000 with 2 000-20 000 entries, random columns: mean ~ 16, max 20 000
   uniform<K>: K random columns per row (10 * rows / K rows)
   python tools/generic_matrix_perf.py [rows=10000000] [cases...]
Matrices are built as CSR-ordered COO entries with numpy and go through HostMatrix -> build_csr_struct -> upload like
any file; sequential-order variants (stream, row-scalar) must agree bit for bit, the others to 1e-12."""
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("spmv_amd_binding", os.path.join(ROOT, "cuda-spmv-benchmark_amd", "binding.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def banded9(rows, rng):
    off = np.arange(-4, 5)
    r = np.repeat(np.arange(rows, dtype=np.int64), 9)
    c = r + np.tile(off, rows)
    keep = (c >= 0) & (c < rows)
    return r[keep], c[keep], rows


def skewed(rows, rng):
    lens = rng.integers(1, 9, size=rows)
    long_rows = rng.choice(rows, size=max(rows // 1000, 1), replace=False)
    lens[long_rows] = rng.integers(2000, 20001, size=len(long_rows))
    r = np.repeat(np.arange(rows, dtype=np.int64), lens)
    c = rng.integers(0, rows, size=len(r), dtype=np.int64)  # duplicates inside a long row are legal COO (two entries, summed by SpMV)
    return r, c, rows


def uniform(k):
    def make(rows, rng):
        r = np.repeat(np.arange(rows, dtype=np.int64), k)
        c = rng.integers(0, rows, size=len(r), dtype=np.int64)
        return r, c, rows
    return make


CASES = {"banded9": banded9, "skewed": skewed}
args = sys.argv[1:]
rows = int(args[0]) if args and args[0].isdigit() else 10_000_000
for a in args:  # uniform<K>: K random columns in every row, 10 * rows / K rows (the same 10 * rows entries whatever K)
    if a.startswith("uniform") and a[7:].isdigit():
        CASES[a] = uniform(int(a[7:]))
cases = [a for a in args if a in CASES] or ["banded9", "skewed"]
B.require_gpu()
out = []
for name in cases:
    rng = np.random.default_rng(42)  # mt19937 seed 42 upstream; any fixed seed serves
    n_rows = rows if not name.startswith("uniform") else 10 * rows // int(name[7:])
    t0 = time.perf_counter()
    r, c, n = CASES[name](n_rows, rng)
    nnz = len(r)
    e = np.empty(nnz, dtype=B.ENTRY_DTYPE)
    e["row"], e["col"] = r, c
    e["value"] = rng.uniform(-2.0, 2.0, nnz)
    del r, c
    lens = np.bincount(e["row"], minlength=n)
    m = B.HostMatrix(e, n, n, -1)
    print(f"== {name}: {n} rows, {nnz} nnz, row length mean {nnz / n:.2f} max {int(lens.max())} (built in {time.perf_counter() - t0:.1f} s)", flush=True)
    alg_csr = 12 * nnz + 4 * (n + 1) + 16 * n
    x = rng.standard_normal(n)
    dx = B.DeviceVector.from_host(x)
    dy = B.DeviceVector(n, fill=0.0)
    B.lib().spmv_amd_reset_host_matrices()
    op = B.Operator("cusparse-csr")
    ref_y = None
    variants = [None, "stream", "adaptive", "row-scalar", "wavefront"]
    first = True
    for var in variants:
        op.select_variant(var)
        if first:
            t0 = time.perf_counter()
            assert op.init(m) == 0
            print(f"   init (build_csr_struct + upload): {time.perf_counter() - t0:.1f} s", flush=True)
            first = False
        op.time_device(dx, dy, 3)
        ms = op.time_device(dx, dy, 10)
        med = float(np.median(ms))
        y = dy.to_host()
        if ref_y is None and var in ("stream", "row-scalar"):
            ref_y = y
        agree = None
        if ref_y is not None:
            agree = "bit-identical" if np.array_equal(y, ref_y) else f"{float(np.max(np.abs(y - ref_y)) / np.max(np.abs(ref_y))):.1e}"
        rec = {"case": name, "operator": "cusparse-csr", "asked": var or "auto", "variant": op.variant(), "rows": n, "nnz": nnz, "median_ms": med,
               "algorithmic_gbs": alg_csr / med / 1e6, "frac_of_8TBs": alg_csr / med / 1e6 / 8000, "vs_sequential_sum": agree}
        out.append(rec)
        print(f"   csr {rec['asked']:10s} -> {rec['variant']:16s} {med:9.3f} ms  {rec['algorithmic_gbs']:8.1f} GB/s ({rec['frac_of_8TBs']:.3f} of 8 TB/s)  y vs sequential: {agree}", flush=True)
    op.select_variant(None)
    op.free()
    width = int(lens.max())
    if width <= 64:  # ELLPACK pads every row to the longest: only sensible for near-uniform rows
        ell = B.Operator("ellpack")
        assert ell.init(m) == 0
        ell.time_device(dx, dy, 3)
        med = float(np.median(ell.time_device(dx, dy, 10)))
        y = dy.to_host()
        alg_ell = n * width * 12 + 16 * n
        rec = {"case": name, "operator": "ellpack", "variant": ell.variant(), "rows": n, "nnz": nnz, "width": width, "median_ms": med,
               "algorithmic_gbs": alg_ell / med / 1e6, "frac_of_8TBs": alg_ell / med / 1e6 / 8000,
               "vs_sequential_sum": "bit-identical" if np.array_equal(y, ref_y) else f"{float(np.max(np.abs(y - ref_y)) / np.max(np.abs(ref_y))):.1e}"}
        out.append(rec)
        print(f"   ellpack (width {width})  {rec['variant']:16s} {med:9.3f} ms  {rec['algorithmic_gbs']:8.1f} GB/s ({rec['frac_of_8TBs']:.3f} of 8 TB/s)  y vs CSR: {rec['vs_sequential_sum']}", flush=True)
        ell.free()
    dx.free(), dy.free()
    B.lib().spmv_amd_reset_host_matrices()
    del e, m
print(json.dumps(out))
