#!/usr/bin/env python3
"""The reference's multi-GPU entry point called the way its own main calls it -- cg_solve_mgpu_partitioned(op, mat, b, x, config, stats)
with the matrix, b and x in HOST memory (reference src/solvers/cg_solver_mgpu_partitioned.cu:259-413 builds the local CSR and uploads
it, :831-833 copies x back) -- timed on the wall around the whole call, beside the solver's own timed region (stats.time_total_ms,
the reference's region and bench.py's). The difference is set-up the boundary imposes: host CSR build, PCIe upload of the slab
(12 B per non-zero + 16 B per row for b and x0), download of x. bench.py's `value` never includes it (inputs resident in HBM).
   python tools/host_entry_rate.py [grid=10000] [calls=3]
The host matrix is the library's own in-HBM stencil (spmv_amd_init_stencil5_synthetic) downloaded and turned into MatrixData entries."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows, nnz = n * n, 5 * n * n - 4 * n
B = load_binding()
B.lib()
B.require_gpu()

t0 = time.perf_counter()
op = B.Operator("stencil5-csr")
assert op.init_synthetic(n) == 0
rp, ci, va = op.download_csr(rows, nnz)
op.free()
entries = np.empty(nnz, dtype=B.ENTRY_DTYPE)
entries["row"] = np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp))
entries["col"] = ci
entries["value"] = va
del rp, ci, va
m = B.HostMatrix(entries, rows, rows, n)
print(f"grid {n}: {rows} rows, {nnz} non-zeros; host matrix ({entries.nbytes / 1e9:.1f} GB of entries) prepared in {time.perf_counter() - t0:.1f} s (not part of any figure below)")

b = np.ones(rows)
wall, inner, its = [], [], []
for k in range(calls):
    x = np.zeros(rows)
    cfg, st = B.CGConfig(1000, 1e-6, 0, 0), B.CGStatsMultiGPU()
    t = time.perf_counter()
    rc = B.lib().spmv_amd_cg_solve_mgpu_partitioned(m.ptr, b.ctypes.data, x.ctypes.data, C.byref(cfg), C.byref(st))
    dt = time.perf_counter() - t
    assert rc == 0 and st.converged == 1
    wall.append(dt * 1e3)
    inner.append(st.time_total_ms)
    its.append(st.iterations)
    print(f"   call {k}: {st.iterations} iterations, wall {dt * 1e3:9.1f} ms around the call, {st.time_total_ms:8.2f} ms in the solver's timed region "
          f"(residual {st.residual_norm:.3e}, sum(x) {st.solution_sum:.6e})")
moved = 12.0 * nnz + 4.0 * (rows + 1) + 3 * 8.0 * rows  # CSR + b + x0 up, x down
w, t_in = float(np.median(wall)), float(np.median(inner))
print(f"median: {its[-1] / (w / 1e3):8.2f} iterations/s with the host interface's set-up inside the clock ({w:.1f} ms per call; {moved / 1e9:.2f} GB "
      f"cross the link per call), {its[-1] / (t_in / 1e3):8.2f} iterations/s in the timed region ({t_in:.2f} ms) -- bench.py's metric")

# where the set-up goes: the same call taken apart through the slab interface (spmv_amd_cg_slab_*), each step on the wall
def step(label, fn):
    t = time.perf_counter()
    out = fn()
    B.lib().spmv_amd_device_synchronize()
    print(f"      {label:58s} {(time.perf_counter() - t) * 1e3:9.1f} ms")
    return out


print("   the same call, step by step:")
slab = step("create: slice the host CSR, upload, verify, place, tune", lambda: B.CgSlab.from_matrix(m))
print("         of which (ms): " + ", ".join(f"{k} {v:.1f}" for k, v in slab.setup_ms().items()) + f"; placement record {slab.placement()}")
step("set_vectors: b and x0 to the GPU", lambda: slab.set_vectors(b, np.zeros(rows)))
step("solve (14 iterations)", lambda: slab.solve())
step("gather: x to the host", lambda: slab.gather())
step("destroy", lambda: slab.destroy())
slab = step("for comparison: create with the matrix generated in HBM", lambda: B.CgSlab.stencil5(n))
print("         of which (ms): " + ", ".join(f"{k} {v:.1f}" for k, v in slab.setup_ms().items()) + f"; placement record {slab.placement()}")
slab.destroy()
