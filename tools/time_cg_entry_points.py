"""The reference's single-GPU CG entry points timed beside the slab solver on one grid, same box, same process:
   cg_solve_device (reference src/solvers/cg_solver.cu:436-706) per operator, the harness rule of
   src/main/cg_solver.cu:154-178 (b = 1, x0 = 0, 3 warm-up solves, 10 timed, >2 sigma dropped, median of
   CGStats.time_total_ms), and spmv_amd_cg_slab_* (what bench.py times) with the same rule.
   python tools/time_cg_entry_points.py 20000 stencil5-csr cusparse-csr ellpack stencil5-ellpack [--host]
Synthetic matrices are generated in HBM (a 20 000^2 .mtx is 48 GB of text); cg_solve_device reads only mat->rows."""
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("spmv_amd_binding", os.path.join(ROOT, "cuda-spmv-benchmark_amd", "binding.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)

args = [a for a in sys.argv[1:] if not a.startswith("--")]
host_too = "--host" in sys.argv
n = int(args[0])
modes = args[1:] or ["stencil5-csr", "cusparse-csr", "ellpack", "stencil5-ellpack"]
rows = n * n
B.require_gpu()


def rule(ms):
    t = np.asarray(ms, dtype=np.float64)
    keep = t[np.abs(t - t.mean()) <= 2.0 * t.std()]
    return float(np.median(keep)), float(t.min()), float(t.max())


out = {"grid": n, "rows": rows, "rule": "3 warm-up solves + 10 timed, >2 sigma dropped, median of time_total_ms", "entries": []}

slab = B.CgSlab.stencil5(n)
for _ in range(3):
    st = slab.solve()
ms = []
for _ in range(10):
    st = slab.solve()
    ms.append(st.time_total_ms)
med, lo, hi = rule(ms)
slab_hist = slab.history()
rec = {"entry": "spmv_amd_cg_slab_solve (cg_solve_mgpu_partitioned, 1 rank)", "operator": slab.variant(), "median_ms": med, "min_ms": lo,
       "max_ms": hi, "iterations": st.iterations, "converged": int(st.converged), "residual": st.residual_norm}
out["entries"].append(rec)
print(f"{'slab solver':34s} {rec['operator']:26s} {med:9.3f} ms  [{lo:.3f}, {hi:.3f}]  {st.iterations} iterations", flush=True)
slab.destroy()

fake = B.HostMatrix(np.empty(0, dtype=B.ENTRY_DTYPE), rows, rows, n)  # cg_solve_device reads mat->rows only
b, x0 = np.ones(rows), np.zeros(rows)
for mode in modes:
    op = B.Operator(mode)
    assert op.init_synthetic(n) == 0
    for device in ([True, False] if host_too else [True]):
        runs = 10 if device else 2
        for _ in range(3 if device else 1):
            x, hist, st = B.cg_solve(op, fake, b, x0, device=device)
        ms, spmv_ms, blas_ms, red_ms = [], [], [], []
        for _ in range(runs):
            x, hist, st = B.cg_solve(op, fake, b, x0, device=device)
            ms.append(st.time_total_ms)
            spmv_ms.append(st.time_spmv_ms)
        med, lo, hi = rule(ms)
        k = min(len(hist), len(slab_hist))
        rel = float(np.max(np.abs(hist[:k] - slab_hist[:k]) / slab_hist[:k])) if k else None
        rec = {"entry": "cg_solve_device" if device else "cg_solve (host interface)", "operator": mode, "variant": op.variant(), "median_ms": med,
               "min_ms": lo, "max_ms": hi, "iterations": st.iterations, "converged": int(st.converged), "residual": st.residual_norm,
               "time_spmv_ms": float(np.median(spmv_ms)), "history_max_rel_diff_vs_slab": rel,
               "history_bit_identical_to_slab": bool(k == len(slab_hist) == len(hist) and np.array_equal(hist, slab_hist)),
               "solution_sum": st.solution_sum, "solution_norm": st.solution_norm}
        out["entries"].append(rec)
        print(f"{rec['entry']:34s} {mode + ' / ' + rec['variant']:26s} {med:9.3f} ms  [{lo:.3f}, {hi:.3f}]  {st.iterations} iterations  "
              f"history vs slab: {rel:.2e}{' (bit-identical)' if rec['history_bit_identical_to_slab'] else ''}", flush=True)
    op.free()
print(json.dumps(out))
