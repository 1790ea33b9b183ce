"""BASELINE.json configs 2 and 5: the SpMV operators side by side on one grid, with the
reference's benchmark rule (x = 1, 5 warm-ups, 10 timed launches, >2 sigma dropped, median;
src/main/main.cu:136-187) and both byte counts: the reference's "effective" formula
(spmv_metrics.cu:85-101) and the algorithmic bytes of each format.
Vectors: each operator's own staging pair, as in the reference harness (run_timed).
   python tools/compare_operators.py 10000 stencil5-csr cusparse-csr
   python tools/compare_operators.py 15000 ellpack stencil5-ellpack stencil5-csr cusparse-csr"""
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("spmv_amd_binding", os.path.join(ROOT, "cuda-spmv-benchmark_amd", "binding.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)

n = int(sys.argv[1])
modes = sys.argv[2:] or ["stencil5-csr", "cusparse-csr", "ellpack", "stencil5-ellpack"]
rows, nnz = n * n, 5 * n * n - 4 * n
algorithmic = {
    "stencil5-csr": 8 * nnz + 16 * rows,
    "cusparse-csr": 12 * nnz + 4 * (rows + 1) + 16 * rows,
    "ellpack": rows * 5 * 12 + 16 * rows,
    "stencil5-ellpack": rows * 5 * 8 + 16 * rows,
}
effective = 12 * nnz + 4 * (rows + 1) + 16 * rows
# what this GPU sustains for each format's byte mix with ideal accesses (csrc/stream_ceiling.hip), same row count
MIX = {"stencil5-csr": "stencil5", "stencil5-ellpack": "stencil5", "cusparse-csr": "csr", "ellpack": "ellpack"}
ceilings = {}
if os.environ.get("SPMV_AMD_COMPARE_CEILINGS", "1") != "0":
    for mix in sorted({MIX[m] for m in modes}):
        ms, nbytes = B.stream_ceiling(rows, warmup=3, reps=10, mix=mix)
        ceilings[mix] = nbytes / float(np.median(ms)) / 1e6
        print(f"stream probe, {mix:8s} byte mix: {float(np.median(ms)):8.3f} ms  {ceilings[mix]:8.1f} GB/s ({ceilings[mix] / 8000.0:.3f} of 8 TB/s)")
out = []
for mode in modes:
    op = B.Operator(mode)
    assert op.init_synthetic(n) == 0
    # the operator's own x (= 1) and y: the vectors run_timed's kernel works on, y placed by the operator at init
    op.time_device(None, None, 5)
    ms = op.time_device(None, None, 10)
    placed = op.placement()
    keep = ms[np.abs(ms - ms.mean()) <= 2.0 * ms.std()]
    med = float(np.median(keep))
    rec = {"operator": mode, "variant": op.variant(), "grid": n, "median_ms": med, "gflops": 2.0 * nnz / med / 1e6,
           "effective_gbs_reference_formula": effective / med / 1e6, "algorithmic_bytes": algorithmic[mode],
           "algorithmic_gbs": algorithmic[mode] / med / 1e6, "frac_of_8TBs": algorithmic[mode] / med / 1e6 / 8000.0}
    if MIX[mode] in ceilings:
        rec["mix_probe_gbs"] = ceilings[MIX[mode]]
        rec["frac_of_mix_probe"] = rec["algorithmic_gbs"] / ceilings[MIX[mode]]
    out.append(rec)
    print(f"{mode:18s} {rec['variant']:22s} {med:8.3f} ms  eff {rec['effective_gbs_reference_formula']:8.1f} GB/s  "
          f"alg {rec['algorithmic_gbs']:8.1f} GB/s ({rec['frac_of_8TBs']:.3f} of 8 TB/s"
          + (f", {rec['frac_of_mix_probe']:.3f} of the {MIX[mode]}-mix stream probe of this run)" if "frac_of_mix_probe" in rec else ")"))
    if placed:
        print(f"{'':18s} output placement: {placed[0]} candidates timed at init, first / kept = {placed[1]:.3f}")
    op.free()
print(json.dumps(out))
