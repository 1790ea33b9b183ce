"""CSR-stream: does an XCD walking a column band pay (as it does for the stencil kernels)? One process, one matrix, the operator
re-initialised with SPMV_AMD_XCD_GROUP = g for runs of 8 * g blocks that cover k grid rows + ~1100 rows (k = 1, 2, 3) and for
dispatch order (g = 1), each setting measured three times in alternation so that drift cannot pass for a gain.
   python tools/ab_csr_runs.py [grid=20000]"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("spmv_amd_binding", os.path.join(ROOT, "cuda-spmv-benchmark_amd", "binding.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rows = n * n
rows_per_block = 176
groups = [1] + [max(2, round((k * n + 1100) / (8 * rows_per_block))) for k in (1, 2, 3)] + [8, 32]
dx, dy = B.DeviceVector(rows, fill=1.0), B.DeviceVector(rows, fill=0.0)
op = B.Operator("cusparse-csr")
results = {g: [] for g in groups}
for rep in range(3):
    for g in groups:
        os.environ["SPMV_AMD_XCD_GROUP"] = str(g)
        assert op.init_synthetic(n) == 0
        op.time_device(dx, dy, 3)
        results[g].append(float(np.median(op.time_device(dx, dy, 10))))
        op.free()
print(f"grid {n}: csr/stream, {rows_per_block} rows per block; median ms of 10 launches, three alternating repetitions per setting")
for g in groups:
    span = 8 * g * rows_per_block
    print(f"   xcd_group {g:4d}  (run = {span:8d} rows = {span / n:6.3f} grid rows)   " + "  ".join(f"{v:.3f}" for v in results[g]) + f"   best {min(results[g]):.3f}")
dx.free(), dy.free()
