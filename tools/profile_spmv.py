"""Small driver for rocprofv3: N launches of one operator's run_device on the synthetic stencil.
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/profile_spmv.py stencil5-csr 20000 5"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("spmv_amd_binding", os.path.join(ROOT, "cuda-spmv-benchmark_amd", "binding.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)

mode = sys.argv[1] if len(sys.argv) > 1 else "stencil5-csr"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
op = B.Operator(mode)
if len(sys.argv) > 4:
    op.select_variant(sys.argv[4])
assert op.init_synthetic(n) == 0
dx, dy = B.DeviceVector(n * n, fill=1.0), B.DeviceVector(n * n, fill=0.0)
ms = op.time_device(dx, dy, reps)
print(mode, op.variant(), "grid", n, "ms:", " ".join(f"{v:.3f}" for v in ms))
dx.free(), dy.free(), op.free()
