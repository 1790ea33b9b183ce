"""Does WHERE the vectors lie relative to the matrix arrays matter? The STENCIL5 operator at one grid with x and y shifted inside
over-sized allocations by a range of byte offsets (same process, same matrix): if the +-1.3 % between processes on one box
(profiles/r03_cpu_baseline_full_size_and_variance.txt) came from channel aliasing between the streams, a systematic pattern in the
offsets would show here.   python tools/ab_placement.py [grid=20000]"""
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
slack = 64 << 20  # bytes
op = B.Operator("stencil5-csr")
assert op.init_synthetic(n) == 0
big_x, big_y = B.DeviceVector(rows + slack // 8, fill=1.0), B.DeviceVector(rows + slack // 8, fill=0.0)


class At:
    def __init__(self, v, off):
        self.ptr = v.ptr + off


offsets = [0, 256, 4096, 8192, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 16 << 20, (16 << 20) + 65536, 48 << 20]
op.time_device(At(big_x, 0), At(big_y, 0), 5)
print(f"grid {n}, variant {op.variant()}: median ms of 8 launches per (x offset, y offset); row = x offset, column = y offset")
print(" " * 12 + "".join(f"{o:>11d}" for o in offsets[::2]))
for ox in offsets:
    line = []
    for oy in offsets[::2]:
        ms = op.time_device(At(big_x, ox), At(big_y, oy), 8)
        line.append(float(np.median(ms)))
    print(f"{ox:>11d} " + "".join(f"{v:11.4f}" for v in line), flush=True)
# and the same launch repeated at offset 0, 0 to show the noise floor
rep = [float(np.median(op.time_device(At(big_x, 0), At(big_y, 0), 8))) for _ in range(6)]
print("offset (0, 0) repeated six times:", " ".join(f"{v:.4f}" for v in rep))
big_x.free(), big_y.free(), op.free()
