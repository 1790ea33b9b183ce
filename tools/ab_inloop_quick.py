#!/usr/bin/env python3
"""operator <false> launch vs slab <true> launch vs in-loop average, 20 000^2 (quick form of tools/ab_inloop.py)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding
B = load_binding(); B.lib(); B.require_gpu()
n = 20000; rows = n * n
op = B.Operator("stencil5-csr"); assert op.init_synthetic(n) == 0
dx, dy = B.DeviceVector(rows, fill=1.0), B.DeviceVector(rows, fill=0.0)
op.time_device(dx, dy, 5); a = float(np.median(op.time_device(dx, dy, 20)))
dx.free(); dy.free(); op.free()
slab = B.CgSlab.stencil5(n); slab.solve(); st = slab.solve()
slab.time_spmv(5); b = float(np.median(slab.time_spmv(20)))
print(json.dumps({"operator<false>": a, "slab<true>_forward": b, "in_loop_avg": st.time_spmv_ms / st.iterations, "solve_ms": st.time_total_ms}))
