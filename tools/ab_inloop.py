#!/usr/bin/env python3
"""Why is the in-loop STENCIL5 launch (stencil5_rowlds_kernel<true>, 3.84 ms at 20 000^2) slower than the standalone
operator launch (<false>, 3.75 ms)? One factor at a time, same process, same matrix bytes:
   kernel   <false> (operator) | <true> (slab: + p.Ap partial per wave)
   x data   all ones | standard-normal values
   sweep    forward | alternating (the CG loop's ping-pong)
   context  back-to-back launches | inside the solve (between the r update and the direction update)
usage: python tools/ab_inloop.py [grid] [reps]      (writes one JSON object to stdout)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
os.dup2(2, 1) if False else None
B = load_binding()
B.lib()
B.require_gpu()
rows = n * n
rng = np.random.default_rng(0)
xr = rng.standard_normal(rows)
out = {"grid": n, "reps": reps, "unit": "ms, median of reps after 5 warm-ups"}


def med(v):
    return float(np.median(v))


op = B.Operator("stencil5-csr")
assert op.init_synthetic(n) == 0
dy = B.DeviceVector(rows, fill=0.0)
for name, dx in (("ones", B.DeviceVector(rows, fill=1.0)), ("normal", B.DeviceVector.from_host(xr)), ("zeros", B.DeviceVector(rows, fill=0.0))):
    op.time_device(dx, dy, 5)
    out[f"operator<false>_x_{name}"] = med(op.time_device(dx, dy, reps))
    dx.free()
dy.free()
op.free()

slab = B.CgSlab.stencil5(n)
st = slab.solve()
out["in_loop_avg_pingpong"] = st.time_spmv_ms / st.iterations
slab.time_spmv(5)
out["slab<true>_p_ones_forward"] = med(slab.time_spmv(reps))     # p = ring[0] = r0 = b = 1 after a solve
y = slab.spmv(xr)                                                 # uploads xr into p (no dot), one launch
slab.time_spmv(5)
out["slab<true>_p_normal_forward"] = med(slab.time_spmv(reps))
slab.destroy()

os.environ["SPMV_AMD_PINGPONG"] = "0"
slab = B.CgSlab.stencil5(n)
slab.solve()
st = slab.solve()
out["in_loop_avg_forward_only"] = st.time_spmv_ms / st.iterations
out["solve_ms_forward_only"] = st.time_total_ms
slab.destroy()
os.environ["SPMV_AMD_PINGPONG"] = "1"
slab = B.CgSlab.stencil5(n)
slab.solve()
st = slab.solve()
out["in_loop_avg_pingpong_2nd_solve"] = st.time_spmv_ms / st.iterations
out["solve_ms_pingpong"] = st.time_total_ms
slab.destroy()
sys.stderr.flush()
print(json.dumps(out, indent=1))
