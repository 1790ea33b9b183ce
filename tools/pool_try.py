#!/usr/bin/env python3
"""The 20 000^2 slab solver in a fresh process, nothing allocated before it: solve time, in-loop SpMV, stage times, the in-loop
SpMV launch by launch. Used for the class-pool A/B (profiles/r04_class_pool_*.txt):
   for p in 1 0; do SPMV_AMD_CLASS_POOL=$p SPMV_AMD_PLACEMENT_VERBOSE=1 python tools/pool_try.py; done"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_binding  # noqa: E402

B = load_binding()
B.lib()
B.require_gpu()
s = B.CgSlab.stencil5(int(sys.argv[1]) if len(sys.argv) > 1 else 20000)
print(s.placement())
for _ in range(3):
    st = s.solve()
ms, sp = [], []
for _ in range(6):
    st = s.solve()
    ms.append(st.time_total_ms)
    sp.append(st.time_spmv_ms / st.iterations)
_, tl = s.timeline_solve()
print("solve", np.median(ms), "spmv", np.median(sp), "r", tl["update_r_us"], "p", tl["direction_update_us"], "flush", tl["final_x_flush_us"], "init", tl["initial_residual_us"], st.iterations)
print(s.spmv_launch_ms())
s.destroy()
