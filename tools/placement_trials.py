#!/usr/bin/env python3
"""Does re-creating the slab (new allocations for the CSR values, the vectors and the direction ring) move the SpMV or the
solve? One process, the 20 000 x 20 000 slab created and destroyed `trials` times; per instance the standalone SpMV (median of 10
launches on the plan's kernel), the in-loop SpMV and the solve (median of 5 solves after 2 warm-ups). VERDICT r03 item 4 asks
whether picking the best of a few allocations at set-up would remove the +-1.3 % between processes of one box.
   python tools/placement_trials.py [grid=20000] [trials=6]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = load_binding()
B.lib()
B.require_gpu()
print(f"grid {n}: slab created and destroyed {trials} times in one process; then two slabs alive at once, alternated")
rows = []
for k in range(trials):
    slab = B.CgSlab.stencil5(n)
    alone = float(np.median(slab.time_spmv(10)))
    for _ in range(2):
        slab.solve()
    ms, sp = [], []
    for _ in range(5):
        st = slab.solve()
        ms.append(st.time_total_ms)
        sp.append(st.time_spmv_ms / st.iterations)
    rows.append((alone, float(np.median(sp)), float(np.median(ms))))
    print(f"   instance {k}: standalone SpMV {alone:.4f} ms   in-loop SpMV {np.median(sp):.4f} ms   solve {np.median(ms):.3f} ms", flush=True)
    slab.destroy()
a = np.array(rows)
for name, col in (("standalone SpMV", 0), ("in-loop SpMV", 1), ("solve", 2)):
    print(f"   {name}: min {a[:, col].min():.4f}  max {a[:, col].max():.4f}  spread {100.0 * (a[:, col].max() / a[:, col].min() - 1.0):.2f} %")
# two instances alive at once: is the difference between them stable over time (a property of the allocations)?
s1, s2 = B.CgSlab.stencil5(n), B.CgSlab.stencil5(n)
for s in (s1, s2):
    s.solve(), s.solve()
t = {1: [], 2: []}
for rnd in range(6):
    for key, s in (((1, s1), (2, s2)) if rnd % 2 == 0 else ((2, s2), (1, s1))):
        t[key].append(s.solve().time_total_ms)
print("   two slabs alive at once, alternated: first  " + " ".join(f"{v:.3f}" for v in t[1]) + f"  median {np.median(t[1]):.3f}")
print("                                        second " + " ".join(f"{v:.3f}" for v in t[2]) + f"  median {np.median(t[2]):.3f}"
      + f"   ({100.0 * (np.median(t[2]) / np.median(t[1]) - 1.0):+.2f} %)")
s1.destroy(), s2.destroy()
