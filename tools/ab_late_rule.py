#!/usr/bin/env python3
"""Late bulk on a stand-in slab of the headline grid (LAB build): off (0), the lead / status / rest protocol in every iteration (1,
rounds 4-5), only where the known residual says convergence is near (2, round 6's rule). One slab, settings alternated.
   python tools/ab_late_rule.py <P> <r> [rounds=8]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

P, r = int(sys.argv[1]), int(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
os.environ["SPMV_AMD_SELF_NEIGHBOUR"] = os.environ["SPMV_AMD_FORCE_COLLECTIVES"] = "1"
B = load_binding().use_lab()
B.lib()
B.require_gpu()
comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
slab = B.CgSlab.stencil5_as(20000, r, P, comm)
slab.set_option("stop_at", 14)
kw = dict(max_iters=14, tol=0.0)
ms = {0: [], 1: [], 2: []}
for v in (0, 1, 2):
    slab.set_option("late_bulk", v)
    slab.solve(**kw)
for rnd in range(rounds):
    for v in ((0, 1, 2) if rnd % 2 == 0 else (2, 1, 0)):
        slab.set_option("late_bulk", v)
        ms[v].append(slab.solve(**kw).time_total_ms)
print(f"slab {r} of {P} ({slab.n_local} rows), stop_at 14, {rounds} rounds, settings alternated")
for v, name in ((0, "off"), (1, "every iteration"), (2, "by prediction (default)")):
    slab.set_option("late_bulk", v)
    tl = slab.timeline_solve(**kw)[1]
    print(f"   late bulk {name:24s} solve ms median {np.median(ms[v]):7.3f}  min {min(ms[v]):7.3f}   direction update {tl['direction_update_us']:6.1f} us  iteration {tl['iteration_us']:7.1f} us")
slab.destroy()
comm.destroy()
