#!/usr/bin/env python3
"""Late bulk (csrc/cg_slab.hip): the direction update of an iteration is enqueued as a LEAD piece, then the host reads the status
record and enqueues the rest only if the loop goes on. The lead piece must outlast the host's read + launch; a slow host (a
throttled container, a noisy neighbour) leaves the GPU idle behind a short one. One slab (LAB build), settings switched between
solves: late bulk off, the default rule (the protocol only in iterations that may converge, judged by the known residual), and
the protocol in every iteration with lead pieces of 2^22 ... 2^27 rows.
   python tools/ab_lead_rows.py [grid=20000] [rounds=6]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = load_binding().use_lab()
B.lib()
B.require_gpu()
slab = B.CgSlab.stencil5(n)
settings = ([("late bulk off", 0, 1 << 25), ("default rule, 2^25", 2, 1 << 25)] +
            [(f"every iteration, 2^{k}", 1, 1 << k) for k in (22, 23, 24, 25, 26, 27)])
ms = {name: [] for name, _, _ in settings}
tl = {}
for _ in range(2):
    slab.solve()
hist = None
for rnd in range(rounds):
    for name, late, lead in (settings if rnd % 2 == 0 else settings[::-1]):
        slab.set_option("late_bulk", late)
        slab.set_option("lead_rows", lead)
        st = slab.solve()
        ms[name].append(st.time_total_ms)
        h = slab.history().copy()
        assert hist is None or np.array_equal(h, hist)
        hist = h
for name, late, lead in settings:
    slab.set_option("late_bulk", late)
    slab.set_option("lead_rows", lead)
    tl[name] = slab.timeline_solve()[1]
print(f"grid {n} ({slab.n_local} rows), one slab, {rounds} rounds, settings alternated; histories bit-identical")
for name, _, _ in settings:
    v = np.array(ms[name])
    t = tl[name]
    print(f"   {name:26s} solve ms median {np.median(v):8.3f}  min {v.min():8.3f}  max {v.max():8.3f}   timeline: direction update {t['direction_update_us']:7.1f} us, "
          f"gap {t['gap_before_next_iteration_us']:5.1f} us, iteration {t['iteration_us']:7.1f} us")
slab.destroy()
