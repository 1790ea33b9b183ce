#!/usr/bin/env python3
"""Soak of the loop's device-flag hand-overs (halo arrival flag, edge rows' ready flag) on stand-in slabs: many solves in a row per
slab shape, both all-reduce shapes, every history compared bit for bit with the one the PLAIN loop shape (no_overlap: everything
on the compute stream, no flag anywhere) produced on the same slab. A lost or late hand-over would show as a different history,
a watchdog exit or a hang. (Rounds 5's reference form, the event-ordered pipeline, left with its switches in round 6.)
   python tools/soak_flags.py [solves=300]
Small grids make the iterations short (tens of microseconds), which is where a race between the streams would have room."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    n, P, r, solves, collectives = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    os.environ["SPMV_AMD_SELF_NEIGHBOUR"] = "1"
    os.environ["SPMV_AMD_FORCE_COLLECTIVES"] = collectives
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_binding
    os.dup2(2, 1)
    B = load_binding().use_lab()  # stand-in slabs and slab options: the LAB build (include/spmv_amd/lab.h)
    B.lib()
    B.require_gpu()
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
    slab = B.CgSlab.stencil5_as(n, r, P, comm)
    kw = dict(max_iters=14, tol=0.0)
    slab.set_option("no_overlap", 1)
    slab.solve(**kw)
    want = slab.history().copy()
    slab.set_option("no_overlap", 0)
    t0 = time.perf_counter()
    bad = 0
    for k in range(solves):
        if k == solves // 2:
            slab.set_option("stop_at", 9)  # second half: converging solves (the last launch writes no direction)
        st = slab.solve(max_iters=14 if k < solves // 2 else 30, tol=0.0)
        h = slab.history()
        ref = want if k < solves // 2 else want[:10]
        if len(h) != len(ref) or not np.array_equal(h, ref):
            bad += 1
    dt = time.perf_counter() - t0
    print(f"grid {n:6d} slab {r} of {P} ({slab.n_local:9d} rows) collectives {collectives}: {solves} solves in {dt:6.2f} s, "
          f"{bad} histories differ from the plain shape's", file=sys.stderr)
    slab.destroy()
    comm.destroy()
    sys.exit(1 if bad else 0)

solves = int(sys.argv[1]) if len(sys.argv) > 1 else 300
failed = 0
for n in (2048, 4096, 10000):
    for P, r in ((2, 0), (4, 1), (8, 3)):
        for collectives in ("1", "0"):
            out = subprocess.run([sys.executable, __file__, "--child", str(n), str(P), str(r), str(solves), collectives], capture_output=True, text=True, timeout=300)
            line = [l for l in out.stderr.splitlines() if l.startswith("grid")]
            print(line[-1] if line else f"grid {n} slab {r} of {P} collectives {collectives}: child ended with {out.returncode}: {out.stderr[-300:]}")
            failed += out.returncode != 0
print(f"{'FAILED: ' + str(failed) + ' shapes' if failed else 'all shapes: every history bit-identical, no watchdog exit'}")
sys.exit(1 if failed else 0)
