#!/usr/bin/env python3
"""Early halo exchange on / off with everything else held still: ONE process, one self-neighbour RCCL communicator, two stand-in
slabs of the same rank alive at once (SPMV_AMD_EARLY_HALO read at slab creation), 14-iteration solves alternated A B A B.
   python tools/ab_early_halo_rigorous.py [grid=20000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
os.environ["SPMV_AMD_SELF_NEIGHBOUR"] = os.environ["SPMV_AMD_FORCE_COLLECTIVES"] = "1"
B = load_binding()
B.lib()
B.require_gpu()
comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
print(f"grid {n}: stand-in slabs, RCCL send / recv with the rank as its own neighbour, ncclAllReduce; two passes (creation order swapped) of six alternating 14-iteration solves per setting")
for P, r in ((8, 3), (8, 0), (4, 1), (2, 1)):
    # two passes with the creation order swapped: WHICH allocation a slab gets is worth up to +-1.3 % by itself
    # (profiles/r03_cpu_baseline_full_size_and_variance.txt), so each setting is measured on both sets of allocations
    ms = {"0": [], "1": []}
    stage = {"0": [], "1": []}
    same = True
    for order in (("0", "1"), ("1", "0")):
        slabs = {}
        for early in order:
            os.environ["SPMV_AMD_EARLY_HALO"] = early
            slabs[early] = B.CgSlab.stencil5_as(n, r, P, comm)
        for s in slabs.values():
            s.solve(max_iters=14, tol=0.0, verbose=0)
            s.solve(max_iters=14, tol=0.0)
        for rnd in range(6):
            for k in (order if rnd % 2 == 0 else order[::-1]):
                ms[k].append(slabs[k].solve(max_iters=14, tol=0.0).time_total_ms)
        for k in slabs:
            stage[k].append(slabs[k].timeline_solve(max_iters=14, tol=0.0)[1])
        same = same and bool(np.array_equal(slabs["0"].history(), slabs["1"].history()))
        for s in slabs.values():
            s.destroy()
    for k in ("0", "1"):
        t = {key: np.mean([d[key] for d in stage[k]]) for key in stage[k][0]}
        print(f"   slab {r} of {P}  early_halo={k}  solve ms (first-created pass | second-created pass) " + " ".join(f"{v:.3f}" for v in ms[k][:6]) + " | " + " ".join(f"{v:.3f}" for v in ms[k][6:])
              + f"  mean {np.mean(ms[k]):.3f}   timeline: iteration {t['iteration_us']:.1f} us, halo wait + boundary rows {t['halo_wait_and_boundary_rows_us']:.1f}, "
              f"direction update {t['direction_update_us']:.1f}, exchange {t['halo_exchange_on_side_stream_us']:.1f}")
    print(f"      early halo vs not: {100.0 * (np.mean(ms['1']) / np.mean(ms['0']) - 1.0):+.2f} % per solve; histories bit-identical: {same}")
comm.destroy()
