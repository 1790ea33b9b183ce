#!/usr/bin/env python3
"""A/B of one loop option of the slab solver on ONE slab -- the same allocations for both settings, switched between solves
with spmv_amd_cg_slab_set_option -- so that placement (worth up to +-1.3 % between two slabs of one process,
profiles/r03_placement.txt) cannot enter. Solves run to the tolerance (14 iterations on the 20000 grid), alternating A B B A.

   python tools/ab_loop_options.py <option> [grid=20000] [as_world=1 as_rank=0] [rounds=8] [collectives=1]
   options: no_overlap (pipeline against the plain loop shape), late_bulk
A slab of a larger job (as_world > 1) is a stand-in slab on a self-neighbour RCCL rank (LAB build, include/spmv_amd/lab.h)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

option = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 1
r = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 8
collectives = sys.argv[6] if len(sys.argv) > 6 else "1"  # "0": no ncclAllReduce call between the sums and the step (the shape of the mailbox path)

comm = None
if P > 1:
    os.environ["SPMV_AMD_SELF_NEIGHBOUR"] = "1"
    os.environ["SPMV_AMD_FORCE_COLLECTIVES"] = collectives
B = load_binding().use_lab()  # stand-in slabs and slab options: the LAB build (include/spmv_amd/lab.h)
B.lib()
B.require_gpu()
if P > 1:
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
    slab = B.CgSlab.stencil5_as(n, r, P, comm)
    kw = dict(max_iters=14, tol=0.0)  # the slab mirrored at its cuts is another system: fixed iteration count
else:
    slab = B.CgSlab.stencil5(n)
    kw = {}
print(f"grid {n}, slab {r} of {P} ({slab.n_local} rows), option {option}: one slab, settings switched between solves, {rounds} rounds A B B A")
ms = {0: [], 1: []}
hist = {}
tl = {}
for v in (0, 1):
    slab.set_option(option, v)
    slab.solve(**kw)
    slab.solve(**kw)
for rnd in range(rounds):
    for v in ((0, 1) if rnd % 2 == 0 else (1, 0)):
        slab.set_option(option, v)
        st = slab.solve(**kw)
        ms[v].append(st.time_total_ms)
        hist[v] = slab.history().copy()
for v in (0, 1):
    slab.set_option(option, v)
    tl[v] = [slab.timeline_solve(**kw)[1] for _ in range(3)]
for v in (0, 1):
    t = {k: float(np.median([d[k] for d in tl[v]])) for k in tl[v][0]}
    print(f"   {option}={v}: solve ms " + " ".join(f"{x:.3f}" for x in ms[v]) + f"  mean {np.mean(ms[v]):.3f}  median {np.median(ms[v]):.3f}")
    print(f"      timeline (median of 3): iteration {t['iteration_us']:.1f} us, SpMV {t['spmv_interior_us']:.1f}, r update {t['update_r_us']:.1f}, "
          f"direction update {t['direction_update_us']:.1f} (over {int(t['direction_updates'])} launches), gap {t['gap_before_next_iteration_us']:.1f}, "
          f"flush {t['final_x_flush_us']:.1f}, iterations {int(t['iterations'])}")
same = bool(np.array_equal(hist[0], hist[1]))
print(f"   {option} 1 vs 0: {100.0 * (np.mean(ms[1]) / np.mean(ms[0]) - 1.0):+.3f} % per solve (means), {100.0 * (np.median(ms[1]) / np.median(ms[0]) - 1.0):+.3f} % (medians); "
      f"histories bit-identical: {same}")
slab.destroy()
if comm is not None:
    comm.destroy()
