#!/usr/bin/env python3
"""What would the CG loop gain if every vector lay in the class of address regions that suits its kernels
(profiles/r04_spmv_regions.txt)? A lab arena of 44 vector slots (141 GB, allocated first), its slots sorted into classes with
the r-update pair kernel; then ONE slab whose Ap, r, coefficient stream and direction buffers are pointed at slots of chosen
classes (spmv_amd_cg_slab_lab_rebind), solved to tolerance; per layout the solve, the in-loop SpMV and the stage times.
   python tools/loop_regions.py [grid=20000]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rows = n * n
os.environ["SPMV_AMD_PLACEMENT_CANDIDATES"] = "1"
B = load_binding()
L = B.lib()
B.require_gpu()
slots = 44
pitch = (rows * 8 + (2 << 20) - 1) // (2 << 20) * (2 << 20) + 4096
arena = L.spmv_amd_device_alloc(C.c_size_t(slots * pitch))
assert arena, "lab arena allocation failed"
slab = B.CgSlab.stencil5(n)
L.spmv_amd_cg_slab_lab_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
L.spmv_amd_cg_slab_lab_rebind.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]


def at(slot):
    return arena + slot * pitch


def pair(a, b, reps=3):
    ms = (C.c_float * reps)()
    assert L.spmv_amd_cg_slab_lab_pair(slab.h, C.c_void_p(at(a)), C.c_void_p(at(b)), rows, reps, ms) == 0
    return float(np.median(ms[:]))


L.spmv_amd_device_fill_f64(C.c_void_p(arena), C.c_size_t(slots * pitch // 8), C.c_double(1.0))
t0 = [pair(m, 0) for m in range(1, slots)]
fast = min(t0)
reps_of, members, loose = [0], {0: [0]}, []
for m in range(1, slots):
    placed = False
    for rep in reps_of:
        t = t0[m - 1] if rep == 0 else pair(m, rep)
        if t < 1.025 * fast:
            members[rep].append(m)
            placed = True
            break
    if not placed:
        if all((t0[m - 1] if rep == 0 else pair(m, rep)) > 1.05 * fast for rep in reps_of):
            reps_of.append(m)
            members[m] = [m]
        else:
            loose.append(m)
classes = sorted(members.values(), key=len, reverse=True)
print(f"grid {n}: lab arena of {slots} slots; classes by size: " + " | ".join(str(c) for c in classes) + f" ; straddling: {loose}")


def measure(tag):
    for _ in range(2):
        slab.solve()
    ms, sp = [], []
    for _ in range(5):
        st = slab.solve()
        ms.append(st.time_total_ms)
        sp.append(st.time_spmv_ms / st.iterations)
    _, tl = slab.timeline_solve()
    print(f"   {tag:58s} solve {np.median(ms):8.3f} ms  in-loop SpMV {np.median(sp):.4f} ms ({22.39936 / np.median(sp) / 8:.3f} of 8 TB/s)  r update {tl['update_r_us']:.1f} us  "
          f"direction update {tl['direction_update_us']:.1f} us  flush {tl['final_x_flush_us']:.0f} us  initial residual {tl['initial_residual_us']:.0f} us", flush=True)
    return slab.history().copy()


def runs(cls, need):
    return [s for s in cls if all(s + k in cls for k in range(need))]


def bind(v_slot, ap, r, ring):
    arr = (C.c_void_p * len(ring))(*[C.c_void_p(at(k)) for k in ring])
    rc = L.spmv_amd_cg_slab_lab_rebind(slab.h, C.c_void_p(at(ap)), C.c_void_p(at(r)), None if v_slot is None else C.c_void_p(at(v_slot)), arr, len(ring))
    assert rc == 0


h0 = measure("as created (vector arena; coefficients allocated separately)")
names = "ABC"
cls_of = {}
for i, c in enumerate(classes[:3]):
    for m in c:
        cls_of[m] = names[i]
ring_cls = classes[0]
if len(ring_cls) < 16:
    print("   the largest class holds fewer than 16 slots: nothing to lay out")
else:
    for v_name, v_cls in zip(names, classes[:3]):
        v_runs = runs(v_cls, 5)
        if v_cls is ring_cls:
            v_runs = [v for v in v_runs if len([s for s in ring_cls if not (v <= s < v + 5)]) >= 16]
        if not v_runs:
            continue
        v = v_runs[-1]
        ring = [s for s in ring_cls if not (v <= s < v + 5)][:16]
        spare = {nm: [s for s in c if s not in ring and not (v <= s < v + 5)] for nm, c in zip(names, classes[:3])}
        for ap_name in names:
            for r_name in names:
                if ap_name not in spare or r_name not in spare:
                    continue
                pool_ap = spare[ap_name]
                pool_r = [s for s in spare[r_name] if not pool_ap or s != pool_ap[0]]
                if not pool_ap or not pool_r:
                    continue
                bind(v, pool_ap[0], pool_r[0], ring)
                h = measure(f"coefficients {v_name}, directions A, Ap {ap_name}, r {r_name}  (slots V {v} Ap {pool_ap[0]} r {pool_r[0]})")
                assert np.array_equal(h, h0)
slab.destroy()
L.spmv_amd_device_free(C.c_void_p(arena))
