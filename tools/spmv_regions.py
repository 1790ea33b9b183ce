#!/usr/bin/env python3
"""The loop's three streaming kernels against the regions of the address space (profiles/r04_arena_probe40.txt: lock-step streams
in different 32 GiB regions are 6.5 % slower). A lab arena of 44 vector slots (141 GB, allocated first), its slots sorted into
classes with the r-update pair kernel (class S = fast with slot 0; the others split again by pairing them with one another),
then the slab's own kernels with each operand taken from a chosen class:
   SpMV (coefficient stream V, x read, y written), r update (Ap read, r read + written), direction update (r, p_in read, p_out written).
   python tools/spmv_regions.py [grid=20000]"""
import ctypes as C
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rows = n * n
os.environ["SPMV_AMD_P_RING"] = "1"  # the slab's own vectors are not used here: keep them small
B = load_binding()
L = B.lib()
B.require_gpu()
slots = 44
pitch = (rows * 8 + (2 << 20) - 1) // (2 << 20) * (2 << 20)
arena = L.spmv_amd_device_alloc(C.c_size_t(slots * pitch))
assert arena, "lab arena allocation failed"
slab = B.CgSlab.stencil5(n)
L.spmv_amd_cg_slab_lab_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
L.spmv_amd_cg_slab_lab_direction.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
L.spmv_amd_cg_slab_lab_spmv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]


def at(slot):
    return C.c_void_p(arena + slot * pitch)


def pair(a, b, reps=3):
    ms = (C.c_float * reps)()
    assert L.spmv_amd_cg_slab_lab_pair(slab.h, at(a), at(b), rows, reps, ms) == 0
    return float(np.median(ms[:]))


def direction(r, p_in, p_out, reps=5):
    ms = (C.c_float * reps)()
    assert L.spmv_amd_cg_slab_lab_direction(slab.h, at(r), at(p_in), at(p_out), rows, reps, ms) == 0
    return float(np.median(ms[:]))


def spmv(a, b, c, reverse=0, reps=5):
    ms = (C.c_float * reps)()
    assert L.spmv_amd_cg_slab_lab_spmv(slab.h, at(a), at(b), at(c), reverse, reps, ms) == 0
    return float(np.median(ms[:]))


L.spmv_amd_device_fill_f64(C.c_void_p(arena), C.c_size_t(slots * pitch // 8), C.c_double(1.0))
# classes: greedy -- a slot joins the first class whose representative it pairs fast with; slots that are fast with nobody
# (they straddle a boundary) stay out
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
        # a new class only if it is cleanly slow with every representative so far
        if all((t0[m - 1] if rep == 0 else pair(m, rep)) > 1.05 * fast for rep in reps_of):
            reps_of.append(m)
            members[m] = [m]
        else:
            loose.append(m)
print(f"grid {n}: lab arena of {slots} slots x {pitch / 1e9:.2f} GB; pair kernel fast mode {fast:.3f} ms; classes (by first member):")
names = {}
for i, rep in enumerate(reps_of):
    names[rep] = "SOQRT"[i]
    print(f"   class {names[rep]}: slots {members[rep]}")
print(f"   straddling a boundary (in no class): {loose}")


def runs_of(cls, need):
    """`need` consecutive slots of one class (for the 5-slot coefficient stream), all starts"""
    m = members[cls]
    return [s for s in m if all(s + k in m for k in range(need))]


cls = {names[r]: r for r in reps_of}
pick = {}
for name, rep in cls.items():
    starts = runs_of(rep, 5)
    singles = [s for s in members[rep]]
    pick[name] = {"V": starts[0] if starts else None, "vec": singles}
print("picked slots: " + ", ".join(f"{k}: V at {v['V']}, vectors {v['vec'][:4]}..{v['vec'][-3:]}" for k, v in pick.items()))


def vec(name, skip, k=0):
    """k-th vector slot of class `name` outside [skip, skip+5) (the coefficient stream's slots)"""
    free = [s for s in pick[name]["vec"] if skip is None or not (skip <= s < skip + 5)]
    return free[-1 - k] if len(free) > k else None  # from the far end of the class: away from the coefficient stream


order = list(cls)
print("SpMV (ms, forward / reverse sweep): classes of (V, x, y)")
for v in order:
    if pick[v]["V"] is None:
        continue
    a = pick[v]["V"]
    for x, y in itertools.product(order, order):
        b, c = vec(x, a, 0), vec(y, a, 1)
        if b is None or c is None:
            continue
        print(f"   V {v} x {x} y {y}   (slots {a:2d} {b:2d} {c:2d})   {spmv(a, b, c, 0):.4f} / {spmv(a, b, c, 1):.4f}", flush=True)
print("r update (ms): classes of (Ap, r)")
for a, r in itertools.product(order, order):
    sa, sr = vec(a, None, 0), vec(r, None, 1)
    if sa is not None and sr is not None:
        print(f"   Ap {a} r {r}   {pair(sa, sr):.4f}")
print("direction update (ms): classes of (r, p_in, p_out)")
for r, pi, po in itertools.product(order, order, order):
    sr, si, so = vec(r, None, 0), vec(pi, None, 1), vec(po, None, 2)
    if None not in (sr, si, so):
        print(f"   r {r} p_in {pi} p_out {po}   {direction(sr, si, so):.4f}", flush=True)
slab.destroy()
L.spmv_amd_device_free(C.c_void_p(arena))
