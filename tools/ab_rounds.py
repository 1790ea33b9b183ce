#!/usr/bin/env python3
"""Same-lease A/B of the single-GPU 20 000 x 20 000 solve between the libraries of earlier rounds and HEAD (BENCH_r03
read 129.9 it/s against 133.1 in r02 -- regression or box noise?). The round libraries are built from their commits into
cuda-spmv-benchmark_amd/lib_ab/libspmv_amd_<tag>.so (git worktree + make; binaries are not committed). Every measurement is a
fresh child process (one library per process, raw ctypes on the entry points all rounds share), the libraries alternated
`rounds` times, so that the process-to-process spread of one box (+-1.3 %, where the allocations land) is sampled for each.
   python tools/ab_rounds.py [grid=20000] [rounds=4] [tags=r02,r03,head]"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cuda-spmv-benchmark_amd")


def lib_of(tag):
    return os.path.join(PKG, "lib", "libspmv_amd.so") if tag == "head" else os.path.join(PKG, "lib_ab", f"libspmv_amd_{tag}.so")


def child(path, n, solves, out_path):
    class Cfg(C.Structure):  # CGConfigMultiGPU, reference include/solvers/cg_solver_mgpu.h:38-46
        _fields_ = [("max_iters", C.c_int), ("tolerance", C.c_double), ("verbose", C.c_int), ("enable_detailed_timers", C.c_int)]

    sys.path.insert(0, PKG)
    import importlib.util
    spec = importlib.util.spec_from_file_location("b", os.path.join(PKG, "binding.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)  # struct layouts only; lib() is never called
    os.dup2(2, 1)
    L = C.CDLL(path)
    L.spmv_amd_cg_slab_create_stencil5.restype = C.c_void_p
    L.spmv_amd_cg_slab_create_stencil5.argtypes = [C.c_int, C.c_void_p]
    L.spmv_amd_cg_slab_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.spmv_amd_cg_slab_destroy.argtypes = [C.c_void_p]
    L.spmv_amd_cg_slab_history.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.spmv_amd_set_device(0)
    s = L.spmv_amd_cg_slab_create_stencil5(n, None)
    cfg, st = Cfg(1000, 1e-6, 0, 0), b.CGStatsMultiGPU()
    for _ in range(3):
        L.spmv_amd_cg_slab_solve(s, C.byref(cfg), C.byref(st))
    ms, spmv = [], []
    for _ in range(solves):
        L.spmv_amd_cg_slab_solve(s, C.byref(cfg), C.byref(st))
        ms.append(st.time_total_ms)
        spmv.append(st.time_spmv_ms / max(st.iterations, 1))
    hist = (C.c_double * 64)()
    count = L.spmv_amd_cg_slab_history(s, hist, 64)
    L.spmv_amd_cg_slab_destroy(s)
    with open(out_path, "w") as f:
        f.write(json.dumps({"ms": ms, "spmv_ms": spmv, "iterations": st.iterations, "history": [float(v).hex() for v in hist[:count]]}) + "\n")


if len(sys.argv) > 1 and sys.argv[1] == "--child":
    child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    sys.exit(0)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tags = (sys.argv[3] if len(sys.argv) > 3 else "r02,r03,head").split(",")
tags = [t for t in tags if os.path.exists(lib_of(t))]
res = {t: [] for t in tags}
hist = {}
for rnd in range(rounds):
    for t in (tags if rnd % 2 == 0 else tags[::-1]):
        out_path = os.path.join(tempfile.gettempdir(), f"ab_rounds_{os.getpid()}_{rnd}_{t}.json")
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib_of(t), str(n), "10", out_path],
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=600)
        data = open(out_path).read() if os.path.exists(out_path) else ""
        if os.path.exists(out_path):
            os.remove(out_path)
        if p.returncode != 0 or not data:
            print(f"   {t}: child failed with {p.returncode}: {p.stderr[-300:]}")
            continue
        rec = json.loads(data)
        res[t].append((float(np.median(rec["ms"])), float(np.median(rec["spmv_ms"])), rec["iterations"]))
        hist[t] = rec["history"]
print(f"grid {n}: one fresh process per measurement (3 warm-ups + 10 solves, medians), libraries alternated {rounds} times on one box")
for t in tags:
    if not res[t]:
        continue
    ms = [v[0] for v in res[t]]
    sp = [v[1] for v in res[t]]
    print(f"   {t:5s} solve ms " + " ".join(f"{v:8.3f}" for v in ms) + f"   median {np.median(ms):8.3f}  best {min(ms):8.3f}   in-loop SpMV ms " + " ".join(f"{v:.3f}" for v in sp)
          + f"   iterations {res[t][0][2]}")
if "head" in res and res["head"]:
    for t in tags:
        if t != "head" and res[t]:
            print(f"   head vs {t}: {100.0 * (np.median([v[0] for v in res['head']]) / np.median([v[0] for v in res[t]]) - 1.0):+.2f} % per solve (medians of the process medians); "
                  f"histories bit-identical: {hist[t] == hist['head']}")
