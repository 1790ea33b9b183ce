#!/usr/bin/env python3
"""An environment switch against the default for ONE slab shape over fresh processes (set-up decisions -- where the vectors lie --
are made per process): per process the slab is created, warmed up and solved `solves` times; median solve time, in-loop SpMV and
the stage timeline. A slab of a larger job (P > 1) is a stand-in slab on a self-neighbour RCCL rank, solved for 14 iterations.
   python tools/ab_env_slab.py <runs> <ENV=VALUE> [grid=20000] [P=1] [rank=0]"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    n, P, r, path = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_binding
    if P > 1:
        os.environ["SPMV_AMD_SELF_NEIGHBOUR"] = os.environ["SPMV_AMD_FORCE_COLLECTIVES"] = "1"
    os.dup2(2, 1)
    B = load_binding().use_lab()  # stand-in slabs and slab options: the LAB build (include/spmv_amd/lab.h)
    B.lib()
    B.require_gpu()
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id()) if P > 1 else None
    slab = B.CgSlab.stencil5_as(n, r, P, comm) if P > 1 else B.CgSlab.stencil5(n)
    kw = dict(max_iters=14, tol=0.0) if P > 1 else {}
    for _ in range(3):
        slab.solve(**kw)
    ms, sp = [], []
    for _ in range(8):
        st = slab.solve(**kw)
        ms.append(st.time_total_ms)
        sp.append(st.time_spmv_ms / st.iterations)
    _, tl = slab.timeline_solve(**kw)
    json.dump({"solve": float(np.median(ms)), "spmv": float(np.median(sp)), "tl": tl}, open(path, "w"))
    slab.destroy()
    sys.exit(0)

runs, switch = int(sys.argv[1]), sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
P = int(sys.argv[4]) if len(sys.argv) > 4 else 1
r = int(sys.argv[5]) if len(sys.argv) > 5 else 0
key, value = switch.split("=", 1)
settings = [("default", {}), (switch, {key: value})]
res = {name: [] for name, _ in settings}
print(f"grid {n}, slab {r} of {P}: {runs} fresh processes per setting, alternated; median of 8 solves per process")
for k in range(runs):
    for name, extra in (settings if k % 2 == 0 else settings[::-1]):
        path = f"/tmp/ab_env_slab_{os.getpid()}_{k}_{len(extra)}.json"
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), str(P), str(r), path], env=dict(os.environ, **extra),
                           capture_output=True, text=True, timeout=600)
        if p.returncode != 0 or not os.path.exists(path):
            print(f"   {name}: failed ({p.returncode}) {p.stderr[-200:]}")
            continue
        rec = json.load(open(path))
        os.remove(path)
        res[name].append(rec["solve"])
        t = rec["tl"]
        note = [l for l in p.stderr.splitlines() if l.startswith("[cg-slab] Ap | r")]
        print(f"   {name:26s} solve {rec['solve']:8.3f} ms  in-loop SpMV {rec['spmv'] * 1e3:7.1f} us  r update {t['update_r_us']:.1f}  direction update {t['direction_update_us']:.1f}  "
              f"iteration {t['iteration_us']:.1f} us" + ("   " + note[0][9:] if note else ""), flush=True)
for name, _ in settings:
    v = np.array(res[name])
    if len(v):
        print(f"{name}: median {np.median(v):.3f} ms, min {v.min():.3f}, max {v.max():.3f}")
if all(len(res[name]) for name, _ in settings):
    print(f"default vs {switch}: {100.0 * (np.median(res['default']) / np.median(res[switch]) - 1.0):+.2f} %")
