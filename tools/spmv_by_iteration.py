#!/usr/bin/env python3
"""The in-loop SpMV launch by launch: which iterations (= which direction buffer as x, which sweep direction) are the slow ones
when a process lands in the slow mode? `procs` fresh processes, each: slab, 3 warm-ups, 5 solves, per-iteration median.
   python tools/spmv_by_iteration.py [procs=6] [grid=20000]"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_binding
    n = int(sys.argv[2])
    os.dup2(2, 1)
    B = load_binding()
    B.lib()
    B.require_gpu()
    # an operator instance first, as bench.py has one before its slab (it shapes where the slab's allocations land)
    if os.environ.get("WITH_OPERATOR", "1") == "1":
        op = B.Operator("stencil5-csr")
        op.init_synthetic(n)
        dx, dy = B.DeviceVector(n * n, fill=1.0), B.DeviceVector(n * n, fill=0.0)
        op.time_device(dx, dy, 5)
        dx.free(), dy.free(), op.free()
    slab = B.CgSlab.stencil5(n)
    for _ in range(3):
        slab.solve()
    per, tot = [], []
    for _ in range(5):
        st = slab.solve()
        per.append(slab.spmv_launch_ms())
        tot.append(st.time_total_ms)
    _, tl = slab.timeline_solve()
    with open(sys.argv[3], "w") as f:
        json.dump({"per": np.median(np.array(per), axis=0).tolist(), "solve": float(np.median(tot)), "r": tl["update_r_us"], "p": tl["direction_update_us"]}, f)
    slab.destroy()
    sys.exit(0)

procs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
print(f"grid {n}: in-loop SpMV per iteration (ms, median of 5 solves), one fresh process per line")
for k in range(procs):
    path = f"/tmp/spmv_by_iteration_{os.getpid()}_{k}.json"
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), path], capture_output=True, text=True, timeout=600)
    if p.returncode != 0 or not os.path.exists(path):
        print(f"   process {k} failed: {p.stderr[-300:]}")
        continue
    rec = json.load(open(path))
    os.remove(path)
    print(f"   solve {rec['solve']:8.3f} ms  r {rec['r']:.0f} us  p {rec['p']:.0f} us  SpMV mean {np.mean(rec['per']):.3f}: " + " ".join(f"{v:.3f}" for v in rec["per"]), flush=True)
