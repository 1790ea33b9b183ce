#!/usr/bin/env python3
"""Round 5, VERDICT item 1: where does a 1/P slab lose against T1/P on ONE GPU?
Every stand-in slab (rank r of P of the headline grid, full RCCL pipeline with the rank as its own neighbour) is solved
 (a) in a FRESH process of its own -- what a rank of a real P-GPU job is -- and
 (b) one after the other inside ONE process, in the order bench.py's scaling probe used in rounds 2-4,
and reported per row: wall and event time per solve, the in-loop SpMV launch by launch (which direction buffer is x, which way
the sweep goes), the same launch standalone, the stage timeline.
   python tools/slab_attribution.py [grid=20000] [solves=5]
   python tools/slab_attribution.py --child <grid> <solves> <P:r,P:r,...>      (internal; also the target of rocprofv3 -- python3 ...)"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(grid, solves, roles):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_binding

    os.environ["SPMV_AMD_SELF_NEIGHBOUR"] = os.environ["SPMV_AMD_FORCE_COLLECTIVES"] = "1"
    json_fd = os.dup(1)
    os.dup2(2, 1)
    B = load_binding().use_lab()  # stand-in slabs and slab options: the LAB build (include/spmv_amd/lab.h)
    B.lib()
    B.require_gpu()
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
    out = []
    for P, r in roles:
        slab = B.CgSlab.stencil5(grid) if P == 1 else B.CgSlab.stencil5_as(grid, r, P, comm)
        slab.set_option("spmv_event_stride", 1)  # every in-loop launch timed, small slabs too
        if P > 1:
            slab.set_option("stop_at", 14)  # the last iteration counts as the converging one, as on the rank of a real job
        for opt in os.environ.get("SLAB_OPTIONS", "").split(","):  # e.g. SLAB_OPTIONS=no_overlap=1 (counter passes: tools/collect_slab_attribution.sh)
            if "=" in opt:
                slab.set_option(opt.split("=")[0], int(opt.split("=")[1]))
        for _ in range(3):
            slab.solve(max_iters=14, tol=0.0)
        B.lib().spmv_amd_device_synchronize()
        wall, event, per = [], [], []
        for _ in range(solves):
            t0 = time.perf_counter()
            st = slab.solve(max_iters=14, tol=0.0)
            B.lib().spmv_amd_device_synchronize()
            wall.append((time.perf_counter() - t0) * 1e3)
            event.append(st.time_total_ms)
            per.append(slab.spmv_launch_ms())
        standalone = slab.time_spmv(12)[2:]
        _, tl = slab.timeline_solve(max_iters=14, tol=0.0)
        rows = slab.n_local
        per_med = np.median(np.array(per), axis=0) if len(per[0]) else np.array([])
        out.append({"P": P, "r": r, "rows": rows, "wall_ms": float(np.median(wall)), "event_ms": float(np.median(event)),
                    "spmv_inloop_us": [round(float(v) * 1e3, 1) for v in per_med], "spmv_standalone_us": round(float(np.median(standalone)) * 1e3, 1),
                    "stage_us": {k: round(v, 1) for k, v in tl.items() if k.endswith("_us")}, "placement": slab.placement()})
        slab.destroy()
    comm.destroy()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def run(grid, solves, roles):
    spec = ",".join(f"{P}:{r}" for P, r in roles)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(grid), str(solves), spec], capture_output=True, text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("[")]
    if p.returncode != 0 or not lines:
        print(f"   child {spec} failed ({p.returncode}): {p.stderr[-400:]}")
        return []
    return json.loads(lines[-1])


def show(rec, full):
    P, rows = rec["P"], rec["rows"]
    per = rec["spmv_inloop_us"]
    ns_row = lambda us: us * 1e3 / rows
    line = (f"   slab {rec['r']} of {P} ({rows:>9d} rows): solve wall {rec['wall_ms']:8.3f} ms, events {rec['event_ms']:8.3f} ms"
            f" | in-loop SpMV mean {np.mean(per):7.1f} us = {ns_row(np.mean(per)) * 1e3:6.2f} ps/row, standalone {rec['spmv_standalone_us']:7.1f} us = "
            f"{ns_row(rec['spmv_standalone_us']) * 1e3:6.2f} ps/row")
    if full is not None:
        line += (f" | vs T1/P: solve {rec['event_ms'] * P / full['event_ms']:.4f}, SpMV {np.mean(per) * P / np.mean(full['spmv_inloop_us']):.4f}")
    print(line)
    print("        in-loop by launch (us): " + " ".join(f"{v:.0f}" for v in per))
    print("        stages (us): " + ", ".join(f"{k[:-3]} {v:.1f}" for k, v in rec["stage_us"].items()))
    sys.stdout.flush()


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    solves = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    roles = [(2, 0), (2, 1), (4, 0), (4, 1), (8, 0), (8, 3)]
    print(f"grid {grid}, 14 iterations per solve, median of {solves} solves after 3 warm-ups; ps/row = picoseconds per matrix row")
    print("(a) every slab in a FRESH process (a rank of a real job is one):")
    full = (run(grid, solves, [(1, 0)]) or [None])[0]
    if full:
        show(full, None)
    fresh = {}
    for role in roles:
        for rec in run(grid, solves, [role]):
            fresh[role] = rec
            show(rec, full)
    print("(b) the same slabs one after the other in ONE process (rounds 2-4's scaling probe):")
    for rec in run(grid, solves, roles):
        show(rec, full)
        f = fresh.get((rec["P"], rec["r"]))
        if f:
            print(f"        same slab, fresh process: solve {f['event_ms']:.3f} ms ({rec['event_ms'] / f['event_ms'] - 1:+.2%} here), "
                  f"in-loop SpMV {np.mean(f['spmv_inloop_us']):.1f} us ({np.mean(rec['spmv_inloop_us']) / np.mean(f['spmv_inloop_us']) - 1:+.2%} here)")
    print("(c) a second fresh process per slab (process-to-process spread):")
    for role in roles:
        for rec in run(grid, solves, [role]):
            f = fresh[role]
            print(f"   slab {role[1]} of {role[0]}: solve {rec['event_ms']:.3f} ms vs {f['event_ms']:.3f}; in-loop SpMV {np.mean(rec['spmv_inloop_us']):.1f} us vs {np.mean(f['spmv_inloop_us']):.1f}")
            sys.stdout.flush()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), [tuple(int(v) for v in t.split(":")) for t in sys.argv[4].split(",")])
    else:
        main()
