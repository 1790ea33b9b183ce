#!/usr/bin/env python3
"""Round 5, VERDICT item 4: STENCIL5 SpMV across grid sizes (the A100 reference is flat: docs/results_spmv_a100_manual.json:9-56).
Per grid, one fresh process per cell: stencil5-csr (row-lds) TWICE (two processes running the same code differ by up to 3 % through
placement alone: the second column is the yardstick for reading the first), once with round 2's run-length rule forced instead of the
set-up trial (SPMV_AMD_ROWLDS_GROUP), the slot-major stencil-aware ELLPACK kernel and the 48:8 stream probe at that row count.
Reference rule per operator: x = 1, 5 warm-ups, 10 launches, > 2 sigma dropped, median (src/main/main.cu:158-187).
   python tools/size_sweep.py [grid ...]"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n, mode):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_binding
    fd = os.dup(1)
    os.dup2(2, 1)
    B = load_binding()
    B.lib()
    B.require_gpu()
    rows = n * n
    if mode == "probe":
        ms, nbytes = B.stream_ceiling(rows, warmup=3, reps=10, mix="stencil5")
        rec = {"ms": float(np.median(ms)), "bytes": nbytes}
    else:
        op = B.Operator(mode)
        assert op.init_synthetic(n) == 0
        op.time_device(None, None, 5)
        ms = op.time_device(None, None, 10)
        keep = ms[np.abs(ms - ms.mean()) <= 2.0 * ms.std()]
        rec = {"ms": float(np.median(keep)), "bytes": 56 * rows - 32 * n, "variant": op.variant()}
        op.free()
    os.write(fd, (json.dumps(rec) + "\n").encode())


def run(n, mode, env=None):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), mode], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **(env or {})))
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return None
    return json.loads(lines[-1])


def main():
    grids = [int(v) for v in sys.argv[1:]] or [10000, 12500, 15000, 16384, 17500, 20000, 20724]
    print("STENCIL5 SpMV by grid size: ms and fraction of 8 TB/s (algorithmic bytes 8 nnz + 16 rows); n*8 mod 128 = how far consecutive grid rows are off a 128-byte line")
    print(f"{'grid':>6s} {'n*8 mod 128':>11s} | {'row-lds (default)':>24s} | {'row-lds, second process':>26s} | {'row-lds, round-2 run rule':>26s} | {'ELLPACK stencil (slot-major)':>28s} | {'48:8 stream probe':>18s}")
    fr = []
    for n in grids:
        cells = []
        old_rule = max(1, min(64, (n + 1100 + 8 * 128 - 1) // (8 * 128))) if n >= 8000 else 4
        for mode, env in (("stencil5-csr", None), ("stencil5-csr", None), ("stencil5-csr", {"SPMV_AMD_ROWLDS_GROUP": str(old_rule)}), ("stencil5-ellpack", None), ("probe", None)):
            r = run(n, mode, env)
            cells.append(r)
        def fmt(r, w):
            return f"{'failed':>{w}s}" if r is None else f"{r['ms']:9.3f} ms  {r['bytes'] / r['ms'] / 1e6 / 8000.0:6.3f}".rjust(w)
        print(f"{n:6d} {n * 8 % 128:11d} | {fmt(cells[0], 24)} | {fmt(cells[1], 26)} | {fmt(cells[2], 26)} | {fmt(cells[3], 28)} | {fmt(cells[4], 18)}", flush=True)
        both = [c["bytes"] / c["ms"] / 1e6 / 8000.0 for c in cells[:2] if c]
        if both:
            fr.append(float(np.mean(both)))
    if fr:
        print(f"row-lds (default, mean of the two processes): best fraction {max(fr):.3f}, worst {min(fr):.3f} = {min(fr) / max(fr):.3f} of the best")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3])
    else:
        main()
