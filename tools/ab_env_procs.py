#!/usr/bin/env python3
"""An environment switch against the default over FRESH processes (where allocations land differs from process to process, so a
switch that changes the layout of allocations has to be judged over several): `tools/compare_operators.py <grid> <modes...>` run
`runs` times per setting, the settings alternated; per operator the median kernel time of every process.
   python tools/ab_env_procs.py <runs> <grid> <ENV=VALUE> [modes...]"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs, grid, switch = int(sys.argv[1]), sys.argv[2], sys.argv[3]
modes = sys.argv[4:] or ["cusparse-csr", "ellpack"]
key, value = switch.split("=", 1)
settings = [("default", {}), (switch, {key: value})]
res = {name: {m: [] for m in modes} for name, _ in settings}
for k in range(runs):
    for name, extra in (settings if k % 2 == 0 else settings[::-1]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_operators.py"), grid] + modes,
                             env=dict(os.environ, SPMV_AMD_COMPARE_CEILINGS="0", **extra), capture_output=True, text=True, timeout=600)
        lines = [l for l in out.stdout.splitlines() if l.startswith("[{")]
        if out.returncode != 0 or not lines:
            print(f"   run failed ({out.returncode}): {out.stderr[-300:]}")
            continue
        for rec in json.loads(lines[-1]):
            res[name][rec["operator"]].append(rec["median_ms"])
print(f"grid {grid}: {runs} fresh processes per setting, alternated; median kernel ms per process (x = 1, 5 warm-ups + 10 launches)")
for m in modes:
    for name, _ in settings:
        v = res[name][m]
        print(f"   {m:18s} {name:24s} " + " ".join(f"{x:.3f}" for x in v) + f"   median {np.median(v):.3f}  min {min(v):.3f}  max {max(v):.3f}")
    a, b = np.median(res["default"][m]), np.median(res[switch][m])
    print(f"   {m:18s} default vs {switch}: {100.0 * (a / b - 1.0):+.2f} % (medians of the process medians)")
