#!/usr/bin/env python3
"""Process-to-process spread of the default benchmark on one box: `runs` consecutive
`python bench.py --no-cpu-baseline --no-scaling-probe` processes (five runs within +-0.5 %?).
   python tools/run_spread.py [runs=5] [steps=10] [ENV=VALUE ...: a second setting, alternated with the default]"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = sys.argv[2] if len(sys.argv) > 2 else "10"
other = dict(a.split("=", 1) for a in sys.argv[3:])
settings = [("default", {})] + ([(" ".join(sys.argv[3:]), other)] if other else [])
res = {name: [] for name, _ in settings}
for k in range(runs):
    for name, extra in (settings if k % 2 == 0 else settings[::-1]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-scaling-probe", "--no-compare", "--steps", steps],
                             env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            print(f"   run failed ({out.returncode}): {out.stderr[-300:]}")
            continue
        line = json.loads(lines[-1])
        res[name].append(line)
        for l in out.stderr.splitlines():
            if l.startswith("[cg-slab] class pool") or l.startswith("[cg-slab] coefficient"):
                print("      " + l[:260])
        st = line["roofline"].get("stages", {})
        print(f"   {name:24s} {line['value']:.2f} it/s  {line['ms_per_step']:.3f} ms/solve  in-loop SpMV {line['roofline']['avg_launch_ms']:.4f} ms "
              f"(frac {line['roofline']['frac']:.3f}, of best stream {line['roofline'].get('frac_of_best_stream', 0):.3f})  standalone SpMV {line['spmv']['median_ms']:.4f} ms  "
              f"r update {st.get('update_r_us', {}).get('us', 0):.1f} us  direction update {st.get('direction_update_us', {}).get('us', 0):.1f} us  "
              f"flush {st.get('final_x_flush_us', {}).get('us', 0):.0f} us  initial residual {line['breakdown']['per_rank'][0].get('initial_residual_us', 0):.0f} us  "
              f"placement {line.get('placement')} / spmv {line['spmv'].get('output_placement')}", flush=True)
for name, _ in settings:
    v = np.array([l["ms_per_step"] for l in res[name]])
    if len(v):
        print(f"{name}: {len(v)} runs, ms per solve min {v.min():.3f} median {np.median(v):.3f} max {v.max():.3f}  spread +-{100.0 * (v.max() - v.min()) / (v.max() + v.min()):.2f} %")
