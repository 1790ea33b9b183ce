#!/usr/bin/env python3
"""What one MI355X can say about the strong scaling of the 20000^2 CG: for P = 2, 4, 8 it solves a square grid with
the row count of one rank's slab (20000^2 / P rows) twice -- plainly, and through the complete multi-rank pipeline
over RCCL with the rank as its own neighbour (SPMV_AMD_SELF_NEIGHBOUR=1: halo send/recv on the side stream under the
interior SpMV, split SpMV launches, both all-reduces issued) -- and projects the P-GPU solve time as
    pipeline time per iteration x 14 iterations + 28 x (assumed inter-device all-reduce latency).
Everything but that latency is measured. Usage: python tools/scaling_projection.py [steps]"""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = sys.argv[1] if len(sys.argv) > 1 else "12"
PORT = [29650]


def bench(grid, pipeline):
    env = dict(os.environ)
    if pipeline:
        PORT[0] += 1
        env.update(SPMV_AMD_BENCH_FORCE_DIST="1", SPMV_AMD_FORCE_COLLECTIVES="1", SPMV_AMD_SELF_NEIGHBOUR="1", RANK="0",
                   WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(PORT[0]))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--grid", str(grid), "--steps", STEPS, "--warmup", "2",
           "--no-cpu-baseline", "--no-spmv"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        raise SystemExit(out.stdout + out.stderr)
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    return rec["ms_per_step"], rec["config"]["iterations_per_solve"]


full_ms, full_it = bench(20000, False)
print(f"P=1  20000^2 ({full_it} iterations): {full_ms:.2f} ms per solve = {full_ms / full_it * 1e3:.0f} us per iteration")
for P in (2, 4, 8):
    grid = int(round(math.sqrt(20000 * 20000 / P)))
    plain_ms, it = bench(grid, False)
    pipe_ms, it2 = bench(grid, True)
    per_it_plain, per_it_pipe = plain_ms / it, pipe_ms / it2
    line = (f"P={P}  slab proxy {grid}^2 ({it} iterations): plain {plain_ms:.2f} ms, pipeline {pipe_ms:.2f} ms "
            f"(+{(per_it_pipe - per_it_plain) * 1e3:.0f} us per iteration); projected {full_it}-iteration solve")
    for lat_us in (10, 20, 30):
        t = per_it_pipe * full_it + 2 * full_it * lat_us / 1e3
        line += f"  {t:.2f} ms -> efficiency {full_ms / (P * t):.3f} @ {lat_us} us/all-reduce;"
    print(line)
