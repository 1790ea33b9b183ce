#!/usr/bin/env python3
"""Per-iteration timeline from a rocprofv3 --kernel-trace CSV of tools/probe_slab.py: for the LAST solve in the trace,
the busy time of each kernel kind, the idle time between consecutive kernels on the GPU, and the wall span.
usage: python tools/trace_gaps.py <..._kernel_trace.csv> [iterations=14]"""
import csv
import collections
import sys

path = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = list(csv.DictReader(open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# a solve starts with its initial SpMV: find the last kernel named cg_init_residual and back up one SpMV
idx = [i for i, e in enumerate(ev) if "cg_init_residual" in e[2]]
start = idx[-1] - 1
while start > 0 and "stencil5" in ev[start - 1][2]:
    start -= 1
solve = ev[start:]
t_begin, t_end = solve[0][0], max(e[1] for e in solve)
busy = collections.OrderedDict()
covered, cursor, gaps = 0, solve[0][0], []
for s, e, name in solve:
    short = name.replace("(anonymous namespace)::", "").replace("spmv_amd::", "").replace("void ", "").split("(")[0]
    busy.setdefault(short, [0, 0])
    busy[short][0] += e - s
    busy[short][1] += 1
    if s > cursor:
        gaps.append((s - cursor, short))
    if e > cursor:
        covered += e - max(s, cursor)
        cursor = e
print(f"last solve: {len(solve)} kernels, span {(t_end - t_begin) / 1e3:.1f} us, GPU busy (union) {covered / 1e3:.1f} us, "
      f"idle {(t_end - t_begin - covered) / 1e3:.1f} us = {(t_end - t_begin - covered) / 1e3 / iters:.1f} us per iteration")
for k, (ns, cnt) in sorted(busy.items(), key=lambda kv: -kv[1][0]):
    print(f"  {ns / 1e3:10.1f} us  {cnt:4d} x {ns / cnt / 1e3:8.2f} us  {k[:100]}")
by_next = collections.defaultdict(lambda: [0, 0])
for g, nxt in gaps:
    by_next[nxt][0] += g
    by_next[nxt][1] += 1
print("idle time in front of each kernel kind:")
for k, (ns, cnt) in sorted(by_next.items(), key=lambda kv: -kv[1][0]):
    print(f"  {ns / 1e3:10.1f} us  {cnt:4d} x {ns / cnt / 1e3:8.2f} us  before {k[:90]}")
