#!/usr/bin/env python3
"""Per-iteration timeline from a rocprofv3 --kernel-trace CSV of tools/probe_slab.py: for the LAST solve in the trace,
the busy time of each kernel kind, the idle time between consecutive kernels on the GPU, and the wall span.
usage: python tools/trace_gaps.py <..._kernel_trace.csv> [iterations=14] [solve=-2]
solve: which solve of the trace, counted from the end; -1 is the LAST one -- in tools/probe_slab.py and bench.py that is the extra solve
with stage-boundary events (spmv_amd_cg_slab_set_timeline), whose event records put a barrier packet (~6 us) between ALL stages;
-2 (default) is the last plain solve, where dependent launches follow each other without such packets."""
import csv
import collections
import sys

path = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 14
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
rows = list(csv.DictReader(open(path)))
# idle time is that of the COMPUTE stream: the queue the SpMV launches run on. Side-stream launches (the halo exchange, its arrival
# flag, round 5's one-thread wait in front of it, which polls for most of an iteration) overlap it and are listed separately.
queues = {}
for r in rows:
    if "stencil5" in r["Kernel_Name"] or "csr_" in r["Kernel_Name"] or "ell_" in r["Kernel_Name"]:
        queues[r.get("Queue_Id", "")] = queues.get(r.get("Queue_Id", ""), 0) + 1
compute_queue = max(queues, key=queues.get) if queues else None
side = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if compute_queue is not None and r.get("Queue_Id", "") != compute_queue),
              key=lambda e: e[0])
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if compute_queue is None or r.get("Queue_Id", "") == compute_queue),
            key=lambda e: e[0])
# A solve starts with its initial SpMV. Anchors, in order of preference: the mode-2 SpMV launches (row-lds / row-planes
# with the initial residual fused in: the default), else cg_init_residual (slabs that do not run row-lds) backed up
# over the SpMV launches in front of it, else cg_scalars_init (which directly follows the initial reductions).
def is_init_spmv(name):
    return "stencil5_rowlds_kernel<2" in name or "stencil5_planes_kernel<2" in name


idx = [i for i, e in enumerate(ev) if is_init_spmv(e[2])]
stop = len(ev)
if idx:
    starts = [i for i in idx if i == 0 or not is_init_spmv(ev[i - 1][2])]  # interior + boundary-row launches of a split initial SpMV count once
    k = which if -len(starts) <= which < 0 else -1
    start = starts[k]
    stop = starts[k + 1] if k + 1 < 0 else len(ev)
else:
    idx = [i for i, e in enumerate(ev) if "cg_init_residual" in e[2]]
    if idx:
        start = idx[-1] - 1
        while start > 0 and "stencil5" in ev[start - 1][2]:
            start -= 1
    else:
        idx = [i for i, e in enumerate(ev) if "cg_scalars_init" in e[2]]
        if not idx:
            sys.exit("trace_gaps: no solve found in this trace (none of: mode-2 SpMV, cg_init_residual, cg_scalars_init)")
        start = idx[-1]
        while start > 0 and ("reduce_" in ev[start - 1][2] or "stencil5" in ev[start - 1][2] or "Generic" in ev[start - 1][2]):
            start -= 1
solve = ev[max(start, 0):stop]
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
print(f"solve {which} of the trace: {len(solve)} kernels, span {(t_end - t_begin) / 1e3:.1f} us, GPU busy (union) {covered / 1e3:.1f} us, "
      f"idle {(t_end - t_begin - covered) / 1e3:.1f} us = {(t_end - t_begin - covered) / 1e3 / iters:.1f} us per iteration")
for k, (ns, cnt) in sorted(busy.items(), key=lambda kv: -kv[1][0]):
    print(f"  {ns / 1e3:10.1f} us  {cnt:4d} x {ns / cnt / 1e3:8.2f} us  {k[:100]}")
by_next = collections.defaultdict(lambda: [0, 0])
for g, nxt in gaps:
    by_next[nxt][0] += g
    by_next[nxt][1] += 1
side_busy = collections.OrderedDict()
for s, e, name in side:
    if s < t_begin or s > t_end:
        continue
    short = name.replace("(anonymous namespace)::", "").replace("spmv_amd::", "").replace("void ", "").split("(")[0]
    side_busy.setdefault(short, [0, 0])
    side_busy[short][0] += e - s
    side_busy[short][1] += 1
for k, (ns, cnt) in side_busy.items():
    print(f"  (other queue) {cnt:4d} x {ns / cnt / 1e3:8.2f} us  {k[:100]}")
print("idle time in front of each kernel kind:")
for k, (ns, cnt) in sorted(by_next.items(), key=lambda kv: -kv[1][0]):
    print(f"  {ns / 1e3:10.1f} us  {cnt:4d} x {ns / cnt / 1e3:8.2f} us  before {k[:90]}")
# one iteration, launch by launch: from the interior SpMV launch of a middle iteration to the next one
spmv = [i for i, e in enumerate(solve) if "stencil5_rowlds_kernel<1" in e[2] or "stencil5_row_" in e[2]]
if len(spmv) >= 3:
    a, b = spmv[len(spmv) // 2], spmv[len(spmv) // 2 + 1]
    print(f"one iteration, launch by launch (start offset, duration, idle before; us) -- {(solve[b][0] - solve[a][0]) / 1e3:.1f} us:")
    prev_end = solve[a - 1][1] if a > 0 else solve[a][0]
    for s, e, name in solve[a:b]:
        short = name.replace("(anonymous namespace)::", "").replace("spmv_amd::", "").replace("void ", "").split("(")[0]
        print(f"  +{(s - solve[a][0]) / 1e3:8.1f}  {(e - s) / 1e3:8.1f}  {(s - prev_end) / 1e3:6.1f}  {short[:90]}")
        prev_end = max(prev_end, e)
