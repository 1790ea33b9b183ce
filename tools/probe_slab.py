#!/usr/bin/env python3
"""One stand-in slab (rank r of P, headline grid) solved a few times through the full multi-rank pipeline with the rank
as its own neighbour -- the target of `rocprofv3 --kernel-trace -- python3 tools/probe_slab.py 8 3 [solves] [mailbox|rccl]`.
Prints the wall time per solve; tools/trace_gaps.py turns the kernel trace into a per-iteration timeline."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_binding  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = int(sys.argv[2]) if len(sys.argv) > 2 else 3
solves = int(sys.argv[3]) if len(sys.argv) > 3 else 5
allreduce = sys.argv[4] if len(sys.argv) > 4 else "mailbox"
grid = int(sys.argv[5]) if len(sys.argv) > 5 else 20000
os.environ["SPMV_AMD_SELF_NEIGHBOUR"] = os.environ["SPMV_AMD_FORCE_COLLECTIVES"] = "1"
B = load_binding().use_lab()  # stand-in slabs and slab options: the LAB build (include/spmv_amd/lab.h)
B.lib()
B.require_gpu()
if P == 1:
    comm = None
    slab = B.CgSlab.stencil5(grid)
else:
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
    if allreduce == "mailbox":
        assert comm.mailbox_enable()
    slab = B.CgSlab.stencil5_as(grid, r, P, comm)
    slab.set_option("stop_at", 14)  # the last iteration counts as the converging one, as on the rank of a real job
for _ in range(2):
    slab.solve(max_iters=14, tol=0.0)
B.lib().spmv_amd_device_synchronize()
t0 = time.perf_counter()
for _ in range(solves):
    st = slab.solve(max_iters=14, tol=0.0)
B.lib().spmv_amd_device_synchronize()
ms = (time.perf_counter() - t0) / solves * 1e3
print(f"slab {r} of {P} ({slab.n_local} rows), all-reduce {allreduce}: {ms:.3f} ms per 14-iteration solve, "
      f"{ms / 14 * 1e3:.1f} us per iteration, SpMV {st.time_spmv_ms / st.iterations * 1e3:.1f} us per launch")
# the solver's own stage timeline (HIP events at the stage boundaries, no host syncs): one more solve
st_t, tl = slab.timeline_solve(max_iters=14, tol=0.0)
print("stage timeline of one extra solve (us per iteration unless named otherwise): " + ", ".join(f"{k}={v:.1f}" for k, v in tl.items()))
slab.destroy()
