"""One rank of a multi-process test; started by tests/test_distributed.py with RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set, rendezvous over gloo on 127.0.0.1.

  mode oracle : (CPU) the partitioned CG algorithm with the ORACLE's per-rank kernels and real
                gloo messages (halo rows by send/recv, dots by all_reduce); rank 0 checks the
                residual history against the serial oracle.
  mode gpu-mailbox / gpu-synthetic-mailbox : as gpu / gpu-synthetic, with the peer-mailbox all-reduce switched on.
  mode gpu    : (GPU) libspmv_amd's slab solver, one rank per process sharing the box's single
                GPU, over the staged communicator whose host callbacks are these gloo calls;
                rank 0 checks against the oracle's partitioned CG.
  mode gpu-rccl / gpu-rccl-mailbox : (>= WORLD_SIZE GPUs) one rank per DEVICE over the RCCL communicator -- ncclSend /
                ncclRecv of the halo rows and ncclAllReduce (or the peer mailbox) between distinct devices, the path
                of BASELINE config 4; rank 0 checks against the oracle's partitioned CG at 1e-10.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from conftest import load_binding, rel_err, hist_err  # noqa: E402


def exchange_halo(rank, world, first_row, last_row):
    """Sends this rank's first grid row to rank-1 and last grid row to rank+1; returns (prev, next)."""
    prev = torch.zeros_like(first_row) if rank > 0 else None
    nxt = torch.zeros_like(last_row) if rank < world - 1 else None
    reqs = []
    if rank > 0:
        reqs += [dist.isend(first_row, rank - 1), dist.irecv(prev, rank - 1)]
    if rank < world - 1:
        reqs += [dist.isend(last_row, rank + 1), dist.irecv(nxt, rank + 1)]
    for r in reqs:
        r.wait()
    return prev, nxt


def allreduce(v):
    t = torch.tensor([v], dtype=torch.float64)
    dist.all_reduce(t)
    return float(t[0])


def run_oracle(n, rank, world):
    rp, ci, va = O.stencil5_csr(n)
    N = n * n
    off, nl = O.partition_rows(N, world, rank)
    base = rp[off]
    lrp = (rp[off:off + nl + 1] - base).astype(np.int32)
    lci, lva = ci[base:rp[off + nl]], va[base:rp[off + nl]]
    x, b = np.zeros(nl), np.ones(nl)

    def spmv(v):
        t = torch.from_numpy(v)
        hp, hn = exchange_halo(rank, world, t[:n].clone(), t[nl - n:].clone())
        return O.spmv_halo(lrp, lci, lva, v, None if hp is None else hp.numpy(), None if hn is None else hn.numpy(), off, N, n)

    r = b - spmv(x)
    p = r.copy()
    rs_old = allreduce(O.dot_host(r, r))
    b_norm = np.sqrt(rs_old)
    hist = [b_norm]
    for it in range(1000):
        Ap = spmv(p)
        alpha = rs_old / allreduce(O.dot_host(p, Ap))
        x = x + alpha * p
        r = r - alpha * Ap
        rs_new = allreduce(O.dot_host(r, r))
        hist.append(np.sqrt(rs_new))
        if np.sqrt(rs_new) / b_norm < 1e-6:
            break
        p = r + (rs_new / rs_old) * p
        rs_old = rs_new
    if rank == 0:
        xs, hs, rs = O.cg_partitioned(rp, ci, va, n, np.ones(N), np.zeros(N), world=world)
        assert len(hist) == len(hs) and rel_err(hist, hs) < 1e-12, (len(hist), len(hs))
        xd, hd, rd = O.cg(rp, ci, va, n, np.ones(N), np.zeros(N))
        assert rd.iterations == len(hist) - 1 and rel_err(hist, hd) < 1e-12
    print(f"rank {rank}: oracle distributed CG ok ({len(hist) - 1} iterations)")


def run_gpu(n, rank, world, synthetic, mailbox=False):
    B = load_binding()
    B.lib()

    comm = B.Comm.staged_over_torch(rank, world, dist)
    assert comm.selftest() == 0  # all-reduce + neighbour exchange + barrier through the communicator
    if mailbox:
        # the dot products' all-reduce as stores between the ranks' hipIpc-mapped mailboxes (here: processes sharing
        # one GPU; on a node: GPUs over xGMI); set up over the communicator itself, self-tested on every rank
        assert comm.mailbox_enable() and comm.mailbox_ready()
        # 3000 all-reduces back to back in one stream, every sum checked: the two slot sets alternate 1500 times
        assert comm.mailbox_selftest(3000) == 0
    N = n * n
    if synthetic:
        slab = B.CgSlab.stencil5(n, comm)
    else:
        m = B.HostMatrix(O.stencil5_coo(n), N, N, n)
        slab = B.CgSlab.from_matrix(m, comm)
        slab.set_vectors(np.ones(N), np.zeros(N))
    assert (slab.row_offset, slab.n_local) == O.partition_rows(N, world, rank)
    aligned = N % (world * n) == 0  # slabs made of whole grid rows: the reference's own domain
    for timers in (0, 1):
        st = slab.solve(timers=timers)
        hist = slab.history()
        x = slab.gather()
        if rank == 0:
            rp, ci, va = O.stencil5_csr(n)
            if aligned:
                xo, ho, ro = O.cg_partitioned(rp, ci, va, n, np.ones(N), np.zeros(N), world=world)
            else:
                # a slab boundary inside a grid row: the reference reads out of bounds there; the
                # mathematically identical single-rank solve is the yardstick
                xo, ho, ro = O.cg(rp, ci, va, n, np.ones(N), np.zeros(N), device_form=False)
            assert st.iterations == ro.iterations and st.converged == 1, (st.iterations, ro.iterations)
            assert hist_err(hist, ho) < 1e-10
            assert np.max(np.abs(x - xo)) <= 1e-10 * np.max(np.abs(xo))
    # the reference entry point over the world communicator
    if not synthetic:
        B.lib().spmv_amd_comm_set_world(comm.handle)
        b, x = np.ones(N), np.zeros(N)
        cfg, st = B.CGConfig(1000, 1e-6, 0, 0), B.CGStatsMultiGPU()
        assert B.lib().spmv_amd_cg_solve_mgpu_partitioned(m.ptr, b.ctypes.data, x.ctypes.data, C.byref(cfg), C.byref(st)) == 0
        if rank == 0:
            assert np.max(np.abs(x - xo)) <= 1e-10 * np.max(np.abs(xo)) and st.solution_norm > 0
        B.lib().spmv_amd_comm_set_world(None)
    slab.destroy()
    comm.destroy()
    print(f"rank {rank}: slab solver over staged/gloo communicator ok")


def run_gpu_rccl(n, rank, world, mailbox):
    B = load_binding()
    L = B.lib()
    assert L.spmv_amd_device_count() >= world, "needs one device per rank"
    L.spmv_amd_set_device(rank)
    box = [B.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm = B.Comm.rccl(rank, world, box[0])
    assert comm is not None and comm.transport() == "rccl" and comm.transport_ranks() == world
    assert comm.selftest() == 0  # all-reduce, neighbour send / recv, loopback, barrier -- between the devices
    used_mailbox = False
    if mailbox:
        # all-or-nothing: hipIpc mapping + a 2048-round self-test across xGMI on every rank, or the transport's all-reduce stays
        used_mailbox = bool(comm.mailbox_enable())
        assert used_mailbox == bool(comm.mailbox_ready())
        everyone = [None] * world
        dist.all_gather_object(everyone, used_mailbox)
        assert all(e == used_mailbox for e in everyone), everyone
        if used_mailbox:
            assert comm.mailbox_selftest(3000) == 0
    N = n * n
    slab = B.CgSlab.stencil5(n, comm)
    assert (slab.row_offset, slab.n_local) == O.partition_rows(N, world, rank)
    # the library's creation check just solved four iterations in the plain order and four in the pipeline BETWEEN THESE DEVICES
    # (halos NaN-poisoned): the flag hand-overs and the in-kernel halo reads must have reproduced the plain order bit for bit
    assert slab.loop_shape() == "pipeline (verified against the plain order at creation)", slab.loop_shape()
    rp, ci, va = O.stencil5_csr(n)
    xo, ho, ro = O.cg_partitioned(rp, ci, va, n, np.ones(N), np.zeros(N), world=world) if rank == 0 else (None, None, None)
    hists = []
    for timers in (0, 1, 0):
        st = slab.solve(timers=timers)
        hist = slab.history()
        hists.append(hist)
        x = slab.gather()
        if rank == 0:
            assert st.iterations == ro.iterations and st.converged == 1, (st.iterations, ro.iterations)
            assert hist_err(hist, ho) < 1e-10
            assert np.max(np.abs(x - xo)) <= 1e-10 * np.max(np.abs(xo))
    assert np.array_equal(hists[0], hists[2])  # fixed-shape reductions + rank-ordered sums: bit-reproducible
    # every rank holds the same bits (the scalars come from all-reduced values)
    everyone = [None] * world
    dist.all_gather_object(everyone, [float(v).hex() for v in hists[0]])
    assert all(e == everyone[0] for e in everyone)
    # stage timeline with the overlap intact
    st_t, tl = slab.timeline_solve()
    assert tl and tl["iterations"] == st_t.iterations and tl["halo_exchange_on_side_stream_us"] > 0
    slab.destroy()
    comm.destroy()
    path = "peer mailbox" if used_mailbox else ("mailbox unavailable between these devices: ncclAllReduce" if mailbox else "ncclAllReduce")
    print(f"rank {rank}: slab solver over RCCL between devices ok ({path}), "
          f"halo exchange {tl['halo_exchange_on_side_stream_us']:.1f} us, iteration {tl['iteration_us']:.1f} us")


def main():
    mode, n = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        if mode == "oracle":
            run_oracle(n, rank, world)
        elif mode == "gpu":
            run_gpu(n, rank, world, synthetic=False)
        elif mode == "gpu-synthetic":
            run_gpu(n, rank, world, synthetic=True)
        elif mode == "gpu-mailbox":
            run_gpu(n, rank, world, synthetic=False, mailbox=True)
        elif mode == "gpu-synthetic-mailbox":
            run_gpu(n, rank, world, synthetic=True, mailbox=True)
        elif mode == "gpu-rccl":
            run_gpu_rccl(n, rank, world, mailbox=False)
        elif mode == "gpu-rccl-mailbox":
            run_gpu_rccl(n, rank, world, mailbox=True)
        else:
            raise SystemExit(f"unknown mode {mode}")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
