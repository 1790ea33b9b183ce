"""World-size > 1 paths. CPU: the partitioned algorithm over real gloo messages (oracle kernels).
GPU: libspmv_amd's slab solver with 2 and 3 ranks sharing the box's GPU over the staged/gloo
communicator, and the RCCL communicator with one rank (a 1-GPU box cannot host two RCCL ranks)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_err, hist_err


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(world, mode, n, timeout=600):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(n)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
    return outs


@pytest.mark.parametrize("world,n", [(2, 64), (2, 130), (4, 64)])
def test_partitioned_cg_over_gloo_cpu(world, n):
    outs = launch(world, "oracle", n)
    assert all("oracle distributed CG ok" in o for o in outs)


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,mode", [(2, 64, "gpu"), (2, 256, "gpu"), (3, 192, "gpu-synthetic"), (2, 512, "gpu-synthetic"),
                                          (4, 512, "gpu-synthetic"), (2, 81, "gpu"), (3, 200, "gpu-synthetic"),
                                          (4, 2048, "gpu-synthetic")])  # the box allows 6 GPU processes incl. pytest itself
def test_slab_solver_multi_rank_one_gpu(world, n, mode):
    outs = launch(world, mode, n)
    assert all("slab solver over staged/gloo communicator ok" in o for o in outs)


@pytest.mark.gpu
def test_rccl_communicator_single_rank(B, O, fresh_host_matrices):
    uid = B.Comm.unique_id()
    assert len(uid) == B.COMM_ID_BYTES and any(uid)
    comm = B.Comm.rccl(0, 1, uid)
    assert B.lib().spmv_amd_comm_rank(comm.handle) == 0 and B.lib().spmv_amd_comm_size(comm.handle) == 1
    assert B.lib().spmv_amd_comm_selftest(comm.handle) == 0  # one all-reduce + barrier through RCCL
    n = 200
    slab = B.CgSlab.stencil5(n, comm)
    st = slab.solve()
    rp, ci, va = O.stencil5_csr(n)
    xo, ho, ro = O.cg_partitioned(rp, ci, va, n, np.ones(n * n), np.zeros(n * n), world=1)
    assert st.iterations == ro.iterations and hist_err(slab.history(), ho) < 1e-10
    slab.destroy()
    comm.destroy()


@pytest.mark.gpu
def test_bench_distributed_path_with_one_rank():
    """bench.py through torch.distributed.run with one rank and the two test hooks: rendezvous,
    unique-id broadcast, RCCL communicator creation and the all-reduces inside the CG loop all run
    (a 1-rank all-reduce is the identity), and the result must equal the plain single-rank run."""
    import json
    env = dict(os.environ, SPMV_AMD_BENCH_FORCE_DIST="1", SPMV_AMD_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--grid", "1024", "--no-cpu-baseline", "--no-spmv"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    dist_line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--grid", "1024",
                            "--no-cpu-baseline", "--no-spmv"], capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stdout + plain.stderr
    plain_line = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    assert dist_line["config"]["transport"] == "rccl" and dist_line["n_gpus"] == 1
    assert dist_line["config"]["residual_history"] == plain_line["config"]["residual_history"]
    assert dist_line["config"]["iterations_per_solve"] == plain_line["config"]["iterations_per_solve"]


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [640, 200])
def test_multi_rank_pipeline_over_rccl_with_the_rank_as_its_own_neighbour(grid):
    """SPMV_AMD_SELF_NEIGHBOUR=1: one RCCL rank that is its own previous and next neighbour. The solver runs the
    complete multi-rank pipeline on this one GPU -- ncclSend / ncclRecv of the halo rows on the side stream under
    the interior SpMV, the event waits, the split SpMV launches, ncclAllReduce of both dot products -- and, because
    the first and last grid row of the global grid have no north / south entry, must reproduce the plain solve
    (the split launches change the order of the dot partials, hence 1e-10 rather than bit equality)."""
    import json
    env = dict(os.environ, SPMV_AMD_BENCH_FORCE_DIST="1", SPMV_AMD_FORCE_COLLECTIVES="1", SPMV_AMD_SELF_NEIGHBOUR="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--grid", str(grid), "--no-cpu-baseline", "--no-spmv"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    nb = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--grid", str(grid),
                            "--no-cpu-baseline", "--no-spmv"], capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stdout + plain.stderr
    one = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    assert nb["config"]["transport"] == "rccl" and nb["config"]["converged"]
    assert nb["config"]["iterations_per_solve"] == one["config"]["iterations_per_solve"]
    assert hist_err(np.array(nb["config"]["residual_history"]), np.array(one["config"]["residual_history"])) < 1e-10


@pytest.mark.gpu
def test_bench_two_ranks_control_flow_on_one_gpu():
    """bench.py --gpus 2 under torch.distributed.run with both ranks pinned to the box's only GPU:
    RCCL refuses two ranks on one device, every rank agrees (over gloo) to switch to the staged
    transport, and the run completes with the same residual history as one rank."""
    import json
    env = dict(os.environ, SPMV_AMD_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--grid", "1024"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    two = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "staged" in two["config"]["transport"]
    assert "cpu_baseline" not in two and "spmv" not in two  # N = 1 only legs
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--grid", "1024",
                          "--no-cpu-baseline", "--no-spmv"], capture_output=True, text=True, timeout=600)
    one = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert two["config"]["iterations_per_solve"] == one["config"]["iterations_per_solve"]
    h2, h1 = np.array(two["config"]["residual_history"]), np.array(one["config"]["residual_history"])
    assert hist_err(h2, h1) < 1e-10
