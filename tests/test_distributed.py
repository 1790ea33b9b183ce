"""World-size > 1 paths. CPU: the partitioned algorithm over real gloo messages (oracle kernels); bench.py's
self-launch and fail-loudly path; the watchdog.
GPU: libspmv_amd's slab solver with 2 and 3 ranks sharing the box's GPU over the staged/gloo
communicator, and the RCCL communicator with one rank (a 1-GPU box cannot host two RCCL ranks)."""
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import pytest

from conftest import ROOT, rel_err, hist_err


def free_port():
    """A TCP port for a rendezvous on 127.0.0.1, taken BELOW the kernel's ephemeral range (32768-60999). A port handed out by
    bind(0) lies inside that range: a rank that starts connecting before rank 0 listens can be given the very same number as
    its source port, connects to itself, and rank 0 then fails with EADDRINUSE (seen once in a round-3 test run)."""
    import random

    rng = random.Random(os.getpid() ^ int(time.time() * 1e6))
    for _ in range(200):
        port = rng.randrange(12000, 32000)
        s = socket.socket()
        try:
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            s.bind(("127.0.0.1", port))
            return port
        except OSError:
            continue
        finally:
            s.close()
    raise RuntimeError("no free port found below the ephemeral range")


def launch(world, mode, n, timeout=300):
    """Starts `world` ranks of tests/dist_worker.py. A rank that fails must not leave the others waiting in a gloo
    collective until some outer limit: as soon as one exits non-zero (or the deadline passes) the rest are ended and
    the failing rank's output is reported."""
    port = free_port()
    procs, logs = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
        log = tempfile.TemporaryFile(mode="w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(n)],
                                      env=env, stdout=log, stderr=subprocess.STDOUT, text=True))
    deadline = time.monotonic() + timeout
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.monotonic() > deadline:
            time.sleep(2.0)  # let the others print what they have
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    outs = []
    for p, log in zip(procs, logs):
        p.wait()
        log.seek(0)
        outs.append(log.read())
        log.close()
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} exited with {p.returncode} (first failure decides; killed ranks show -9):\n" + \
            "\n".join(f"--- rank {r} ---\n{o[-3000:]}" for r, o in enumerate(outs))
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
@pytest.mark.parametrize("world,n,mode", [(2, 256, "gpu-mailbox"), (3, 192, "gpu-synthetic-mailbox"), (4, 1024, "gpu-synthetic-mailbox"),
                                          (2, 2048, "gpu-synthetic-mailbox")])
def test_slab_solver_with_the_peer_mailbox_all_reduce(world, n, mode):
    """The dot products' all-reduce as direct stores between hipIpc-mapped mailboxes (csrc/mailbox.hip), between 2-4
    processes sharing the box's GPU: handles exchanged over the communicator, every rank's self-test green, then the
    same parity bar as every other transport (iteration count, ||r_k|| and x at 1e-10 against the oracle)."""
    outs = launch(world, mode, n)
    assert all("slab solver over staged/gloo communicator ok" in o for o in outs)


@pytest.mark.gpu
def test_a_mailbox_whose_peer_never_answers_fails_its_self_test_and_is_dropped(tmp_path):
    """Rank 0 of a two-rank communicator sets its mailbox up by hand with a 'peer' that is its own second allocation
    and never writes: either the mapping is refused, or the self-test gives up after its 5 s probation limit and
    reports failure -- the process lives on and the communicator keeps its transport's all-reduce. Run in a child
    process with a deadline, so that a defect here shows as a failure, not as a stalled test run."""
    script = tmp_path / "dead_peer.py"
    script.write_text(
        "import sys, time\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "from conftest import load_binding\n"
        "B = load_binding(); B.lib(); B.require_gpu()\n"
        "comm = B.Comm.staged(0, 2, lambda *a: 0, lambda *a: 0)\n"
        "other = B.Comm.staged(1, 2, lambda *a: 0, lambda *a: 0)   # same process: stands in for a dead peer\n"
        "h0, h1 = comm.mailbox_prepare(), other.mailbox_prepare()\n"
        "assert h0 is not None and h1 is not None\n"
        "print('prepared', flush=True)\n"
        "t0 = time.monotonic()\n"
        "if comm.mailbox_connect([h0, h1]):\n"
        "    print('connected', flush=True)\n"
        "    assert comm.mailbox_selftest(4) != 0          # rank 1 never contributes\n"
        "    assert time.monotonic() - t0 < 30\n"
        "    comm.mailbox_disable()\n"
        "else:\n"
        "    print('mapping refused', flush=True)\n"
        "assert not comm.mailbox_ready()\n"
        "comm.destroy(); other.destroy()\n"
        "print('alive', flush=True)\n")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "alive" in out.stdout, out.stdout + out.stderr


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
    env = dict(os.environ, SPMV_AMD_BENCH_FORCE_DIST="1", SPMV_AMD_FORCE_COLLECTIVES="1", SPMV_AMD_LIB="lab", HSA_ENABLE_IPC_MODE_LEGACY="0")
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
    env = dict(os.environ, SPMV_AMD_BENCH_FORCE_DIST="1", SPMV_AMD_FORCE_COLLECTIVES="1", SPMV_AMD_SELF_NEIGHBOUR="1", SPMV_AMD_LIB="lab",
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


def _json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


TIMELINE_KEYS = ["iterations", "solve_ms", "initial_residual_us", "spmv_interior_us", "halo_wait_and_boundary_rows_us", "reduce_pAp_and_allreduce_us",
                 "update_r_us", "reduce_rr_allreduce_and_scalar_step_us", "direction_update_us", "gap_before_next_iteration_us", "iteration_us",
                 "halo_exchange_on_side_stream_us", "final_x_flush_us", "direction_updates"]


def check_multi_rank_line(line, world, mailbox_must_work=True, with_ab=False):
    """What a multi-rank bench line must carry so that the number arrives with its own parity statement: the golden
    comparison, the ranks' agreement, the per-rank stage breakdown and -- only when --allreduce-ab asked for the second
    leg -- both all-reduce figures."""
    assert line["value"] is not None and line["n_gpus"] == world and line["transport"] == "rccl"
    assert line["allreduce"] == "ncclAllReduce"  # north_star's path is the headline
    p = line["parity_vs_golden"]
    assert p["available"] and p["ok"] and p["max_rel_err"] <= 1e-10 and p["iterations"] == p["golden_iterations"]
    assert line["ranks_agree_on_history"] is True
    b = line["breakdown"]
    assert len(b["per_rank"]) == world and [r["rank"] for r in b["per_rank"]] == list(range(world))
    for r in b["per_rank"]:
        assert all(k in r for k in TIMELINE_KEYS) and r["spmv_interior_us"] > 0 and r["iteration_us"] > 0
    assert set(TIMELINE_KEYS) - {"iterations"} <= set(b["max_over_ranks"]) and b["max_over_ranks"]["iteration_us"] >= b["min_over_ranks"]["iteration_us"]
    if not with_ab:
        assert "allreduce_ab" not in line  # a default run measures north_star's path and nothing else
        return
    ab = line["allreduce_ab"]
    assert ab["headline"] == "rccl" and ab["rccl"] == line["ms_per_step"]
    o = ab["other_leg"]
    if ab["mailbox"] is None and not mailbox_must_work:
        # between devices the mailbox may be refused (all-or-nothing set-up); the leg must then SAY so, and the headline stands
        assert "mailbox" in o.get("error", ""), ab
        return
    assert ab["mailbox"] is not None and ab["mailbox"] > 0, ab  # the child leg ran and passed its own parity gate
    assert "peer mailbox" in o["allreduce"] and o["parity_vs_golden"]["ok"] and len(o["breakdown"]["per_rank"]) == world


def read_process_log(path):
    return [json.loads(l) for l in open(path).read().splitlines() if l.strip()] if os.path.exists(path) else []


# the complete multi-rank pipeline on one GPU needs the LAB build's hooks (self-neighbour communicator, forced collectives, fault
# injection): SPMV_AMD_LIB=lab makes binding.py -- and with it bench.py -- load lib/libspmv_amd_lab.so
FORCED_MULTI_ENV = dict(SPMV_AMD_BENCH_FORCE_DIST="1", SPMV_AMD_FORCE_COLLECTIVES="1", SPMV_AMD_SELF_NEIGHBOUR="1", SPMV_AMD_LIB="lab",
                        HSA_ENABLE_IPC_MODE_LEGACY="0")


@pytest.mark.gpu
def test_bench_multi_rank_default_run_is_lean(tmp_path):
    """One GPU, the complete multi-rank pipeline (SPMV_AMD_BENCH_FORCE_DIST + SPMV_AMD_SELF_NEIGHBOUR: RCCL send / recv of the
    halo rows on the side stream, split SpMV launches, ncclAllReduce of both dot products) on the 2000 x 2000 grid, for which a
    golden history is committed, WITHOUT any option: exactly one process touched the GPU (the rank's leg child: the rank process
    itself is a supervisor that only rendezvouses; no second leg, no probe), ONE line, with parity_vs_golden and the stage
    breakdown and no allreduce_ab, degraded or first_leg_failure."""
    log = tmp_path / "procs.jsonl"
    env = dict(os.environ, SPMV_AMD_BENCH_PROCESS_LOG=str(log), **FORCED_MULTI_ENV)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--grid", "2000",
                          "--no-cpu-baseline", "--no-spmv"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = _json_lines(out.stdout)
    assert len(lines) == 1
    check_multi_rank_line(lines[0], 1)
    assert lines[0]["breakdown"]["per_rank"][0]["halo_exchange_on_side_stream_us"] > 0
    assert lines[0]["rccl_ranks"] == 1 and "scaling_probe" not in lines[0] and "degraded" not in lines[0] and "first_leg_failure" not in lines[0]
    assert [p["role"] for p in read_process_log(log)] == ["rank", "leg-child"]
    assert "mix_probe_gbs" in lines[0]["roofline"] and lines[0]["devices"][0]["rank"] == 0  # measured by the leg child, carried by its record


@pytest.mark.gpu
def test_bench_restarts_fresh_ranks_without_overlap_after_a_watchdog_exit(tmp_path):
    """VERDICT r04 item 5. The overlapped pipeline (halo exchange on a side stream, two RCCL communicators driven from two
    streams) has never run between devices; if its ranks end through the library's watchdog, bench.py must not lose the run: the
    rank processes are supervisors, the leg runs in their children, and after a watchdog exit FRESH children are started once
    with SPMV_AMD_NO_OVERLAP=1. Here the side-stream exchange is wedged on the host (SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE; the
    GPU is not involved) and the watchdog limit is 5 s: one line, measured, parity-checked, marked degraded with the
    watchdog's own sentence and the first attempt's outcome; 2N = 2 processes touched the GPU over the run's life."""
    log = tmp_path / "procs.jsonl"
    env = dict(os.environ, SPMV_AMD_BENCH_PROCESS_LOG=str(log), SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE="1", SPMV_AMD_WATCHDOG_S="5", **FORCED_MULTI_ENV)
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--grid", "2000",
                          "--no-cpu-baseline", "--no-spmv"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and time.monotonic() - t0 < 300, out.stdout + out.stderr
    lines = _json_lines(out.stdout)
    assert len(lines) == 1
    line = lines[0]
    check_multi_rank_line(line, 1)
    assert line["degraded"].startswith("no-overlap after no progress for ") and "halo exchange" in line["degraded"], line["degraded"]
    first = line["first_leg_failure"]["per_rank"]
    assert len(first) == 1 and first[0]["rc"] == 1 and "halo exchange" in first[0]["watchdog"]
    assert line["vs_baseline"] is None
    assert line["breakdown"]["per_rank"][0]["halo_exchange_on_side_stream_us"] == 0  # the exchange ran on the compute stream
    assert [p["role"] for p in read_process_log(log)] == ["rank", "leg-child", "leg-child"]
    assert "[spmv_amd watchdog]" in out.stderr and "starting fresh ranks once with SPMV_AMD_NO_OVERLAP=1" in out.stderr


@pytest.mark.gpu
def test_bench_restarts_fresh_ranks_without_overlap_after_wrong_numbers(tmp_path):
    """Round 5: the overlapped pipeline hands rows between its streams by device flags (halo arrival, edge rows ready), proven on
    one device only. If its ranks FINISH but with a residual history off the committed golden (or ranks disagreeing), the
    supervisors start fresh ranks once with SPMV_AMD_NO_OVERLAP=1 as after a watchdog exit. Here the overlapped pipeline is made
    to deliver wrong numbers by a test hook (SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE=2: b[0] nudged by the first side-stream
    exchange): one line, measured, parity-checked, marked degraded with the parity sentence."""
    log = tmp_path / "procs.jsonl"
    env = dict(os.environ, SPMV_AMD_BENCH_PROCESS_LOG=str(log), SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE="2", **FORCED_MULTI_ENV)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--grid", "2000",
                          "--no-cpu-baseline", "--no-spmv"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = _json_lines(out.stdout)
    assert len(lines) == 1
    line = lines[0]
    check_multi_rank_line(line, 1)
    assert line["degraded"].startswith("no-overlap after residual history of the timed solves differs from the committed golden"), line["degraded"]
    first = line["first_leg_failure"]["per_rank"]
    assert len(first) == 1 and first[0]["rc"] != 0 and first[0]["watchdog"] is None and "golden" in first[0]["error"]
    assert line["parity_vs_golden"]["ok"] and line["breakdown"]["per_rank"][0]["halo_exchange_on_side_stream_us"] == 0
    assert [p["role"] for p in read_process_log(log)] == ["rank", "leg-child", "leg-child"]
    assert "produced wrong numbers" in out.stderr and "starting fresh ranks once with SPMV_AMD_NO_OVERLAP=1" in out.stderr


@pytest.mark.gpu
def test_bench_multi_rank_line_carries_both_allreduce_legs_when_asked():
    """--allreduce-ab: the headline line first, alone; then the mailbox leg in a child process and the line again with
    allreduce_ab added."""
    env = dict(os.environ, **FORCED_MULTI_ENV)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--grid", "2000",
                          "--no-cpu-baseline", "--no-spmv", "--allreduce-ab"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = _json_lines(out.stdout)
    assert len(lines) == 2
    check_multi_rank_line(lines[0], 1)
    check_multi_rank_line(lines[1], 1, with_ab=True)
    assert lines[1]["value"] == lines[0]["value"] and lines[1]["config"] == lines[0]["config"]


@pytest.mark.gpu
def test_bench_headline_survives_a_second_leg_that_is_killed(tmp_path):
    """--allreduce-ab with the child leg ending itself by SIGKILL after its warm-ups (SPMV_AMD_BENCH_TEST_KILL_LEG_CHILD), the way
    a process guard or an out-of-memory kill would: the headline line was already written, the exit status is 0, and the
    second line says what happened to the leg."""
    log = tmp_path / "procs.jsonl"
    env = dict(os.environ, SPMV_AMD_BENCH_TEST_KILL_LEG_CHILD="1", SPMV_AMD_BENCH_PROCESS_LOG=str(log), **FORCED_MULTI_ENV)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--grid", "2000",
                          "--no-cpu-baseline", "--no-spmv", "--allreduce-ab"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = _json_lines(out.stdout)
    assert len(lines) == 2 and lines[0]["value"] is not None and lines[0]["parity_vs_golden"]["ok"] and "allreduce_ab" not in lines[0]
    ab = lines[1]["allreduce_ab"]
    assert lines[1]["value"] == lines[0]["value"] and ab["mailbox"] is None and "signal 9" in ab["other_leg"]["error"], ab
    assert sorted(p["role"] for p in read_process_log(log)) == ["leg-child", "leg-child", "rank"]  # headline leg, killed extra leg, supervisor


@pytest.mark.gpu
def test_bench_refuses_a_history_that_misses_the_golden(tmp_path):
    """parity gate: the same run against a golden history that is off by 1e-9 in one entry is UNMEASURED (exit 3)."""
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))
    golden["cases"]["512:5.0"]["cg"]["history"][3] *= 1.0 + 1e-9
    bad = tmp_path / "known_answers.json"
    bad.write_text(json.dumps(golden))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--grid", "512", "--no-cpu-baseline", "--no-spmv"]
    good = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert good.returncode == 0 and _json_lines(good.stdout)[-1]["parity_vs_golden"]["ok"], good.stdout + good.stderr
    out = subprocess.run(cmd, env=dict(os.environ, SPMV_AMD_BENCH_GOLDEN=str(bad)), capture_output=True, text=True, timeout=600)
    line = _json_lines(out.stdout)[-1]
    assert out.returncode == 3 and line["value"] is None and "golden" in line["unmeasured"], out.stdout + out.stderr


def gpus_visible():
    from conftest import load_binding
    B = load_binding()
    return B.lib().spmv_amd_device_count() if os.path.exists(B.LIB_PATH) else 0


def need_gpus(world):
    have = gpus_visible()
    if have < world:
        pytest.skip(f"needs {world} MI355X in one node for RCCL between DISTINCT devices (BASELINE config 4); this box shows {have} -- "
                    "the same solver, transport calls and bench.py logic run above on one GPU (staged ranks, self-neighbour RCCL rank)")


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,mode", [(2, 512, "gpu-rccl"), (2, 2000, "gpu-rccl"), (2, 2000, "gpu-rccl-mailbox"), (4, 2000, "gpu-rccl"),
                                          (4, 2000, "gpu-rccl-mailbox")])
def test_slab_solver_over_rccl_between_devices(world, n, mode):
    """BASELINE config 4's data path on real links: one rank per device, halo rows by ncclSend / ncclRecv over xGMI on the
    side stream, dot products by ncclAllReduce (or the peer mailbox: uncached memory mapped through hipIpc, system-scope
    stores), against the oracle's partitioned CG at 1e-10, bit-reproducible, all ranks holding the same history.
    The mailbox is all-or-nothing by design: where the devices cannot map one another's mailboxes it must refuse on EVERY rank
    and the solve must run -- and pass -- on ncclAllReduce; the worker says which it was. (8 ranks are left to bench.py under
    the driver: the test pool limits how many processes a test run may put on the GPUs.)"""
    need_gpus(world)
    outs = launch(world, mode, n, timeout=600)
    assert all("slab solver over RCCL between devices ok" in o for o in outs)
    if mode.endswith("mailbox"):
        said = {("peer mailbox" in o, "mailbox unavailable" in o) for o in outs}
        assert len(said) == 1 and sum(next(iter(said))) == 1, outs  # every rank took the same path, and says which


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_bench_over_rccl_between_devices(world, tmp_path):
    """`python bench.py --gpus N` as the driver runs it, on N devices, at the 2000 x 2000 grid (golden history committed):
    rccl_ranks == N, parity_vs_golden green, breakdown for every rank, exactly N processes on the GPUs (the ranks' leg children;
    the rank processes themselves are supervisors that never touch a GPU)."""
    need_gpus(world)
    log = tmp_path / "procs.jsonl"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SPMV_AMD_BENCH_PROCESS_LOG=str(log))
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--grid", "2000"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = _json_lines(out.stdout)
    assert len(lines) == 1 and lines[0]["rccl_ranks"] == world
    check_multi_rank_line(lines[0], world)
    assert len({d["pci_bus_id"] for d in lines[0]["devices"]}) == world  # distinct devices
    assert sorted(p["role"] for p in read_process_log(log)) == ["launcher"] + ["leg-child"] * world + ["rank"] * world  # N processes on the GPUs, no more
    assert "degraded" not in lines[0]


def test_bench_self_launches_its_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2`, the way the driver invokes it, with no GPU in sight: the parent starts two ranks
    (no torch, no GPU call in the parent), they rendezvous over gloo, agree that nothing can be measured, rank 0 prints
    ONE line with value null and the reason of every rank, and the exit status is non-zero."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--grid", "256"],
                         env=env, capture_output=True, text=True, timeout=600)
    lines = _json_lines(out.stdout)
    assert out.returncode == 3, out.stdout + out.stderr
    assert len(lines) == 1 and lines[0]["value"] is None and lines[0]["n_gpus"] == 2
    assert "rank 0:" in lines[0]["unmeasured"] and "rank 1:" in lines[0]["unmeasured"]
    assert "UNMEASURED" in out.stderr


def test_bench_default_multi_gpu_command_starts_exactly_n_rank_processes(tmp_path):
    """`python bench.py --gpus 2` with every default: the processes of the run are the launcher (no torch, no GPU), two rank
    processes (supervisors: gloo rendezvous, no GPU) and ONE leg child each -- no second leg, no probe, nothing else that could
    touch a GPU (no GPU here: the leg children report UNMEASURED, and that is no watchdog exit, so nothing is restarted)."""
    log = tmp_path / "procs.jsonl"
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", SPMV_AMD_BENCH_PROCESS_LOG=str(log))
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 3, out.stdout + out.stderr
    procs = read_process_log(log)
    assert sorted(p["role"] for p in procs) == ["launcher", "leg-child", "leg-child", "rank", "rank"]
    assert sorted(p["rank"] for p in procs if p["role"] == "rank") == [0, 1] == sorted(p["rank"] for p in procs if p["role"] == "leg-child")


def test_bench_default_eight_gpu_command_is_boring_without_gpus(tmp_path):
    """`python bench.py --gpus 8` with every default -- the command the driver's scaling tier runs (the reference's launch line is
    `mpirun -np 8 cg_solver_mgpu_stencil`, README.md:345) -- on a machine with no GPU in sight: exactly 1 launcher + 8 supervisors
    + 8 leg children, ONE line that says nothing was measured and names all eight ranks, exit status 3, long before --launch-timeout."""
    log = tmp_path / "procs.jsonl"
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", SPMV_AMD_BENCH_PROCESS_LOG=str(log), OMP_NUM_THREADS="1")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=600)
    took = time.monotonic() - t0
    lines = _json_lines(out.stdout)
    assert out.returncode == 3, out.stdout + out.stderr
    assert took < 300, f"{took:.0f} s: the default launch limit is 420 s"
    assert len(lines) == 1 and lines[0]["value"] is None and lines[0]["n_gpus"] == 8 and lines[0]["scaling"] == "strong"
    for rank in range(8):
        assert f"rank {rank}: device {rank} wanted, 0 HIP device(s) visible" in lines[0]["unmeasured"]
    procs = read_process_log(log)
    assert sorted(p["role"] for p in procs) == ["launcher"] + ["leg-child"] * 8 + ["rank"] * 8
    assert sorted(p["rank"] for p in procs if p["role"] == "rank") == list(range(8)) == sorted(p["rank"] for p in procs if p["role"] == "leg-child")


def load_bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_extra_leg_runner_survives_a_killed_a_silent_and_a_slow_child(tmp_path):
    """run_child_for_record() is how bench.py starts everything that is evidence only (the --allreduce-ab leg, the N = 1
    scaling probe). A child that is SIGKILLed mid-run, one that prints nothing, one that overruns its limit: each comes back
    as a record with "error", within the limit, never as an exception -- the caller's headline does not depend on it."""
    bench = load_bench_module()
    killed = tmp_path / "killed.py"
    killed.write_text("import os, signal, sys\nprint('{\"partial\": 1}', flush=True)\nos.kill(os.getpid(), signal.SIGKILL)\n")
    rec = bench.run_child_for_record([sys.executable, str(killed)], dict(os.environ), 60, "child leg")
    assert rec["partial"] == 1 and "signal 9" in rec["error"]
    silent = tmp_path / "silent.py"
    silent.write_text("import sys\nsys.exit(0)\n")
    assert "error" in bench.run_child_for_record([sys.executable, str(silent)], dict(os.environ), 60, "child leg")
    slow = tmp_path / "slow.py"
    slow.write_text("import time\ntime.sleep(600)\n")
    t0 = time.monotonic()
    rec = bench.run_child_for_record([sys.executable, str(slow)], dict(os.environ), 2, "child leg")
    assert "did not finish within 2 s" in rec["error"] and time.monotonic() - t0 < 30
    good = tmp_path / "good.py"
    good.write_text("print('noise')\nprint('{\"ms_per_step\": 1.5}')\n")
    assert bench.run_child_for_record([sys.executable, str(good)], dict(os.environ), 60, "child leg") == {"ms_per_step": 1.5}


def test_bench_default_timeouts_fit_the_drivers_limit():
    """The driver ends bench.py after 600 s: every limit bench.py sets for itself must expire first, so that its own line
    (with the reason) is what the driver reads."""
    import re
    defaults = {m.group(1): float(m.group(2)) for m in re.finditer(r'"--([a-z-]+-timeout)", type=float, default=([0-9.]+)', open(os.path.join(ROOT, "bench.py")).read())}
    assert set(defaults) == {"ab-timeout", "probe-timeout", "launch-timeout", "leg-timeout"}
    assert defaults["launch-timeout"] < 600 and defaults["ab-timeout"] < defaults["launch-timeout"] and defaults["probe-timeout"] + 180 < 600
    assert 2 * defaults["leg-timeout"] + 60 < defaults["launch-timeout"]  # two attempts at the headline leg (overlapped, then not) fit


def test_bench_self_launch_reports_a_rank_that_died():
    """One of the self-launched ranks dies before the rendezvous: the parent gives the others a moment, ends them,
    prints ONE line saying nothing was measured and with which statuses the ranks ended, and exits non-zero itself."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", SPMV_AMD_BENCH_TEST_CRASH_RANK="1")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--grid", "256",
                          "--launch-grace", "3"], env=env, capture_output=True, text=True, timeout=300)
    lines = _json_lines(out.stdout)
    assert out.returncode != 0 and time.monotonic() - t0 < 120, out.stdout + out.stderr
    assert len(lines) == 1 and lines[0]["value"] is None and "7" in lines[0]["unmeasured"] and lines[0]["n_gpus"] == 2


def test_watchdog_turns_a_wedged_barrier_into_a_diagnosable_exit(tmp_path):
    """A staged communicator whose barrier callback never returns (a peer that died): the watchdog names the rank and
    the stage and ends the process with a non-zero status after SPMV_AMD_WATCHDOG_S seconds. No GPU involved."""
    script = tmp_path / "wedged_barrier.py"
    script.write_text(
        "import sys, time\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "from conftest import load_binding\n"
        "B = load_binding(); B.lib()\n"
        "def never(user):\n"
        "    time.sleep(3600)\n"
        "    return 0\n"
        "c = B.Comm.staged(1, 2, lambda *a: 0, lambda *a: 0, None, never)\n"
        "print('transport', c.transport(), flush=True)\n"
        "c.barrier()\n"
        "print('unreachable', flush=True)\n")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, SPMV_AMD_WATCHDOG_S="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and time.monotonic() - t0 < 60, out.stdout + out.stderr
    assert "transport staged" in out.stdout and "unreachable" not in out.stdout
    assert "[spmv_amd watchdog] rank 1: no progress" in out.stderr and "stage 'barrier'" in out.stderr


@pytest.mark.gpu
def test_watchdog_names_the_stage_of_a_wedged_all_reduce(tmp_path):
    """Two-rank staged communicator whose all-reduce callback stops answering in its third call (initial r.r, p.Ap of
    iteration 0, then r.r of iteration 0): the solve ends with the watchdog's report naming the all-reduce and the
    iteration, with the state of both streams, instead of hanging."""
    script = tmp_path / "wedged_allreduce.py"
    script.write_text(
        "import sys, time\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "from conftest import load_binding\n"
        "B = load_binding(); B.lib()\n"
        "calls = [None]\n"
        "def allreduce(user, buf, count):\n"
        "    if calls[0] is None:\n"
        "        return 0\n"  # creation (the library's check of the pipeline against the plain order solves a few iterations)
        "    calls[0] += 1\n"
        "    if calls[0] >= 3:\n"
        "        time.sleep(3600)\n"
        "    return 0\n"
        "c = B.Comm.staged(0, 2, lambda *a: 0, allreduce)\n"
        "slab = B.CgSlab.stencil5(256, c)\n"
        "calls[0] = 0\n"
        "print('solving', flush=True)\n"
        "slab.solve()\n"
        "print('unreachable', flush=True)\n")
    out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, SPMV_AMD_WATCHDOG_S="2"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 1, out.stdout + out.stderr
    assert "solving" in out.stdout and "unreachable" not in out.stdout
    assert "[spmv_amd watchdog] rank 0: no progress" in out.stderr and "all-reduce of" in out.stderr
    assert "compute stream:" in out.stderr and "side (halo) stream:" in out.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_refuses_to_publish_a_staged_number():
    """`python bench.py --gpus 2` (self-launched) with both ranks pinned to the box's only GPU: RCCL cannot span two
    ranks on one device. Without SPMV_AMD_BENCH_ALLOW_STAGED the run must NOT fall back: one line with value null, the
    reason, exit status 3."""
    env = dict(os.environ, SPMV_AMD_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SPMV_AMD_BENCH_ALLOW_STAGED", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--grid", "1024"],
                         env=env, capture_output=True, text=True, timeout=600)
    lines = _json_lines(out.stdout)
    assert out.returncode == 3, out.stdout + out.stderr
    assert len(lines) == 1 and lines[0]["value"] is None and "RCCL" in lines[0]["unmeasured"] and lines[0]["n_gpus"] == 2


@pytest.mark.gpu
def test_bench_two_ranks_control_flow_on_one_gpu():
    """`python bench.py --gpus 2`, self-launched, both ranks pinned to the box's only GPU and the staged transport
    explicitly allowed: the ranks agree (over gloo) to switch, the run completes with the same residual history as one
    rank, and the line is marked degraded and carries no vs_baseline."""
    env = dict(os.environ, SPMV_AMD_BENCH_DEVICE="0", SPMV_AMD_BENCH_ALLOW_STAGED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--grid", "1024"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = _json_lines(out.stdout)
    assert len(lines) == 1
    two = lines[0]
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "staged" in two["transport"] and "degraded" in two
    assert two["vs_baseline"] is None and two["rccl_ranks"] == 0 and "self-launched" in two["launched_by"]
    assert [d["rank"] for d in two["devices"]] == [0, 1] and all(d["device"] == 0 for d in two["devices"])
    assert "cpu_baseline" not in two and "spmv" not in two  # N = 1 only legs
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--grid", "1024",
                          "--no-cpu-baseline", "--no-spmv"], capture_output=True, text=True, timeout=600)
    one = _json_lines(one.stdout)[-1]
    assert two["config"]["iterations_per_solve"] == one["config"]["iterations_per_solve"]
    h2, h1 = np.array(two["config"]["residual_history"]), np.array(one["config"]["residual_history"])
    assert hist_err(h2, h1) < 1e-10


@pytest.mark.gpu
def test_bench_two_ranks_under_an_external_launcher_on_one_gpu():
    """The contract's other launch form: torch.distributed.run starts the ranks (RANK / WORLD_SIZE in the environment),
    bench.py must not start any itself."""
    env = dict(os.environ, SPMV_AMD_BENCH_DEVICE="0", SPMV_AMD_BENCH_ALLOW_STAGED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--grid", "1024"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    two = _json_lines(out.stdout)[-1]
    assert two["n_gpus"] == 2 and "external launcher" in two["launched_by"] and two["config"]["converged"]


@pytest.mark.gpu
@pytest.mark.parametrize("n,as_rank,as_world", [(640, 0, 2), (640, 1, 4), (640, 3, 4), (1024, 3, 8)])
def test_stand_in_slab_is_the_real_slab_of_that_rank(Blab, O, monkeypatch, n, as_rank, as_world):
    """spmv_amd_cg_slab_create_stencil5_as (the scaling probe's slab): rank r-of-P's rows, CSR bytes and halo sides on a
    single self-neighbour RCCL rank. Its slab SpMV on caller data (halos filled from the full vector, no exchange) must
    be bit-identical to the oracle's halo kernel for that rank; a solve with tolerance 0 runs exactly max_iters
    iterations through the send/recv + all-reduce pipeline."""
    monkeypatch.setenv("SPMV_AMD_SELF_NEIGHBOUR", "1")
    monkeypatch.setenv("SPMV_AMD_FORCE_COLLECTIVES", "1")
    B = Blab
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
    assert comm is not None and comm.transport() == "rccl" and comm.transport_ranks() == 1
    slab = B.CgSlab.stencil5_as(n, as_rank, as_world, comm)
    off, nl = O.partition_rows(n * n, as_world, as_rank)
    assert (slab.row_offset, slab.n_local) == (off, nl)
    rp, ci, va = O.stencil5_csr(n)
    x = np.random.default_rng(5).standard_normal(n * n)
    base = rp[off]
    lrp = (rp[off:off + nl + 1] - base).astype(np.int32)
    hp = x[off - n:off] if as_rank > 0 else None
    hn = x[off + nl:off + nl + n] if as_rank < as_world - 1 else None
    want = O.spmv_halo(lrp, ci[base:], va[base:], x[off:off + nl], hp, hn, off, n * n, n)
    assert np.array_equal(slab.spmv(x), want)
    st = slab.solve(max_iters=6, tol=0.0)
    assert st.iterations == 6 and st.converged == 0 and np.all(np.isfinite(slab.history()))
    assert slab.history()[-1] < slab.history()[0]  # the mirrored slab is SPD too: CG makes progress
    slab.destroy()
    comm.destroy()


def stand_in_system(O, n, as_rank, as_world):
    """The linear system a stand-in slab solves, as a general CSR matrix of the slab's own rows. The slab is rank r of P of the
    n x n stencil carried by ONE self-neighbour RCCL rank: the halo exchange sends the first grid row "to the previous rank" and the
    last grid row "to the next rank", both of which are this rank, and same-peer send / recv pairs match in the order they were
    issued (csrc/comm.hip, halo_exchange) -- so the previous-rank halo receives the slab's own FIRST grid row and the next-rank
    halo its own LAST one. A north entry of the first grid row (global column c in [off - n, off)) therefore multiplies the row's
    own element (local column c - off + n), a south entry of the last grid row local column c - off - n: the slab mirrored at both
    cuts (diagonal 4 instead of 5 there: still symmetric positive definite). Rows and values are the global stencil's
    (reference slab: cg_solver_mgpu_partitioned.cu:306-329); the column map is the halo kernel's (spmv_stencil_partitioned_halo_kernel.cu:44-68)
    with the halos' contents written out."""
    off, nl = O.partition_rows(n * n, as_world, as_rank)
    rp, ci, va = O.stencil5_csr(n)
    base, end = int(rp[off]), int(rp[off + nl])
    lrp = (rp[off:off + nl + 1] - base).astype(np.int32)
    local = ci[base:end].astype(np.int64) - off
    if as_rank == 0:
        assert local.min() >= 0  # the global first grid row has no north entry
    if as_rank == as_world - 1:
        assert local.max() < nl
    local = np.where(local < 0, local + n, local)
    local = np.where(local >= nl, local - n, local)
    assert local.min() >= 0 and local.max() < nl
    return lrp, local.astype(np.int32), va[base:end].copy(), nl


@pytest.mark.gpu
@pytest.mark.parametrize("ring", [16, 4])
@pytest.mark.parametrize("n,as_rank,as_world", [(1024, 0, 2), (1024, 1, 2), (1024, 1, 4), (2048, 3, 8)])
def test_stand_in_slab_over_rccl_solves_its_system_like_the_oracle(Blab, O, monkeypatch, n, as_rank, as_world, ring):
    """The RCCL transport's multi-rank shapes against the ORACLE: a stand-in slab under self-neighbour RCCL
    with DEFAULT options -- halo rows by ncclSend / ncclRecv on the side stream, the boundary waves' in-kernel wait on the arrival
    flag, the direction update's edge rows written through + the side stream's wait kernel, both ncclAllReduce -- is the one
    configuration on a one-GPU box in which the RECEIVED halo values enter the result over RCCL (a whole-grid self-neighbour
    solve cannot notice stale or lost halo rows: its first / last grid row have no north / south entry). The system it solves,
    written out as a general CSR (stand_in_system), goes through oracle_cg's CSR loop with the partitioned solver's BLAS1 forms
    (device_form = False: p = r + beta p as axpby, cg_solver_mgpu_partitioned.cu:680-703); every ||r_k|| must agree to 1e-10.
    Ring 16 (no flush inside 12 iterations: the scalar step rides in the direction update's launch every time) and ring 4 (every
    fourth iteration flushes x first: the step is a launch of its own there)."""
    monkeypatch.setenv("SPMV_AMD_SELF_NEIGHBOUR", "1")
    monkeypatch.setenv("SPMV_AMD_FORCE_COLLECTIVES", "1")
    monkeypatch.setenv("SPMV_AMD_P_RING", str(ring))
    B = Blab
    iterations = 12
    lrp, lci, lva, nl = stand_in_system(O, n, as_rank, as_world)
    _, want, res = O.cg(lrp, lci, lva, -1, np.ones(nl), np.zeros(nl), max_iters=iterations, tol=0.0, device_form=False)
    assert res.iterations == iterations and len(want) == iterations + 1 and want[-1] < 1e-3 * want[0]
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
    assert comm is not None and comm.transport() == "rccl"
    slab = B.CgSlab.stencil5_as(n, as_rank, as_world, comm)
    try:
        assert slab.n_local == nl and slab.variant() == "stencil5/row-lds"
        for _ in range(3):  # back to back: flags and sequence numbers carry over from solve to solve
            st = slab.solve(max_iters=iterations, tol=0.0)
            got = slab.history()
            assert st.iterations == iterations and st.converged == 0 and len(got) == iterations + 1
            assert hist_err(got, want) < 1e-10, (got, want)
    finally:
        slab.destroy()
        comm.destroy()


@pytest.mark.gpu
def test_a_ranks_set_up_fits_the_leg_timeout_many_times_over(Blab, monkeypatch):
    """What one rank of the default `bench.py --gpus 8` run does before its first timed solve, on the real P = 8 slab of the headline
    grid (rank 3 of 8: 5e7 rows, two neighbours; a stand-in on this one GPU): two RCCL communicators + the self-test, the slab
    generated in HBM, the placement and tile-run trials of creation, 3 warm-up solves + 10 steps. bench.py ends an attempt at the
    headline leg after --leg-timeout = 170 s (two attempts fit the 420 s launch limit): the whole of it must take a small
    fraction of that, or a slow box would turn a healthy run into an UNMEASURED line."""
    monkeypatch.setenv("SPMV_AMD_SELF_NEIGHBOUR", "1")
    monkeypatch.setenv("SPMV_AMD_FORCE_COLLECTIVES", "1")
    B = Blab
    t0 = time.monotonic()
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
    assert comm is not None and comm.selftest() == 0
    t_comm = time.monotonic() - t0
    slab = B.CgSlab.stencil5_as(20000, 3, 8, comm)
    t_slab = time.monotonic() - t0 - t_comm
    phases = slab.setup_ms()
    for _ in range(13):
        st = slab.solve(max_iters=14, tol=0.0)
    total = time.monotonic() - t0
    slab.destroy()
    comm.destroy()
    assert st.iterations == 14 and slab is not None
    assert t_comm < 30 and t_slab < 30 and total < 60, (t_comm, t_slab, total, phases)
    assert sum(phases.values()) / 1e3 <= t_slab + 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("fault", ["0", "3", "4"])
def test_creation_check_refuses_a_pipeline_that_loses_halo_rows(Blab, O, monkeypatch, capfd, fault):
    """The library's own guard for the constructions no one-GPU box can prove between devices: the first slab created
    on a communicator that exchanges halos solves four iterations in the plain order and four in the pipeline; the shapes are
    bit-identical by construction, so a difference refuses the pipeline for that communicator. Healthy (fault 0): the pipeline
    is kept and says it was verified. Fault injection of the LAB build: 3 = the rows of a side-stream exchange never travel,
    4 = the exchange's arrival flag never comes (the boundary waves give up after 2 s in the check, not 20 s, and the check --
    not the process -- ends): the slab says so on stderr, runs the plain order (no side-stream exchange in the timeline) and its
    solve still matches the oracle on the stand-in's system, where the received rows matter."""
    monkeypatch.setenv("SPMV_AMD_SELF_NEIGHBOUR", "1")
    monkeypatch.setenv("SPMV_AMD_FORCE_COLLECTIVES", "1")
    monkeypatch.setenv("SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE", fault)
    B = Blab
    n, r, P, iterations = 1024, 1, 4, 10
    lrp, lci, lva, nl = stand_in_system(O, n, r, P)
    _, want, _ = O.cg(lrp, lci, lva, -1, np.ones(nl), np.zeros(nl), max_iters=iterations, tol=0.0, device_form=False)
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
    t0 = time.monotonic()
    slab = B.CgSlab.stencil5_as(n, r, P, comm)
    assert time.monotonic() - t0 < 60
    err = capfd.readouterr().err
    try:
        shape = slab.loop_shape()
        if fault == "0":
            assert shape == "pipeline (verified against the plain order at creation)" and "refused" not in err
        else:
            assert shape.startswith("plain: the pipeline's residual history differed"), shape
            assert "did NOT reproduce the plain order's residual history" in err and "refused the overlapped pipeline" in err
        st, tl = slab.timeline_solve(max_iters=iterations, tol=0.0)
        assert st.iterations == iterations and hist_err(slab.history(), want) < 1e-10
        assert (tl["halo_exchange_on_side_stream_us"] > 0) == (fault == "0")
        second = B.CgSlab.stencil5_as(n, 0, 2, comm)  # the verdict belongs to the communicator: a later slab does not check again
        assert second.loop_shape() == shape
        second.destroy()
    finally:
        slab.destroy()
        comm.destroy()
