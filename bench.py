#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X, one JSON line on stdout (rank 0).

Workload (BASELINE.json metric "SpMV effective GB/s (fp64, 20k x 20k STENCIL5); CG iters/sec at
1/2/4/8 GPUs", reference numbers in BASELINE.md): unpreconditioned CG on the 20000 x 20000
5-point stencil (400 M unknowns, centre 5 / neighbours -1, b = 1, x0 = 0, tol 1e-6), the
configuration of reference docs/PROBLEM_SIZE_SCALING_RESULTS.md:40-47, on N row slabs.

  step      one complete CG solve (14 iterations to tolerance), the unit the reference times;
            inputs (slab CSR, b, x0) are resident in HBM before the timed region; every step starts
            from the stored x0 and ends with the solution x in HBM (the deferred x update's flush pass
            is inside the solve).
  value     CG iterations per second, whole job = K * iterations / (max over ranks of the time
            of K steps, bracketed by barrier + torch.cuda.synchronize() on both sides).
  scaling   strong: the 400 M-unknown problem is fixed, slabs shrink as N grows.
  roofline  the dominant kernel of the timed region, the STENCIL5 row-lds SpMV (fused with the
            p.Ap partials): algorithmic bytes of one launch (8*nnz + 8*cols + 8*rows of the slab,
            SURVEY.md 8d) / its average duration, from HIP events recorded on the solver's stream
            around every in-loop launch of the timed steps. Peak 8 TB/s (MI355X_MICROARCH.md).
  spmv      (N = 1 only) the reference's other headline: stencil5-csr operator, x = 1, 5 warm-ups
            + 10 timed launches, >2 sigma outliers dropped, median -> "effective" GB/s by both of
            the reference's byte formulas (spmv_metrics.cu:85-101 and the published 12*nnz+16*rows).
  scaling_probe  (N = 1 only) per-rank slab sizes of 2 / 4 / 8 GPUs solved on this GPU, plainly and through the full
            RCCL pipeline with the rank as its own neighbour, and the strong-scaling efficiency that projects for an
            assumed 20 us per inter-device all-reduce (evidence for DESIGN.md section 5; not part of `value`).
  cpu_baseline  (N = 1, rank 0) the serial C oracle's CG (oracle/spmv_oracle.c, 1 core) on a
            bounded sample (10 000 x 10 000 = 1/4 of the rows, ~10 s), scaled by rows to the 400 M-unknown problem;
            `all_cores` = the same loops under OpenMP on up to 16 threads (~1-2 s).

Multi-GPU: launched by torch.distributed.run with one rank per GPU. torch.distributed (gloo) is
used only for rendezvous, the unique-id broadcast, barriers and the max-over-ranks; the data path
(halo send/recv + all-reduce) is RCCL inside libspmv_amd.so.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Published by the reference on A100 hardware; no MI355X number exists.
A100_CG_ITERS_PER_S = {1: 14 / 0.5314, 2: 14 / 0.2693, 4: 14 / 0.1363, 8: 14 / 0.0710}
A100_SPMV_EFFECTIVE_GBS_PUBLISHED_FORMULA = 2364.16
HBM_PEAK_GBS = 8000.0


def load_binding():
    spec = importlib.util.spec_from_file_location("spmv_amd_binding", os.path.join(ROOT, "cuda-spmv-benchmark_amd", "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_stats(times_ms):
    """benchmark_with_stats' rule: drop > 2 sigma, median of the rest (benchmark_stats.cu:60-83)."""
    t = np.asarray(times_ms, dtype=np.float64)
    keep = t[np.abs(t - t.mean()) <= 2.0 * t.std()]
    return float(np.median(keep)), int(len(t) - len(keep))


def spmv_headline(B, n, warmup=5, runs=10):
    """The reference's SpMV benchmark (src/main/main.cu:136-187) on the stencil5-csr operator."""
    rows, nnz = n * n, 5 * n * n - 4 * n
    op = B.Operator("stencil5-csr")
    if op.init_synthetic(n) != 0:
        raise RuntimeError("stencil5-csr synthetic init failed")
    dx, dy = B.DeviceVector(rows, fill=1.0), B.DeviceVector(rows, fill=0.0)
    op.time_device(dx, dy, warmup)
    ms = op.time_device(dx, dy, runs)
    median_ms, dropped = reference_stats(ms)
    # checksum without pulling 3.2 GB to the host: sum(y) through a dot with ones on the device
    y_sum = None
    if rows <= 50_000_000:
        y = dy.to_host()
        y_sum = float(y.sum())
    variant = op.variant()
    dx.free(), dy.free(), op.free()
    secs = median_ms / 1e3
    bytes_today = 8 * nnz + 4 * nnz + 4 * (rows + 1) + 8 * rows + 8 * rows
    bytes_published = 12 * nnz + 16 * rows
    bytes_algorithmic = 8 * nnz + 8 * rows + 8 * rows
    return {
        "operator": "stencil5-csr", "variant": variant, "grid": n, "median_ms": median_ms, "outliers_removed": dropped,
        "all_ms": [round(float(v), 4) for v in ms], "gflops": 2.0 * nnz / secs / 1e9,
        "effective_gbs": bytes_today / secs / 1e9, "effective_gbs_published_formula": bytes_published / secs / 1e9,
        "algorithmic_gbs": bytes_algorithmic / secs / 1e9, "frac_of_hbm_peak": bytes_algorithmic / secs / 1e9 / HBM_PEAK_GBS,
        "vs_a100_published": bytes_published / secs / 1e9 / A100_SPMV_EFFECTIVE_GBS_PUBLISHED_FORMULA, "sum_y": y_sum,
    }


def cpu_baseline(sample_grid, full_rows):
    """Oracle CG (serial C, 1 core) on a sample grid; cost is linear in rows, so iterations/s at the
    full size = iterations/s on the sample * sample_rows / full_rows."""
    from oracle import oracle as O

    rp, ci, va = O.stencil5_csr(sample_grid)
    rows = sample_grid * sample_grid
    t0 = time.perf_counter()
    x, hist, res = O.cg(rp, ci, va, sample_grid, np.ones(rows), np.zeros(rows), device_form=True)
    dt = time.perf_counter() - t0
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    rec = {
        "value": res.iterations / dt * rows / full_rows, "unit": "CG iterations/s (scaled to 400M unknowns)", "cores": 1, "kind": "port",
        "sample": f"oracle_cg on the {sample_grid}x{sample_grid} stencil ({rows} rows = 1/{full_rows // rows} of the workload), "
                  f"{res.iterations} iterations in {dt:.2f} s on 1 core of {os.cpu_count()} ({cpu})",
    }
    # the same loops spread over the cores this job may use (a one-GPU box's CPU share is 16), BASELINE.md section 3
    try:
        threads = max(1, min(16, os.cpu_count() or 1))
        t0 = time.perf_counter()
        x2, hist2, res2 = O.cg_all_cores(rp, ci, va, sample_grid, np.ones(rows), np.zeros(rows), threads)
        dt2 = time.perf_counter() - t0
        rec["all_cores"] = {"value": res2.iterations / dt2 * rows / full_rows, "cores": threads, "kind": "port (OpenMP over the oracle's loops)",
                            "sample": f"same sample, {res2.iterations} iterations in {dt2:.2f} s on {threads} threads"}
    except Exception as e:  # optional figure
        rec["all_cores"] = {"error": repr(e)}
    return rec


def scaling_probe(B, torch, grid, full_ms, full_iterations, steps=6):
    """What this one GPU can say about strong scaling (N = 1 only): for P = 2, 4, 8 a square grid with the row count
    of one rank's slab is solved plainly and through the complete multi-rank pipeline over RCCL with the rank as its
    own neighbour (SPMV_AMD_SELF_NEIGHBOUR: halo send/recv on the side stream under the interior SpMV, split SpMV
    launches, both all-reduces issued). Everything but the latency of an all-reduce BETWEEN
    devices is measured; the projection adds 2 per iteration at an assumed 20 us."""
    import math

    out = {"method": "per-rank slab sizes solved on one GPU, plain vs full RCCL pipeline with the rank as its own neighbour; "
                     "projection = pipeline time per iteration x iterations of the full problem + 2 all-reduces per iteration at the assumed latency",
           "assumed_allreduce_latency_us": 20.0, "slabs": []}

    def run(n, comm):
        slab = B.CgSlab.stencil5(n, comm)
        for _ in range(2):
            st = slab.solve()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            st = slab.solve()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        it = st.iterations
        slab.destroy()
        return ms, it

    for P in (2, 4, 8):
        n = int(round(math.sqrt(grid * grid / P)))
        plain_ms, it = run(n, None)
        saved = {k: os.environ.get(k) for k in ("SPMV_AMD_SELF_NEIGHBOUR", "SPMV_AMD_FORCE_COLLECTIVES")}
        os.environ["SPMV_AMD_SELF_NEIGHBOUR"] = os.environ["SPMV_AMD_FORCE_COLLECTIVES"] = "1"
        try:
            comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if comm is None:
            out["slabs"].append({"gpus": P, "error": "RCCL communicator could not be created"})
            continue
        pipe_ms, it2 = run(n, comm)
        comm.destroy()
        projected = pipe_ms / it2 * full_iterations + 2 * full_iterations * out["assumed_allreduce_latency_us"] / 1e3
        out["slabs"].append({"gpus": P, "slab_proxy_grid": n, "iterations": it, "plain_ms_per_solve": plain_ms, "pipeline_ms_per_solve": pipe_ms,
                             "pipeline_overhead_us_per_iteration": (pipe_ms / it2 - plain_ms / it) * 1e3,
                             "projected_ms_per_solve": projected, "projected_strong_scaling_efficiency": full_ms / (P * projected)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=20000, help="n of the n x n stencil (default: the 400M-unknown headline)")
    ap.add_argument("--cpu-sample-grid", type=int, default=10000, help="grid of the CPU-baseline sample (10000: ~10 s on one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spmv", action="store_true", help="skip the N=1 SpMV headline leg")
    ap.add_argument("--no-scaling-probe", action="store_true", help="skip the N=1 one-GPU strong-scaling probe")
    ap.add_argument("--scaling-probe-only", nargs=2, metavar=("FULL_MS", "FULL_ITERATIONS"), default=None,
                    help="(internal) run only the scaling probe and print its JSON object")
    args = ap.parse_args()

    # The library prints progress lines the way the reference harness does ("[stencil5-csr] Cleaning up", ...):
    # send everything written to fd 1 from here on to stderr and keep the real stdout for the one JSON line.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    if args.scaling_probe_only is not None:
        import torch

        B = load_binding()
        B.lib()
        B.require_gpu()
        torch.cuda.set_device(0)
        B.lib().spmv_amd_set_device(0)
        emit(scaling_probe(B, torch, args.grid, float(args.scaling_probe_only[0]), int(args.scaling_probe_only[1])))
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world

    import torch  # first: libspmv_amd then shares the HIP runtime torch has loaded
    import torch.distributed as dist

    B = load_binding()
    if not os.path.exists(B.LIB_PATH):
        if rank == 0:
            B.build()
    # SPMV_AMD_BENCH_FORCE_DIST=1 (test hook): take the distributed path with one rank too, i.e.
    # rendezvous, unique-id broadcast and an RCCL communicator, so a 1-GPU box exercises it.
    multi = world > 1 or os.environ.get("SPMV_AMD_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    L = B.lib()
    B.require_gpu()
    # SPMV_AMD_BENCH_DEVICE (test hook): put every rank on one device, so a 1-GPU box can walk the
    # N > 1 control flow (RCCL refuses two ranks on one device -> the staged transport takes over)
    device = int(os.environ.get("SPMV_AMD_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device)
    L.spmv_amd_set_device(device)

    def barrier():
        if multi:
            dist.barrier()

    n = args.grid
    rows, nnz = n * n, 5 * n * n - 4 * n

    spmv = None
    if world == 1 and not args.no_spmv:
        spmv = spmv_headline(B, n)

    comm, transport = None, "single rank (no communicator)"
    if multi:
        box = [B.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = B.Comm.rccl(rank, world, box[0])
        # every rank must have its RCCL communicators and pass the collective self-test (all-reduce,
        # neighbour send/recv, barrier); otherwise all ranks switch together to the staged transport
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 1:
            ok = torch.tensor([1 if comm.selftest() == 0 else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 1:
            transport = "rccl"
        else:
            if comm is not None:
                comm.destroy()
            comm = B.Comm.staged_over_torch(rank, world, dist)
            transport = "staged over torch.distributed/gloo (RCCL unavailable on this node)"
            if rank == 0:
                print("bench.py: RCCL communicator unusable, using the host-staged transport", file=sys.stderr)
    slab = B.CgSlab.stencil5(n, comm)

    iterations = None
    for _ in range(args.warmup):
        st = slab.solve()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    spmv_ms, spmv_launches = 0.0, 0
    for _ in range(args.steps):
        st = slab.solve()
        spmv_ms += st.time_spmv_ms
        spmv_launches += st.iterations
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    iterations = st.iterations
    if multi:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    hist = slab.history()
    kernel_name = {"stencil5/row-lds": "stencil5_rowlds_kernel<true>", "stencil5/row-direct": "stencil5_rowdirect_kernel<1, true>"}.get(
        slab.variant(), slab.variant()) + " (SpMV + p.Ap partials)"

    # dominant kernel: STENCIL5 SpMV of this rank's slab, average over the launches of the timed steps
    local_rows, local_nnz = slab.n_local, slab.local_nnz if slab.local_nnz > 0 else None
    if local_nnz is None:  # nnz of the slab does not fit the int the info call returns (single rank, 20k)
        local_nnz = nnz // world
    alg_bytes = 8 * local_nnz + 8 * local_rows + 8 * local_rows
    avg_spmv_ms = spmv_ms / max(spmv_launches, 1)
    achieved = alg_bytes / (avg_spmv_ms / 1e3) / 1e9 if avg_spmv_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath))
            if rec.get("grid") == n and rec.get("n_gpus") == world:
                traffic = rec.get("bytes_per_launch")  # measured with rocprofv3 --pmc, see the file's "source"
        except (OSError, ValueError):
            pass
    roofline = {
        "bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
        "avg_launch_ms": avg_spmv_ms, "launches_timed": spmv_launches,
        "traffic_note": "L2-to-fabric bytes per launch from separate rocprofv3 --pmc passes (profiles/hbm_traffic.json); Infinity-Cache hits included" if traffic else None,
    }

    out = None
    if rank == 0:
        value = args.steps * iterations / dt
        out = {
            "metric": "cg_iterations_per_second", "value": value, "unit": "CG iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": value / A100_CG_ITERS_PER_S[world] if world in A100_CG_ITERS_PER_S and n == 20000 else None,
            "baseline_note": "reference's published CG iters/s on the same problem at the same GPU count, A100-SXM4-80GB (BASELINE.md); no MI355X number is published",
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"CG on the {n}x{n} 5-point stencil ({rows} unknowns, {nnz} nnz), b=1, x0=0, tol 1e-6",
                       "grid": n, "unknowns": rows, "nnz": nnz, "partition": f"{world} row slab(s)", "transport": transport,
                       "iterations_per_solve": iterations, "converged": bool(st.converged), "final_residual": st.residual_norm,
                       "residual_history": [float(v) for v in hist]},
            "roofline": roofline,
        }
        if spmv is not None:
            out["spmv"] = spmv
    slab.destroy()
    if comm is not None:
        comm.destroy()

    if rank == 0 and world == 1 and not multi and not args.no_scaling_probe and n >= 8192:
        # evidence only, in a child process: whatever happens to it, the benchmark line above is printed
        import subprocess
        try:
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--grid", str(n), "--scaling-probe-only",
                                    repr(out["ms_per_step"]), str(iterations)], capture_output=True, text=True, timeout=600)
            lines = [l for l in child.stdout.splitlines() if l.startswith("{")]
            out["scaling_probe"] = json.loads(lines[-1]) if child.returncode == 0 and lines else {
                "error": f"probe exited with {child.returncode}", "stderr_tail": child.stderr[-400:]}
        except Exception as e:
            out["scaling_probe"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_grid, rows)
    if rank == 0:
        emit(out)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
