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
  roofline  the dominant kernel of the timed region, the solver's STENCIL5 SpMV (row-lds kernel, fused with the
            p.Ap partials): algorithmic bytes of one launch (8*nnz + 8*cols + 8*rows of the slab,
            SURVEY.md 8d) / its average duration, from HIP events recorded on the solver's stream
            around every in-loop launch of the timed steps. Peak 8 TB/s (MI355X_MICROARCH.md).
            `frac` is of that peak and is the figure to read. Three yardsticks measured in the same run say what
            the part sustains for streams: mix_probe_gbs (48 B read : 8 B written per row with ideal accesses,
            csrc/stream_ceiling.hip), read_only_probe_gbs (the same arrays, read only) and the rate the loop's own
            r update reaches (stages.update_r_us); best_stream_gbs_this_run is their maximum and
            frac_of_best_stream the SpMV's share of it -- a yardstick, not a ceiling. traffic: fabric bytes per
            launch from profiles/hbm_traffic.json, only while that file's hash of csrc/spmv_kernels.hip matches.
            spmv_standalone: BASELINE's FIRST metric, the `spmv` leg below in brief (median ms, effective GB/s by
            both of the reference's formulas, algorithmic GB/s, frac of peak); placement: what set-up placement did.
  spmv      (N = 1 only; runs AFTER the CG leg, so that the solver's slab is the first large allocation of the process)
            the reference's other headline: stencil5-csr operator on its own staging vectors (the ones run_timed's kernel
            works on), x = 1, 5 warm-ups + 10 timed launches, >2 sigma outliers dropped, median -> "effective" GB/s by both
            of the reference's byte formulas (spmv_metrics.cu:85-101 and the published 12*nnz+16*rows).
  scaling_probe  (N = 1 only; a PROJECTION, never part of `value`; the role processes load the LAB build of the library,
            lib/libspmv_amd_lab.so, the only one with stand-in slabs) the real per-rank slabs of a 2 / 4 / 8-GPU
            run of this problem (rows [r*N/P, (r+1)*N/P), 160 KB halos), edge rank and a two-neighbour rank,
            each in a FRESH process of its own (what a rank of a real job is: inside one process the second slab
            inherits the first one's freed memory and measured up to 3 % off either way, profiles/r05_slab_attribution.txt),
            solved for the full iteration count on this GPU through the complete RCCL pipeline with the rank as its
            own neighbour, the last iteration counting as the converging one as in the real solve (rounds 2-4 let the
            stand-in run a 14th direction update + halo exchange no real rank runs: +1.3 % per slab solve).
            Clock: the solver's timed region (HIP events from the barrier to the end of the x flush,
            the reference's :405-413 -> :728-731), median of 5 solves, for the slab and for the full problem alike.
            The latency of an all-reduce BETWEEN devices cannot be measured on one GPU: the efficiency is given for a
            range of latencies.
  cpu_baseline  (N = 1, rank 0) the serial C oracle (oracle/spmv_oracle.c, 1 core): its CG on the FULL 400 M-unknown
            problem when the host's MemAvailable allows (~45 GB, ~33 s on one EPYC core; `history_vs_committed_golden_ok` then
            re-verifies the golden history on the spot), else on a 10 000 x 10 000 sample scaled by rows (`sample` says which);
            `spmv` = its STENCIL5 and CSR SpMV on the same matrix (1 warm-up + 3 runs, median) in the reference's
            effective-GB/s formula and in algorithmic GB/s; `all_cores` = the same loops under OpenMP.
  compare   (N = 1; after the spmv leg) BASELINE configs 2 and 5 through get_operator with the reference's 5 + 10 / median rule,
            each operator checksum-gated (sum y = n^2 + 4n, ||y||^2 exact): 10 000^2 stencil5-csr / cusparse-csr, 15 000^2 ellpack /
            stencil5-ellpack / stencil5-csr / cusparse-csr. FLAT scalars in `roofline` (the driver's record keeps scalars of
            roofline / config only): stencil10k_ms, csr10k_ms, stencil15k_ms, csr15k_ms, ell15k_ms, ell_stencil15k_ms, each
            with its *_frac (algorithmic bytes of the format / median / 8 TB/s), beside spmv20k_median_ms,
            spmv20k_effective_gbs, spmv20k_effective_gbs_published_formula, spmv20k_frac (BASELINE's first metric).

  placement (rank 0's slab) and spmv.output_placement: on this part a SpMV is ~4.5 % faster when the vector it writes lies in another
            class of 32 GiB address regions than the data it reads, and only hipMalloc decides the class (profiles/r04_spmv_regions.txt):
            at set-up, outside every timed region, the operator times its kernel on three allocations for y, one 32 GiB region apart, and keeps the fastest;
            the slab does the same for its coefficient stream, which must not share Ap's class (csrc/cg_slab.hip). Addresses only, same bits.

  config.loop_shape  which shape the solver's loop took and who decided ("single rank", "pipeline (verified against the plain order
            at creation)", "plain: ..."; include/spmv_amd/api.h): the library compares its overlapped pipeline with the plain order on
            the first slab of every communicator and falls back by itself -- the line then says `degraded`.
  config.host_throttled_periods_in_timed_steps / _usec_  CPU throttling of this container (cgroup cpu.stat) over the timed steps: a
            stalled host thread idles the GPU (profiles/r06_throttle_probe.txt); zero rules the host out of a slow `value`.
            The timed solves must agree on (iterations, verdict, final residual) or the run is UNMEASURED.
  parity_vs_golden  every run, every N: the residual history of the timed solves against the committed CPU-oracle
            history of the same grid (tests/golden/known_answers.json: 3, 81, 512, 2000, 10000, 15000, 20000); above 1e-10
            relative, or on another iteration count, the run is UNMEASURED (exit 3), whatever it timed.
  breakdown per rank, from ONE extra solve after the timed region with HIP events at the stage boundaries of every
            iteration and around every halo exchange (no host syncs, overlap intact): interior SpMV, wait for the halo +
            boundary rows, p.Ap sum + all-reduce, r update, r.r sum + all-reduce + scalar step, direction update, gap
            before the next iteration; max / min over the ranks as the reference reports its timers (mgpu :748-800).
            Each of those events is a barrier packet (~6 us): the breakdown solve is ~40 us per iteration slower than a timed one.
  allreduce_ab  (opt-in: --allreduce-ab, multi-rank runs) `value` is north_star's path: RCCL send/recv halos + ncclAllReduce on
            the two dot products. With the flag the same K steps are then repeated with the peer-mailbox all-reduce
            (csrc/mailbox.hip) in CHILD processes, one per rank: {"rccl": ms, "mailbox": ms | null, "other_leg": {...}}. The
            headline line is written BEFORE that leg starts and written again, augmented, if the leg returns; a default run
            never starts it, so a default `--gpus N` run has exactly N processes on the GPUs (N = 1: one more for the probe).
            SPMV_AMD_BENCH_ALLREDUCE=mailbox swaps the two legs.

Wall time, worst case of every DEFAULT path (the driver's limit for this file is 600 s):
  N = 1   torch import + library (<= 120 s on a fresh box) + SpMV leg (~5 s) + (W + K + 1) solves of ~0.11 s + stream ceiling
          (~0.2 s) + compare leg (six operator set-ups, ~15 s) + scaling probe (8 role processes one after the other under one
          child, the child ended after --probe-timeout = 240 s; ~30 s when healthy) + CPU baseline (full size: ~70 s of host
          work, no GPU; sample: ~25 s) : < 480 s, ~150 s when healthy.
  N > 1   the same import + rendezvous (gloo, 120 s limit for the store) + (W + K + 1) solves + nothing else: rank 0 prints
          the line as soon as the ranks have agreed on the measured leg. Every wait on a peer inside a solve ends after
          SPMV_AMD_WATCHDOG_S = 60 s with a report; gloo collectives give up after 180 s; a self-launched run is ended by
          its parent after --launch-timeout = 420 s, with a line that says so. < 420 s in every case, ~20 s when healthy.
  --allreduce-ab adds one child leg, ended after --ab-timeout = 180 s; it cannot delay or remove the headline line.

Multi-GPU: one process per GPU. Either the driver starts the ranks (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`, RANK / WORLD_SIZE in the environment) or `python bench.py --gpus N`
starts them itself: the parent process touches neither torch nor the GPU, spawns N children of this file with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's JSON line and exits with the worst
child status. torch.distributed (gloo) is used only for rendezvous, the unique-id broadcast, barriers and the
max-over-ranks; the data path (halo send/recv + all-reduce) is RCCL inside libspmv_amd.so.

No silent degradation: if RCCL cannot be created or fails its self-test on any rank, if fewer GPUs are visible
than ranks, or if RCCL spans fewer ranks than asked, every rank stops, rank 0 prints a line with "value": null and
"unmeasured": "<reason>", and the exit status is 3. The host-staged transport (the reference's own D2H -> host
exchange -> H2D scheme) is taken only when SPMV_AMD_BENCH_ALLOW_STAGED=1 asks for it (tests on a 1-GPU box),
and then the line says "degraded" and carries no vs_baseline.
"""
import os as _os

# Thread pools sized by the host's core count (256 on the GPU boxes) inside a container with a 16-CPU quota get the whole
# cgroup CPU-throttled, the solver's host thread included (tools/throttle_probe.sh): keep the numeric libraries' pools small
# unless the caller decided otherwise. Must happen before numpy / torch are imported. The CPU baseline sets its own thread count.
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    _os.environ.setdefault(_k, "8")

import argparse
import hashlib
import importlib.util
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Published by the reference on A100 hardware; no MI355X number exists.
A100_CG_ITERS_PER_S = {1: 14 / 0.5314, 2: 14 / 0.2693, 4: 14 / 0.1363, 8: 14 / 0.0710}
A100_SPMV_EFFECTIVE_GBS_PUBLISHED_FORMULA = 2364.16
HBM_PEAK_GBS = 8000.0
EXIT_UNMEASURED = 3
PARITY_TOL = 1e-10  # north_star: fp64 CG residuals within 1e-10 relative of the CPU path
# SPMV_AMD_BENCH_GOLDEN (test hook): another fixture file, to show that the parity gate closes
GOLDEN_ANSWERS = os.environ.get("SPMV_AMD_BENCH_GOLDEN") or os.path.join(ROOT, "tests", "golden", "known_answers.json")
KERNEL_SOURCE = os.path.join(ROOT, "cuda-spmv-benchmark_amd", "csrc", "spmv_kernels.hip")


def load_binding():
    spec = importlib.util.spec_from_file_location("spmv_amd_binding", os.path.join(ROOT, "cuda-spmv-benchmark_amd", "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def log_process(role, rank=None):
    """SPMV_AMD_BENCH_PROCESS_LOG=<file> (test hook): every process of a bench.py run appends one line saying what it is, so
    a test can count the processes a command really started (tests/test_distributed.py)."""
    path = os.environ.get("SPMV_AMD_BENCH_PROCESS_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"pid": os.getpid(), "role": role, "rank": rank, "argv": sys.argv[1:]}) + "\n")


def reference_stats(times_ms):
    """benchmark_with_stats' rule: drop > 2 sigma, median of the rest (benchmark_stats.cu:60-83)."""
    t = np.asarray(times_ms, dtype=np.float64)
    keep = t[np.abs(t - t.mean()) <= 2.0 * t.std()]
    return float(np.median(keep)), int(len(t) - len(keep))


def spmv_byte_formulas(rows, nnz):
    """(reference's effective bytes today, its published formula, algorithmic bytes) of one SpMV."""
    return (8 * nnz + 4 * nnz + 4 * (rows + 1) + 8 * rows + 8 * rows, 12 * nnz + 16 * rows, 8 * nnz + 8 * rows + 8 * rows)


def spmv_headline(B, n, warmup=5, runs=10):
    """The reference's SpMV benchmark (src/main/main.cu:136-187) on the stencil5-csr operator."""
    rows, nnz = n * n, 5 * n * n - 4 * n
    op = B.Operator("stencil5-csr")
    if op.init_synthetic(n) != 0:
        raise RuntimeError("stencil5-csr synthetic init failed")
    # the operator's own staging vectors, as in the reference harness (main.cu:158-187 times run_timed, whose kernel reads and
    # writes the vectors the operator owns): x = 1; y was placed by the operator at init (output placement, csrc/device_runtime.hpp)
    op.time_device(None, None, warmup)
    ms = op.time_device(None, None, runs)
    median_ms, dropped = reference_stats(ms)
    placement = op.placement()
    # checksum gate on caller-owned vectors (the harness prints Sum(y), main.cu:176-183): x = 1 on the generator's stencil gives
    # y[i] in {1, 2, 3}, sum(y) = n^2 + 4n exactly; 3.2 GB of y come back to the host at 20 000^2
    dx, dy = B.DeviceVector(rows, fill=1.0), B.DeviceVector(rows, fill=-7.0)
    op.run_device(dx, dy)
    y = dy.to_host()
    y_sum = float(y.sum())
    del y
    dx.free(), dy.free()
    variant = op.variant()
    op.free()
    secs = median_ms / 1e3
    bytes_today, bytes_published, bytes_algorithmic = spmv_byte_formulas(rows, nnz)
    return {
        "operator": "stencil5-csr", "variant": variant, "grid": n, "median_ms": median_ms, "outliers_removed": dropped,
        "all_ms": [round(float(v), 4) for v in ms], "gflops": 2.0 * nnz / secs / 1e9,
        "effective_gbs": bytes_today / secs / 1e9, "effective_gbs_published_formula": bytes_published / secs / 1e9,
        "algorithmic_gbs": bytes_algorithmic / secs / 1e9, "frac_of_hbm_peak": bytes_algorithmic / secs / 1e9 / HBM_PEAK_GBS,
        "vs_a100_published": bytes_published / secs / 1e9 / A100_SPMV_EFFECTIVE_GBS_PUBLISHED_FORMULA, "sum_y": y_sum,
        "vectors": "the operator's own x (= 1) and y, the ones run_timed's kernel works on",
        "output_placement": None if placement is None else {"candidates_timed": placement[0], "first_candidate_over_kept": placement[1]},
    }


COMPARE_CASES = ((10000, ("stencil5-csr", "cusparse-csr")),  # BASELINE config 2: STENCIL5 against the CSR kernel
                 (15000, ("ellpack", "stencil5-ellpack", "stencil5-csr", "cusparse-csr")))  # config 5: ELLPACK against STENCIL5 and CSR
COMPARE_KEYS = {"stencil5-csr": "stencil", "cusparse-csr": "csr", "ellpack": "ell", "stencil5-ellpack": "ell_stencil"}


def algorithmic_bytes(mode, rows, nnz, width=5):
    """Algorithmic bytes of one SpMV per format: what the kernel must move, padded ELLPACK slots included."""
    return {"stencil5-csr": 8 * nnz + 16 * rows, "cusparse-csr": 12 * nnz + 4 * (rows + 1) + 16 * rows,
            "ellpack": rows * width * 12 + 16 * rows, "stencil5-ellpack": rows * width * 8 + 16 * rows}[mode]


def compare_leg(B, cases=COMPARE_CASES, warmup=5, runs=10):
    """BASELINE configs 2 and 5 in brief, on the library the CG leg just ran on: every operator through get_operator, the
    reference's rule (src/main/main.cu:136-187: x = 1, 5 warm-ups, 10 timed launches, > 2 sigma dropped, median) on the operator's
    own staging vectors, and a checksum gate on caller-owned vectors: with the generator's stencil (centre 5, neighbours -1) and
    x = 1 every y[i] is 1, 2 or 3, so sum(y) = n^2 + 4n and ||y||^2 = (n-2)^2 + 16(n-2) + 36 exactly. An operator
    whose checksum is off gets no time. Returns (flat scalars for the roofline object, per-operator records)."""
    flat, records, all_ok = {}, [], True
    for n, modes in cases:
        rows, nnz = n * n, 5 * n * n - 4 * n
        want_sum, want_sq = float(n * n + 4 * n), float((n - 2) * (n - 2) + 16 * (n - 2) + 36)
        dx, dy = B.DeviceVector(rows, fill=1.0), B.DeviceVector(rows, fill=0.0)
        try:
            for mode in modes:
                key = f"{COMPARE_KEYS[mode]}{n // 1000}k"
                rec = {"operator": mode, "grid": n}
                try:
                    op = B.Operator(mode)
                    if op.init_synthetic(n) != 0:
                        raise RuntimeError("synthetic init failed")
                    try:
                        B.lib().spmv_amd_device_fill_f64(dy.ptr, rows, -7.0)  # a launch that writes nothing cannot pass
                        if op.run_device(dx, dy) != 0:
                            raise RuntimeError("run_device failed")
                        y = dy.to_host()
                        rec["sum_y"], rec["sumsq_y"] = float(y.sum()), float(np.dot(y, y))
                        del y
                        rec["checksum_ok"] = rec["sum_y"] == want_sum and rec["sumsq_y"] == want_sq
                        if rec["checksum_ok"]:
                            op.time_device(None, None, warmup)
                            ms = op.time_device(None, None, runs)
                            med, dropped = reference_stats(ms)
                            alg = algorithmic_bytes(mode, rows, nnz)
                            rec.update(variant=op.variant(), median_ms=med, outliers_removed=dropped, algorithmic_bytes=alg,
                                       algorithmic_gbs=alg / med / 1e6, frac=alg / med / 1e6 / HBM_PEAK_GBS,
                                       effective_gbs=spmv_byte_formulas(rows, nnz)[0] / med / 1e6, gflops=2.0 * nnz / med / 1e6)
                            flat[key + "_ms"], flat[key + "_frac"] = med, rec["frac"]
                    finally:
                        op.free()
                except Exception as e:  # evidence only: the CG line stands
                    rec["error"] = repr(e)
                all_ok = all_ok and bool(rec.get("checksum_ok"))
                records.append(rec)
        finally:
            dx.free(), dy.free()
    flat["compare_checksums_ok"] = all_ok
    return flat, records


def stream_ceiling(B, rows=200_000_000, reps=20, mix="stencil5"):
    """The stream probes on the benchmark's data (csrc/stream_ceiling.hip), ~100 ms of GPU time each at 4e8 rows."""
    ms, nbytes = B.stream_ceiling(rows, warmup=3, reps=reps, mix=mix)
    med, _ = reference_stats(ms)
    what = {"stencil5": "48 B read : 8 B written per row", "read-only": "48 B read per row, one 8-byte partial per wave written"}[mix]
    return {"gbs": nbytes / (med / 1e3) / 1e9, "median_ms": med, "rows": rows, "bytes_per_launch": nbytes, "launches": reps,
            "what": what + ", coalesced 8-byte nontemporal accesses, one-wave workgroups, coefficients [-1 -1 5 -1 -1] and x = 1 "
                           "(non-zero data), no neighbour reads"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_spmv_baseline(O, rp, ci, va, n, threads):
    """Serial oracle SpMV, stencil5 path (spmv_stencil_csr_direct.cu:76-123) and CSR path
    (cg_solver_mgpu_partitioned.cu:40-56): 1 warm-up + 3 runs, median; then the threaded build."""
    rows, nnz = n * n, len(va)
    x, y = np.ones(rows), np.empty(rows)
    eff, published, alg_stencil = spmv_byte_formulas(rows, nnz)
    alg = {"stencil5": alg_stencil, "csr": eff}  # the CSR kernel really reads col_idx and row_ptr

    def timed(kind, L):
        g = n if kind == "stencil5" else None
        O.spmv_into(rp, ci, va, x, y, g, L)  # warm-up
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.spmv_into(rp, ci, va, x, y, g, L)
            ts.append(time.perf_counter() - t0)
        s = float(np.median(ts))
        return {"median_ms": s * 1e3, "runs_ms": [round(t * 1e3, 2) for t in ts], "gflops": 2.0 * nnz / s / 1e9,
                "effective_gbs": eff / s / 1e9, "effective_gbs_published_formula": published / s / 1e9,
                "algorithmic_gbs": alg[kind] / s / 1e9, "sum_y": float(y.sum())}

    rec = {"grid": n, "rows": rows, "nnz": nnz, "rule": "x = 1, 1 warm-up + 3 runs, median (BASELINE.md section 3)", "cores": 1,
           "stencil5": timed("stencil5", None), "csr": timed("csr", None)}
    try:
        L = O.omp_lib(threads)
        rec["all_cores"] = {"cores": threads, "stencil5": timed("stencil5", L), "csr": timed("csr", L)}
    except Exception as e:  # optional figure
        rec["all_cores"] = {"error": repr(e)}
    return rec


def mem_available_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def cpu_baseline(sample_grid, full_grid):
    """Oracle CG (serial C, 1 core). On the FULL workload when the host has the memory (CSR 12 B per non-zero + row pointers +
    six vectors: ~45 GB at 20 000^2, ~33 s on one EPYC core; asked for 1.6x that as MemAvailable) -- the history of that run is
    then compared with the committed golden history on the spot; else on a sample grid, cost being linear in rows: iterations/s
    at the full size = iterations/s on the sample * sample_rows / full_rows. Plus the oracle's SpMV on the same matrix.
    --cpu-sample-grid N forces a sample."""
    from oracle import oracle as O

    full_rows = full_grid * full_grid
    need = 12 * (5 * full_rows - 4 * full_grid) + 4 * (full_rows + 1) + 6 * 8 * full_rows
    have = mem_available_bytes()
    at_full_size = sample_grid is None and have >= 1.6 * need
    grid = full_grid if at_full_size else (sample_grid or min(10000, full_grid))
    rp, ci, va = O.stencil5_csr(grid)
    rows = grid * grid
    t0 = time.perf_counter()
    x, hist, res = O.cg(rp, ci, va, grid, np.ones(rows), np.zeros(rows), device_form=True)
    dt = time.perf_counter() - t0
    del x
    cpu = cpu_model()
    what = (f"the full {grid}x{grid} workload, nothing scaled" if grid == full_grid else
            f"the {grid}x{grid} stencil ({rows} rows = 1/{full_rows // rows} of the workload; host MemAvailable {have / 2**30:.0f} GiB, "
            f"{1.6 * need / 2**30:.0f} GiB wanted for the full size)" if sample_grid is None else
            f"the {grid}x{grid} stencil ({rows} rows = 1/{full_rows // rows} of the workload, --cpu-sample-grid)")
    rec = {
        "value": res.iterations / dt * rows / full_rows, "unit": "CG iterations/s" + ("" if grid == full_grid else f" (scaled by rows to {full_rows} unknowns)"),
        "cores": 1, "kind": "port", "sample": f"oracle_cg on {what}: {res.iterations} iterations in {dt:.2f} s on 1 core of {os.cpu_count()} ({cpu})",
        "grid": grid, "full_size": grid == full_grid, "seconds": dt, "iterations": int(res.iterations),
    }
    # the oracle run that was just timed IS the checker of the GPU histories: its history against the committed fixture
    # (tests/golden/known_answers.json was written by the same oracle on another machine; bit-equal unless libm / compiler differ)
    gold = parity_vs_golden(grid, hist, res.iterations)
    if gold.get("available"):
        rec["history_vs_committed_golden_max_rel_err"] = gold["max_rel_err"]
        rec["history_vs_committed_golden_ok"] = gold["ok"]
    # the same loops spread over the cores this job may use (a one-GPU box's CPU share is 16), BASELINE.md section 3
    threads = max(1, min(16, os.cpu_count() or 1))
    try:
        t0 = time.perf_counter()
        x2, hist2, res2 = O.cg_all_cores(rp, ci, va, grid, np.ones(rows), np.zeros(rows), threads)
        dt2 = time.perf_counter() - t0
        del x2
        rec["all_cores"] = {"value": res2.iterations / dt2 * rows / full_rows, "cores": threads, "kind": "port (OpenMP over the oracle's loops)",
                            "sample": f"same matrix, {res2.iterations} iterations in {dt2:.2f} s on {threads} threads"}
        rec["all_cores_value"], rec["all_cores_cores"] = rec["all_cores"]["value"], threads
    except Exception as e:  # optional figure
        rec["all_cores"] = {"error": repr(e)}
    try:
        rec["spmv"] = cpu_spmv_baseline(O, rp, ci, va, grid, threads)
        rec["spmv_stencil5_ms"], rec["spmv_csr_ms"] = rec["spmv"]["stencil5"]["median_ms"], rec["spmv"]["csr"]["median_ms"]
        rec["spmv_stencil5_effective_gbs"] = rec["spmv"]["stencil5"]["effective_gbs"]
    except Exception as e:
        rec["spmv"] = {"error": repr(e)}
    return rec


def probe_role(B, grid, P, r, full_iterations, mailbox, steps=5):
    """ONE stand-in slab (rank r of P of this grid) in THIS process, which was started for it: the complete multi-rank pipeline
    over RCCL with the rank as its own neighbour (halo ncclSend / ncclRecv on the side stream under the interior SpMV, split SpMV
    launches, both all-reduces issued with one rank), `full_iterations` iterations per solve (tolerance 0: the slab's neighbours
    being its own grid rows, the system solved is the slab mirrored at its cuts, not the global one -- only the time is used), the last
    of which counts as the converging one, as in the real solve."""
    os.environ["SPMV_AMD_SELF_NEIGHBOUR"] = os.environ["SPMV_AMD_FORCE_COLLECTIVES"] = "1"
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
    if comm is None:
        return {"error": "RCCL communicator could not be created"}
    if mailbox and not comm.mailbox_enable():
        comm.destroy()
        return {"error": "peer mailbox could not be set up"}
    slab = B.CgSlab.stencil5_as(grid, r, P, comm)
    # the rank of a real job converges in its last iteration: no direction update, no halo exchange behind it. The stand-in's
    # mirrored system would not: its last iteration is declared the converging one (timing aid, csrc/cg_slab.hip "stop_at")
    slab.set_option("stop_at", full_iterations)
    for _ in range(2):
        st = slab.solve(max_iters=full_iterations, tol=0.0)
    B.lib().spmv_amd_device_synchronize()
    wall, event, spmv_ms = [], [], 0.0
    for _ in range(steps):
        t0 = time.perf_counter()
        st = slab.solve(max_iters=full_iterations, tol=0.0)
        B.lib().spmv_amd_device_synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        event.append(st.time_total_ms)
        spmv_ms += st.time_spmv_ms
    ms = float(np.median(event))
    rec = {"as_rank": r, "neighbours": (1 if r > 0 else 0) + (1 if r < P - 1 else 0), "row_offset": slab.row_offset, "rows": slab.n_local,
           "halo_doubles": grid, "iterations": st.iterations, "ms_per_solve": ms, "wall_ms_per_solve": float(np.median(wall)),
           "us_per_iteration": ms / max(st.iterations, 1) * 1e3, "spmv_us_per_launch": spmv_ms / steps / max(st.iterations, 1) * 1e3,
           "spmv_us_per_launch_covers": "the interior rows' launch (all but one or two grid rows of the slab), every 7th launch timed, phase advancing with every solve"}
    try:  # where this slab's iteration goes: one more solve with stage-boundary events (no host syncs)
        _, tl = slab.timeline_solve(max_iters=full_iterations, tol=0.0)
        rec["stage_us"] = {k: round(v, 1) for k, v in tl.items() if k.endswith("_us")}
    except Exception as e:
        rec["stage_us"] = {"error": repr(e)}
    slab.destroy()
    comm.destroy()
    return rec


def scaling_probe(grid, full_event_ms, full_iterations, role_timeout_s=45.0):
    """What ONE GPU can say about strong scaling (N = 1 only; a projection). For P = 2, 4, 8 the real slab of the edge rank (one
    neighbour) and of a middle rank (two neighbours) of this grid -- rows [r*N/P, (r+1)*N/P), same CSR bytes, `grid`-double
    halos -- each solved in a FRESH process (probe_role). This function touches no GPU: it starts the role processes one after
    the other. Not measured: the latency of a send/recv and of an all-reduce BETWEEN devices."""
    out = {"method": "real per-rank slabs of the headline grid on one GPU, each in a fresh process of its own: full RCCL pipeline with the "
                     f"rank as its own neighbour, {full_iterations} iterations per solve (tolerance 0), median of 5 solves after 2 warm-ups",
           "clock": "the solver's timed region (HIP events: barrier -> end of the x flush), for the slabs and for the full problem",
           "projection": True, "measured_between_devices": False, "grid": grid, "full_problem_ms_per_solve": full_event_ms,
           "allreduces_per_solve": 2 * full_iterations + 1, "slabs": []}

    def run(P, r, mailbox=False):
        argv = [sys.executable, os.path.abspath(__file__), "--grid", str(grid), "--probe-role", str(P), str(r), str(full_iterations), "1" if mailbox else "0"]
        return run_child_for_record(argv, dict(os.environ), role_timeout_s, f"probe role {r} of {P}")

    # the headline path of a multi-GPU run: two ncclAllReduce launches per iteration (with one rank: their launch cost only)
    out["allreduce_path"] = "ncclAllReduce (one rank: launch cost only)"
    for P in (2, 4, 8):
        ranks = sorted({0, min(P - 1, max(1, P // 2 - 1))})  # edge rank and (P > 2) one with two neighbours: 1 of 4, 3 of 8
        roles = [run(P, r) for r in ranks]
        entry = {"gpus": P, "roles": roles, "ideal_ms_per_solve": full_event_ms / P}
        if all("ms_per_solve" in v for v in roles):
            slowest = max(v["ms_per_solve"] for v in roles)
            entry["slowest_role_ms_per_solve"] = slowest
            entry["projected_efficiency_by_allreduce_latency"] = {
                f"{lat}us": full_event_ms / (P * (slowest + out["allreduces_per_solve"] * lat / 1e3)) for lat in (0, 5, 10, 25, 50)}
        out["slabs"].append(entry)
    # the same P = 8 slabs with the peer mailbox (world = 1: its launch cost, no peer latency): what folding the two
    # all-reduces into the reduction kernels saves on this GPU
    roles = [run(8, r, mailbox=True) for r in (0, 3)]
    if all("ms_per_solve" in v for v in roles):
        out["p8_with_peer_mailbox"] = {"roles": roles, "slowest_role_ms_per_solve": max(v["ms_per_solve"] for v in roles)}
    else:
        out["p8_with_peer_mailbox"] = {"roles": roles}
    return out


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


def self_launch(args):
    """`python bench.py --gpus N` with no RANK / WORLD_SIZE around: start the N ranks ourselves. Nothing in this
    process imports torch or touches the GPU (a process that has initialised the GPU must not be replaced by or
    turned into a launcher on this pool); the ranks are ordinary children, never an exec."""
    import tempfile

    n = args.gpus
    log_process("launcher")
    port = free_port()
    argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    # rank 0's stdout goes to a file, not a pipe: a pipe nobody reads until the ranks have exited would block rank 0 in
    # write() as soon as its line outgrows the pipe buffer (64 KiB: a long residual history would do)
    out0_file = tempfile.TemporaryFile()
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPMV_AMD_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport needs it on this host driver
        env.setdefault("OMP_NUM_THREADS", "4")
        # rank 0's stdout carries the JSON line; the other ranks have nothing to say there
        procs.append(subprocess.Popen(argv, env=env, stdout=out0_file if rank == 0 else sys.stderr.fileno()))
    deadline = time.monotonic() + args.launch_timeout
    failed_at = None
    ended_by_deadline = set()
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        now = time.monotonic()
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = now  # the other ranks notice through gloo / the solver's watchdog; give them a moment to report
        if (failed_at is not None and now - failed_at > args.launch_grace) or now > deadline:
            for i, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()  # exactly the children started above
                    if failed_at is None:
                        ended_by_deadline.add(i)
            break
    codes = [p.wait() for p in procs]
    out0_file.seek(0)
    out0 = out0_file.read().decode(errors="replace")
    out0_file.close()
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    worst = max(codes, key=abs) if codes else 1
    if lines:
        # the last line rank 0 wrote: the headline, or the headline augmented by an opt-in extra leg that completed
        sys.stdout.write(lines[-1] + "\n")
        try:
            measured = json.loads(lines[-1]).get("value") is not None
        except ValueError:
            measured = False
        if measured and ended_by_deadline and all(c == 0 for i, c in enumerate(codes) if i not in ended_by_deadline):
            # the measurement was complete and printed; only an opt-in extra was still running when the deadline came
            print(f"bench.py: launcher deadline ({args.launch_timeout:.0f} s) reached after the headline line was written; "
                  f"ranks {sorted(ended_by_deadline)} ended", file=sys.stderr)
            worst = 0
    else:
        reason = f"ranks exited with {codes} and rank 0 printed no result line" + (" (launcher deadline reached)" if time.monotonic() > deadline else "")
        sys.stdout.write(json.dumps({"metric": "cg_iterations_per_second", "value": None, "unit": "CG iterations/s", "n_gpus": n,
                                     "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong",
                                     "unmeasured": reason}) + "\n")
        worst = worst or EXIT_UNMEASURED
    sys.stdout.flush()
    return worst


def cgroup_throttle_counters():
    """(periods in which this container was CPU-throttled, microseconds throttled) from cgroup v2's cpu.stat, or None. A throttled
    container stalls the solver's host thread, which sits on the critical path once per iteration (status record -> next launches):
    the difference over the timed steps tells a slow number caused by the host from one caused by the GPU."""
    try:
        rec = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(rec["nr_throttled"]), int(rec["throttled_usec"])
    except (OSError, KeyError, ValueError):
        return None


def parity_vs_golden(n, hist, iterations):
    """The residual history of the timed solves against the committed oracle fixture (tests/golden/known_answers.json, written
    by tests/golden/make_golden.py from oracle/spmv_oracle.c on a CPU: data, not code). north_star's bar: 1e-10 relative on
    every ||r_k||, and the reference's "14 iterations on every GPU count" (docs/scaling_summary.md:17)."""
    try:
        case = json.load(open(GOLDEN_ANSWERS))["cases"].get(f"{n}:5.0")
    except (OSError, ValueError) as e:
        return {"available": False, "note": f"golden fixture unreadable: {e!r}"}
    if case is None:
        return {"available": False, "note": f"no committed golden history for grid {n} (fixtures exist for 3, 81, 512, 2000, 10000, 15000, 20000)"}
    g = np.asarray(case["cg"]["history"], dtype=np.float64)
    h = np.asarray(hist, dtype=np.float64)
    k = min(len(g), len(h))
    err = float(np.max(np.abs(h[:k] - g[:k]) / np.abs(g[:k]))) if k else float("inf")
    ok = bool(len(h) == len(g) and iterations == case["cg"]["iterations"] and err <= PARITY_TOL)
    return {"available": True, "max_rel_err": err, "iterations": int(iterations), "golden_iterations": int(case["cg"]["iterations"]),
            "entries_compared": int(k), "tolerance": PARITY_TOL, "ok": ok,
            "fixture": f"tests/golden/known_answers.json cases[{n}:5.0].cg.history (CPU oracle)"}


class Unmeasured(Exception):
    """A leg cannot produce a number every rank stands behind; the reason is the same text on all ranks."""


class Ctx:
    """What every leg of this process shares: rank, the gloo group, the binding, the device."""


def setup_process(args):
    """Rendezvous (gloo), library, device. Returns a Ctx; calls nothing that needs another rank's data path."""
    c = Ctx()
    c.args = args
    c.rank = int(os.environ.get("RANK", "0"))
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("SPMV_AMD_BENCH_TEST_CRASH_RANK") == str(c.rank):  # test hook: a rank that dies before the rendezvous
        print(f"bench.py: rank {c.rank} exiting with status 7 on request (SPMV_AMD_BENCH_TEST_CRASH_RANK)", file=sys.stderr)
        os._exit(7)
    c.local_rank = int(os.environ.get("LOCAL_RANK", str(c.rank)))
    log_process("leg-child" if args.leg_only else "rank", c.rank)
    if c.world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={c.world}: running {c.world} rank(s)", file=sys.stderr)
        args.gpus = c.world

    import datetime

    import torch  # first: libspmv_amd then shares the HIP runtime torch has loaded
    import torch.distributed as dist

    c.torch, c.dist = torch, dist
    c.B = B = load_binding()
    if not os.path.exists(B.LIB_PATH):
        if c.rank == 0:
            B.build()
    # SPMV_AMD_BENCH_FORCE_DIST=1 (test hook): take the distributed path with one rank too, i.e.
    # rendezvous, unique-id broadcast and an RCCL communicator, so a 1-GPU box exercises it.
    c.multi = c.world > 1 or os.environ.get("SPMV_AMD_BENCH_FORCE_DIST") == "1"
    if c.multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if c.world == 1 and "MASTER_PORT" not in os.environ:  # SPMV_AMD_BENCH_FORCE_DIST without a launcher
            os.environ["MASTER_PORT"] = str(free_port())
        # collectives of a healthy run take milliseconds; a rank that is gone must not hold the others past the driver's limit
        dist.init_process_group("gloo", rank=c.rank, world_size=c.world,
                                timeout=datetime.timedelta(seconds=120 if args.leg_only else 180))
        dist.barrier()
    c.L = B.lib()
    return c


def agree(c, ok, reason):
    """All ranks learn whether every rank is fine; returns the failing ranks' reasons, or None."""
    if not c.multi:
        return None if ok else reason
    box = [None] * c.world
    c.dist.all_gather_object(box, None if ok else f"rank {c.rank}: {reason}")
    bad = [b for b in box if b is not None]
    return "; ".join(bad) if bad else None


def gather(c, value):
    if not c.multi:
        return [value]
    box = [None] * c.world
    c.dist.all_gather_object(box, value)
    return box


def pick_device(c):
    # SPMV_AMD_BENCH_DEVICE (test hook): put every rank on one device, so a 1-GPU box can walk the N > 1 control flow
    forced = os.environ.get("SPMV_AMD_BENCH_DEVICE")
    device = int(forced) if forced is not None else c.local_rank
    visible = c.L.spmv_amd_device_count()
    why = agree(c, visible > device, f"device {device} wanted, {visible} HIP device(s) visible")
    if why:
        raise Unmeasured(why)
    c.torch.cuda.set_device(device)
    c.L.spmv_amd_set_device(device)
    c.device_index, c.pci = c.B.current_device()


def measure_leg(c, allreduce_kind):
    """One complete measurement with one all-reduce path: communicator, slab, warm-ups, K timed solves between barriers,
    agreement of the ranks' histories, parity against the golden history, one extra solve with the stage timeline.
    allreduce_kind: "rccl" (north_star: ncclAllReduce) or "mailbox" (peer stores, csrc/mailbox.hip). Raises Unmeasured."""
    args, B, dist, torch, rank, world, multi = c.args, c.B, c.dist, c.torch, c.rank, c.world, c.multi
    n = args.grid
    allow_staged = os.environ.get("SPMV_AMD_BENCH_ALLOW_STAGED") == "1"
    comm, transport, degraded = None, "single rank (no communicator)", None
    if multi:
        box = [B.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = B.Comm.rccl(rank, world, box[0])
        # every rank must have its RCCL communicators, see all `world` ranks in them and pass the collective
        # self-test (all-reduce, neighbour send/recv, loopback on a side stream, barrier)
        why = agree(c, comm is not None, "RCCL communicator creation failed (see the [comm/rccl] line on stderr)")
        if why is None:
            got = comm.transport_ranks()
            why = agree(c, got == world, f"RCCL communicators span {got} rank(s), {world} wanted")
        if why is None:
            why = agree(c, comm.selftest() == 0, "RCCL self-test (all-reduce / neighbour send-recv / barrier) returned wrong data")
        if why is None:
            transport = "rccl"
        elif allow_staged:
            if comm is not None:
                comm.destroy()
            comm = B.Comm.staged_over_torch(rank, world, dist)
            transport = "staged over torch.distributed/gloo"
            degraded = f"host-staged transport instead of RCCL (SPMV_AMD_BENCH_ALLOW_STAGED=1): {why}"
            if rank == 0:
                print(f"bench.py: {degraded}", file=sys.stderr)
        else:
            raise Unmeasured(f"RCCL unusable, and the host-staged transport is not a substitute for the headline number: {why}")
    # The two 8-byte all-reduces per iteration. Default and headline: the transport's own, ncclAllReduce on the device scalars
    # (north_star). "mailbox": stores between the GPUs' hipIpc-mapped memory inside the reduction kernel -- set up
    # all-or-nothing behind a 2048-round self-test; the line says which one ran.
    allreduce = "none (single rank)"
    if comm is not None:
        allreduce = "ncclAllReduce" if transport == "rccl" else "host callbacks (staged)"
        if allreduce_kind == "mailbox":
            if not comm.mailbox_enable():
                comm.destroy()
                raise Unmeasured("the peer mailbox could not be set up on every rank, or failed its self-test (see the [mailbox] lines on stderr)")
            allreduce = "peer mailbox (system-scope stores into hipIpc-mapped device memory)"
        if rank == 0:
            print(f"bench.py: dot-product all-reduce: {allreduce}", file=sys.stderr)
    slab = B.CgSlab.stencil5(n, comm)
    # which loop shape the library chose: its creation check solves a few iterations in both shapes on the first slab of a
    # communicator and refuses the overlapped pipeline if the histories differ (include/spmv_amd/api.h)
    loop_shape = slab.loop_shape()
    if loop_shape.startswith("plain: the pipeline"):
        degraded = (degraded + "; " if degraded else "") + loop_shape
        if rank == 0:
            print(f"bench.py: {loop_shape}", file=sys.stderr)
    placement = slab.placement()  # set-up work, outside the timed region (csrc/cg_slab.hip, place_coefficients)
    tile_runs = slab.tile_runs()  # ditto: row-lds tiles per XCD and run, the rule's neighbours timed at creation
    setup_ms = slab.setup_ms()    # wall ms of creation's phases; none of it inside the timed region

    def barrier():
        if multi:
            dist.barrier()

    try:
        for _ in range(args.warmup):
            st = slab.solve()
        if args.leg_only and args.leg_role == "ab" and os.environ.get("SPMV_AMD_BENCH_TEST_KILL_LEG_CHILD") == "1":  # test hook: the extra leg dies mid-run
            import signal

            os.kill(os.getpid(), signal.SIGKILL)
        barrier()
        torch.cuda.synchronize()
        throttle0 = cgroup_throttle_counters()
        t0 = time.perf_counter()
        # in-loop SpMV launches are event-timed on their own stream inside the solver: every 7th launch, the phase moving on by
        # one with every solve, so the timed steps cover every iteration of the loop (each pair of events costs two barrier
        # packets next to the launch; csrc/cg_slab.hip, spmv_event_stride)
        spmv_each, event_ms, outcomes = [], [], []
        for _ in range(args.steps):
            st = slab.solve()
            spmv_each.extend(float(v) for v in slab.spmv_launch_ms())
            event_ms.append(st.time_total_ms)
            outcomes.append((int(st.iterations), int(st.converged), float(st.residual_norm).hex()))
        spmv_ms, spmv_launches = float(np.sum(spmv_each)), len(spmv_each)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        throttle1 = cgroup_throttle_counters()
        host_throttled = None if throttle0 is None or throttle1 is None else {"periods": throttle1[0] - throttle0[0], "usec": throttle1[1] - throttle0[1]}
        rank_ms = [float(v) for v in gather(c, dt / args.steps * 1e3)]  # every rank's own wall time per step: the spread is load imbalance
        if multi:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        # every timed solve is the same solve: same iteration count, same verdict, the same final residual to the bit (the parity
        # gate below looks at the LAST solve's history; a solve that ended early would be a fast solve, not a right one)
        if len(set(outcomes)) != 1:
            raise Unmeasured(f"the {args.steps} timed solves disagree on (iterations, converged, final residual): {sorted(set(outcomes))}")
        hist = slab.history()
        # every rank computes its scalars from the same all-reduced values: the histories must agree bit for bit
        hexes = gather(c, [float(v).hex() for v in hist])
        if any(h != hexes[0] for h in hexes):
            raise Unmeasured(f"ranks hold different residual histories after the timed solves: the {allreduce} all-reduce does not "
                             "deliver the same sums everywhere")
        parity = parity_vs_golden(n, hist, st.iterations)
        if parity["available"] and not parity["ok"]:
            raise Unmeasured(f"residual history of the timed solves differs from the committed golden history: max relative error "
                             f"{parity['max_rel_err']:.3e} (bar {PARITY_TOL:g}), {parity['iterations']} iterations against {parity['golden_iterations']}")
        # one more solve with stage-boundary events (outside the timed region): where an iteration's time goes on this rank
        st_t, tl = slab.timeline_solve()
        breakdown = gather(c, dict(tl, rank=rank, rows=slab.n_local, timeline_solve_ms=st_t.time_total_ms))
        leg = {"allreduce": allreduce, "allreduce_kind": allreduce_kind if comm is not None else "none", "transport": transport, "degraded": degraded,
               "dt": dt, "ms_per_step": dt / args.steps * 1e3, "iterations": st.iterations, "converged": bool(st.converged),
               "final_residual": st.residual_norm, "history": [float(v) for v in hist], "rank_ms": rank_ms, "parity_vs_golden": parity,
               "ranks_agree_on_history": True if multi else None, "breakdown": breakdown, "variant": slab.variant(),
               "local_rows": slab.n_local, "local_nnz": slab.local_nnz, "spmv_ms": spmv_ms, "spmv_launches": spmv_launches,
               "event_ms_per_solve": float(np.median(event_ms)),
               "rccl_ranks": comm.transport_ranks() if (comm is not None and transport == "rccl") else 0,
               "placement": placement, "tile_runs": tile_runs, "setup_ms": setup_ms, "loop_shape": loop_shape,
               "host_throttled": host_throttled}
    finally:
        slab.destroy()
        if comm is not None:
            comm.destroy()
    leg["devices"] = gather(c, {"rank": rank, "device": c.device_index, "pci_bus_id": c.pci})
    # yardsticks, not ceilings: what THIS run sustains for ideal streams of the slab's row count (the rate depends on it:
    # 6.2-6.4 TB/s for 2e8 / 4e8 rows). In round 4's run the loop's r update streamed 7 % faster than the 48:8 probe.
    leg["probes"] = {}
    if rank == 0 and not args.no_ceiling and leg["local_rows"] >= 1_000_000:
        for key, mix in (("mix_probe", "stencil5"), ("read_only_probe", "read-only")):
            try:
                leg["probes"][key] = stream_ceiling(B, rows=leg["local_rows"], mix=mix)
            except Exception as e:
                leg["probes"][key] = {"error": repr(e)}
    return leg


def breakdown_summary(per_rank):
    """max / min over the ranks of every stage (the reference reduces its six timers with MPI_MAX / MPI_MIN,
    cg_solver_mgpu_partitioned.cu:748-800), next to the per-rank records."""
    keys = [k for k in per_rank[0] if k not in ("rank", "rows", "iterations")]
    return {"unit": "us per iteration, average over the counted iterations of ONE extra solve with stage-boundary HIP events (no host "
                    "syncs, outside the timed region); solve_ms / timeline_solve_ms in ms; a stage runs from the end of the previous "
                    "stage to the end of its own last kernel, so waiting is inside the stage that waits; every event record is a barrier packet "
                    "on the stream (~6 us), so this solve is ~40 us per iteration slower than a timed one and the small stages are mostly "
                    "that packet (profiles/r05_slab_timeline_p8.txt)",
            "max_over_ranks": {k: max(r[k] for r in per_rank) for k in keys}, "min_over_ranks": {k: min(r[k] for r in per_rank) for k in keys},
            "per_rank": per_rank}


def run_child_for_record(argv, env, timeout_s, what):
    """Runs one evidence-only child of this file and returns the JSON object of its last `{` line. Whatever happens to the
    child -- non-zero exit, killed by a signal, no output, not done within timeout_s (it is then killed) -- comes back as a
    record with an "error" key, never as an exception: the caller's own line does not depend on it."""
    rec = None
    try:
        child = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=timeout_s)
        lines = [l for l in child.stdout.splitlines() if l.startswith("{")]
        if lines:
            try:
                rec = json.loads(lines[-1])
            except ValueError:
                rec = None
        if rec is None or child.returncode != 0:
            how = f"killed by signal {-child.returncode}" if child.returncode < 0 else f"exited with {child.returncode}"
            rec = dict(rec or {}, error=(rec or {}).get("error") or f"{what} {how}", stderr_tail=child.stderr[-600:])
    except subprocess.TimeoutExpired:
        rec = {"error": f"{what} did not finish within {timeout_s:.0f} s and was ended"}
    except Exception as e:  # evidence only
        rec = {"error": repr(e)}
    return rec


WATCHDOG_LINE = re.compile(r"\[spmv_amd watchdog\] rank \d+: (no progress for [0-9.]+ s in stage '[^']*'(?: \(CG iteration \d+\))?)")


def run_leg_children(c, kind, role, extra_env, timeout_s):
    """One measurement leg in CHILD processes, one per rank, on a fresh rendezvous port: every rank process of a multi-rank run
    (a supervisor: it never touches the GPU) starts its own child and waits for it. Whatever happens to a child -- an
    UNMEASURED exit, a watchdog exit of a wedged RCCL call, a kill from outside, the time limit -- comes back as a record:
    {"rec": the JSON object of the child's last `{` line or None, "rc": exit status, "watchdog": the library watchdog's own
    sentence if the child ended through it, else None}. The child's stderr is passed on."""
    port = [free_port() if c.rank == 0 else None]
    if c.multi:
        c.dist.broadcast_object_list(port, src=0)
    env = dict(os.environ, RANK=str(c.rank), LOCAL_RANK=str(c.local_rank), WORLD_SIZE=str(c.world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port[0]), **extra_env)
    env.pop("SPMV_AMD_BENCH_TEST_CRASH_RANK", None)
    if c.world > 1:
        env.setdefault("NCCL_DEBUG", "WARN")  # RCCL's own words on stderr when a communicator misbehaves
    # Under torch.distributed.run the ranks carry TORCHELASTIC_* variables; TORCHELASTIC_USE_AGENT_STORE makes env://
    # rendezvous CONNECT to the launcher's store at MASTER_PORT instead of creating one. The children rendezvous among
    # themselves on a fresh port, so they must not see any of that.
    for k in [k for k in env if k.startswith("TORCHELASTIC_") or k in ("GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "ROLE_WORLD_SIZE",
                                                                        "GROUP_WORLD_SIZE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")]:
        env.pop(k)
    argv = [sys.executable, os.path.abspath(__file__), "--gpus", str(c.world), "--steps", str(c.args.steps), "--warmup", str(c.args.warmup),
            "--grid", str(c.args.grid), "--leg-only", kind, "--leg-role", role] + (["--no-ceiling"] if c.args.no_ceiling else [])
    rec, rc, err = None, None, ""
    try:
        child = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=timeout_s)
        rc, err = child.returncode, child.stderr
        lines = [l for l in child.stdout.splitlines() if l.startswith("{")]
        if lines:
            try:
                rec = json.loads(lines[-1])
            except ValueError:
                rec = None
    except subprocess.TimeoutExpired as e:
        rc, err = -9, (e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or ""))
        rec = {"error": f"{role} leg did not finish within {timeout_s:.0f} s and was ended"}
    sys.stderr.write(err)
    sys.stderr.flush()
    m = WATCHDOG_LINE.search(err)
    if rc != 0 and (rec is None or "error" not in rec):
        how = f"killed by signal {-rc}" if rc < 0 else f"exited with {rc}"
        rec = dict(rec or {}, error=f"{role} leg child {how}" + (f": {m.group(1)}" if m else ""))
    return {"rec": rec, "rc": rc, "watchdog": m.group(1) if m else None}


def supervise_headline(c, kind, timeout_s):
    """The headline leg of a multi-rank run, in child processes. Returns (leg or None, reason or None, extras).
    If a rank of the first attempt ended through the library's watchdog -- the overlapped pipeline drives two RCCL communicators
    from two streams, the one construction of this solver no run between devices has exercised yet -- or the
    attempt finished with WRONG NUMBERS (ranks disagreeing on the residual history, or a history off the committed golden:
    the overlapped pipeline hands rows between streams by device flags, proven on one device only), the ranks are started ONCE more, as fresh processes on a fresh rendezvous, with SPMV_AMD_NO_OVERLAP=1 (halo exchange on the
    compute stream: the reference's own, non-overlapped shape, cg_solver_mgpu_partitioned.cu:173-231). A line measured that way
    says so: "degraded": "no-overlap after <the watchdog's sentence>", with the first attempt's outcome per rank beside it."""
    first = run_leg_children(c, kind, "headline", {}, timeout_s)
    outcomes = gather(c, {"rank": c.rank, "rc": first["rc"], "watchdog": first["watchdog"], "error": (first["rec"] or {}).get("error")})
    if all(o["rc"] == 0 for o in outcomes):
        return (first["rec"] or {}).get("leg"), None, {}
    reasons = "; ".join(f"rank {o['rank']}: {o['error']}" for o in outcomes if o["rc"] != 0)
    sentences = [o["watchdog"] for o in outcomes if o["watchdog"]]
    wrong = [o["error"] for o in outcomes if o["error"] and ("residual histories" in o["error"] or "residual history" in o["error"])]
    if c.multi and not sentences and wrong:
        sentences = [wrong[0]]  # (every rank raises the same sentence: the histories are gathered before they are compared)
    if not sentences or os.environ.get("SPMV_AMD_NO_OVERLAP") == "1":
        return None, reasons, {}
    if c.rank == 0:
        print(f"bench.py: the overlapped leg {'produced wrong numbers' if wrong and sentences[0] == wrong[0] else 'ended through the watchdog'} "
              f"({sentences[0]}); starting fresh ranks once with SPMV_AMD_NO_OVERLAP=1", file=sys.stderr)
    second = run_leg_children(c, kind, "headline", {"SPMV_AMD_NO_OVERLAP": "1"}, timeout_s)
    outcomes2 = gather(c, {"rank": c.rank, "rc": second["rc"], "watchdog": second["watchdog"], "error": (second["rec"] or {}).get("error")})
    extras = {"degraded": f"no-overlap after {sentences[0]}", "first_leg_failure": {"per_rank": outcomes}}
    if all(o["rc"] == 0 for o in outcomes2):
        return (second["rec"] or {}).get("leg"), None, extras
    return None, "after the no-overlap retry: " + "; ".join(f"rank {o['rank']}: {o['error']}" for o in outcomes2 if o["rc"] != 0) + f" (first attempt: {reasons})", extras


def other_allreduce_leg(c, kind, timeout_s):
    """--allreduce-ab: the same K steps with the OTHER all-reduce path, in child processes (one per rank, fresh rendezvous on a
    new port), started only after the headline line is on stdout: whatever happens there -- a failed set-up, a watchdog exit,
    a GPU fault in an unproven path, a kill from outside -- the headline stands. Returns rank 0's record of the child leg
    (or the reason there is none)."""
    out = run_leg_children(c, kind, "ab", {}, timeout_s)
    rec = out["rec"] or {"error": "child leg printed no record"}
    if "leg" in rec:
        leg = rec["leg"]
        rec = {k: leg[k] for k in ("allreduce", "ms_per_step", "iterations", "rank_ms", "parity_vs_golden", "ranks_agree_on_history", "rccl_ranks")} | {
            "breakdown": breakdown_summary(leg["breakdown"])}
    if c.multi:
        try:
            c.dist.barrier()
        except Exception as e:  # a rank lost in the extra leg must not take the (already printed) headline's exit status with it
            rec = dict(rec, barrier_after_leg=repr(e))
    return rec


def build_line(args, base, leg, spmv, extras, world, n, rows, nnz, compare=None):
    """The benchmark line from a finished leg (rank 0 only; touches neither the GPU nor another rank)."""
    # template argument: <kMode = 1 (SpMV + p.Ap partials)>
    kernel_symbol = {"stencil5/row-lds": "stencil5_rowlds_kernel<1>", "stencil5/row-direct": "stencil5_rowdirect_kernel<true>"}.get(leg["variant"], leg["variant"])
    transport, allreduce, degraded, dt, iterations = leg["transport"], leg["allreduce"], extras.get("degraded") or leg["degraded"], leg["dt"], leg["iterations"]

    # dominant kernel: STENCIL5 SpMV of this rank's slab, average over the launches of the timed steps
    local_rows, local_nnz = leg["local_rows"], leg["local_nnz"] if leg["local_nnz"] > 0 else None
    if local_nnz is None:  # nnz of the slab does not fit the int the info call returns (single rank, 20k)
        local_nnz = nnz // world
    alg_bytes = 8 * local_nnz + 8 * local_rows + 8 * local_rows
    avg_spmv_ms = leg["spmv_ms"] / max(leg["spmv_launches"], 1)
    achieved = alg_bytes / (avg_spmv_ms / 1e3) / 1e9 if avg_spmv_ms > 0 else 0.0
    # fabric bytes per launch from the committed PMC passes -- only if they were taken on THIS kernel source
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(tpath))
        source_hash = hashlib.sha256(open(KERNEL_SOURCE, "rb").read()).hexdigest()
        if rec.get("grid") != n or rec.get("n_gpus") != world or rec.get("kernel") != kernel_symbol:
            traffic_note = "profiles/hbm_traffic.json holds another configuration or kernel: traffic not reported"
        elif rec.get("kernel_source_sha256") != source_hash:
            traffic_note = "profiles/hbm_traffic.json is stale (csrc/spmv_kernels.hip changed since the PMC passes): traffic not reported"
        else:
            traffic = rec.get("bytes_per_launch")
            traffic_note = ("L2-to-fabric bytes per launch from separate rocprofv3 --pmc passes (profiles/hbm_traffic.json, taken on this "
                            "kernel source); Infinity-Cache hits included")
    except (OSError, ValueError):
        traffic_note = "no profiles/hbm_traffic.json"
    roofline = {
        "bound": "hbm", "kernel": kernel_symbol + " (SpMV + p.Ap partials)", "achieved": achieved, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
        "avg_launch_ms": avg_spmv_ms, "launches_timed": leg["spmv_launches"], "traffic_note": traffic_note,
    }
    for key, probe in (leg.get("probes") or {}).items():  # the stream yardsticks, measured by the process that owned the GPU
        roofline[key] = probe
        if "gbs" in probe:
            roofline[key + "_gbs"] = probe["gbs"]
    if "mix_probe_gbs" in roofline:
        roofline["frac_of_mix_probe"] = achieved / roofline["mix_probe_gbs"]

    # every streaming stage of the loop against the same peak, from rank 0's stage timeline (one extra solve, HIP events; a stage
    # includes the queue gap in front of its kernel): algorithmic bytes per row of the stage / its duration
    try:
        t0_ = leg["breakdown"][0]
        stage_bytes = {"spmv_interior_us": ("SpMV + p.Ap partials (interior rows)", 56.0), "update_r_us": ("r -= alpha Ap + r.r partials", 24.0),
                       "direction_update_us": ("p' = r + beta p (out of place)", 24.0)}
        roofline["stages"] = {k: {"what": what, "bytes_per_row": bpr, "us": t0_[k], "gbs": bpr * t0_["rows"] / (t0_[k] * 1e-6) / 1e9,
                                  "frac": bpr * t0_["rows"] / (t0_[k] * 1e-6) / 1e9 / HBM_PEAK_GBS}
                              for k, (what, bpr) in stage_bytes.items() if t0_.get(k, 0) > 0}
        its = max(int(t0_["iterations"]), 1)
        if t0_.get("final_x_flush_us", 0) > 0:
            bpr = 8.0 * its + 16.0  # one read of every direction of the window + x in + x out
            roofline["stages"]["final_x_flush_us"] = {"what": f"x = x0 + sum of {its} alpha_k p_k (deferred x update)", "bytes_per_row": bpr, "us": t0_["final_x_flush_us"],
                                                      "gbs": bpr * t0_["rows"] / (t0_["final_x_flush_us"] * 1e-6) / 1e9,
                                                      "frac": bpr * t0_["rows"] / (t0_["final_x_flush_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS}
    except Exception as e:  # evidence only
        roofline["stages"] = {"error": repr(e)}
    rates = {"mix_probe_gbs": roofline.get("mix_probe_gbs"), "read_only_probe_gbs": roofline.get("read_only_probe_gbs"),
             "update_r_in_the_loop_gbs": (roofline.get("stages") or {}).get("update_r_us", {}).get("gbs")}
    rates = {k: v for k, v in rates.items() if isinstance(v, (int, float)) and v > 0}
    if rates:
        best = max(rates, key=rates.get)
        roofline["best_stream_gbs_this_run"] = rates[best]
        roofline["best_stream_is"] = best
        roofline["frac_of_best_stream"] = achieved / rates[best]
    if leg.get("placement") is not None:
        roofline["placement"] = {"coefficient_stream": leg["placement"]}
    if leg.get("tile_runs") is not None:
        roofline["tiles_per_xcd_run"] = leg["tile_runs"]
    if leg.get("setup_ms") is not None:
        # what creating the slab cost on the wall (rank 0): set-up, as the reference's build + upload before its timed region
        roofline["setup_ms_outside_timed_region"] = {k: round(v, 1) for k, v in leg["setup_ms"].items()}
    if spmv is not None and "median_ms" in spmv:
        # BASELINE.json's FIRST metric inside an object the driver keeps: SpMV effective GB/s (fp64, 20k x 20k STENCIL5), the
        # reference's rule (src/main/main.cu:158-187: 5 warm-ups, 10 launches, > 2 sigma dropped, median) and byte formulas
        # (src/spmv/spmv_metrics.cu:85-101; the published 12 nnz + 16 rows)
        roofline["spmv_standalone"] = {k: spmv[k] for k in ("operator", "variant", "grid", "median_ms", "effective_gbs", "effective_gbs_published_formula",
                                                             "algorithmic_gbs", "gflops", "vs_a100_published")} | {"frac": spmv["frac_of_hbm_peak"]}
        # ... and as FLAT scalars: the driver's record keeps the scalars of `roofline` and `config` and drops nested objects
        # (BENCH_r05.json.parsed lost spmv_standalone and config.spmv_effective_gbs). spmv20k_* at the headline grid.
        tag = f"spmv{n // 1000}k" if n % 1000 == 0 else f"spmv{n}"
        roofline[tag + "_median_ms"] = spmv["median_ms"]
        roofline[tag + "_effective_gbs"] = spmv["effective_gbs"]
        roofline[tag + "_effective_gbs_published_formula"] = spmv["effective_gbs_published_formula"]
        roofline[tag + "_frac"] = spmv["frac_of_hbm_peak"]
        roofline[tag + "_vs_a100_published"] = spmv["vs_a100_published"]
        if spmv.get("sum_y") is not None:
            roofline[tag + "_checksum_ok"] = spmv["sum_y"] == float(n * n + 4 * n)
        if spmv.get("output_placement") is not None:
            roofline.setdefault("placement", {})["spmv_output_vector"] = spmv["output_placement"]

    if compare is not None:
        # BASELINE configs 2 and 5 (10 000^2 STENCIL5 vs CSR; 15 000^2 ELLPACK vs STENCIL5 vs CSR), flat for the same reason:
        # <operator><grid>k_ms = median kernel ms by the reference's rule, <...>_frac = algorithmic bytes / that / 8 TB/s
        roofline.update(compare[0])
    devices = leg.get("devices")
    if True:
        value = args.steps * iterations / dt
        headline = world in A100_CG_ITERS_PER_S and n == 20000 and degraded is None
        out = dict(base, value=value, ms_per_step=dt / args.steps * 1e3,
                   vs_baseline=value / A100_CG_ITERS_PER_S[world] if headline else None,
                   baseline_note="reference's published CG iters/s on the same problem at the same GPU count, A100-SXM4-80GB (BASELINE.md); no MI355X number is published",
                   config={"workload": f"CG on the {n}x{n} 5-point stencil ({rows} unknowns, {nnz} nnz), b=1, x0=0, tol 1e-6",
                           # BASELINE's first metric as scalars (reference formula spmv_metrics.cu:85-101 / the published 12 nnz + 16 rows)
                           "spmv_effective_gbs": None if spmv is None or "effective_gbs" not in spmv else spmv["effective_gbs"],
                           "spmv_effective_gbs_published_formula": None if spmv is None or "effective_gbs" not in spmv else spmv["effective_gbs_published_formula"],
                           "spmv_median_ms": None if spmv is None or "median_ms" not in spmv else spmv["median_ms"],
                           "grid": n, "unknowns": rows, "nnz": nnz, "partition": f"{world} row slab(s)", "transport": transport,
                           "loop_shape": leg.get("loop_shape"),
                           # CPU throttling of this container during the timed steps (cgroup cpu.stat; rank 0's view): a stalled host
                           # thread leaves the GPU idle once per iteration -- non-zero values explain a slow `value`, zero rules the host out
                           "host_throttled_periods_in_timed_steps": None if not leg.get("host_throttled") else leg["host_throttled"]["periods"],
                           "host_throttled_usec_in_timed_steps": None if not leg.get("host_throttled") else leg["host_throttled"]["usec"],
                           "iterations_per_solve": iterations, "converged": leg["converged"], "final_residual": leg["final_residual"],
                           # the solver's own clock (HIP events around the reference's timed region, median over the timed steps, rank 0):
                           # `value` is the wall clock around the K steps, which also holds the host's work between two solves
                           "solver_clock_ms_per_solve": leg.get("event_ms_per_solve"),
                           "residual_history": leg["history"]},
                   transport=transport, allreduce=allreduce, ranks_agree_on_history=leg["ranks_agree_on_history"], parity_vs_golden=leg["parity_vs_golden"],
                   rccl_ranks=leg["rccl_ranks"], devices=devices, rank_ms_per_step=leg["rank_ms"], breakdown=breakdown_summary(leg["breakdown"]),
                   launched_by="bench.py (self-launched ranks)" if os.environ.get("SPMV_AMD_BENCH_SELF_LAUNCHED") == "1" else
                   ("external launcher (RANK/WORLD_SIZE in the environment)" if "RANK" in os.environ else "single process"),
                   roofline=roofline)
        if leg.get("placement") is not None:
            out["placement"] = leg["placement"]
        if degraded:
            out["degraded"] = degraded
        if extras.get("first_leg_failure"):
            out["first_leg_failure"] = extras["first_leg_failure"]
        if spmv is not None:
            out["spmv"] = spmv
        if compare is not None:
            out["compare"] = compare[1]

    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=20000, help="n of the n x n stencil (default: the 400M-unknown headline)")
    ap.add_argument("--cpu-sample-grid", type=int, default=None,
                    help="grid of a CPU-baseline SAMPLE, scaled by rows (default: the full workload when the host's MemAvailable allows -- ~33 s on one "
                         "core at 20000 -- else 10000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spmv", action="store_true", help="skip the N=1 SpMV headline leg")
    ap.add_argument("--no-compare", action="store_true", help="skip the N=1 leg of BASELINE configs 2 and 5 (10k STENCIL5 vs CSR, 15k ELLPACK vs STENCIL5 vs CSR)")
    ap.add_argument("--no-scaling-probe", action="store_true", help="skip the N=1 one-GPU strong-scaling probe")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the in-run stream-ceiling probe")
    ap.add_argument("--allreduce-ab", action="store_true",
                    help="multi-rank runs, opt-in: after the headline line is printed, repeat the K steps with the other all-reduce path in child processes")
    ap.add_argument("--no-allreduce-ab", action="store_true", help="(accepted for older command lines; the second leg is off unless --allreduce-ab)")
    ap.add_argument("--ab-timeout", type=float, default=180.0, help="--allreduce-ab: seconds before the child leg is ended")
    ap.add_argument("--probe-timeout", type=float, default=240.0, help="N = 1: seconds before the scaling-probe child is ended")
    ap.add_argument("--launch-timeout", type=float, default=420.0,
                    help="self-launch only: seconds before the parent ends the ranks (below the driver's 600 s limit for this file)")
    ap.add_argument("--launch-grace", type=float, default=90.0,
                    help="self-launch only: seconds the other ranks get to report after one rank has exited non-zero")
    ap.add_argument("--scaling-probe-only", nargs=2, metavar=("FULL_MS", "FULL_ITERATIONS"), default=None,
                    help="(internal) run only the scaling probe and print its JSON object")
    ap.add_argument("--probe-role", nargs=4, metavar=("P", "R", "ITERATIONS", "MAILBOX"), default=None,
                    help="(internal) one stand-in slab of the scaling probe in this process; prints its record")
    ap.add_argument("--leg-only", choices=["rccl", "mailbox"], default=None,
                    help="(internal) run one measurement leg with that all-reduce path and print its record")
    ap.add_argument("--leg-role", choices=["headline", "ab"], default="ab", help="(internal) which leg of the parent this child is")
    ap.add_argument("--leg-timeout", type=float, default=170.0,
                    help="multi-rank runs: seconds before an attempt at the headline leg (child processes) is ended; two attempts at most")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0 or args.gpus < 1 or args.grid < 2:
        ap.error("--steps >= 1, --warmup >= 0, --gpus >= 1 and --grid >= 2 are required")

    if args.scaling_probe_only is None and args.probe_role is None and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    # The library prints progress lines the way the reference harness does ("[stencil5-csr] Cleaning up", ...):
    # send everything written to fd 1 from here on to stderr and keep the real stdout for the one JSON line.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    if args.scaling_probe_only is not None:  # starts the role processes; no torch, no GPU here
        log_process("probe-child")
        emit(scaling_probe(args.grid, float(args.scaling_probe_only[0]), int(args.scaling_probe_only[1])))
        return
    if args.probe_role is not None:
        B = load_binding().use_lab()  # stand-in slabs, self-neighbour communicator, stop_at: the LAB build (include/spmv_amd/lab.h)
        B.lib()
        B.require_gpu()
        B.lib().spmv_amd_set_device(0)
        log_process("probe-role")
        P, r, its, mailbox = (int(v) for v in args.probe_role)
        emit(probe_role(B, args.grid, P, r, its, mailbox != 0))
        return

    c = setup_process(args)
    rank, world, multi, B, dist = c.rank, c.world, c.multi, c.B, c.dist
    n = args.grid
    rows, nnz = n * n, 5 * n * n - 4 * n
    base = {"metric": "cg_iterations_per_second", "unit": "CG iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "strong", "dtype": "f64", "data": "synthetic"}

    def leave(status):
        if multi:
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception:
                pass
        sys.exit(status)

    def give_up(reason):
        """No measurement: say so on the one line the driver reads and leave with a non-zero status."""
        if rank == 0:
            print(f"bench.py: UNMEASURED -- {reason}", file=sys.stderr)
            if args.leg_only:
                emit({"error": reason})
            else:
                emit(dict(base, value=None, ms_per_step=None, vs_baseline=None, unmeasured=reason,
                          config={"workload": f"CG on the {n}x{n} 5-point stencil, b=1, x0=0, tol 1e-6", "grid": n, "partition": f"{world} row slab(s)"}))
        leave(EXIT_UNMEASURED)

    # north_star's path (RCCL send/recv halos + ncclAllReduce) is the headline; SPMV_AMD_BENCH_ALLREDUCE=mailbox swaps the legs
    headline_kind = "mailbox" if os.environ.get("SPMV_AMD_BENCH_ALLREDUCE", "rccl") == "mailbox" else "rccl"
    extras = {}
    if args.leg_only:  # a supervisor's child: the process that owns the GPU. One leg, one record, nothing else
        try:
            pick_device(c)
            leg = measure_leg(c, args.leg_only)
        except Unmeasured as e:
            give_up(str(e))
        if rank == 0:
            emit({"leg": leg})
        leave(0)
    elif multi:
        # A rank process of a multi-rank run is a SUPERVISOR: rendezvous (gloo) and bookkeeping here, everything that touches
        # the GPU in a child process -- so that a leg whose ranks end through the library's watchdog can be followed, once, by
        # fresh ranks with the halo exchange on the compute stream (supervise_headline), never by a re-exec and never silently.
        leg, reason, extras = supervise_headline(c, headline_kind, args.leg_timeout)
        if leg is None and rank == 0 and reason is None:
            reason = "the leg's rank 0 printed no record"
        verdict = [reason if rank == 0 else None]
        dist.broadcast_object_list(verdict, src=0)
        if verdict[0] is not None:
            if rank == 0 and extras:
                print(f"bench.py: {json.dumps(extras)}", file=sys.stderr)
            give_up(verdict[0])
        shared = [{k: leg[k] for k in ("transport",)} if rank == 0 else None]
        dist.broadcast_object_list(shared, src=0)
        if rank != 0:
            leg = shared[0]
    else:
        # single process: the headline leg FIRST, so that the slab is the first large allocation of the process and gets the
        # device's free memory in one piece (what a solver run by itself gets; behind another leg's allocations and frees the
        # same solve measured 0.5 % slower, profiles/r04_class_pool_clean_process.txt against r04_class_pool_1gib_chunks.txt)
        try:
            pick_device(c)
            leg = measure_leg(c, headline_kind)
        except Unmeasured as e:
            give_up(str(e))
    transport = leg["transport"]

    spmv = None
    if world == 1 and not multi and not args.no_spmv:
        try:
            spmv = spmv_headline(B, n)
        except Exception as e:  # the CG measurement above stands whatever happens to this leg
            spmv = {"error": repr(e)}

    compare = None
    if world == 1 and not multi and not args.no_compare:
        try:
            compare = compare_leg(B)
        except Exception as e:  # evidence only
            compare = ({"compare_error": repr(e)[:200]}, [])

    out = build_line(args, base, leg, spmv, extras, world, n, rows, nnz, compare) if rank == 0 else None
    iterations = leg["iterations"] if rank == 0 else None

    if multi:
        # A multi-rank run has nothing left to measure: the line goes out NOW, while every rank process is still alive and
        # before anything optional can start. A default run ends here with exactly `world` processes on the GPUs.
        if rank == 0:
            emit(out)
        if args.allreduce_ab and transport == "rccl":
            # opt-in second leg: the same K steps with the other all-reduce path, in child processes; if it returns, the
            # line is written once more with the comparison added (a reader takes the last line; the first one stands alone)
            other = "mailbox" if headline_kind == "rccl" else "rccl"
            child = other_allreduce_leg(c, other, timeout_s=args.ab_timeout)
            if rank == 0:
                out["allreduce_ab"] = {headline_kind: leg["ms_per_step"], other: child.get("ms_per_step"), "headline": headline_kind,
                                       "unit": "ms per step (one CG solve), max over ranks, same K steps after the same warm-ups",
                                       "other_leg": child,
                                       "note": "the other leg runs in child processes after the headline line has been written: "
                                               "its failure cannot touch `value`"}
                emit(out)
        leave(0)

    # single process (N = 1): the evidence-only extras, each bounded, then the ONE line
    if rank == 0 and not args.no_scaling_probe and n >= 8192:
        # in a child process: whatever happens to it, the benchmark line is printed
        out["scaling_probe"] = run_child_for_record(
            [sys.executable, os.path.abspath(__file__), "--grid", str(n), "--scaling-probe-only", repr(leg["event_ms_per_solve"]), str(iterations)],
            dict(os.environ), args.probe_timeout, "scaling probe")
    if rank == 0 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_grid, n)
        except Exception as e:  # the checker failing to build or load must not take the GPU measurement with it
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        emit(out)
    leave(0)


if __name__ == "__main__":
    main()
