"""The harness binaries (cuda-spmv-benchmark_amd/apps): same command lines, checksums and JSON keys
as the reference's spmv_bench / cg_solver / cg_solver_mgpu_stencil."""
import json
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, ROOT

BIN = os.path.join(ROOT, "cuda-spmv-benchmark_amd", "bin")


def run(args, **kw):
    return subprocess.run(args, capture_output=True, text=True, timeout=600, **kw)


def test_binaries_exist_and_print_usage(B):
    for exe in ("spmv_bench", "cg_solver", "cg_solver_mgpu_stencil", "generate_matrix"):
        path = os.path.join(BIN, exe)
        assert os.path.exists(path), f"{exe} not built (make -C cuda-spmv-benchmark_amd)"
        out = run([path])
        assert out.returncode != 0 and "Usage" in (out.stdout + out.stderr)
    out = run([os.path.join(BIN, "spmv_bench"), os.path.join(GOLDEN, "example81x81.mtx"), "--mode=nonsense"])
    assert out.returncode != 0 and "Unknown mode" in out.stderr  # modes are validated before the matrix is read


@pytest.mark.gpu
def test_spmv_bench_on_shipped_matrix(tmp_path):
    js = tmp_path / "res.json"
    out = run([os.path.join(BIN, "spmv_bench"), os.path.join(GOLDEN, "example81x81.mtx"),
               "--mode=cusparse-csr,stencil5-csr,ellpack", f"--json={js}", f"--csv={tmp_path / 'res.csv'}"])
    assert out.returncode == 0, out.stdout + out.stderr
    sums = re.findall(r"Sum\(y\):\s+(\S+)", out.stdout)
    assert len(sums) == 3 and all(float(s) == -52164.0 for s in sums)
    for mode in ("cusparse-csr", "stencil5-csr", "ellpack"):
        d = json.load(open(tmp_path / f"res_{mode}.json"))
        assert d["benchmark"]["operator"] == mode and d["benchmark"]["matrix"]["nnz"] == 32481
        assert d["benchmark"]["performance"]["execution_time_ms"] > 0  # key scraped by scripts/run_all.sh
        assert d["benchmark"]["validation"]["sum_y"] == -52164.0
        assert d["gpu"]["name"] and d["gpu"]["multiprocessor_count"] > 0 and d["system"]["cpu_model"]
        assert os.path.exists(tmp_path / f"res_{mode}.csv")


@pytest.mark.gpu
def test_cg_solver_binary(tmp_path):
    out = run([os.path.join(BIN, "cg_solver"), "--stencil=81", "--mode=stencil5-csr,cusparse-csr", f"--json={tmp_path / 'cg.json'}"])
    assert out.returncode == 0, out.stdout + out.stderr
    assert re.findall(r"Converged: YES in (\d+) iterations", out.stdout) == ["18", "18"]
    d = json.load(open(tmp_path / "cg_stencil5-csr.json"))
    assert d["convergence"]["iterations"] == 18 and d["timing"]["median_ms"] > 0 and d["statistics"]["valid_runs"] >= 3
    host = run([os.path.join(BIN, "cg_solver"), os.path.join(GOLDEN, "example81x81.mtx"), "--host"])
    assert host.returncode == 0 and "Converged: YES in 40 iterations" in host.stdout


@pytest.mark.gpu
def test_cg_solver_mgpu_binary_single_rank(tmp_path):
    js = tmp_path / "m.json"
    out = run([os.path.join(BIN, "cg_solver_mgpu_stencil"), "--stencil=300", "--runs=5", f"--json={js}"])
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Converged: YES in 17 iterations" in out.stdout
    d = json.load(open(js))
    assert d["num_gpus"] == 1 and d["convergence"]["iterations"] == 17 and d["timing"]["median_ms"] > 0
    mtx = run([os.path.join(BIN, "cg_solver_mgpu_stencil"), os.path.join(GOLDEN, "example81x81.mtx"), "--runs=3"])
    assert mtx.returncode == 0 and "Converged: YES in 40 iterations" in mtx.stdout
    s = float(re.search(r"Sum\(x\):\s+(\S+)", mtx.stdout).group(1))
    assert abs(s - (-8.2608388842537738e+02)) < 1e-8


def test_generate_matrix_binary_matches_reference_generator(O, tmp_path):
    mine = tmp_path / "mine.mtx"
    assert run([os.path.join(BIN, "generate_matrix"), "12", str(mine)]).returncode == 0
    assert mine.read_text().split("\n")[1] == "% STENCIL_GRID_SIZE 12"
    ref_gen = O.REF_GEN_PATH
    if os.path.exists(ref_gen):  # the reference's own generate_matrix, compiled in place (oracle/_ref)
        theirs = tmp_path / "theirs.mtx"
        assert run([ref_gen, "12", str(theirs)]).returncode == 0
        assert mine.read_bytes() == theirs.read_bytes()


def test_trace_gaps_finds_the_last_solve_by_every_anchor(tmp_path):
    """tools/trace_gaps.py (per-kernel busy / idle table of the last solve in a rocprofv3 kernel trace) must find the solve
    whichever way it starts: the fused initial residual (mode-2 row-lds launches, the default), cg_init_residual (slabs
    that do not run row-lds), or only cg_scalars_init. CPU only: synthetic traces."""
    import subprocess
    import sys
    from conftest import ROOT

    def trace(first_kernels):
        rows, t = [], 1000
        def add(name, dur):
            nonlocal t
            rows.append((t, t + dur, name))
            t += dur + 5
        for solve in range(2):
            for name in first_kernels:
                add(name, 400)
            add("spmv_amd::(anonymous namespace)::cg_scalars_init_kernel(spmv_amd::CgScalars*, double*)", 4)
            for it in range(3):
                add("void spmv_amd::(anonymous namespace)::stencil5_rowlds_kernel<1, true>(spmv_amd::SlabCsr, ...)", 450)
                add("spmv_amd::(anonymous namespace)::reduce_slices_kernel(double const*, int, int, double*, int const*)", 6)
                add("spmv_amd::(anonymous namespace)::cg_update_r_kernel(unsigned long, ...)", 180)
                add("spmv_amd::(anonymous namespace)::cg_update_p_ring_kernel(unsigned long, ...)", 175)
        path = tmp_path / f"trace_{abs(hash(tuple(first_kernels)))}_kernel_trace.csv"
        with open(path, "w") as f:
            f.write('"Kernel_Name","Start_Timestamp","End_Timestamp"\n')
            for s, e, name in rows:
                f.write(f'"{name}",{s},{e}\n')
        return path

    fused = ["void spmv_amd::(anonymous namespace)::stencil5_rowlds_kernel<2, true>(spmv_amd::SlabCsr, ...)",
             "void spmv_amd::(anonymous namespace)::stencil5_rowlds_kernel<2, false>(spmv_amd::SlabCsr, ...)"]
    unfused = ["void spmv_amd::(anonymous namespace)::stencil5_rowdirect_kernel<1, false>(spmv_amd::SlabCsr, ...)",
               "spmv_amd::(anonymous namespace)::cg_init_residual_kernel(unsigned long, ...)"]
    bare = ["spmv_amd::(anonymous namespace)::reduce_partials_kernel(double const*, ...)"]
    for first, kernels_in_solve in ((fused, 2 + 1 + 12), (unfused, 2 + 1 + 12), (bare, 1 + 1 + 12)):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_gaps.py"), str(trace(first)), "3", "-1"], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        assert f"solve -1 of the trace: {kernels_in_solve} kernels" in out.stdout, out.stdout
    # the default is the last solve BUT ONE (the last one of a probe's trace carries stage-boundary events): same length here
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_gaps.py"), str(trace(fused)), "3"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "solve -2 of the trace: 15 kernels" in out.stdout, out.stdout + out.stderr


def test_bench_parity_gate_logic_on_the_cpu(golden):
    """bench.py's parity_vs_golden: the committed history passes; one residual off by 1e-9, a missing residual or another
    iteration count does not; a grid without a committed history is reported as such (and does not block a run)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for grid in (512, 2000, 10000, 20000):
        g = golden["cases"][f"{grid}:5.0"]["cg"]
        ok = bench.parity_vs_golden(grid, g["history"], g["iterations"])
        assert ok["available"] and ok["ok"] and ok["max_rel_err"] == 0.0 and ok["entries_compared"] == g["iterations"] + 1
        off = list(g["history"])
        off[len(off) // 2] *= 1.0 + 1e-9
        assert not bench.parity_vs_golden(grid, off, g["iterations"])["ok"]
        near = [v * (1.0 + 5e-11) for v in g["history"]]
        assert bench.parity_vs_golden(grid, near, g["iterations"])["ok"]
        assert not bench.parity_vs_golden(grid, g["history"][:-1], g["iterations"] - 1)["ok"]
        assert not bench.parity_vs_golden(grid, g["history"], g["iterations"] + 1)["ok"]
    none = bench.parity_vs_golden(777, [1.0, 0.5], 1)
    assert none["available"] is False and "no committed golden" in none["note"]
    # the per-rank breakdown summary: max / min over ranks of every stage
    s = bench.breakdown_summary([{"rank": 0, "rows": 10, "iterations": 14.0, "iteration_us": 900.0, "update_r_us": 190.0},
                                 {"rank": 1, "rows": 10, "iterations": 14.0, "iteration_us": 950.0, "update_r_us": 185.0}])
    assert s["max_over_ranks"] == {"iteration_us": 950.0, "update_r_us": 190.0} and s["min_over_ranks"]["iteration_us"] == 900.0
    assert [r["rank"] for r in s["per_rank"]] == [0, 1]


def test_bench_line_carries_every_baseline_number_as_a_flat_scalar():
    """The driver's record keeps the SCALARS of `roofline` / `config` / `cpu_baseline` and drops nested objects (BENCH_r05.json lost
    roofline.spmv_standalone and config.spmv_effective_gbs that way). BASELINE's first metric (20k STENCIL5 SpMV: median ms,
    effective GB/s by both of the reference's formulas, frac) and the numbers of configs 2 and 5 must therefore be flat keys."""
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module_flat", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n, rows, nnz = 20000, 20000 * 20000, 5 * 20000 * 20000 - 4 * 20000
    # the byte counts SURVEY 8d states
    assert bench.algorithmic_bytes("stencil5-csr", rows, nnz) == 22_399_360_000
    assert bench.algorithmic_bytes("cusparse-csr", 15000 ** 2, 5 * 15000 ** 2 - 60000) == 17_999_280_004
    assert bench.algorithmic_bytes("ellpack", 15000 ** 2, 0) == 17_100_000_000
    assert bench.spmv_byte_formulas(rows, nnz)[:2] == (31_999_040_004, 30_399_040_000)
    leg = {"variant": "stencil5/row-lds", "transport": "single rank (no communicator)", "allreduce": "none (single rank)", "degraded": None, "dt": 1.04,
           "iterations": 14, "local_rows": rows, "local_nnz": -1, "spmv_ms": 36.4, "spmv_launches": 10, "probes": {"mix_probe": {"gbs": 6177.0}},
           "breakdown": [{"rank": 0, "rows": rows, "iterations": 14.0, "spmv_interior_us": 3650.0, "update_r_us": 1440.0, "direction_update_us": 1490.0,
                          "final_x_flush_us": 8300.0}],
           "placement": None, "tile_runs": None, "setup_ms": None, "converged": True, "final_residual": 0.0108, "event_ms_per_solve": 103.9,
           "history": [2e4, 1e-2], "ranks_agree_on_history": None, "parity_vs_golden": {"ok": True}, "rccl_ranks": 0, "devices": [], "rank_ms": [104.0]}
    secs = 3.64e-3
    spmv = {"operator": "stencil5-csr", "variant": "stencil5/row-lds", "grid": n, "median_ms": 3.64, "effective_gbs": 31_999_040_004 / secs / 1e9,
            "effective_gbs_published_formula": 30_399_040_000 / secs / 1e9, "algorithmic_gbs": 22_399_360_000 / secs / 1e9, "gflops": 2 * nnz / secs / 1e9,
            "vs_a100_published": 3.5, "frac_of_hbm_peak": 22_399_360_000 / secs / 1e9 / 8000.0, "sum_y": float(n * n + 4 * n), "output_placement": None}
    compare = ({"stencil10k_ms": 0.93, "stencil10k_frac": 0.75, "csr10k_ms": 1.38, "csr10k_frac": 0.72, "stencil15k_ms": 2.1, "stencil15k_frac": 0.75,
                "csr15k_ms": 3.2, "csr15k_frac": 0.7, "ell15k_ms": 2.97, "ell15k_frac": 0.72, "ell_stencil15k_ms": 1.98, "ell_stencil15k_frac": 0.79,
                "compare_checksums_ok": True}, [{"operator": "stencil5-csr", "grid": 10000}])
    args = argparse.Namespace(steps=10, warmup=3)
    base = {"metric": "cg_iterations_per_second", "unit": "CG iterations/s", "n_gpus": 1, "steps": 10, "warmup": 3}
    line = bench.build_line(args, base, leg, spmv, {}, 1, n, rows, nnz, compare)

    def scalars(obj):  # what the driver keeps of an object
        return {k: v for k, v in obj.items() if isinstance(v, (int, float, str, bool)) or v is None}

    roof, conf = scalars(line["roofline"]), scalars(line["config"])
    for key in ("spmv20k_median_ms", "spmv20k_effective_gbs", "spmv20k_effective_gbs_published_formula", "spmv20k_frac", "spmv20k_checksum_ok",
                "stencil10k_ms", "stencil10k_frac", "csr10k_ms", "csr10k_frac", "stencil15k_ms", "stencil15k_frac", "csr15k_ms", "csr15k_frac",
                "ell15k_ms", "ell15k_frac", "ell_stencil15k_ms", "ell_stencil15k_frac", "compare_checksums_ok", "frac", "achieved", "peak", "traffic"):
        assert key in roof, key
    assert roof["spmv20k_median_ms"] == 3.64 and abs(roof["spmv20k_effective_gbs"] - 8790.9) < 0.1 and roof["spmv20k_checksum_ok"] is True
    assert abs(roof["spmv20k_frac"] - 0.7692) < 1e-4 and roof["bound"] == "hbm" and roof["unit"] == "GB/s"
    assert conf["spmv_effective_gbs"] == roof["spmv20k_effective_gbs"] and conf["spmv_median_ms"] == 3.64
    assert json.dumps(line)  # one JSON line


def test_reference_mpi_main_finds_its_ranks_and_fails_loudly_without_gpus():
    """The reference's MPI main (oracle/_ref/ref_cg_solver_mgpu: src/main/cg_solver_mgpu_stencil.cu minus its nsys capture window)
    under `mpiexec -np 2` with no GPU in sight: every rank loads the matrix, calls cg_solve_mgpu_partitioned with nothing but
    MPI_Init around it, and the LIBRARY finds rank and size through the process's own MPI library (csrc/mpi_bootstrap.cpp: dlsym,
    no link-time MPI dependency) -- as the reference's solver does (cg_solver_mgpu_partitioned.cu:240-259) -- then stops the way
    the reference's cudaSetDevice(rank) would: a message naming rank, size and the devices it found, non-zero exit."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_cg_solver_mgpu")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not os.path.exists(exe) or not os.path.exists(mpiexec):
        pytest.skip("oracle/_ref/ref_cg_solver_mgpu not built, or no mpiexec (reference sources / MPI absent)")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([mpiexec, "-np", "2", exe, os.path.join(GOLDEN, "example81x81.mtx")], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0, out.stdout + out.stderr
    # (rank 0's own progress lines sit in its stdio buffer when the launcher ends it because the other rank has exited:
    # what must be there is the library's sentence on the unbuffered stderr)
    assert "Loading matrix" in out.stdout
    # (the launcher ends the other rank as soon as the first one has exited: at least one of them got its sentence out)
    assert any(f"[cg-mgpu] rank {rank} of 2 (MPI: MPICH ABI): 0 HIP device(s) visible, one per rank is required" in out.stderr for rank in (0, 1)), out.stderr
    # the library itself has no MPI dependency
    ldd = subprocess.run(["ldd", os.path.join(ROOT, "cuda-spmv-benchmark_amd", "lib", "libspmv_amd.so")], capture_output=True, text=True).stdout
    assert "libmpi" not in ldd
