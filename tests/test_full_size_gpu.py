"""BASELINE.json's full-size configurations, checked through size-independent properties (the
oracle cannot run 400 M rows inside a test): exact checksums of A*1, linearity, agreement of the
three operators, the published 14 CG iterations, and the committed n = 10^4 golden history."""
import numpy as np
import pytest

from conftest import hist_err

pytestmark = pytest.mark.gpu


def device_checks(B, op, n):
    """y = A*1: interior rows 1, edges 2, corners 3 -> sum = n^2 + 4n, sum of squares closed form.
    Returned through two CG-style reductions on the device would need more API; rows <= 2.25e8 fit
    a host copy comfortably (1.8 GB), 4e8 too (3.2 GB)."""
    rows = n * n
    dx, dy = B.DeviceVector(rows, fill=1.0), B.DeviceVector(rows, fill=-1.0)
    assert op.run_device(dx, dy) == 0
    y = dy.to_host()
    dx.free(), dy.free()
    return y


@pytest.mark.parametrize("mode,n", [("stencil5-csr", 10000), ("cusparse-csr", 10000), ("ellpack", 15000),
                                    ("stencil5-ellpack", 15000), ("stencil5-csr", 20000)])
def test_spmv_full_size_checksums(B, mode, n):
    op = B.Operator(mode)
    assert op.init_synthetic(n) == 0
    y = device_checks(B, op, n)
    # exact: every entry is a small integer
    assert y.sum() == n * n + 4 * n
    assert float(np.dot(y, y)) == (n - 2) ** 2 + 16 * (n - 2) + 36
    counts = np.bincount(y.astype(np.int64), minlength=4)
    assert list(counts[:4]) == [0, (n - 2) ** 2, 4 * (n - 2), 4]
    # structure of the answer: corners 3, edges 2
    assert y[0] == 3 and y[n - 1] == 3 and y[-1] == 3 and y[1] == 2 and y[n] == 2 and y[n + 1] == 1
    op.free()


MAX_GRID = 20724  # largest n with 5 n^2 - 4 n = 2 147 337 984 <= 2^31 - 1: the reference's int CSR ends here


@pytest.mark.parametrize("mode", ["stencil5-csr", "cusparse-csr", "stencil5-ellpack"])
def test_largest_grid_32_bit_csr_indices_allow(B, mode):
    """Maximum size: n = 20 724 (4.29e8 rows, nnz 145 663 short of 2^31). Every byte offset and CSR position must be
    computed in 64 bits and still land in int32 arrays; y = A*1 has its exact closed-form checksums. One more column
    (n = 20 725) does not fit and must be refused with an error code, not attempted."""
    n = MAX_GRID
    assert 5 * n * n - 4 * n <= 2**31 - 1 < 5 * (n + 1) ** 2 - 4 * (n + 1)
    op = B.Operator(mode)
    assert op.init_synthetic(n) == 0
    y = device_checks(B, op, n)
    assert y.sum() == n * n + 4 * n
    counts = np.bincount(y.astype(np.int64), minlength=4)
    assert list(counts[:4]) == [0, (n - 2) ** 2, 4 * (n - 2), 4]
    assert y[0] == 3 and y[-1] == 3 and y[n * n - n] == 3 and y[n * n - 2] == 2 and y[n * (n - 1) - 1] == 2
    op.free()
    assert op.init_synthetic(n + 1) != 0


def test_largest_grid_through_the_slab_solver(B):
    """The same maximum through the CG slab solver (in-HBM generator, launch plans, reductions over 3.36 M partials):
    five iterations, residuals strictly decreasing from ||b|| = n, and the generator refuses n + 1."""
    n = MAX_GRID
    slab = B.CgSlab.stencil5(n)
    st = slab.solve(max_iters=5)
    h = slab.history()
    assert st.iterations == 5 and h[0] == float(n) and np.all(np.diff(h) < 0)
    slab.destroy()
    with pytest.raises(RuntimeError):
        B.CgSlab.stencil5(n + 1)


def test_spmv_full_size_linearity_and_operator_agreement(B):
    """A(a*u + b*v) == a*A(u) + b*A(v) to rounding, and stencil5-csr == cusparse-csr == ellpack on the
    same random x at 10k (the reference's own cross-check, tests/test_wrapper_basic.cpp:159-193)."""
    n = 10000
    rows = n * n
    rng = np.random.default_rng(7)
    u, v = rng.standard_normal(rows), rng.standard_normal(rows)
    ys = {}
    for mode in ("stencil5-csr", "cusparse-csr", "ellpack"):
        op = B.Operator(mode)
        assert op.init_synthetic(n) == 0
        yu, _ = op.run_timed(u)
        if mode == "stencil5-csr":
            yv, _ = op.run_timed(v)
            yw, _ = op.run_timed(2.0 * u - 0.5 * v)
            assert np.max(np.abs(yw - (2.0 * yu - 0.5 * yv))) <= 1e-12 * np.max(np.abs(yw))
        ys[mode] = yu
        op.free()
    scale = np.max(np.abs(ys["stencil5-csr"]))
    assert np.max(np.abs(ys["stencil5-csr"] - ys["cusparse-csr"])) <= 1e-12 * scale
    assert np.max(np.abs(ys["stencil5-csr"] - ys["ellpack"])) <= 1e-12 * scale
    # independent evaluation of the 5-point formula with numpy on a few grid rows
    U = u.reshape(n, n)
    for i in (1, n // 2, n - 2):
        want = 5.0 * U[i, 1:-1] - U[i, :-2] - U[i, 2:] - U[i - 1, 1:-1] - U[i + 1, 1:-1]
        got = ys["stencil5-csr"].reshape(n, n)[i, 1:-1]
        assert np.max(np.abs(got - want)) <= 1e-12 * scale


def test_cg_10k_matches_committed_golden_history(B, golden):
    g = golden["cases"].get("10000:5.0")
    if g is None:
        pytest.skip("10k golden not generated")
    slab = B.CgSlab.stencil5(10000)
    st = slab.solve()
    assert st.iterations == g["cg"]["iterations"] == 14 and st.converged == 1
    assert hist_err(slab.history(), g["cg"]["history"]) < 1e-10
    slab.destroy()


def test_cg_15k_matches_committed_golden_history(B, golden):
    """BASELINE config 5's grid (a grid of n != 0 mod 16 columns: the row-lds tiles of most grid rows start off a 128-byte line);
    the reference publishes CG there too (docs/PROBLEM_SIZE_SCALING_RESULTS.md:31-38). History from tests/golden/make_golden.py --with-15k."""
    g = golden["cases"].get("15000:5.0")
    if g is None:
        pytest.skip("15k golden not generated")
    slab = B.CgSlab.stencil5(15000)
    st = slab.solve()
    assert st.iterations == g["cg"]["iterations"] == 14 and st.converged == 1
    assert hist_err(slab.history(), g["cg"]["history"]) < 1e-10
    assert slab.tile_runs() is not None  # the run-length trial ran on this slab (2.25e8 rows)
    slab.destroy()


def test_cg_20k_published_iteration_count_and_invariants(B):
    """400 M unknowns: 14 iterations on every GPU count (README.md:62); ||r0|| = sqrt(n^2) exactly;
    residuals decrease monotonically for this SPD system; the solve is bit-reproducible."""
    n = 20000
    slab = B.CgSlab.stencil5(n)
    st = slab.solve()
    h = slab.history()
    assert st.iterations == 14 and st.converged == 1 and len(h) == 15
    assert h[0] == float(n) and np.all(np.diff(h) < 0) and h[-1] / h[0] < 1e-6 <= h[-2] / h[0]
    st2 = slab.solve()
    assert st2.iterations == 14 and np.array_equal(slab.history(), h)
    slab.destroy()


def test_cg_20k_matches_committed_golden_history(B, golden):
    """BASELINE config 3: 20 000 x 20 000 CG, 14-iteration residual match vs the CPU oracle
    (history generated once on the GPU box's host by tests/golden/make_golden.py --with-20k)."""
    g = golden["cases"].get("20000:5.0")
    if g is None:
        pytest.skip("20k golden not generated")
    slab = B.CgSlab.stencil5(20000)
    st = slab.solve()
    assert st.iterations == g["cg"]["iterations"] == 14 and st.converged == 1
    assert hist_err(slab.history(), g["cg"]["history"]) < 1e-10
    slab.destroy()


def test_cg_solve_device_20k_matches_golden_history_and_checksums(B, golden):
    """BASELINE config 3 through the reference's single-GPU entry point, cg_solve_device (cg_solver.cu:436-706), on the
    stencil5-csr operator generated in HBM: 14 iterations, every ||r_k|| within 1e-10 of the CPU oracle's committed
    history, and Sum(x) / Norm2(x) -- the checksums the reference main prints -- within 1e-10 of the oracle's."""
    g = golden["cases"].get("20000:5.0")
    if g is None:
        pytest.skip("20k golden not generated")
    n = 20000
    op = B.Operator("stencil5-csr")
    assert op.init_synthetic(n) == 0
    shell = B.HostMatrix(np.empty(0, dtype=B.ENTRY_DTYPE), n * n, n * n, n)  # cg_solve_device reads mat->rows only
    x, hist, st = B.cg_solve(op, shell, np.ones(n * n), np.zeros(n * n), device=True)
    assert st.iterations == g["cg"]["iterations"] == 14 and st.converged == 1
    assert hist_err(hist, g["cg"]["history"]) < 1e-10
    assert abs(st.solution_sum - g["cg"]["solution_sum"]) <= 1e-10 * abs(g["cg"]["solution_sum"])
    assert abs(st.solution_norm - g["cg"]["solution_norm"]) <= 1e-10 * g["cg"]["solution_norm"]
    op.free()


def test_sizes_beyond_int32_csr_are_refused(B):
    """nnz = 5n^2 - 4n must fit the reference's signed 32-bit CSR indices (hard parts):
    n = 20 724 is the last grid that does."""
    last_ok = max(n for n in range(20700, 20760) if 5 * n * n - 4 * n <= 2**31 - 1)
    assert last_ok == 20724
    op = B.Operator("stencil5-csr")
    assert op.init_synthetic(last_ok + 1) != 0  # refused, nothing allocated
    with pytest.raises(RuntimeError):
        B.CgSlab.stencil5(last_ok + 1)
