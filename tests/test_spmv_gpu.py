"""Parity of the HIP SpMV operators with the CPU oracle, through the C-ABI (get_operator ->
init / run_timed / run_device / free). Bit-exact wherever the summation order is defined by the
reference (stencil5-csr, ELLPACK, CSR row-scalar), <= 1e-12 relative for the sub-wavefront CSR
kernels whose shuffle tree re-orders the sum (cuSPARSE's own order is unpinned)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err
import matrices as M

pytestmark = pytest.mark.gpu

STENCIL_SIZES = [2, 3, 5, 81, 127, 128, 129, 130, 200, 257, 300, 513, 640]


def random_stencil(O, n, seed):
    rng = np.random.default_rng(seed)
    e = O.stencil5_coo(n)
    e["value"] = rng.uniform(-3.0, 3.0, len(e))
    return e, rng.standard_normal(n * n)


@pytest.mark.parametrize("n", STENCIL_SIZES)
def test_stencil5_csr_bit_exact_random_values(B, O, fresh_host_matrices, n):
    e, x = random_stencil(O, n, n)
    m = B.HostMatrix(e, n * n, n * n, n)
    op = B.Operator("stencil5-csr")
    assert op.init(m) == 0
    assert op.variant() == ("stencil5/row-lds" if n >= 512 else "stencil5/row-direct")
    rp, ci, va = O.build_csr(e, n * n)
    want = O.spmv_stencil5(rp, ci, va, x, n)
    got, ms = op.run_timed(x)
    assert ms > 0 and np.array_equal(got, want)
    # device-native entry point, including x/y at odd (8-byte aligned) addresses -> scalar-load variant
    dx, dy = B.DeviceVector.from_host(np.concatenate([[0.0], x])), B.DeviceVector(n * n + 1, fill=-7.0)
    class Shift:  # view one double into the allocation
        def __init__(self, v): self.ptr = v.ptr + 8
    assert op.run_device(Shift(dx), Shift(dy)) == 0
    assert np.array_equal(dy.to_host()[1:], want)
    # the same matrix forced through the other kernel variants (aligned and odd-address vectors)
    for forced in ("row-lds", "row-direct", "row-generic"):
        op.select_variant(forced)
        if n >= 2:
            assert op.variant() == "stencil5/" + forced
        got2, _ = op.run_timed(x)
        assert np.array_equal(got2, want), forced
        assert op.run_device(Shift(dx), Shift(dy)) == 0
        assert np.array_equal(dy.to_host()[1:], want), forced
    op.select_variant(None)
    dx.free(), dy.free(), op.free()


def test_stencil5_csr_shipped_81_and_generator_checksums(B, O, golden, fresh_host_matrices):
    m = B.load_matrix_market(os.path.join(GOLDEN, "example81x81.mtx"))
    op = B.Operator("stencil5-csr")
    assert op.init(m) == 0
    y, _ = op.run_timed(np.ones(6561))
    assert y.sum() == -52164.0 and abs(np.sqrt((y * y).sum()) - golden["survey_8c"]["81:-4.0"]["norm2_y"]) < 1e-9
    op.free()
    for n in (3, 81, 512):
        B.lib().spmv_amd_reset_host_matrices()
        m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
        assert op.init(m) == 0
        y, _ = op.run_timed(np.ones(n * n))
        g = golden["cases"][f"{n}:5.0"]
        assert y.sum() == g["sum_y"] == n * n + 4 * n and np.sqrt((y * y).sum()) == g["norm2_y"]
        op.free()


def test_stencil5_csr_non_stencil_inputs_take_the_csr_loop(B, O, fresh_host_matrices):
    """grid_size = -1 (no comment in the file), or a grid_size that does not describe the
    structure: every row goes through the CSR loop, results equal plain CSR SpMV bit for bit."""
    cases = []
    for make in (M.identity, M.diagonal, M.tridiagonal, M.upper_triangular):
        e, r, c, expect = make()
        cases.append((e, r, c, -1, expect))
    e, r, c = M.random_sparse(500, 400, 7, seed=21)
    cases.append((e, r, c, -1, None))
    e, r, c = M.unbalanced()
    cases.append((e, r, c, -1, None))
    e, r, c = M.random_sparse(400, 400, 5, seed=22)
    cases.append((e, r, c, 20, None))  # claims to be a 20x20 stencil but is not
    op = B.Operator("stencil5-csr")
    for e, r, c, grid, expect in cases:
        B.lib().spmv_amd_reset_host_matrices()
        m = B.HostMatrix(e, r, c, grid)
        assert op.init(m) == 0
        assert "csr-loop" in op.variant()
        x = np.ones(c) if expect is not None else np.random.default_rng(r).standard_normal(c)
        rp, ci, va = O.build_csr(e, r)
        got, _ = op.run_timed(x)
        assert np.array_equal(got, O.spmv_csr(rp, ci, va, x))
        if expect is not None:
            assert got.sum() == expect
        op.free()


@pytest.mark.parametrize("variant", [None, "stream", "adaptive", "row-scalar", "wavefront"])
def test_csr_operator(B, O, fresh_host_matrices, variant):
    op = B.Operator("cusparse-csr")
    op.select_variant(variant)
    mats = []
    e, x = random_stencil(O, 130, 5)
    mats.append((e, 130 * 130, 130 * 130, 130, x))
    e, r, c = M.unbalanced()
    mats.append((e, r, c, -1, np.random.default_rng(3).standard_normal(c)))
    e, r, c = M.random_sparse(1000, 777, 33, seed=9)
    mats.append((e, r, c, -1, np.random.default_rng(4).standard_normal(c)))
    # rows far longer than the stream kernel's 1024-entry LDS strip (its chunked walk), next to short and empty ones
    # (rows 0 and 1 are both long and share a block: one long row ends and the next begins inside the same chunk; row 137 sits alone)
    e, r, c = M.random_sparse(300, 6000, lambda row, rng: (3500 if row == 0 else 2900) if row in (0, 1, 137) else (0 if row % 11 == 0 else int(rng.integers(1, 9))), seed=12)
    mats.append((e, r, c, -1, np.random.default_rng(6).standard_normal(c)))
    for e, r, c, grid, x in mats:
        B.lib().spmv_amd_reset_host_matrices()
        m = B.HostMatrix(e, r, c, grid)
        assert op.init(m) == 0
        rp, ci, va = O.build_csr(e, r)
        want = O.spmv_csr(rp, ci, va, x)
        got, ms = op.run_timed(x)
        if variant in ("row-scalar", "stream"):  # sequential fma order: bit-exact
            assert np.array_equal(got, want)
        else:
            scale = np.maximum(np.abs(want), O.spmv_csr(rp, ci, np.abs(va), np.abs(x)))
            assert np.max(np.abs(got - want) / np.maximum(scale, 1e-300)) <= 1e-12
        # integer-valued data: exact under any summation order
        xi = np.ones(c)
        ei = e.copy()
        ei["value"] = np.round(e["value"])
        B.lib().spmv_amd_reset_host_matrices()
        mi = B.HostMatrix(ei, r, c, grid)
        assert op.init(mi) == 0
        rpi, cii, vai = O.build_csr(ei, r)
        goti, _ = op.run_timed(xi)
        assert np.array_equal(goti, O.spmv_csr(rpi, cii, vai, xi))
        op.free()
    op.select_variant(None)


@pytest.mark.parametrize("name", ["ellpack", "stencil5-ellpack"])
def test_ellpack_operators_bit_exact(B, O, fresh_host_matrices, name):
    op = B.Operator(name)
    for n in (3, 5, 81, 130, 257):
        B.lib().spmv_amd_reset_host_matrices()
        e, x = random_stencil(O, n, 100 + n)
        m = B.HostMatrix(e, n * n, n * n, n)
        assert op.init(m) == 0
        rp, ci, va = O.build_csr(e, n * n)
        w, idx, val = O.build_ell(rp, ci, va)
        want_ell = O.spmv_ell(n * n, w, idx, val, x)
        got, _ = op.run_timed(x)
        if name == "stencil5-ellpack":
            assert op.variant() == "ell/stencil5-direct"
            # interior rows use the stencil order W,C,E,N,S: compare with the stencil oracle
            assert np.array_equal(got, O.spmv_stencil5(rp, ci, va, x, n))
        else:
            assert np.array_equal(got, want_ell) and np.array_equal(want_ell, O.spmv_csr(rp, ci, va, x))
        op.free()
    # generic, ragged rows with padding
    B.lib().spmv_amd_reset_host_matrices()
    e, r, c = M.unbalanced()
    m = B.HostMatrix(e, r, c)
    assert op.init(m) == 0 and op.variant() == "ell/slot-major"
    x = np.random.default_rng(8).standard_normal(c)
    rp, ci, va = O.build_csr(e, r)
    got, _ = op.run_timed(x)
    assert np.array_equal(got, O.spmv_csr(rp, ci, va, x))
    op.free()


@pytest.mark.parametrize("n", [2, 3, 81, 128, 300])
def test_synthetic_device_csr_equals_host_built_csr(B, O, fresh_host_matrices, n):
    """The in-HBM generator must produce the bytes load + build_csr_struct + upload produce."""
    op = B.Operator("stencil5-csr")
    assert op.init_synthetic(n) == 0
    orp, oci, ova = O.build_csr(O.stencil5_coo(n), n * n)
    rp, ci, va = op.download_csr(n * n, len(oci))
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and np.array_equal(va, ova)
    x = np.random.default_rng(n).standard_normal(n * n)
    got, _ = op.run_timed(x)
    assert np.array_equal(got, O.spmv_stencil5(orp, oci, ova, x, n))
    cm = B.csr_mat()
    assert (cm.nb_rows, cm.nb_nonzeros) == (n * n, len(oci))
    op.free()
    for other in ("cusparse-csr", "ellpack", "stencil5-ellpack"):
        o2 = B.Operator(other)
        assert o2.init_synthetic(n) == 0
        got, _ = o2.run_timed(np.ones(n * n))
        assert np.array_equal(got, O.spmv_csr(orp, oci, ova, np.ones(n * n)))
        o2.free()


def test_run_device_leaves_no_sync_and_time_helper(B, O, fresh_host_matrices):
    n = 512
    op = B.Operator("stencil5-csr")
    assert op.init_synthetic(n) == 0
    dx, dy = B.DeviceVector(n * n, fill=1.0), B.DeviceVector(n * n, fill=0.0)
    ms = op.time_device(dx, dy, 5)
    assert len(ms) == 5 and (ms > 0).all()
    y = dy.to_host()
    assert y.sum() == n * n + 4 * n
    dx.free(), dy.free(), op.free()


def test_degenerate_inputs(B, O, fresh_host_matrices):
    """1x1, an all-zero matrix (every row empty), a single dense row, and the 1x1 'grid'."""
    cases = [
        (M.entries([(0, 0, 2.5)]), 1, 1, -1, np.array([4.0])),
        (M.entries([]), 3, 3, -1, np.array([1.0, 2.0, 3.0])),
        (M.entries([(1, j, float(j + 1)) for j in range(300)]), 3, 300, -1, np.arange(300, dtype=np.float64)),
        (M.entries([(0, 0, 5.0)]), 1, 1, 1, np.array([-2.0])),
    ]
    for e, r, c, grid, x in cases:
        rp, ci, va = O.build_csr(e, r)
        want = O.spmv_csr(rp, ci, va, x)
        for mode in ("stencil5-csr", "cusparse-csr", "ellpack", "stencil5-ellpack"):
            B.lib().spmv_amd_reset_host_matrices()
            op = B.Operator(mode)
            m = B.HostMatrix(e, r, c, grid)
            assert op.init(m) == 0, mode
            got, _ = op.run_timed(x)
            if mode == "cusparse-csr" and len(e) == 300:
                assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))  # long row: sub-wavefront tree
            else:
                assert np.array_equal(got, want), (mode, r, c)
            op.free()
    op = B.Operator("stencil5-csr")
    assert op.init_synthetic(1) == 0
    got, _ = op.run_timed(np.array([3.0]))
    assert got[0] == 15.0
    op.free()


@pytest.mark.parametrize("name", ["ellpack", "stencil5-ellpack"])
def test_ellpack_alpha_beta_contract(B, O, fresh_host_matrices, name):
    """y = alpha*A*x + beta*y (reference include/spmv_stencil.h:25-42), bit for bit against the oracle."""
    n = 130
    e, x = random_stencil(O, n, 77)
    m = B.HostMatrix(e, n * n, n * n, n)
    op = B.Operator(name)
    assert op.init(m) == 0
    rp, ci, va = O.build_csr(e, n * n)
    y0 = np.random.default_rng(5).standard_normal(n * n)
    base = O.spmv_stencil5(rp, ci, va, x, n) if name == "stencil5-ellpack" else O.spmv_csr(rp, ci, va, x)
    for alpha, beta in ((1.0, 0.0), (2.5, 0.0), (1.0, 1.0), (-0.75, 3.0)):
        dx, dy = B.DeviceVector.from_host(x), B.DeviceVector.from_host(y0)
        assert B.lib().spmv_amd_ellpack_run_device_scaled(name.encode(), dx.ptr, dy.ptr, alpha, beta) == 0
        got = dy.to_host()
        want = alpha * base if beta == 0.0 else np.array([np.float64(alpha) * s + np.float64(beta) * y for s, y in zip(base, y0)])
        if beta == 0.0:
            assert np.array_equal(got, want)
        else:  # fma(alpha, sum, beta*y): one rounding fewer than the numpy expression
            assert np.max(np.abs(got - want)) <= 4e-16 * np.max(np.abs(want)) * 4
        dx.free(), dy.free()
    op.free()


@pytest.mark.parametrize("n,min_grid", [(640, None), (513, None), (1029, None), (130, "2"), (259, "2"), (7, "2")])
def test_rowlds_edge_geometry_is_bit_exact(B, O, fresh_host_matrices, monkeypatch, n, min_grid):
    """The row-lds tile on the geometries that stress its clamped coefficient runs: a last tile of 1 / 5 / 3 live columns, grids
    narrower than one tile and of a single partial tile (forced onto row-lds through SPMV_AMD_ROWLDS_MIN_GRID), and a run length
    per XCD that does not divide the tile count (SPMV_AMD_ROWLDS_GROUP=3). Random coefficients: bit-exact against the oracle."""
    if min_grid:
        monkeypatch.setenv("SPMV_AMD_ROWLDS_MIN_GRID", min_grid)
    e, x = random_stencil(O, n, 100 + n)
    m = B.HostMatrix(e, n * n, n * n, n)
    rp, ci, va = O.build_csr(e, n * n)
    want = O.spmv_stencil5(rp, ci, va, x, n)
    for group in (None, "3"):
        if group:
            monkeypatch.setenv("SPMV_AMD_ROWLDS_GROUP", group)
        op = B.Operator("stencil5-csr")
        assert op.init(m) == 0 and op.variant() == "stencil5/row-lds"
        got, _ = op.run_timed(x)
        assert np.array_equal(got, want), group
        op.free()


@pytest.mark.parametrize("variant", [None, "stream", "adaptive", "row-scalar", "wavefront"])
def test_csr_rows_without_column_zero_never_touch_x0(B, O, fresh_host_matrices, variant):
    """Padding slots of the CSR kernels' chunked loads must not gather x[0]: with x[0] = inf a folded 0 * x[0] would be NaN.
    Long rows (> the 1024-entry strip, ending inside a partial chunk: the adaptive kernel's no-staging fold), short and empty
    rows, none of them with an entry in column 0: y must be finite and equal the oracle's, which reads x only where A has an entry."""
    e, r, c = M.random_sparse(300, 6000, lambda row, rng: (3500 if row == 0 else 2900) if row in (0, 1, 137) else (0 if row % 11 == 0 else int(rng.integers(1, 9))), seed=21)
    e = e[e["col"] != 0]
    x = np.random.default_rng(8).standard_normal(c)
    x[0] = np.inf
    op = B.Operator("cusparse-csr")
    op.select_variant(variant)
    m = B.HostMatrix(e, r, c, -1)
    assert op.init(m) == 0
    if variant is None:
        assert op.variant() == "csr/adaptive"  # the automatic choice for rows beyond the strip
    rp, ci, va = O.build_csr(e, r)
    want = O.spmv_csr(rp, ci, va, x)
    got, _ = op.run_timed(x)
    assert np.all(np.isfinite(want)) and np.all(np.isfinite(got))
    scale = np.maximum(np.abs(want), O.spmv_csr(rp, ci, np.abs(va), np.abs(np.where(np.isfinite(x), x, 0.0))))
    assert np.max(np.abs(got - want) / np.maximum(scale, 1e-300)) <= 1e-12
    op.free()
    op.select_variant(None)
    # ELLPACK pads with index -1: same property
    B.lib().spmv_amd_reset_host_matrices()
    short = e[(e["row"] != 0) & (e["row"] != 1) & (e["row"] != 137)]
    ell = B.Operator("ellpack")
    assert ell.init(B.HostMatrix(short, r, c, -1)) == 0
    rp, ci, va = O.build_csr(short, r)
    got, _ = ell.run_timed(x)
    assert np.array_equal(got, O.spmv_csr(rp, ci, va, x))
    ell.free()


@pytest.mark.parametrize("fixture", ["stencil_9point", "banded", "dense_blocks", "ill_conditioned"])
def test_structured_non_stencil_fixtures_through_every_operator(B, O, fresh_host_matrices, fixture):
    """Structured matrices after the ideas of the reference's (stale) fixture file, tests/helpers/matrix_fixtures.cpp:181-338:
    a 9-point stencil that CLAIMS a grid size (the stencil operators must notice it is not the 5-point pattern), a banded SPD
    matrix, dense diagonal blocks (rows of 37 entries: the sub-wavefront CSR kernels), an ill-conditioned tridiagonal one
    (12 decades on the diagonal). Every operator against the oracle's CSR loop: bit-exact where the summation order is the
    sequential one, 1e-12 (scaled) otherwise; and the analytic checksum of A * 1 to rounding."""
    e, r, c, checksum = {"stencil_9point": lambda: M.stencil_9point(41), "banded": lambda: M.banded(3000, 6),
                         "dense_blocks": lambda: M.dense_blocks(1000, 37), "ill_conditioned": lambda: M.ill_conditioned(2500)}[fixture]()
    grid = 41 if fixture == "stencil_9point" else -1
    rng = np.random.default_rng(17)
    x = rng.standard_normal(c)
    rp, ci, va = O.build_csr(e, r)
    want, want1 = O.spmv_csr(rp, ci, va, x), O.spmv_csr(rp, ci, va, np.ones(c))
    assert abs(want1.sum() - checksum) <= 1e-9 * max(1.0, np.abs(va).sum())
    scale = np.maximum(np.abs(want), O.spmv_csr(rp, ci, np.abs(va), np.abs(x)))
    for mode in ("stencil5-csr", "cusparse-csr", "ellpack", "stencil5-ellpack"):
        B.lib().spmv_amd_reset_host_matrices()
        m = B.HostMatrix(e, r, c, grid)
        op = B.Operator(mode)
        assert op.init(m) == 0, mode
        if mode == "stencil5-csr":
            assert "csr-loop" in op.variant()
        got, _ = op.run_timed(x)
        sequential = mode != "cusparse-csr" or op.variant() in ("csr/stream", "csr/row-scalar")
        if sequential:
            assert np.array_equal(got, want), (mode, op.variant())
        else:
            assert np.max(np.abs(got - want) / np.maximum(scale, 1e-300)) <= 1e-12, (mode, op.variant())
        op.free()
