"""Parity of the CG solvers with the CPU oracle: iteration counts equal, per-iteration ||r_k||
within 1e-10 relative (BASELINE.json north_star), solution within 1e-10 of the oracle's."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, rel_err, hist_err
import matrices as M

pytestmark = pytest.mark.gpu
TOL = 1e-10  # the north_star's bar on fp64 CG residuals


def check(O, rp, ci, va, n, hist, x, stats, device_form=True):
    xo, ho, ro = O.cg(rp, ci, va, n, np.ones(n * n), np.zeros(n * n), device_form=device_form)
    assert stats.iterations == ro.iterations and stats.converged == ro.converged == 1
    assert len(hist) == len(ho) and hist_err(hist, ho) < TOL
    assert np.max(np.abs(x - xo)) <= TOL * np.max(np.abs(xo))
    assert abs(stats.solution_sum - ro.solution_sum) <= TOL * abs(ro.solution_sum)
    assert abs(stats.solution_norm - ro.solution_norm) <= TOL * ro.solution_norm
    assert hist_err([ho[0], stats.residual_norm], [ho[0], ro.residual_norm]) < TOL


@pytest.mark.parametrize("n", [3, 81, 200, 512])
@pytest.mark.parametrize("mode", ["stencil5-csr", "cusparse-csr", "ellpack"])
def test_cg_solve_device(B, O, fresh_host_matrices, n, mode):
    m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
    op = B.Operator(mode)
    assert op.init(m) == 0
    rp, ci, va = O.stencil5_csr(n)
    x, hist, st = B.cg_solve(op, m, np.ones(n * n), np.zeros(n * n), device=True)
    check(O, rp, ci, va, n, hist, x, st)
    op.free()


def test_cg_solve_host_path_and_shipped_matrix(B, O, golden, fresh_host_matrices):
    m = B.load_matrix_market(os.path.join(GOLDEN, "example81x81.mtx"))
    op = B.Operator("stencil5-csr")
    assert op.init(m) == 0
    x, hist, st = B.cg_solve(op, m, np.ones(6561), np.zeros(6561), device=False)
    s = golden["survey_8c"]["81:-4.0"]
    assert st.iterations == s["cg_iterations"] == 40 and st.converged == 1
    assert abs(st.solution_sum - s["solution_sum"]) < 1e-10 * abs(s["solution_sum"])
    assert abs(st.solution_norm - s["solution_norm"]) < 1e-10 * s["solution_norm"]
    g = golden["cases"]["81:-4.0"]["cg"]
    assert hist_err(hist, g["history"]) < TOL
    op.free()


def test_cg_not_converged_reports_like_reference(B, O, fresh_host_matrices):
    n = 81
    m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
    op = B.Operator("stencil5-csr")
    assert op.init(m) == 0
    x, hist, st = B.cg_solve(op, m, np.ones(n * n), np.zeros(n * n), max_iters=5, device=True)
    rp, ci, va = O.stencil5_csr(n)
    xo, ho, ro = O.cg(rp, ci, va, n, np.ones(n * n), np.zeros(n * n), max_iters=5)
    assert st.iterations == ro.iterations == 5 and st.converged == ro.converged == 0
    assert st.residual_norm == ro.residual_norm == ho[0]  # device form keeps ||r0|| when it never converged
    assert hist_err(hist, ho) < TOL
    op.free()


@pytest.mark.parametrize("n", [3, 64, 81, 130, 512])
def test_slab_solver_single_rank(B, O, fresh_host_matrices, n):
    """The multi-GPU solver with one rank: the configuration behind the reference's '1 GPU' numbers."""
    m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
    slab = B.CgSlab.from_matrix(m)
    slab.set_vectors(np.ones(n * n), np.zeros(n * n))
    st = slab.solve()
    rp, ci, va = O.stencil5_csr(n)
    xo, ho, ro = O.cg_partitioned(rp, ci, va, n, np.ones(n * n), np.zeros(n * n), world=1)
    hist, x = slab.history(), slab.gather()
    assert st.iterations == ro.iterations and st.converged == 1
    assert hist_err(hist, ho) < TOL and np.max(np.abs(x - xo)) <= TOL * np.max(np.abs(xo))
    # solving again from the stored x0 reproduces the run bit for bit (fixed-shape reductions)
    st2 = slab.solve()
    assert st2.iterations == st.iterations and np.array_equal(slab.history(), hist) and np.array_equal(slab.gather(), x)
    slab.destroy()
    # synthetic slab == host-matrix slab
    s2 = B.CgSlab.stencil5(n) if n >= 2 else None
    st3 = s2.solve()
    assert st3.iterations == st.iterations and np.array_equal(s2.history(), hist) and np.array_equal(s2.gather(), x)
    s2.destroy()


@pytest.mark.parametrize("n,maxiter", [(81, 1000), (200, 1000), (200, 7), (64, 0), (130, 5)])
def test_direction_ring_is_bit_identical_to_in_place_updates(B, O, fresh_host_matrices, monkeypatch, n, maxiter):
    """Deferred x update: with any ring length the solver evaluates, element for element, the fma chain
    x <- fma(alpha_k, p_k, x) of the per-iteration update (ring length 1), so x, the history and the iteration
    count agree bit for bit -- through several wrap-arounds of short rings, with a non-zero x0, and when the
    iteration limit cuts the solve short (window flushed at the limit)."""
    rng = np.random.default_rng(n)
    e = O.stencil5_coo(n)
    b, x0 = rng.standard_normal(n * n), 0.1 * rng.standard_normal(n * n)
    results = {}
    for ring in (1, 2, 3, 4, 5, 16):
        monkeypatch.setenv("SPMV_AMD_P_RING", str(ring))
        B.lib().spmv_amd_reset_host_matrices()
        m = B.HostMatrix(e, n * n, n * n, n)
        slab = B.CgSlab.from_matrix(m)
        slab.set_vectors(b, x0)
        st = slab.solve(max_iters=maxiter)
        results[ring] = (st.iterations, st.converged, slab.history().copy(), slab.gather().copy())
        st2 = slab.solve(max_iters=maxiter)  # a second solve starts from x0 again, ring rewound
        assert st2.iterations == st.iterations and np.array_equal(slab.gather(), results[ring][3])
        slab.destroy()
    it1, conv1, h1, x1 = results[1]
    if maxiter == 0:
        assert it1 == 0 and np.array_equal(x1, x0)
    for ring, (it, conv, h, x) in results.items():
        assert (it, conv) == (it1, conv1) and np.array_equal(h, h1) and np.array_equal(x, x1), ring
    rp, ci, va = O.build_csr(e, n * n)
    if maxiter == 1000:
        xo, ho, ro = O.cg_partitioned(rp, ci, va, n, b, x0, world=1)
        assert it1 == ro.iterations and hist_err(h1, ho) < TOL and np.max(np.abs(x1 - xo)) <= TOL * np.max(np.abs(xo))


@pytest.mark.parametrize("n", [130, 640])
def test_cg_solve_device_runs_the_slab_solvers_loop(B, O, fresh_host_matrices, n):
    """cg_solve_device (reference cg_solver.cu:436-706) is the slab solver's loop around the caller's operator. With
    this library's stencil5-csr it takes the operator's fused launch and differs from the slab solver in ONE rounding
    per element and iteration -- p = fma(beta, p, r) (update_p_kernel, cg_solver.cu:90-95) instead of
    fma(1.0, r, beta*p) (mgpu :136-140) -- so the two histories agree far below the 1e-10 bar; with a random
    right-hand side and a non-zero x0 it equals the oracle's device form; and a second call re-uses the cached vectors."""
    rng = np.random.default_rng(n)
    e = O.stencil5_coo(n)
    m = B.HostMatrix(e, n * n, n * n, n)
    op = B.Operator("stencil5-csr")
    assert op.init(m) == 0
    b, x0 = rng.standard_normal(n * n), 0.1 * rng.standard_normal(n * n)
    x, hist, st = B.cg_solve(op, m, b, x0, device=True)
    x2, hist2, st2 = B.cg_solve(op, m, b, x0, device=True)  # cached workspace, same bits
    assert st2.iterations == st.iterations and np.array_equal(hist2, hist) and np.array_equal(x2, x)
    rp, ci, va = O.build_csr(e, n * n)
    xo, ho, ro = O.cg(rp, ci, va, n, b, x0, device_form=True)
    assert st.iterations == ro.iterations and st.converged == 1 and hist_err(hist, ho) < TOL
    assert np.max(np.abs(x - xo)) <= TOL * np.max(np.abs(xo))
    slab = B.CgSlab.from_matrix(m)
    slab.set_vectors(b, x0)
    sst = slab.solve()
    assert sst.iterations == st.iterations and hist_err(hist, slab.history()) < 1e-12
    assert np.max(np.abs(x - slab.gather())) <= 1e-12 * np.max(np.abs(x))
    slab.destroy()
    op.free()


def test_cg_solve_device_through_a_callers_own_operator_table(B, O, fresh_host_matrices):
    """An SpmvOperator table this library does not own (here: a ctypes-made struct whose run_device forwards to the CSR
    operator) offers nothing beyond run_device(d_x, d_y): the solver drives it through the vtable plus a dot kernel, and
    still matches the oracle at 1e-10. Detailed timers fill the three categories without changing a bit."""
    n = 200
    m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
    real = B.Operator("cusparse-csr")
    assert real.init(m) == 0
    calls = []

    def run_device(d_x, d_y):
        calls.append(1)
        return real.op.contents.run_device(d_x, d_y)

    keep = (B.INIT_FN(lambda mat: 0), B.RUN_TIMED_FN(lambda x, y, ms: 1), B.RUN_DEVICE_FN(run_device), B.FREE_FN(lambda: None))
    table = B.SpmvOperator(b"callers-own", *keep)

    class Foreign:  # what B.cg_solve needs of an operator: a pointer to the table
        op = C.pointer(table)

    rp, ci, va = O.stencil5_csr(n)
    x, hist, st = B.cg_solve(Foreign, m, np.ones(n * n), np.zeros(n * n), device=True)
    check(O, rp, ci, va, n, hist, x, st)
    assert len(calls) == st.iterations + 1  # one SpMV per iteration + the initial residual's, all through the vtable
    x2, hist2, st2 = B.cg_solve(Foreign, m, np.ones(n * n), np.zeros(n * n), device=True, timers=1)
    assert np.array_equal(hist2, hist) and np.array_equal(x2, x)
    assert st2.time_spmv_ms > 0 and st2.time_blas1_ms > 0 and st2.time_reductions_ms > 0
    # switching to the library's own stencil operator on the same workspace picks the fused launch up again
    own = B.Operator("stencil5-csr")
    assert own.init(m) == 0
    x3, hist3, st3 = B.cg_solve(own, m, np.ones(n * n), np.zeros(n * n), device=True)
    check(O, rp, ci, va, n, hist3, x3, st3)
    own.free()
    real.free()


@pytest.mark.parametrize("device", [True, False])
def test_a_failing_run_device_is_a_status_not_the_end_of_the_callers_process(B, O, fresh_host_matrices, device):
    """A caller's operator table whose run_device starts failing in its third call: cg_solve_device / cg_solve hand back a
    non-zero status (the reference ignores the value, cg_solver.cu:498,541; ending the caller's process over it is not a
    library's call to make), this process lives on, and the next solve on a healthy table is right."""
    n = 120
    m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
    real = B.Operator("cusparse-csr")
    assert real.init(m) == 0
    calls = []

    def run_device(d_x, d_y):
        calls.append(1)
        return 1 if len(calls) >= 3 and fail[0] else real.op.contents.run_device(d_x, d_y)

    keep = (B.INIT_FN(lambda mat: 0), B.RUN_TIMED_FN(lambda x, y, ms: 1), B.RUN_DEVICE_FN(run_device), B.FREE_FN(lambda: None))
    table = B.SpmvOperator(b"callers-own", *keep)

    class Foreign:
        op = C.pointer(table)

    fail = [True]
    with pytest.raises(RuntimeError, match="cg_solve -> 1"):
        B.cg_solve(Foreign, m, np.ones(n * n), np.zeros(n * n), device=device)
    assert len(calls) == 3  # nothing was enqueued after the failing call
    fail[0] = False
    rp, ci, va = O.stencil5_csr(n)
    x, hist, st = B.cg_solve(Foreign, m, np.ones(n * n), np.zeros(n * n), device=device)
    xo, ho, ro = O.cg(rp, ci, va, n, np.ones(n * n), np.zeros(n * n), device_form=device)
    assert st.iterations == ro.iterations and st.converged == 1 and hist_err(hist, ho) < TOL
    assert np.max(np.abs(x - xo)) <= TOL * np.max(np.abs(xo))
    # a caller with its own table never reaches this library's operator free(): the workspace has its own release call
    B.lib().spmv_amd_cg_release_workspace()
    B.lib().spmv_amd_cg_release_workspace()  # idempotent
    x2, hist2, _ = B.cg_solve(Foreign, m, np.ones(n * n), np.zeros(n * n), device=device)
    assert np.array_equal(hist2, hist) and np.array_equal(x2, x)
    real.free()


def test_release_asked_for_from_inside_run_device_is_deferred_not_a_deadlock(B, O, fresh_host_matrices):
    """cg_solve_device holds the workspace lock for the whole solve, callbacks into a foreign run_device included.
    A caller's run_device that calls spmv_amd_cg_release_workspace() -- a documented way to give the memory back -- must neither
    hang on that (non-recursive) lock nor pull the vectors from under the running loop: the release is noted, the solve finishes
    with the right numbers, and the workspace is gone afterwards (the next solve builds a fresh one and agrees bit for bit)."""
    n = 120
    m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
    real = B.Operator("cusparse-csr")
    assert real.init(m) == 0
    calls = []

    def run_device(d_x, d_y):
        calls.append(1)
        if len(calls) == 2 and release[0]:
            B.lib().spmv_amd_cg_release_workspace()  # from inside the solve, on the solving thread
        return real.op.contents.run_device(d_x, d_y)

    keep = (B.INIT_FN(lambda mat: 0), B.RUN_TIMED_FN(lambda x, y, ms: 1), B.RUN_DEVICE_FN(run_device), B.FREE_FN(lambda: None))
    table = B.SpmvOperator(b"callers-own", *keep)

    class Foreign:
        op = C.pointer(table)

    release = [True]
    rp, ci, va = O.stencil5_csr(n)
    x, hist, st = B.cg_solve(Foreign, m, np.ones(n * n), np.zeros(n * n), device=True)
    xo, ho, ro = O.cg(rp, ci, va, n, np.ones(n * n), np.zeros(n * n), device_form=True)
    assert st.iterations == ro.iterations and st.converged == 1 and hist_err(hist, ho) < TOL
    assert np.max(np.abs(x - xo)) <= TOL * np.max(np.abs(xo))
    release[0] = False
    x2, hist2, _ = B.cg_solve(Foreign, m, np.ones(n * n), np.zeros(n * n), device=True)
    assert np.array_equal(hist2, hist) and np.array_equal(x2, x)
    B.lib().spmv_amd_cg_release_workspace()
    real.free()


def test_placement_at_set_up_changes_addresses_only(B, O, fresh_host_matrices, monkeypatch):
    """Round 4's set-up work -- the vector arena, the placement of the coefficient stream (slabs) and of the vector run_timed's
    kernel writes (operators), both by timing a few allocations one 32 GiB region apart -- may change where data lies and nothing
    else. A grid large enough for the trials to run (4200^2 = 17.6 M rows >= 16 Mi): with 3 candidates and with the trials
    off, the operator's y and the solver's history and solution are bit-identical, and the records say what was done."""
    n = 4200
    out = {}
    for cand in ("3", "1"):
        monkeypatch.setenv("SPMV_AMD_PLACEMENT_CANDIDATES", cand)
        op = B.Operator("stencil5-csr")
        assert op.init_synthetic(n) == 0
        placed = op.placement()
        assert placed is not None and placed[0] == int(cand) and placed[1] > 0.9
        ms = op.time_device(None, None, 3)  # the operator's own vectors: x = 1
        assert len(ms) == 3 and np.all(ms > 0)
        dx, dy = B.DeviceVector(n * n, fill=1.0), B.DeviceVector(n * n, fill=0.0)
        op.run_device(dx, dy)
        y = dy.to_host()
        assert y.sum() == n * n + 4 * n  # the generator stencil's analytic checksum
        dx.free(), dy.free(), op.free()
        slab = B.CgSlab.stencil5(n)
        rec = slab.placement()
        assert (rec is None) if cand == "1" else (rec["kind"] == "coefficient candidates" and rec["candidates"] == 3 and rec["spmv_ms_kept"] <= rec["spmv_ms_before"] * 1.001)
        runs = slab.tile_runs()  # row-lds tiles per XCD and run: the rule's neighbours timed at creation, the fastest kept
        assert runs is not None and runs["rule"] == 4 and 1 <= runs["kept"] <= 6 and runs["spmv_ms_kept"] <= runs["spmv_ms_rule"]
        phases = slab.setup_ms()  # wall time of creation's phases, reported by bench.py beside the timed region
        assert len(phases) == 5 and all(v >= 0.0 for v in phases.values()) and phases["matrix_to_hbm"] > 0 and phases["tile_runs"] > 0
        assert (phases["coefficient_placement"] > 1.0) == (cand == "3")  # the trial copies and times 3 x 141 MB; without it the phase is empty
        st = slab.solve()
        each = slab.spmv_launch_ms()
        assert 0 < len(each) <= st.iterations and np.all(each > 0)
        out[cand] = (y, slab.history().copy(), slab.gather(), st.iterations)
        slab.destroy()
    assert np.array_equal(out["3"][0], out["1"][0]) and np.array_equal(out["3"][1], out["1"][1])
    assert np.array_equal(out["3"][2], out["1"][2]) and out["3"][3] == out["1"][3]


def test_a_placement_candidate_that_does_not_fit_ends_the_trial_not_the_process(Blab, monkeypatch):
    """every further candidate of the coefficient placement is an OPTIONAL copy -- when the device cannot provide
    it (LAB build's SPMV_AMD_PLACEMENT_FAIL_AFTER=1: the first further candidate "does not fit") the slab keeps the array it has,
    says so in its record and solves to the same bits."""
    B = Blab
    n = 5000  # 2.5e7 rows: above the 16 Mi-row threshold of the trial
    out = {}
    for fail in ("0", "1"):
        monkeypatch.setenv("SPMV_AMD_PLACEMENT_FAIL_AFTER", fail)
        slab = B.CgSlab.stencil5(n)
        rec = slab.placement()
        st = slab.solve()
        assert st.converged == 1
        out[fail] = (rec, slab.history().copy(), slab.gather())
        slab.destroy()
    assert out["0"][0]["candidates"] >= 1 and out["1"][0]["candidates"] == 1
    assert out["1"][0]["spmv_ms_kept"] == out["1"][0]["spmv_ms_before"]
    assert np.array_equal(out["0"][1], out["1"][1]) and np.array_equal(out["0"][2], out["1"][2])


def test_slab_timeline_accounts_for_the_solve(B, O, fresh_host_matrices):
    """spmv_amd_cg_slab_set_timeline: stage-boundary events without host syncs. The stages of an iteration add up to the
    iteration, the iterations (+ initial residual + flush) to the solve, and the numbers are those of a plain solve."""
    n = 1024
    slab = B.CgSlab.stencil5(n)
    st = slab.solve()
    hist = slab.history().copy()
    st_t, t = slab.timeline_solve()
    names = B.lib().spmv_amd_cg_slab_timeline_names().decode().split(",")
    assert list(t.keys()) == names and t["iterations"] == st.iterations == st_t.iterations
    assert np.array_equal(slab.history(), hist)
    stages = ["spmv_interior_us", "halo_wait_and_boundary_rows_us", "reduce_pAp_and_allreduce_us", "update_r_us",
              "reduce_rr_allreduce_and_scalar_step_us", "direction_update_us", "gap_before_next_iteration_us"]
    assert all(t[k] >= 0.0 for k in stages) and t["spmv_interior_us"] > 0 and t["update_r_us"] > 0
    # The converging iteration's direction update does no work (the reference tests convergence before its p update,
    # cg_solver_mgpu_partitioned.cu:652-676): the stage is averaged over the launches that did, one fewer than the iterations.
    assert st_t.converged == 1 and t["direction_updates"] == st.iterations - 1 and t["direction_update_us"] > 0
    # a solve that is stopped by max_iters runs the update in every iteration, and its stages add up to the iteration
    st_f, tf = slab.timeline_solve(max_iters=st.iterations - 2)
    assert st_f.converged == 0 and tf["direction_updates"] == tf["iterations"] == st.iterations - 2
    assert abs(sum(tf[k] for k in stages) - tf["iteration_us"]) <= 1e-3 * tf["iteration_us"] + 1.0
    whole = t["initial_residual_us"] + t["iterations"] * t["iteration_us"] + t["final_x_flush_us"]
    assert 0.9 * whole <= t["solve_ms"] * 1e3 <= 1.1 * whole + 50.0
    assert slab.solve().iterations == st.iterations and B.lib().spmv_amd_cg_slab_timeline(slab.h, None, 0) == 0  # off again
    # degenerate solves: no iteration at all, and a single one
    st0, t0 = slab.timeline_solve(max_iters=0)
    assert st0.iterations == 0 and t0["iterations"] == 0 and t0["iteration_us"] == 0.0 and t0["initial_residual_us"] > 0
    st1, t1 = slab.timeline_solve(max_iters=1)
    assert st1.iterations == 1 and t1["iterations"] == 1 and t1["iteration_us"] > 0 and t1["direction_updates"] == 1
    slab.destroy()


@pytest.mark.parametrize("n,P,r", [(1024, 1, 0), (1000, 1, 0), (1024, 4, 1), (1024, 2, 0), (1024, 2, 1), (1001, 1, 0), (2048, 8, 3)])
def test_both_loop_shapes_leave_every_bit_alone(Blab, monkeypatch, n, P, r):
    """The loop has two shapes (csrc/cg_slab.hip, LoopShape): the PIPELINE -- halo exchange on the side stream under the interior
    rows, boundary rows inside the reducing launch behind the arrival flag, the direction update's edge rows + first piece (+ the
    scalar step on the RCCL path) in one launch that releases the exchange by a device flag -- and the PLAIN order (no_overlap:
    everything on the compute stream, the exchange behind the whole direction update: the reference's own, and bench.py's
    fallback). Plus the one split both may take: the direction update as a lead piece + (after the status record) the rest (late
    bulk), forced on with short leads so that small grids take it in both sweep directions. History and solution must be
    bit-identical under every combination, with the direction ring and with the in-place x / p update (ring 1), on a plain slab
    (even and odd row counts) and on stand-in slabs of a larger job (one and two neighbours; LAB build). Round 6's host-side
    rules ride along: the host one iteration ahead of the status records while convergence is far (run_ahead, default on; a
    stand-in solve with tolerance 0 is "far" in every iteration, a real one until its last few), and the late bulk by prediction
    (late_bulk = 2) against the protocol in every iteration (1). The A/B switches of
    rounds 2-5 (two-launch reductions, event-ordered hand-overs, three-launch direction update, no sweep alternation) left with
    their code in round 6."""
    B = Blab
    comm = None
    if P > 1:
        monkeypatch.setenv("SPMV_AMD_SELF_NEIGHBOUR", "1")
        monkeypatch.setenv("SPMV_AMD_FORCE_COLLECTIVES", "1")
    for ring in ("16", "1"):
        monkeypatch.setenv("SPMV_AMD_P_RING", ring)
        if P > 1:
            comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
            slab = B.CgSlab.stencil5_as(n, r, P, comm)
            kw = dict(max_iters=9, tol=0.0)
        else:
            slab = B.CgSlab.stencil5(n)
            kw = {}
        slab.set_option("late_bulk", 0)
        st0 = slab.solve(**kw)  # the default shape of this slab
        h0, x0 = slab.history().copy(), slab.gather() if P == 1 else None
        for opts in ({"late_bulk": 1, "lead_rows": 50000}, {"late_bulk": 1, "lead_rows": 512}, {"no_overlap": 1},
                     {"no_overlap": 1, "late_bulk": 1, "lead_rows": 512}, {}, {"run_ahead": 0}, {"late_bulk": 2, "lead_rows": 512},
                     {"late_bulk": 2, "lead_rows": 512, "run_ahead": 0, "no_overlap": 1}):
            for k in ("late_bulk", "no_overlap"):
                slab.set_option(k, opts.get(k, 0))
            slab.set_option("run_ahead", opts.get("run_ahead", 1))  # default: the host one iteration ahead of the status records
            slab.set_option("lead_rows", opts.get("lead_rows", 1 << 24))
            st = slab.solve(**kw)
            assert (st.iterations, st.converged) == (st0.iterations, st0.converged) and np.array_equal(slab.history(), h0), (ring, opts)
            if P == 1:
                assert np.array_equal(slab.gather(), x0), (ring, opts)
            _, tl = slab.timeline_solve(**kw)
            assert np.array_equal(slab.history(), h0) and tl["direction_updates"] == st0.iterations - st0.converged
            if P > 1:  # the exchange really ran where the shape says: on the side stream (timed there) or on the compute stream
                pipeline = not opts.get("no_overlap") and ring == "16"  # the in-place form (ring 1) takes the plain order
                assert (tl["halo_exchange_on_side_stream_us"] > 0) == pipeline, (ring, opts, tl)
                assert slab.loop_shape().startswith("pipeline" if pipeline else "plain"), slab.loop_shape()
            st_d = slab.solve(timers=1, **kw)  # the reference's detailed timers: the plain shape with a host sync per stage
            assert st_d.iterations == st0.iterations and np.array_equal(slab.history(), h0) and st_d.time_spmv_ms > 0
        with pytest.raises(ValueError):
            slab.set_option("no_such_option", 1)
        for gone in ("halo_flag", "edges_in_step", "reduce_one_launch", "early_halo", "pingpong"):
            with pytest.raises(ValueError):
                slab.set_option(gone, 0)
        slab.destroy()
        if comm is not None:
            comm.destroy()


@pytest.mark.parametrize("n,P,r", [(1024, 1, 0), (1001, 1, 0), (1024, 4, 1)])
def test_a_wrong_guess_costs_an_iteration_of_no_ops_and_nothing_else(Blab, monkeypatch, n, P, r):
    """Round 6: the host does not wait for the status record of an iteration the known residual says cannot converge; it runs one
    iteration ahead. If the guess is WRONG the iteration after the converging one is enqueued in full: its kernels -- SpMV, sums,
    step, direction update, halo exchange, all-reduces -- see the flag and do nothing, the loop ends one record later, and the x
    flush counts on the device how many directions are real. LAB hook run_ahead = 2 makes the guess wrong in EVERY solve. Iteration
    count, verdict, residual history and solution must equal the solve that waits for every record (run_ahead = 0), bit for
    bit, for every ring length (the x flush falls on, before and after the iteration that overshoots) and for the in-place form;
    on a stand-in slab (stop_at declares the converging iteration) with the pipeline and with the plain order."""
    B = Blab
    comm = None
    if P > 1:
        monkeypatch.setenv("SPMV_AMD_SELF_NEIGHBOUR", "1")
        monkeypatch.setenv("SPMV_AMD_FORCE_COLLECTIVES", "1")
    for ring in ("16", "5", "4", "3", "2", "1"):
        monkeypatch.setenv("SPMV_AMD_P_RING", ring)
        if P > 1:
            comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
            slab = B.CgSlab.stencil5_as(n, r, P, comm)
            slab.set_option("stop_at", 8)
            kw = dict(max_iters=30, tol=0.0)
        else:
            slab = B.CgSlab.stencil5(n)
            kw = {}
        slab.set_option("run_ahead", 0)
        st0 = slab.solve(**kw)
        want = (st0.iterations, st0.converged, slab.history().copy(), slab.gather() if P == 1 else None)
        assert st0.converged == 1 and (P == 1 or st0.iterations == 8)
        for ahead, no_overlap in ((2, 0), (1, 0), (2, 1)):
            slab.set_option("run_ahead", ahead)
            slab.set_option("no_overlap", no_overlap)
            for _ in range(2):
                st = slab.solve(**kw)
                assert (st.iterations, st.converged) == want[:2] and np.array_equal(slab.history(), want[2]), (ring, ahead, no_overlap)
                if P == 1:
                    assert np.array_equal(slab.gather(), want[3]), (ring, ahead, no_overlap)
            st_t, tl = slab.timeline_solve(**kw)
            assert st_t.iterations == want[0] and np.array_equal(slab.history(), want[2]) and tl["iterations"] == want[0]
        slab.destroy()
        if comm is not None:
            comm.destroy()


@pytest.mark.parametrize("collectives", ["1", "0"])
@pytest.mark.parametrize("n,P,r", [(4096, 2, 0), (4096, 2, 1), (6000, 4, 1)])
def test_direction_update_in_one_launch_with_the_step(Blab, monkeypatch, collectives, n, P, r):
    """On a slab with neighbours the rows they wait for, the first piece of the rest of the direction update and -- on the RCCL
    path (collectives forced), where the scalar step follows an ncclAllReduce -- the step itself share ONE launch (kernels.hpp,
    DirectionLaunch). Nobody in it waits for the step: every workgroup derives beta and the convergence verdict from scalars the
    step does not write, with the step's own expressions. Its first workgroups write the edge rows through and raise the flag
    that releases the halo exchange on the side stream. Against the PLAIN shape (no_overlap: step | whole direction update |
    exchange, one stream), with the direction ring wrapping (ring 4: every fourth iteration keeps the step a launch of its own
    in front of the x flush) and not, with the late bulk's lead piece as the first piece, on wide slabs with one and two
    neighbours: the residual history (every direction feeds the next residual) bit-identical. collectives = 0: no all-reduce
    call between sum and step (the shape of the peer-mailbox path): the step stays in the sum's launch, the rest is fused."""
    B = Blab
    monkeypatch.setenv("SPMV_AMD_SELF_NEIGHBOUR", "1")
    monkeypatch.setenv("SPMV_AMD_FORCE_COLLECTIVES", collectives)
    for ring in ("16", "4"):
        monkeypatch.setenv("SPMV_AMD_P_RING", ring)
        comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
        slab = B.CgSlab.stencil5_as(n, r, P, comm)
        kw = dict(max_iters=11, tol=0.0)
        slab.set_option("no_overlap", 1)
        st0 = slab.solve(**kw)
        h0 = slab.history().copy()
        slab.set_option("no_overlap", 0)
        for opts in ({}, {"late_bulk": 0}, {"late_bulk": 1, "lead_rows": 4096}, {"late_bulk": 1, "lead_rows": 512}):
            slab.set_option("late_bulk", opts.get("late_bulk", 1))
            slab.set_option("lead_rows", opts.get("lead_rows", 1 << 24))
            st = slab.solve(**kw)
            assert st.iterations == st0.iterations and np.array_equal(slab.history(), h0), (ring, opts)
        # a converging solve: the launch of the converging iteration writes no direction, the loop ends on the step's record
        slab.set_option("stop_at", 7)
        st = slab.solve(max_iters=20, tol=0.0)
        assert st.iterations == 7 and st.converged == 1 and np.array_equal(slab.history(), h0[:8])
        slab.destroy()
        comm.destroy()


def test_reference_entry_point_cg_solve_mgpu_partitioned(B, O, fresh_host_matrices):
    import ctypes as C
    n = 100
    m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
    b, x = np.ones(n * n), np.zeros(n * n)
    cfg, st = B.CGConfig(1000, 1e-6, 0, 0), B.CGStatsMultiGPU()
    assert B.lib().spmv_amd_cg_solve_mgpu_partitioned(m.ptr, b.ctypes.data, x.ctypes.data, C.byref(cfg), C.byref(st)) == 0
    rp, ci, va = O.stencil5_csr(n)
    xo, ho, ro = O.cg_partitioned(rp, ci, va, n, b, np.zeros(n * n), world=1)
    assert st.iterations == ro.iterations and st.converged == 1 and st.time_total_ms > 0
    assert np.max(np.abs(x - xo)) <= TOL * np.max(np.abs(xo))
    assert abs(st.solution_sum - ro.solution_sum) <= TOL * abs(ro.solution_sum)
    # detailed timers fill the per-category fields and do not change the numbers
    x2 = np.zeros(n * n)
    cfg2, st2 = B.CGConfig(1000, 1e-6, 0, 1), B.CGStatsMultiGPU()
    assert B.lib().spmv_amd_cg_solve_mgpu_partitioned(m.ptr, b.ctypes.data, x2.ctypes.data, C.byref(cfg2), C.byref(st2)) == 0
    assert np.array_equal(x2, x) and st2.time_spmv_ms > 0 and st2.time_blas1_ms > 0


@pytest.mark.parametrize("n,rowlds_min_grid,variant", [(256, None, "stencil5/row-direct"), (640, None, "stencil5/row-lds"),
                                                       (130, "2", "stencil5/row-lds"), (260, "2", "stencil5/row-lds"),
                                                       (64, "2", "stencil5/row-lds")])
def test_slab_spmv_matches_halo_oracle(B, O, fresh_host_matrices, monkeypatch, n, rowlds_min_grid, variant):
    """Slab-local SpMV with halos for every rank of a 1/2/4-way split, one rank at a time on this
    GPU (staged communicator with trivial callbacks: spmv() fills the halos from the full vector).
    Random coefficients, so a coefficient taken from the wrong CSR position cannot go unnoticed; the
    row-lds kernel is also forced onto small grids (one or two clamped tiles per grid row)."""
    if rowlds_min_grid is not None:
        monkeypatch.setenv("SPMV_AMD_ROWLDS_MIN_GRID", rowlds_min_grid)
    e = O.stencil5_coo(n)
    rng = np.random.default_rng(1)
    e["value"] = rng.uniform(-3, 3, len(e))
    x = rng.standard_normal(n * n)
    rp, ci, va = O.build_csr(e, n * n)
    full = O.spmv_csr(rp, ci, va, x)
    m = B.HostMatrix(e, n * n, n * n, n)
    for world in (2, 4):
        for rank in range(world):
            comm = B.Comm.staged(rank, world, lambda *a: 0, lambda *a: 0)
            slab = B.CgSlab.from_matrix(m, comm)
            off, nl = O.partition_rows(n * n, world, rank)
            assert (slab.row_offset, slab.n_local) == (off, nl)
            if (n * n) % (world * n) == 0:  # slabs of whole grid rows; others run row-generic
                assert slab.variant() == variant
            base = rp[off]
            lrp = (rp[off:off + nl + 1] - base).astype(np.int32)
            hp = x[off - n:off] if rank > 0 else None
            hn = x[off + nl:off + nl + n] if rank < world - 1 else None
            want = O.spmv_halo(lrp, ci[base:], va[base:], x[off:off + nl], hp, hn, off, n * n, n)
            got = slab.spmv(x)
            assert np.array_equal(got, want)
            # interior stencil order vs plain CSR order: same values up to rounding
            assert rel_err(got, full[off:off + nl]) < 1e-12 or np.allclose(got, full[off:off + nl], rtol=1e-12, atol=1e-12)
            slab.destroy()
            comm.destroy()


def test_reference_harness_binary_runs_on_this_library(O):
    """oracle/_ref/ref_cg_solver is the reference's own src/main/cg_solver.cu, unmodified, compiled
    against this repo's include/ and linked against libspmv_amd.so (oracle/Makefile). Run it on the
    reference's shipped matrix. The reference main leaves the solution of its last warm-up solve in
    x, and cg_benchmark_with_stats_device backs that up as the start vector (cg_solver.cu(main):
    157-172, benchmark_stats.cu:112,124): its timed runs restart CG from the converged x. The oracle
    run the same way gives the numbers the binary must print."""
    import re
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_cg_solver")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_cg_solver not built (reference sources absent at build time)")
    out = subprocess.run([exe, os.path.join(GOLDEN, "example81x81.mtx"), "--mode=stencil5-csr"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    rp, ci, va = O.stencil5_csr(81, -4.0, -1.0)
    x1, h1, r1 = O.cg(rp, ci, va, 81, np.ones(6561), np.zeros(6561))
    x2, h2, r2 = O.cg(rp, ci, va, 81, np.ones(6561), x1)
    assert r1.iterations == 40
    m = re.search(r"Converged: YES in (\d+) iterations", text)
    assert m and int(m.group(1)) == r2.iterations, text[-2000:]
    sx = float(re.search(r"Sum\(x\):\s+(\S+)", text).group(1))
    nx = float(re.search(r"Norm2\(x\):\s+(\S+)", text).group(1))
    assert abs(sx - r2.solution_sum) <= 1e-10 * abs(r2.solution_sum)
    assert abs(nx - r2.solution_norm) <= 1e-10 * r2.solution_norm
    assert "valid runs" in text  # the reference main went through cg_benchmark_with_stats_device


def test_reference_spmv_bench_main_runs_on_this_library(golden, tmp_path):
    """oracle/_ref/ref_spmv_bench is the reference's src/main/main.cu minus its one unused `#include <cuda_runtime.h>`
    (dropped in a sed pipe by oracle/Makefile; nothing else is touched), compiled against this repo's include/ and
    linked against libspmv_amd.so. Run on the reference's shipped 81 x 81 matrix with both of its modes: the checksums
    it prints are the reference's known answers (Sum(y) = -52164), its JSON export carries the keys the
    reference's scripts scrape."""
    import json
    import re
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_spmv_bench")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_spmv_bench not built (reference sources absent at build time)")
    base = tmp_path / "res.json"
    out = subprocess.run([exe, os.path.join(GOLDEN, "example81x81.mtx"), "--mode=cusparse-csr,stencil5-csr", f"--json={base}"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    want = golden["survey_8c"]["81:-4.0"]
    sums = [float(v) for v in re.findall(r"Sum\(y\)\s*[:=]\s*(\S+)", out.stdout)]
    norms = [float(v) for v in re.findall(r"Norm2\(y\)\s*[:=]\s*(\S+)", out.stdout)]
    assert len(sums) == 2 and all(v == want["sum_y"] for v in sums), out.stdout[-3000:]
    assert len(norms) == 2 and all(abs(v - want["norm2_y"]) <= 1e-12 * want["norm2_y"] for v in norms)
    assert out.stdout.count("valid runs") == 2  # benchmark_with_stats ran for both modes
    for mode in ("cusparse-csr", "stencil5-csr"):
        rec = json.load(open(tmp_path / f"res_{mode}.json"))
        text = json.dumps(rec)
        assert "execution_time_ms" in text and "gflops" in text and "bandwidth_gb_s" in text  # keys scripts/run_all.sh scrapes
        assert rec["benchmark"]["validation"]["sum_y"] == want["sum_y"] if "benchmark" in rec else '"sum_y": -52164.0' in text


def test_reference_mpi_main_runs_on_this_library_unchanged(golden, tmp_path):
    """oracle/_ref/ref_cg_solver_mgpu is the reference's MPI main src/main/cg_solver_mgpu_stencil.cu -- the binary behind every
    published CG number -- minus its nsys capture window (`#include <cuda_profiler_api.h>`, cudaProfilerStart(), cudaProfilerStop():
    dropped in a sed pipe by oracle/Makefile, nothing else touched), compiled against this repo's include/ and the build
    container's MPI. It calls cg_solve_mgpu_partitioned with only MPI_Init around it (:23-27,105-131): the library finds the MPI
    world through the process's own MPI library (csrc/mpi_bootstrap.cpp). One rank here (a one-GPU box): the shipped 81 x 81 matrix
    converges in 40 iterations with the checksums SURVEY 8c lists; 3 warm-ups + 1 + 10 solves through
    cg_benchmark_with_stats_mgpu_partitioned; the JSON export carries the reference's keys."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_cg_solver_mgpu")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_cg_solver_mgpu not built (reference sources or an MPI absent at build time)")
    if not os.path.exists(mpiexec) or not os.path.exists(os.path.realpath(os.path.join(ROOT, "oracle", "_ref", "mpi", "libmpi.so.12"))):
        pytest.skip("no MPI on this box (/opt/conda/bin/mpiexec, libmpi.so.12)")
    js = tmp_path / "mgpu.json"
    out = subprocess.run([mpiexec, "-np", "1", exe, os.path.join(GOLDEN, "example81x81.mtx"), f"--json={js}"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    want = golden["survey_8c"]["81:-4.0"]
    assert "Profiled run: converged in 40 iterations" in text and "Converged: YES in 40 iterations" in text, text[-3000:]
    sx = float(re.search(r"Sum\(x\):\s+(\S+)", text).group(1))
    nx = float(re.search(r"Norm2\(x\):\s+(\S+)", text).group(1))
    assert want["cg_iterations"] == 40 and abs(sx - want["solution_sum"]) <= 1e-10 * abs(want["solution_sum"]) and abs(sx + 826.0838884253774) < 1e-7
    assert abs(nx - want["solution_norm"]) <= 1e-10 * want["solution_norm"]
    assert "10 valid runs" in text or "valid runs" in text
    rec = json.load(open(js))
    assert "median_ms" in json.dumps(rec)  # key scraped by scripts/run_all.sh:222-242


def test_slab_solver_on_a_general_spd_matrix(B, O, fresh_host_matrices):
    """No stencil announced (grid_size = -1): the slab solver runs the CSR loop for every row and the
    plain dot kernel; results still match the oracle's CG on the same matrix."""
    n = 400
    rng = np.random.default_rng(12)
    rows, cols, vals = [], [], []
    for i in range(n):  # symmetric, strictly diagonally dominant => SPD
        for j in rng.choice(n, size=4, replace=False):
            if j != i:
                v = float(rng.uniform(-1.0, 1.0))
                rows += [i, int(j)]
                cols += [int(j), i]
                vals += [v, v]
    import collections
    acc = collections.defaultdict(float)
    for r, c, v in zip(rows, cols, vals):
        acc[(r, c)] += v
    rowsum = np.zeros(n)
    for (r, c), v in acc.items():
        rowsum[r] += abs(v)
    for i in range(n):
        acc[(i, i)] = rowsum[i] + 1.0
    import matrices as M
    e = M.entries([(r, c, v) for (r, c), v in sorted(acc.items(), key=lambda kv: rng.random())])
    m = B.HostMatrix(e, n, n, -1)
    rp, ci, va = O.build_csr(e, n)
    b = rng.standard_normal(n)
    xo, ho, ro = O.cg(rp, ci, va, -1, b, np.zeros(n), tol=1e-9)
    slab = B.CgSlab.from_matrix(m)
    slab.set_vectors(b, np.zeros(n))
    st = slab.solve(tol=1e-9)
    assert st.iterations == ro.iterations and st.converged == 1
    assert hist_err(slab.history(), ho) < TOL and np.max(np.abs(slab.gather() - xo)) <= 1e-9 * np.max(np.abs(xo))
    slab.destroy()
    op = B.Operator("cusparse-csr")
    assert op.init(m) == 0
    x, hist, st2 = B.cg_solve(op, m, b, np.zeros(n), tol=1e-9, device=True)
    assert st2.iterations == ro.iterations and hist_err(hist, ho) < TOL
    op.free()


def test_slab_solver_nonzero_initial_guess_and_zero_iterations(B, O, fresh_host_matrices):
    """x0 != 0 (the first x update reads the stored x0, x is never pre-copied) and max_iters = 0."""
    n = 130
    rng = np.random.default_rng(3)
    b, x0 = rng.standard_normal(n * n), rng.standard_normal(n * n)
    m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
    slab = B.CgSlab.from_matrix(m)
    slab.set_vectors(b, x0)
    rp, ci, va = O.stencil5_csr(n)
    xo, ho, ro = O.cg_partitioned(rp, ci, va, n, b, x0, world=1)
    for _ in range(2):  # the second solve must start from x0 again, not from the first solution
        st = slab.solve()
        assert st.iterations == ro.iterations and hist_err(slab.history(), ho) < TOL
        assert np.max(np.abs(slab.gather() - xo)) <= TOL * np.max(np.abs(xo))
    st0 = slab.solve(max_iters=0)
    assert st0.iterations == 0 and st0.converged == 0 and np.array_equal(slab.gather(), x0)
    slab.destroy()


@pytest.mark.parametrize("name", ["sym_spd40.mtx", "sym_stencil8.mtx", "sym_hand3.mtx"])
def test_symmetric_matrix_market_file_through_the_operators_and_cg(B, O, fresh_host_matrices, name):
    """a `coordinate real symmetric` file -> load_matrix_market (expanded to general, the reference reader's
    own expansion order, checked against reference src/io/io.cu:189-310 in test_host_logic.py) -> every operator:
    SpMV bit-exact against oracle_spmv_csr on the sorted CSR, CG (the two SPD files) at 1e-10 against the oracle."""
    m = B.load_matrix_market(os.path.join(GOLDEN, name))
    rows = m.c.rows
    rp, ci, va = O.build_csr(m.entries, rows)
    x = np.random.default_rng(3).standard_normal(rows)
    want = O.spmv_csr(rp, ci, va, x)
    grid = m.c.grid_size
    for mode in ("cusparse-csr", "ellpack", "stencil5-csr", "stencil5-ellpack"):
        op = B.Operator(mode)
        assert op.init(m) == 0
        got, ms = op.run_timed(x)
        if mode.startswith("stencil5") and grid > 0:   # both stencil-aware operators: interior rows in W,C,E,N,S order
            assert np.array_equal(got, O.spmv_stencil5(rp, ci, va, x, grid))
        else:
            assert np.array_equal(got, want), mode
        op.free()
    if name == "sym_hand3.mtx":
        return  # the hand case has no (3,3) entry: symmetric but not positive definite, SpMV only
    op = B.Operator("cusparse-csr")
    assert op.init(m) == 0
    xs, hist, st = B.cg_solve(op, m, np.ones(rows), np.zeros(rows), device=True)
    xo, ho, ro = O.cg(rp, ci, va, -1, np.ones(rows), np.zeros(rows), device_form=True)
    assert st.iterations == ro.iterations and st.converged == 1
    assert hist_err(hist, ho) < TOL and np.max(np.abs(xs - xo)) <= TOL * np.max(np.abs(xo))
    op.free()
    # and the slab solver on the same host matrix
    B.lib().spmv_amd_reset_host_matrices()
    slab = B.CgSlab.from_matrix(m)
    st2 = slab.solve()
    assert st2.iterations == ro.iterations and hist_err(slab.history(), ho) < TOL
    assert np.max(np.abs(slab.gather() - xo)) <= TOL * np.max(np.abs(xo))
    slab.destroy()


def test_stop_at_makes_a_stand_in_slab_do_a_converging_solves_work(Blab, monkeypatch):
    """Timing aid of the scaling probe (LAB build, set_option "stop_at"): a stand-in slab's mirrored system does not converge in 14 iterations, so with max_iters
    alone it runs one direction update + halo exchange more than the rank of a real job, whose last iteration converges. With
    stop_at = k the k-th iteration counts as the converging one: k iterations, converged, k - 1 direction updates, and the same
    residuals up to there as the free-running solve."""
    monkeypatch.setenv("SPMV_AMD_SELF_NEIGHBOUR", "1")
    monkeypatch.setenv("SPMV_AMD_FORCE_COLLECTIVES", "1")
    B = Blab
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
    slab = B.CgSlab.stencil5_as(1024, 1, 4, comm)
    free = slab.solve(max_iters=9, tol=0.0)
    h_free = slab.history().copy()
    assert free.iterations == 9 and free.converged == 0
    slab.set_option("stop_at", 9)
    st = slab.solve(max_iters=9, tol=0.0)
    assert st.iterations == 9 and st.converged == 1 and np.array_equal(slab.history(), h_free)
    _, tl = slab.timeline_solve(max_iters=9, tol=0.0)
    assert tl["iterations"] == 9 and tl["direction_updates"] == 8
    slab.set_option("stop_at", 0)
    assert slab.solve(max_iters=9, tol=0.0).converged == 0
    slab.destroy()
    comm.destroy()


def test_the_product_library_has_no_hooks(B, monkeypatch):
    """The product library neither exports the LAB entry points nor listens to the LAB build's environment switches: with
    SPMV_AMD_SELF_NEIGHBOUR / SPMV_AMD_FORCE_COLLECTIVES / SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE set, a single-rank RCCL
    communicator is a plain single rank (no halo exchange, no all-reduce) and the solve is the committed golden one."""
    for name in B.LAB_ONLY_SYMBOLS:
        assert not hasattr(B.lib(), name), name
    with pytest.raises(RuntimeError):
        B.CgSlab.stencil5_as(640, 0, 2, None)
    monkeypatch.setenv("SPMV_AMD_SELF_NEIGHBOUR", "1")
    monkeypatch.setenv("SPMV_AMD_FORCE_COLLECTIVES", "1")
    monkeypatch.setenv("SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE", "2")
    comm = B.Comm.rccl(0, 1, B.Comm.unique_id())
    assert comm is not None
    slab = B.CgSlab.stencil5(512, comm)
    st, tl = slab.timeline_solve()
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))["cases"]["512:5.0"]["cg"]
    assert st.iterations == gold["iterations"] and hist_err(slab.history(), np.array(gold["history"])) < TOL
    assert tl["halo_exchange_on_side_stream_us"] == 0.0  # nothing was exchanged: the switch was not read
    slab.destroy()
    comm.destroy()


def test_long_row_matrices_take_the_wavefront_kernel_and_the_unfused_loop(B, O, fresh_host_matrices):
    """a CSR matrix whose MEAN row length exceeds 192 (here a band of half-width 160: ~312 entries per row) is
    sent to the one-row-per-wavefront kernel by the auto rule. That kernel's row sum is a tree (not the sequential sum: <= 1e-12
    instead of bit-exact) and it writes no dot partials, so cg_solve_device runs the unfused loop on it (run_device + a dot pass,
    +16 B/row per iteration). Both must still match the oracle."""
    e, r, c, _ = M.banded(3000, 160, diagonal=12.0)
    m = B.HostMatrix(e, r, c, -1)
    op = B.Operator("cusparse-csr")
    assert op.init(m) == 0 and op.variant() == "csr/wavefront"
    rp, ci, va = O.build_csr(e, r)
    x = np.random.default_rng(11).standard_normal(c)
    want = O.spmv_csr(rp, ci, va, x)
    got, _ = op.run_timed(x)
    scale = np.maximum(np.abs(want), O.spmv_csr(rp, ci, np.abs(va), np.abs(x)))
    assert np.max(np.abs(got - want) / scale) <= 1e-12
    b = np.random.default_rng(12).standard_normal(r)
    xs, hist, st = B.cg_solve(op, m, b, np.zeros(r), device=True)
    xo, ho, ro = O.cg(rp, ci, va, -1, b, np.zeros(r), device_form=True)
    assert st.iterations == ro.iterations and st.converged == 1 and hist_err(hist, ho) < TOL
    assert np.max(np.abs(xs - xo)) <= TOL * np.max(np.abs(xo))
    op.free()


@pytest.mark.parametrize("mode", ["cusparse-csr", "ellpack"])
def test_cg_solve_device_on_a_banded_spd_matrix(B, O, fresh_host_matrices, mode):
    """cg_solve_device on a matrix that is no stencil at all (banded, SPD, grid_size = -1), through the CSR and ELLPACK
    operators' fused launches, against the oracle's device-form CG on the same CSR arrays."""
    e, r, c, _ = M.banded(20000, 5)
    m = B.HostMatrix(e, r, c, -1)
    op = B.Operator(mode)
    assert op.init(m) == 0
    b = np.random.default_rng(9).standard_normal(r)
    x, hist, st = B.cg_solve(op, m, b, np.zeros(r), device=True)
    rp, ci, va = O.build_csr(e, r)
    xo, ho, ro = O.cg(rp, ci, va, -1, b, np.zeros(r), device_form=True)
    assert st.iterations == ro.iterations and st.converged == 1 and hist_err(hist, ho) < TOL
    assert np.max(np.abs(x - xo)) <= TOL * np.max(np.abs(xo))
    op.free()


def test_cg_solve_host_interface_with_and_without_a_device_entry_point(B, O, fresh_host_matrices):
    """cg_solve (reference cg_solver.cu:154-378: host scalars). An operator WITHOUT run_device (legal upstream: callers check
    for NULL) is driven through run_timed with host arrays exactly as the reference does; one that has run_device keeps the
    vector on the GPU. Same numbers either way, equal to the oracle's host form; cg_solve_device refuses the first kind."""
    n = 130
    m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
    real = B.Operator("stencil5-csr")
    assert real.init(m) == 0
    timed_calls = []

    def run_timed(x, y, ms):
        timed_calls.append(1)
        return real.op.contents.run_timed(x, y, ms)

    keep = (B.INIT_FN(lambda mat: 0), B.RUN_TIMED_FN(run_timed), B.FREE_FN(lambda: None))
    table = B.SpmvOperator(b"host-only", keep[0], keep[1], B.RUN_DEVICE_FN(), keep[2])  # run_device = NULL

    class HostOnly:
        op = C.pointer(table)

    rp, ci, va = O.stencil5_csr(n)
    xo, ho, ro = O.cg(rp, ci, va, n, np.ones(n * n), np.zeros(n * n), device_form=False)
    x1, h1, s1 = B.cg_solve(HostOnly, m, np.ones(n * n), np.zeros(n * n), device=False)
    assert len(timed_calls) == s1.iterations + 1  # every SpMV went through run_timed
    x2, h2, s2 = B.cg_solve(real, m, np.ones(n * n), np.zeros(n * n), device=False)
    assert len(timed_calls) == s1.iterations + 1  # ... and none of the second solve's did
    for x, h, s in ((x1, h1, s1), (x2, h2, s2)):
        assert s.iterations == ro.iterations and s.converged == 1 and hist_err(h, ho) < TOL
        assert np.max(np.abs(x - xo)) <= TOL * np.max(np.abs(xo))
    assert np.array_equal(h1, h2) and np.array_equal(x1, x2)
    with pytest.raises(RuntimeError):
        B.cg_solve(HostOnly, m, np.ones(n * n), np.zeros(n * n), device=True)
    real.free()
