"""The CG building blocks one at a time against the oracle's element-wise forms (reference kernels
cg_solver.cu:38-132,384-409; mgpu :125-140): a wrong-but-compensating element-wise operation would survive a
whole-solve comparison at 1e-10, not these. Element-wise kernels must be BIT-exact (the fusions nvcc makes are
written out as fma() on both sides); dot products re-order the sum and are held to 1e-13 relative of sum|x_i y_i|."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 63, 64, 65, 127, 128, 129, 1000, 4097, 65536 + 3, 1_000_001]


def dev(B, a):
    return B.DeviceVector.from_host(a)


@pytest.mark.parametrize("n", SIZES)
def test_axpy_axpby_axpy_dev_update_p_bit_exact(B, O, n):
    rng = np.random.default_rng(n)
    x, y, r = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    a, b = float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2))
    L = B.lib()
    dx, dy = dev(B, x), dev(B, y)
    assert L.spmv_amd_blas1_axpy(n, a, dx.ptr, dy.ptr) == 0
    assert np.array_equal(dy.to_host(), O.axpy(a, x, y))                  # y = fma(a, x, y)
    dz = B.DeviceVector(n, fill=0.0)
    dy2 = dev(B, y)
    assert L.spmv_amd_blas1_axpby(n, a, dx.ptr, b, dy2.ptr, dz.ptr) == 0
    assert np.array_equal(dz.to_host(), O.axpby(a, x, b, y))              # z = fma(a, x, b*y)
    for subtract, want in ((0, O.axpy(a, x, y)), (1, O.axpy_sub(a, x, y))):
        dy3 = dev(B, y)
        assert L.spmv_amd_blas1_axpy_dev(n, a, dx.ptr, dy3.ptr, subtract) == 0
        assert np.array_equal(dy3.to_host(), want)                        # y = fma(+-a, x, y), a read from the device
        dy3.free()
    dr, dp = dev(B, r), dev(B, y)
    assert L.spmv_amd_blas1_update_p_dev(n, dr.ptr, b, dp.ptr) == 0
    assert np.array_equal(dp.to_host(), O.update_p(r, b, y))              # p = fma(b, p, r)
    for v in (dx, dy, dy2, dz, dr, dp):
        v.free()


@pytest.mark.parametrize("n", SIZES)
def test_dot_against_both_reference_shapes(B, O, n):
    rng = np.random.default_rng(100 + n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    dx, dy = dev(B, x), dev(B, y)
    got = C.c_double()
    assert B.lib().spmv_amd_blas1_dot(n, dx.ptr, dy.ptr, C.byref(got)) == 0
    scale = float(np.sum(np.abs(x * y)))
    for want in (O.dot_device(x, y), O.dot_host(x, y)):   # final_sum_kernel shape and host-loop shape
        assert abs(got.value - want) <= 1e-13 * scale
    again = C.c_double()
    assert B.lib().spmv_amd_blas1_dot(n, dx.ptr, dy.ptr, C.byref(again)) == 0
    assert again.value == got.value                        # fixed reduction shape: bit-reproducible
    ones = dev(B, np.ones(n))
    assert B.lib().spmv_amd_blas1_dot(n, ones.ptr, ones.ptr, C.byref(got)) == 0 and got.value == float(n)
    for v in (dx, dy, ones):
        v.free()


@pytest.mark.parametrize("n", [1, 2, 65, 128, 1000, 4097, 1_000_001])
@pytest.mark.parametrize("reverse", [0, 1])
def test_fused_slab_steps_equal_the_unfused_reference_kernels(B, O, n, reverse):
    """The slab solver's fused passes, element for element: r -= alpha*Ap (alpha = rr_old/pAp, one IEEE division) is
    axpy_kernel(-alpha, Ap, r) (mgpu :612); p_out = r + beta*p_in is axpby_kernel(1.0, r, beta, p) (:682);
    r = b - Ap, p = r is axpy_kernel(-1, Ap, b) + copy (:475-476). Sweep direction must not matter."""
    rng = np.random.default_rng(7 * n + reverse)
    Ap, r, p, b = (rng.standard_normal(n) for _ in range(4))
    rr_old, pAp, beta = 3.25, 7.5, 0.3125
    sc = (C.c_double * 3)(rr_old, pAp, beta)
    L = B.lib()
    dot = C.c_double()
    # which = 0
    dA, dr = dev(B, Ap), dev(B, r)
    assert L.spmv_amd_cg_fused_step(0, n, sc, dA.ptr, dr.ptr, None, reverse, C.byref(dot)) == 0
    want_r = O.axpy(-(rr_old / pAp), Ap, r)
    got_r = dr.to_host()
    assert np.array_equal(got_r, want_r)
    assert abs(dot.value - O.dot_device(want_r, want_r)) <= 1e-13 * float(want_r @ want_r)
    # which = 1
    dr2, dp, dq = dev(B, r), dev(B, p), B.DeviceVector(n, fill=0.0)
    assert L.spmv_amd_cg_fused_step(1, n, sc, dr2.ptr, dp.ptr, dq.ptr, reverse, None) == 0
    assert np.array_equal(dq.to_host(), O.axpby(1.0, r, beta, p))
    assert np.array_equal(dp.to_host(), p)  # written out of place: the old direction is untouched
    # which = 2
    db, dA2, drp = dev(B, b), dev(B, Ap), B.DeviceVector(2 * n, fill=0.0)
    assert L.spmv_amd_cg_fused_step(2, n, sc, db.ptr, dA2.ptr, drp.ptr, 0, C.byref(dot)) == 0
    out = drp.to_host()
    want0 = O.axpy(-1.0, Ap, b)
    assert np.array_equal(out[:n], want0) and np.array_equal(out[n:], want0)
    assert abs(dot.value - O.dot_device(want0, want0)) <= 1e-13 * float(want0 @ want0)
    for v in (dA, dr, dr2, dp, dq, db, dA2, drp):
        v.free()
