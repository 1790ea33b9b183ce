"""ctypes view of oracle/liboracle.so (and of oracle/_ref when it has been built).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. The product (cuda-spmv-benchmark_amd/) never
imports this module; see oracle/spmv_oracle.h for what each function restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
OMP_LIB_PATH = os.path.join(HERE, "liboracle_omp.so")  # threaded build: bench.py's all-cores baseline only
REF_IO_PATH = os.path.join(HERE, "_ref", "libref_io.so")
REF_GEN_PATH = os.path.join(HERE, "_ref", "generate_matrix")

ENTRY_DTYPE = np.dtype([("row", np.int32), ("col", np.int32), ("value", np.float64)], align=True)
assert ENTRY_DTYPE.itemsize == 16


class CGResult(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("converged", C.c_int),
        ("residual_norm", C.c_double),
        ("b_norm", C.c_double),
        ("solution_sum", C.c_double),
        ("solution_norm", C.c_double),
    ]


class BenchStats(C.Structure):
    _fields_ = [
        ("median_ms", C.c_double),
        ("mean_ms", C.c_double),
        ("std_dev_ms", C.c_double),
        ("min_ms", C.c_double),
        ("max_ms", C.c_double),
        ("valid_runs", C.c_int),
        ("outliers_removed", C.c_int),
    ]


def build(force=False):
    """Compiles liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
        os.path.join(HERE, "spmv_oracle.c")
    ):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so", "liboracle_omp.so"], stdout=subprocess.DEVNULL)
    if not os.path.exists(REF_IO_PATH):
        subprocess.call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_stencil5_nnz.restype = C.c_longlong
        _lib.oracle_stencil5_coo.restype = C.c_longlong
        _lib.oracle_dot_host.restype = C.c_double
        _lib.oracle_dot_device.restype = C.c_double
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def _ip(a):
    return _p(a, C.c_int)


def _dp(a):
    return _p(a, C.c_double)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------------------------------------------------------- matrices


def stencil5_coo(n, center=5.0, off=-1.0):
    nnz = lib().oracle_stencil5_nnz(n)
    e = np.zeros(nnz, dtype=ENTRY_DTYPE)
    got = lib().oracle_stencil5_coo(n, C.c_double(center), C.c_double(off), e.ctypes.data_as(C.c_void_p))
    assert got == nnz
    return e


def stencil5_csr(n, center=5.0, off=-1.0):
    nnz = lib().oracle_stencil5_nnz(n)
    rp = np.empty(n * n + 1, dtype=np.int32)
    ci = np.empty(nnz, dtype=np.int32)
    va = np.empty(nnz, dtype=np.float64)
    lib().oracle_stencil5_csr(n, C.c_double(center), C.c_double(off), _ip(rp), _ip(ci), _dp(va))
    return rp, ci, va


def build_csr(entries, rows):
    entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
    nnz = len(entries)
    rp = np.empty(rows + 1, dtype=np.int32)
    ci = np.empty(nnz, dtype=np.int32)
    va = np.empty(nnz, dtype=np.float64)
    rc = lib().oracle_build_csr(entries.ctypes.data_as(C.c_void_p), rows, nnz, _ip(rp), _ip(ci), _dp(va))
    assert rc == 0
    return rp, ci, va


def interior_csr_offset(row, n):
    return lib().oracle_interior_csr_offset(int(row), int(n))


def partition_rows(n, world, rank):
    off, nl = C.c_int(), C.c_int()
    lib().oracle_partition_rows(n, world, rank, C.byref(off), C.byref(nl))
    return off.value, nl.value


# ---------------------------------------------------------------- SpMV


def spmv_csr(rp, ci, va, x):
    x = _f64(x)
    y = np.empty(len(rp) - 1, dtype=np.float64)
    lib().oracle_spmv_csr(len(rp) - 1, _ip(rp), _ip(ci), _dp(va), _dp(x), _dp(y))
    return y


def spmv_stencil5(rp, ci, va, x, grid_size, alpha=1.0):
    x = _f64(x)
    y = np.empty(len(rp) - 1, dtype=np.float64)
    lib().oracle_spmv_stencil5(len(rp) - 1, _ip(rp), _ip(ci), _dp(va), _dp(x), _dp(y), int(grid_size), C.c_double(alpha))
    return y


def spmv_halo(rp_local, ci, va, x_local, halo_prev, halo_next, row_offset, n_total, grid_size):
    x_local = _f64(x_local)
    hp = _f64(halo_prev) if halo_prev is not None else None
    hn = _f64(halo_next) if halo_next is not None else None
    n_local = len(rp_local) - 1
    y = np.empty(n_local, dtype=np.float64)
    lib().oracle_spmv_halo(_ip(rp_local), _ip(ci), _dp(va), _dp(x_local), _dp(hp), _dp(hn), _dp(y), n_local, int(row_offset), int(n_total), int(grid_size))
    return y


def build_ell(rp, ci, va):
    rows = len(rp) - 1
    w = lib().oracle_ell_width(rows, _ip(rp))
    idx = np.empty(rows * w, dtype=np.int32)
    val = np.empty(rows * w, dtype=np.float64)
    lib().oracle_build_ell(rows, _ip(rp), _ip(ci), _dp(va), w, _ip(idx), _dp(val))
    return w, idx, val


def spmv_ell(rows, width, idx, val, x, y0=None, alpha=1.0, beta=0.0):
    x = _f64(x)
    y = np.zeros(rows, dtype=np.float64) if y0 is None else _f64(y0).copy()
    lib().oracle_spmv_ell(rows, width, _ip(idx), _dp(val), _dp(x), _dp(y), C.c_double(alpha), C.c_double(beta))
    return y


def axpy(alpha, x, y):
    y = _f64(y).copy()
    lib().oracle_axpy(len(y), C.c_double(alpha), _dp(_f64(x)), _dp(y))
    return y


def axpby(alpha, x, beta, y):
    x, y = _f64(x), _f64(y)
    z = np.empty_like(x)
    lib().oracle_axpby(len(x), C.c_double(alpha), _dp(x), C.c_double(beta), _dp(y), _dp(z))
    return z


def axpy_sub(alpha, x, y):
    y = _f64(y).copy()
    lib().oracle_axpy_sub(len(y), C.c_double(alpha), _dp(_f64(x)), _dp(y))
    return y


def update_p(r, beta, p):
    p = _f64(p).copy()
    lib().oracle_update_p(len(p), _dp(_f64(r)), C.c_double(beta), _dp(p))
    return p


def dot_host(x, y):
    x, y = _f64(x), _f64(y)
    return lib().oracle_dot_host(len(x), _dp(x), _dp(y))


def dot_device(x, y):
    x, y = _f64(x), _f64(y)
    return lib().oracle_dot_device(len(x), _dp(x), _dp(y))


# ---------------------------------------------------------------- CG


def cg(rp, ci, va, grid_size, b, x0, max_iters=1000, tol=1e-6, device_form=True):
    n = len(rp) - 1
    b = _f64(b)
    x = _f64(x0).copy()
    hist = np.zeros(max_iters + 1, dtype=np.float64)
    res = CGResult()
    rc = lib().oracle_cg(n, _ip(rp), _ip(ci), _dp(va), int(grid_size), _dp(b), _dp(x), max_iters, C.c_double(tol), int(bool(device_form)), _dp(hist), len(hist), C.byref(res))
    assert rc == 0
    return x, hist[: res.iterations + 1].copy(), res


def cg_all_cores(rp, ci, va, grid_size, b, x0, threads, max_iters=1000, tol=1e-6):
    """oracle_cg (device form) from the threaded build, for timing the all-cores CPU baseline. Its dot products are
    summed per thread, so compare it with the serial oracle at 1e-10, never bit for bit."""
    if not os.path.exists(OMP_LIB_PATH):
        subprocess.check_call(["make", "-C", HERE, "liboracle_omp.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(OMP_LIB_PATH)
    C.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
    n = len(rp) - 1
    b = _f64(b)
    x = _f64(x0).copy()
    hist = np.zeros(max_iters + 1, dtype=np.float64)
    res = CGResult()
    rc = L.oracle_cg(n, _ip(rp), _ip(ci), _dp(va), int(grid_size), _dp(b), _dp(x), max_iters, C.c_double(tol), 1, _dp(hist), len(hist), C.byref(res))
    assert rc == 0
    return x, hist[: res.iterations + 1].copy(), res


def omp_lib(threads):
    """The threaded build (liboracle_omp.so): timing of the all-cores CPU baseline only."""
    if not os.path.exists(OMP_LIB_PATH):
        subprocess.check_call(["make", "-C", HERE, "liboracle_omp.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(OMP_LIB_PATH)
    C.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
    return L


def spmv_into(rp, ci, va, x, y, grid_size=None, L=None):
    """oracle_spmv_stencil5 (grid_size given) or oracle_spmv_csr into a caller-owned y, without allocating: the form
    bench.py times. L = another build of the oracle (omp_lib) or None for the serial one. Rows are independent,
    so the threaded build's SpMV results are bit-identical to the serial ones."""
    L = L or lib()
    if grid_size is None:
        L.oracle_spmv_csr(len(rp) - 1, _ip(rp), _ip(ci), _dp(va), _dp(x), _dp(y))
    else:
        L.oracle_spmv_stencil5(len(rp) - 1, _ip(rp), _ip(ci), _dp(va), _dp(x), _dp(y), int(grid_size), C.c_double(1.0))
    return y


def cg_partitioned(rp, ci, va, grid_size, b, x0, world, max_iters=1000, tol=1e-6):
    n = len(rp) - 1
    b = _f64(b)
    x = _f64(x0).copy()
    hist = np.zeros(max_iters + 1, dtype=np.float64)
    res = CGResult()
    rc = lib().oracle_cg_partitioned(n, _ip(rp), _ip(ci), _dp(va), int(grid_size), _dp(b), _dp(x), max_iters, C.c_double(tol), int(world), _dp(hist), len(hist), C.byref(res))
    assert rc == 0, f"oracle_cg_partitioned rc={rc}"
    return x, hist[: res.iterations + 1].copy(), res


# ---------------------------------------------------------------- harness


def bench_stats(times):
    t = _f64(times)
    out = BenchStats()
    rc = lib().oracle_bench_stats(_dp(t), len(t), C.byref(out))
    return rc, out


def spmv_metrics(ms, rows, cols, nnz):
    g, b = C.c_double(), C.c_double()
    lib().oracle_spmv_metrics(C.c_double(ms), rows, cols, nnz, C.byref(g), C.byref(b))
    return g.value, b.value


# ---------------------------------------------------------------- oracle/_ref (the reference's own io.cu)


class RefMatrixData(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("nnz", C.c_int), ("grid_size", C.c_int), ("entries", C.c_void_p)]


def ref_io_available():
    return os.path.exists(REF_IO_PATH)


_ref = None


def ref_io():
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_IO_PATH)
    return _ref


def ref_load_matrix_market(path):
    """load_matrix_market of the reference's own io.cu (compiled into oracle/_ref)."""
    m = RefMatrixData()
    rc = ref_io().load_matrix_market(path.encode(), C.byref(m))
    assert rc == 0
    e = np.ctypeslib.as_array((C.c_byte * (16 * m.nnz)).from_address(m.entries)).view(ENTRY_DTYPE).copy()
    return m.rows, m.cols, m.nnz, m.grid_size, e


def ref_write_stencil5(n, path):
    return ref_io().write_matrix_market_stencil5(int(n), path.encode())


def ref_read_matrix_symtogen(path):
    """The reference's own symmetric reader (src/io/io.cu:189-310, compiled in place into oracle/_ref):
    returns rows, cols, stored nnz, expanded nnz and its CSR arrays (columns in a row unsorted, io.cu:288-307)."""
    L = ref_io()
    m = RefMatrixData()
    rows, cols, nnz, full = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rp, ci, va = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_double)()
    L.read_matrix_symtogen(C.byref(m), os.fsencode(path), C.byref(rows), C.byref(cols), C.byref(nnz), C.byref(rp), C.byref(ci), C.byref(va), C.byref(full))
    out = (rows.value, cols.value, nnz.value, full.value, np.ctypeslib.as_array(rp, shape=(rows.value + 1,)).copy(),
           np.ctypeslib.as_array(ci, shape=(full.value,)).copy(), np.ctypeslib.as_array(va, shape=(full.value,)).copy())
    libc = C.CDLL(None)
    for ptr in (rp, ci, va):
        libc.free(ptr)
    return out
