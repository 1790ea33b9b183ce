"""ctypes binding of libspmv_amd.so, used by tests/, bench.py and __graft_entry__.py.

This module is plumbing only: every compute call goes through the C-ABI declared in
include/spmv_amd/api.h into hand-written HIP kernels. There is no CPU fallback; if the
library has not been built, or no GPU is visible when a compute entry point is called, the
call fails loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# The LAB build: the same sources with the test and measurement hooks compiled in (include/spmv_amd/lab.h). A second instance of
# this module bound to it: use_lab() below (tests' `Blab` fixture, tools/, bench.py's scaling probe).
LAB_LIB_PATH = os.path.join(PKG_DIR, "lib", "libspmv_amd_lab.so")
# SPMV_AMD_LIB: measurement aid, lets another build of the library be loaded instead ("lab" = the LAB build)
LIB_PATH = os.environ.get("SPMV_AMD_LIB") or os.path.join(PKG_DIR, "lib", "libspmv_amd.so")
if LIB_PATH == "lab":
    LIB_PATH = LAB_LIB_PATH

ENTRY_DTYPE = np.dtype([("row", np.int32), ("col", np.int32), ("value", np.float64)], align=True)
assert ENTRY_DTYPE.itemsize == 16
COMM_ID_BYTES = 256


# ---------------------------------------------------------------- structs (include/spmv_amd/types.h)
class MatrixData(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("nnz", C.c_int), ("grid_size", C.c_int), ("entries", C.c_void_p)]


class CSRMatrix(C.Structure):
    _fields_ = [("nb_rows", C.c_int), ("nb_cols", C.c_int), ("nb_nonzeros", C.c_int), ("row_ptr", C.POINTER(C.c_int)), ("col_indices", C.POINTER(C.c_int)), ("values", C.POINTER(C.c_double))]


class ELLPACKMatrix(C.Structure):
    _fields_ = [("nb_rows", C.c_int), ("nb_cols", C.c_int), ("ell_width", C.c_int), ("grid_size", C.c_int), ("indices", C.POINTER(C.c_int)), ("nb_nonzeros", C.c_int), ("values", C.POINTER(C.c_double))]


INIT_FN = C.CFUNCTYPE(C.c_int, C.POINTER(MatrixData))
RUN_TIMED_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double))
RUN_DEVICE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
FREE_FN = C.CFUNCTYPE(None)


class SpmvOperator(C.Structure):
    _fields_ = [("name", C.c_char_p), ("init", INIT_FN), ("run_timed", RUN_TIMED_FN), ("run_device", RUN_DEVICE_FN), ("free", FREE_FN)]


class BenchmarkStats(C.Structure):
    _fields_ = [("median_ms", C.c_double), ("mean_ms", C.c_double), ("std_dev_ms", C.c_double), ("min_ms", C.c_double), ("max_ms", C.c_double), ("valid_runs", C.c_int), ("outliers_removed", C.c_int)]


class CGConfig(C.Structure):
    _fields_ = [("max_iters", C.c_int), ("tolerance", C.c_double), ("verbose", C.c_int), ("enable_detailed_timers", C.c_int)]


class CGStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("residual_norm", C.c_double), ("time_total_ms", C.c_double), ("time_spmv_ms", C.c_double), ("time_blas1_ms", C.c_double), ("time_reductions_ms", C.c_double), ("converged", C.c_int), ("solution_sum", C.c_double), ("solution_norm", C.c_double)]


CGConfigMultiGPU = CGConfig  # same layout (include/spmv_amd/types.h)


class CGStatsMultiGPU(C.Structure):
    _fields_ = [
        ("iterations", C.c_int), ("residual_norm", C.c_double), ("time_total_ms", C.c_double), ("time_spmv_ms", C.c_double),
        ("time_blas1_ms", C.c_double), ("time_reductions_ms", C.c_double), ("time_allreduce_ms", C.c_double),
        ("time_allgather_ms", C.c_double), ("converged", C.c_int), ("time_dot_rs_initial_ms", C.c_double),
        ("time_dot_pAp_ms", C.c_double), ("time_dot_rs_new_ms", C.c_double), ("time_axpy_update_x_ms", C.c_double),
        ("time_axpy_update_r_ms", C.c_double), ("time_axpby_update_p_ms", C.c_double), ("time_initial_r_ms", C.c_double),
        ("solution_sum", C.c_double), ("solution_norm", C.c_double),
    ]


HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int))
BARRIER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)

# Every extern "C" symbol include/spmv_amd/api.h declares (the CPU test-suite checks them all).
DECLARED_SYMBOLS = [
    "csr_mat", "ellpack_matrix", "build_ellpack_from_csr_local", "ensure_ellpack_structure_built", "get_operator",
    "calculate_spmv_metrics", "get_gpu_properties", "print_benchmark_metrics", "print_metrics_json", "print_metrics_csv",
    "read_matrix_type", "read_matrix_general", "read_matrix_symtogen", "load_matrix_market", "convert_csr_to_ellpack",
    "write_matrix_market_stencil5", "benchmark_with_stats", "cg_benchmark_with_stats_device",
    "cg_benchmark_with_stats_mgpu_partitioned", "export_cg_json", "export_cg_mgpu_json", "export_cg_csv",
    "spmv_amd_build_csr_struct", "spmv_amd_build_ellpack_from_csr_struct", "spmv_amd_cg_solve", "spmv_amd_cg_solve_device",
    "spmv_amd_cg_solve_mgpu_partitioned", "spmv_amd_reset_host_matrices", "spmv_amd_interior_csr_offset",
    "spmv_amd_partition_rows", "spmv_amd_device_count", "spmv_amd_set_device", "spmv_amd_current_device", "spmv_amd_stream_ceiling", "spmv_amd_stream_ceiling_mix", "spmv_amd_device_alloc", "spmv_amd_device_free",
    "spmv_amd_copy_to_device", "spmv_amd_copy_to_host", "spmv_amd_device_fill_f64", "spmv_amd_device_synchronize",
    "spmv_amd_init_stencil5_synthetic", "spmv_amd_ellpack_run_device_scaled", "spmv_amd_download_device_csr", "spmv_amd_time_run_device", "spmv_amd_operator_variant",
    "spmv_amd_operator_select_variant", "spmv_amd_cg_last_history", "spmv_amd_comm_unique_id", "spmv_amd_comm_create_rccl",
    "spmv_amd_comm_create_staged", "spmv_amd_comm_destroy", "spmv_amd_comm_set_world", "spmv_amd_comm_rank", "spmv_amd_comm_size", "spmv_amd_comm_selftest",
    "spmv_amd_comm_barrier", "spmv_amd_comm_transport", "spmv_amd_comm_transport_ranks",
    "spmv_amd_comm_mailbox_enable", "spmv_amd_comm_mailbox_prepare", "spmv_amd_comm_mailbox_connect", "spmv_amd_comm_mailbox_selftest",
    "spmv_amd_comm_mailbox_disable", "spmv_amd_comm_mailbox_ready",
    "spmv_amd_cg_slab_create", "spmv_amd_cg_slab_create_stencil5", "spmv_amd_cg_slab_set_vectors", "spmv_amd_cg_slab_solve",
    "spmv_amd_cg_slab_gather", "spmv_amd_cg_slab_loop_shape", "spmv_amd_cg_slab_history", "spmv_amd_cg_slab_spmv", "spmv_amd_cg_slab_info",
    "spmv_amd_cg_slab_time_spmv", "spmv_amd_cg_slab_set_timeline", "spmv_amd_operator_placement", "spmv_amd_cg_slab_placement", "spmv_amd_cg_slab_tile_runs", "spmv_amd_cg_slab_setup_ms", "spmv_amd_cg_slab_spmv_launch_ms",  "spmv_amd_cg_release_workspace", "spmv_amd_cg_slab_timeline_names", "spmv_amd_cg_slab_timeline", "spmv_amd_cg_slab_variant", "spmv_amd_cg_slab_destroy", "spmv_amd_version", "spmv_amd_write_stencil5_values",
    "spmv_amd_blas1_axpy", "spmv_amd_blas1_axpby", "spmv_amd_blas1_axpy_dev", "spmv_amd_blas1_update_p_dev", "spmv_amd_blas1_dot",
    "spmv_amd_cg_fused_step",
]
# What the LAB build exports on top of that (include/spmv_amd/lab.h); the product library must NOT have these.
LAB_ONLY_SYMBOLS = ["spmv_amd_cg_slab_create_stencil5_as", "spmv_amd_cg_slab_set_option"]
# C++-linkage entry points kept under the reference's own names (Itanium-mangled).
DECLARED_CXX_SYMBOLS = [
    "SPMV_CSR", "SPMV_STENCIL5_CSR", "SPMV_STENCIL_HALO_MGPU", "SPMV_ELLPACK", "SPMV_STENCIL5_ELLPACK",
    "_Z16build_csr_structP10MatrixData",
    "_Z29build_ellpack_from_csr_structPK9CSRMatrixP13ELLPACKMatrixPi",
    "_Z8cg_solveP12SpmvOperatorP10MatrixDataPKdPd8CGConfigP7CGStats",
    "_Z15cg_solve_deviceP12SpmvOperatorP10MatrixDataPKdPd8CGConfigP7CGStats",
    "_Z25cg_solve_mgpu_partitionedP12SpmvOperatorP10MatrixDataPKdPd16CGConfigMultiGPUP15CGStatsMultiGPU",
]


def build(force=False):
    """Compiles libspmv_amd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", PKG_DIR, "clean"], stdout=subprocess.DEVNULL)
    done = subprocess.run(["make", "-C", PKG_DIR, "-j8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if done.returncode != 0:
        raise RuntimeError(f"building libspmv_amd.so failed (make exit {done.returncode}); compiler output:\n{done.stdout[-8000:]}")
    for path in (LIB_PATH, LAB_LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(f"{os.path.basename(path)} was not produced by the build")


_lib = None


def lib():
    """The loaded library. Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)")
    L = C.CDLL(LIB_PATH)
    L.get_operator.restype = C.POINTER(SpmvOperator)
    L.get_operator.argtypes = [C.c_char_p]
    L.spmv_amd_version.restype = C.c_char_p
    L.spmv_amd_operator_variant.restype = C.c_char_p
    L.spmv_amd_operator_variant.argtypes = [C.c_char_p]
    L.spmv_amd_operator_select_variant.argtypes = [C.c_char_p, C.c_char_p]
    L.spmv_amd_device_alloc.restype = C.c_void_p
    L.spmv_amd_device_alloc.argtypes = [C.c_size_t]
    L.spmv_amd_device_free.argtypes = [C.c_void_p]
    L.spmv_amd_blas1_axpy.argtypes = [C.c_size_t, C.c_double, C.c_void_p, C.c_void_p]
    L.spmv_amd_blas1_axpby.argtypes = [C.c_size_t, C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    L.spmv_amd_blas1_axpy_dev.argtypes = [C.c_size_t, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
    L.spmv_amd_blas1_update_p_dev.argtypes = [C.c_size_t, C.c_void_p, C.c_double, C.c_void_p]
    L.spmv_amd_blas1_dot.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    L.spmv_amd_cg_fused_step.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.spmv_amd_current_device.argtypes = [C.c_char_p, C.c_int]
    L.spmv_amd_stream_ceiling.restype = C.c_double
    L.spmv_amd_stream_ceiling.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.spmv_amd_stream_ceiling_mix.restype = C.c_double
    L.spmv_amd_stream_ceiling_mix.argtypes = [C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.spmv_amd_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.spmv_amd_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.spmv_amd_device_fill_f64.argtypes = [C.c_void_p, C.c_size_t, C.c_double]
    L.spmv_amd_init_stencil5_synthetic.argtypes = [C.c_char_p, C.c_int]
    L.spmv_amd_download_device_csr.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.spmv_amd_ellpack_run_device_scaled.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
    L.spmv_amd_time_run_device.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.spmv_amd_cg_solve.argtypes = [C.POINTER(SpmvOperator), C.POINTER(MatrixData), C.c_void_p, C.c_void_p, C.POINTER(CGConfig), C.POINTER(CGStats)]
    L.spmv_amd_cg_solve_device.argtypes = L.spmv_amd_cg_solve.argtypes
    L.spmv_amd_cg_solve_mgpu_partitioned.argtypes = [C.POINTER(MatrixData), C.c_void_p, C.c_void_p, C.POINTER(CGConfig), C.POINTER(CGStatsMultiGPU)]
    L.spmv_amd_cg_last_history.argtypes = [C.c_void_p, C.c_int]
    L.spmv_amd_comm_unique_id.argtypes = [C.c_void_p]
    L.spmv_amd_comm_create_rccl.restype = C.c_void_p
    L.spmv_amd_comm_create_rccl.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.spmv_amd_comm_create_staged.restype = C.c_void_p
    L.spmv_amd_comm_create_staged.argtypes = [C.c_int, C.c_int, HALO_FN, ALLREDUCE_FN, GATHER_FN, BARRIER_FN, C.c_void_p]
    L.spmv_amd_comm_destroy.argtypes = [C.c_void_p]
    L.spmv_amd_comm_set_world.argtypes = [C.c_void_p]
    L.spmv_amd_comm_rank.argtypes = [C.c_void_p]
    L.spmv_amd_comm_size.argtypes = [C.c_void_p]
    L.spmv_amd_comm_selftest.argtypes = [C.c_void_p]
    L.spmv_amd_comm_barrier.argtypes = [C.c_void_p]
    L.spmv_amd_comm_mailbox_enable.argtypes = [C.c_void_p]
    L.spmv_amd_comm_mailbox_prepare.argtypes = [C.c_void_p, C.c_void_p]
    L.spmv_amd_comm_mailbox_connect.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.spmv_amd_comm_mailbox_selftest.argtypes = [C.c_void_p, C.c_int]
    L.spmv_amd_comm_mailbox_disable.argtypes = [C.c_void_p]
    L.spmv_amd_comm_mailbox_ready.argtypes = [C.c_void_p]
    L.spmv_amd_comm_transport.restype = C.c_char_p
    L.spmv_amd_comm_transport.argtypes = [C.c_void_p]
    L.spmv_amd_comm_transport_ranks.argtypes = [C.c_void_p]
    if hasattr(L, "spmv_amd_cg_slab_create_stencil5_as"):  # LAB build only
        L.spmv_amd_cg_slab_create_stencil5_as.restype = C.c_void_p
        L.spmv_amd_cg_slab_create_stencil5_as.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.spmv_amd_cg_slab_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
        L.spmv_amd_cg_slab_set_option.restype = C.c_int
    L.spmv_amd_cg_slab_create.restype = C.c_void_p
    L.spmv_amd_cg_slab_create.argtypes = [C.POINTER(MatrixData), C.c_void_p]
    L.spmv_amd_cg_slab_create_stencil5.restype = C.c_void_p
    L.spmv_amd_cg_slab_create_stencil5.argtypes = [C.c_int, C.c_void_p]
    L.spmv_amd_cg_slab_set_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.spmv_amd_cg_slab_solve.argtypes = [C.c_void_p, C.POINTER(CGConfig), C.POINTER(CGStatsMultiGPU)]
    L.spmv_amd_cg_slab_gather.argtypes = [C.c_void_p, C.c_void_p]
    L.spmv_amd_cg_slab_history.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.spmv_amd_cg_slab_spmv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.spmv_amd_cg_slab_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.spmv_amd_cg_slab_time_spmv.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    L.spmv_amd_cg_slab_destroy.argtypes = [C.c_void_p]
    L.spmv_amd_cg_slab_set_timeline.argtypes = [C.c_void_p, C.c_int]
    L.spmv_amd_cg_slab_set_timeline.restype = None
    L.spmv_amd_cg_slab_timeline_names.restype = C.c_char_p
    L.spmv_amd_cg_slab_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.spmv_amd_cg_slab_variant.restype = C.c_char_p
    L.spmv_amd_cg_slab_variant.argtypes = [C.c_void_p]
    L.spmv_amd_cg_slab_loop_shape.restype = C.c_char_p
    L.spmv_amd_cg_slab_loop_shape.argtypes = [C.c_void_p]
    L.load_matrix_market.argtypes = [C.c_char_p, C.POINTER(MatrixData)]
    L.write_matrix_market_stencil5.argtypes = [C.c_int, C.c_char_p]
    L.spmv_amd_write_stencil5_values.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
    L.read_matrix_type.argtypes = [C.c_char_p]
    L.spmv_amd_partition_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.benchmark_with_stats.argtypes = [RUN_TIMED_FN, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(BenchmarkStats)]
    _lib = L
    return L


def is_lab():
    """True when the loaded library is the LAB build (it has the hooks of include/spmv_amd/lab.h)."""
    return hasattr(lib(), "spmv_amd_cg_slab_set_option")


def use_lab():
    """A SECOND instance of this module bound to the LAB build (this one stays on the product library). Both libraries can be
    loaded in one process: each has its own globals (ctypes loads them RTLD_LOCAL)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("spmv_amd_binding_lab", os.path.abspath(__file__))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.LIB_PATH = LAB_LIB_PATH
    return mod


def require_gpu():
    if lib().spmv_amd_device_count() < 1:
        raise RuntimeError("no HIP device visible: libspmv_amd has no CPU path")


def current_device():
    """(device index, PCI bus id) of the calling thread's current HIP device."""
    buf = C.create_string_buffer(64)
    dev = lib().spmv_amd_current_device(buf, 64)
    return dev, buf.value.decode()


def stream_ceiling(rows, warmup=3, reps=20, mix="stencil5"):
    """Per-launch milliseconds and bytes of the stream probe (csrc/stream_ceiling.hip) for a format's byte mix:
    "stencil5" 48:8 B/row, "csr" 72:8 (values, column indices, row pointers, x : y), "ellpack" 68:8, "read-only" 48:0."""
    require_gpu()
    ms = (C.c_float * reps)()
    nbytes = lib().spmv_amd_stream_ceiling_mix({"stencil5": 0, "csr": 1, "ellpack": 2, "read-only": 3}[mix], int(rows), int(warmup), int(reps), ms)
    if nbytes <= 0:
        raise RuntimeError("spmv_amd_stream_ceiling failed")
    return np.array(ms[:], dtype=np.float64), nbytes


def csr_mat():
    return CSRMatrix.in_dll(lib(), "csr_mat")


def ellpack_matrix():
    return ELLPACKMatrix.in_dll(lib(), "ellpack_matrix")


# ---------------------------------------------------------------- host-side helpers
class HostMatrix:
    """A MatrixData whose entries live in a numpy array owned by this object."""

    def __init__(self, entries, rows, cols, grid_size=-1):
        self.entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
        self.c = MatrixData(int(rows), int(cols), len(self.entries), int(grid_size), self.entries.ctypes.data)

    @property
    def ptr(self):
        return C.byref(self.c)


def load_matrix_market(path):
    m = MatrixData()
    rc = lib().load_matrix_market(os.fsencode(path), C.byref(m))
    if rc != 0:
        raise IOError(f"load_matrix_market({path}) -> {rc}")
    e = np.ctypeslib.as_array((C.c_byte * (16 * m.nnz)).from_address(m.entries)).view(ENTRY_DTYPE).copy()
    C.CDLL(None).free(C.c_void_p(m.entries))
    return HostMatrix(e, m.rows, m.cols, m.grid_size)


def read_matrix_symtogen(path):
    """read_matrix_symtogen with every output requested: (rows, cols, stored nnz, expanded nnz, row_ptr, col_idx,
    values, expanded entries). The CSR arrays are the reference reader's (columns in a row in file order)."""
    m = MatrixData()
    rows, cols, nnz, full = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rp, ci, va = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_double)()
    lib().read_matrix_symtogen(C.byref(m), os.fsencode(path), C.byref(rows), C.byref(cols), C.byref(nnz), C.byref(rp), C.byref(ci),
                               C.byref(va), C.byref(full))
    if not m.entries:
        raise IOError(f"read_matrix_symtogen({path}) failed")
    out = (rows.value, cols.value, nnz.value, full.value, np.ctypeslib.as_array(rp, shape=(rows.value + 1,)).copy(),
           np.ctypeslib.as_array(ci, shape=(full.value,)).copy(), np.ctypeslib.as_array(va, shape=(full.value,)).copy(),
           np.ctypeslib.as_array((C.c_byte * (16 * m.nnz)).from_address(m.entries)).view(ENTRY_DTYPE).copy())
    libc = C.CDLL(None)
    for ptr in (rp, ci, va, C.c_void_p(m.entries)):
        libc.free(ptr)
    return out


def host_csr_arrays():
    """Copies of the process-wide csr_mat built by build_csr_struct."""
    m = csr_mat()
    rp = np.ctypeslib.as_array(m.row_ptr, shape=(m.nb_rows + 1,)).copy()
    ci = np.ctypeslib.as_array(m.col_indices, shape=(m.nb_nonzeros,)).copy()
    va = np.ctypeslib.as_array(m.values, shape=(m.nb_nonzeros,)).copy()
    return rp, ci, va


def host_ell_arrays():
    m = ellpack_matrix()
    n = m.nb_rows * m.ell_width
    return m.ell_width, np.ctypeslib.as_array(m.indices, shape=(n,)).copy(), np.ctypeslib.as_array(m.values, shape=(n,)).copy()


def partition_rows(n, world, rank):
    off, nl = C.c_int(), C.c_int()
    lib().spmv_amd_partition_rows(n, world, rank, C.byref(off), C.byref(nl))
    return off.value, nl.value


# ---------------------------------------------------------------- device-side helpers
class DeviceVector:
    def __init__(self, count, fill=None):
        self.count = int(count)
        self.ptr = lib().spmv_amd_device_alloc(self.count * 8)
        if fill is not None:
            lib().spmv_amd_device_fill_f64(self.ptr, self.count, float(fill))

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        v = cls(len(a))
        lib().spmv_amd_copy_to_device(v.ptr, a.ctypes.data, a.nbytes)
        return v

    def to_host(self):
        out = np.empty(self.count, dtype=np.float64)
        lib().spmv_amd_device_synchronize()
        lib().spmv_amd_copy_to_host(out.ctypes.data, self.ptr, out.nbytes)
        return out

    def free(self):
        if self.ptr:
            lib().spmv_amd_device_free(self.ptr)
            self.ptr = None


class Operator:
    """get_operator(name) with numpy-friendly wrappers around the vtable."""

    def __init__(self, name):
        self.name = name
        p = lib().get_operator(name.encode())
        if not p:
            raise KeyError(name)
        self.op = p
        self.rows = self.cols = 0

    @property
    def canonical_name(self):
        return self.op.contents.name.decode()

    def init(self, host_matrix):
        require_gpu()
        rc = self.op.contents.init(host_matrix.ptr)
        self.rows, self.cols = host_matrix.c.rows, host_matrix.c.cols
        return rc

    def init_synthetic(self, n):
        require_gpu()
        rc = lib().spmv_amd_init_stencil5_synthetic(self.name.encode(), int(n))
        self.rows = self.cols = n * n
        return rc

    def run_timed(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert len(x) == self.cols
        y = np.zeros(self.rows, dtype=np.float64)
        ms = C.c_double()
        rc = self.op.contents.run_timed(x.ctypes.data, y.ctypes.data, C.byref(ms))
        if rc != 0:
            raise RuntimeError(f"run_timed -> {rc}")
        return y, ms.value

    def run_device(self, d_x, d_y):
        return self.op.contents.run_device(d_x.ptr, d_y.ptr)

    def time_device(self, d_x, d_y, reps):
        """Kernel-only ms of `reps` run_device launches; d_x / d_y None = the operator's own staging vectors (x = 1), the
        ones run_timed's kernel works on."""
        ms = (C.c_float * reps)()
        rc = lib().spmv_amd_time_run_device(self.name.encode(), None if d_x is None else d_x.ptr, None if d_y is None else d_y.ptr, reps, ms)
        if rc != 0:
            raise RuntimeError(f"time_run_device -> {rc}")
        return np.array(ms[:], dtype=np.float64)

    def placement(self):
        """(candidates timed for the operator's y vector at init, kernel time on the first / on the one kept)"""
        cand, gain = C.c_int(), C.c_double()
        lib().spmv_amd_operator_placement.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
        ok = lib().spmv_amd_operator_placement(self.name.encode(), C.byref(cand), C.byref(gain))
        return (cand.value, gain.value) if ok else None

    def variant(self):
        return lib().spmv_amd_operator_variant(self.name.encode()).decode()

    def select_variant(self, variant):
        return lib().spmv_amd_operator_select_variant(self.name.encode(), None if variant is None else variant.encode())

    def download_csr(self, rows, nnz):
        rp = np.empty(rows + 1, dtype=np.int32)
        ci = np.empty(nnz, dtype=np.int32)
        va = np.empty(nnz, dtype=np.float64)
        rc = lib().spmv_amd_download_device_csr(self.name.encode(), rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
        assert rc == 0
        return rp, ci, va

    def free(self):
        self.op.contents.free()


def cg_solve(op, host_matrix, b, x0, max_iters=1000, tol=1e-6, device=True, verbose=0, timers=0):
    """cg_solve_device (device=True) or cg_solve through the C-ABI; returns x, history, stats."""
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    cfg = CGConfig(max_iters, tol, verbose, timers)
    st = CGStats()
    fn = lib().spmv_amd_cg_solve_device if device else lib().spmv_amd_cg_solve
    rc = fn(op.op, host_matrix.ptr, b.ctypes.data, x.ctypes.data, C.byref(cfg), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"cg_solve -> {rc}")
    hist = np.zeros(max_iters + 1, dtype=np.float64)
    count = lib().spmv_amd_cg_last_history(hist.ctypes.data, len(hist))
    return x, hist[:count].copy(), st


class Comm:
    """Owner of an SpmvAmdComm handle (and of the ctypes callbacks of a staged one)."""

    def __init__(self, handle, keep=()):
        self.handle = handle
        self._keep = keep

    @classmethod
    def rccl(cls, rank, world, id_bytes):
        """RCCL communicator, or None if RCCL could not create it."""
        buf = C.create_string_buffer(bytes(id_bytes), COMM_ID_BYTES)
        h = lib().spmv_amd_comm_create_rccl(rank, world, buf)
        return cls(h) if h else None

    @classmethod
    def staged_over_torch(cls, rank, world, dist):
        """Staged communicator whose host exchange is torch.distributed (CPU tensors, e.g. gloo):
        the transport of the multi-rank tests on a 1-GPU box, and bench.py's way out if RCCL
        cannot be initialised."""
        import torch

        def as_np(ptr, count):
            return np.ctypeslib.as_array(ptr, shape=(count,))

        def halo_cb(user, sp, sn, rp_, rn_, count):
            reqs, prev, nxt = [], None, None
            if rank > 0:
                prev = torch.zeros(count, dtype=torch.float64)
                reqs += [dist.isend(torch.from_numpy(as_np(sp, count).copy()), rank - 1), dist.irecv(prev, rank - 1)]
            if rank < world - 1:
                nxt = torch.zeros(count, dtype=torch.float64)
                reqs += [dist.isend(torch.from_numpy(as_np(sn, count).copy()), rank + 1), dist.irecv(nxt, rank + 1)]
            for r in reqs:
                r.wait()
            if prev is not None:
                as_np(rp_, count)[:] = prev.numpy()
            if nxt is not None:
                as_np(rn_, count)[:] = nxt.numpy()
            return 0

        def allreduce_cb(user, buf, count):
            a = as_np(buf, count)
            t = torch.from_numpy(a.copy())
            dist.all_reduce(t)
            a[:] = t.numpy()
            return 0

        def gather_cb(user, send, n_send, recv, counts, displs):
            mine = torch.from_numpy(as_np(send, n_send).copy())
            if rank == 0:
                out = as_np(recv, sum(counts[r] for r in range(world)))
                out[displs[0]:displs[0] + counts[0]] = mine.numpy()
                for r in range(1, world):
                    t = torch.zeros(counts[r], dtype=torch.float64)
                    dist.recv(t, r)
                    out[displs[r]:displs[r] + counts[r]] = t.numpy()
            else:
                dist.send(mine, 0)
            return 0

        def barrier_cb(user):
            dist.barrier()
            return 0

        return cls.staged(rank, world, halo_cb, allreduce_cb, gather_cb, barrier_cb)

    def selftest(self):
        return lib().spmv_amd_comm_selftest(self.handle)

    def barrier(self):
        return lib().spmv_amd_comm_barrier(self.handle)

    def mailbox_enable(self):
        """Collective: peer-mailbox all-reduce over hipIpc-mapped device memory; True when every rank's passed its self-test."""
        return lib().spmv_amd_comm_mailbox_enable(self.handle) == 1

    def mailbox_prepare(self):
        """Step 1 of the manual set-up: allocate this rank's mailbox, return its 64-byte hipIpc handle (None on failure)."""
        buf = C.create_string_buffer(64)
        return buf.raw if lib().spmv_amd_comm_mailbox_prepare(self.handle, buf) == 0 else None

    def mailbox_connect(self, handles):
        """Step 2: map every rank's mailbox (handles: list of 64-byte blobs in rank order); True on success."""
        blob = b"".join(handles)
        return lib().spmv_amd_comm_mailbox_connect(self.handle, C.create_string_buffer(blob, len(blob)), len(handles)) == 0

    def mailbox_selftest(self, rounds=8):
        """Collective: `rounds` back-to-back mailbox all-reduces with known sums; 0 = all correct."""
        return lib().spmv_amd_comm_mailbox_selftest(self.handle, int(rounds))

    def mailbox_ready(self):
        return lib().spmv_amd_comm_mailbox_ready(self.handle) == 1

    def mailbox_disable(self):
        lib().spmv_amd_comm_mailbox_disable(self.handle)

    def transport(self):
        return lib().spmv_amd_comm_transport(self.handle).decode()

    def transport_ranks(self):
        """Ranks the device transport itself reports (ncclCommCount); 0 for staged / self."""
        return lib().spmv_amd_comm_transport_ranks(self.handle)

    @classmethod
    def staged(cls, rank, world, halo, allreduce, gather=None, barrier=None):
        cbs = (HALO_FN(halo), ALLREDUCE_FN(allreduce), GATHER_FN(gather) if gather else GATHER_FN(), BARRIER_FN(barrier) if barrier else BARRIER_FN())
        h = lib().spmv_amd_comm_create_staged(rank, world, cbs[0], cbs[1], cbs[2], cbs[3], None)
        if not h:
            raise RuntimeError("spmv_amd_comm_create_staged failed")
        return cls(h, cbs)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        lib().spmv_amd_comm_unique_id(buf)
        return buf.raw

    def destroy(self):
        if self.handle:
            lib().spmv_amd_comm_destroy(self.handle)
            self.handle = None


class CgSlab:
    """Resident multi-GPU CG state (spmv_amd_cg_slab_*)."""

    def __init__(self, handle, n_rows):
        if not handle:
            raise RuntimeError("spmv_amd_cg_slab_create failed")
        self.h = handle
        self.n_rows = n_rows
        off, nl, nz = C.c_int(), C.c_int(), C.c_int()
        lib().spmv_amd_cg_slab_info(self.h, C.byref(off), C.byref(nl), C.byref(nz))
        self.row_offset, self.n_local, self.local_nnz = off.value, nl.value, nz.value

    @classmethod
    def from_matrix(cls, host_matrix, comm=None):
        require_gpu()
        return cls(lib().spmv_amd_cg_slab_create(host_matrix.ptr, comm.handle if comm else None), host_matrix.c.rows)

    @classmethod
    def stencil5(cls, n, comm=None):
        require_gpu()
        return cls(lib().spmv_amd_cg_slab_create_stencil5(int(n), comm.handle if comm else None), n * n)

    @classmethod
    def stencil5_as(cls, n, as_rank, as_world, comm):
        """The slab rank `as_rank` of an `as_world`-GPU run would own, on a single self-neighbour rank (LAB build only)."""
        require_gpu()
        if not is_lab():
            raise RuntimeError("stand-in slabs exist in the LAB build only (binding.use_lab(), lib/libspmv_amd_lab.so)")
        return cls(lib().spmv_amd_cg_slab_create_stencil5_as(int(n), int(as_rank), int(as_world), comm.handle), n * n)

    def set_vectors(self, b=None, x0=None):
        b = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
        x0 = None if x0 is None else np.ascontiguousarray(x0, dtype=np.float64)
        lib().spmv_amd_cg_slab_set_vectors(self.h, None if b is None else b.ctypes.data, None if x0 is None else x0.ctypes.data)

    def solve(self, max_iters=1000, tol=1e-6, verbose=0, timers=0):
        cfg = CGConfig(max_iters, tol, verbose, timers)
        st = CGStatsMultiGPU()
        rc = lib().spmv_amd_cg_slab_solve(self.h, C.byref(cfg), C.byref(st))
        if rc != 0:
            raise RuntimeError(f"cg_slab_solve -> {rc}")
        return st

    def history(self, cap=1001):
        h = np.zeros(cap, dtype=np.float64)
        count = lib().spmv_amd_cg_slab_history(self.h, h.ctypes.data, cap)
        return h[: min(count, cap)].copy()

    def gather(self):
        x = np.zeros(self.n_rows, dtype=np.float64)
        lib().spmv_amd_cg_slab_gather(self.h, x.ctypes.data)
        return x

    def spmv(self, x_full):
        x_full = np.ascontiguousarray(x_full, dtype=np.float64)
        y = np.zeros(self.n_local, dtype=np.float64)
        lib().spmv_amd_cg_slab_spmv(self.h, x_full.ctypes.data, y.ctypes.data)
        return y

    def variant(self):
        return lib().spmv_amd_cg_slab_variant(self.h).decode()

    def loop_shape(self):
        """"single rank" | "pipeline ..." | "plain: <who decided>" (include/spmv_amd/api.h)."""
        return lib().spmv_amd_cg_slab_loop_shape(self.h).decode()

    def timeline_solve(self, max_iters=1000, tol=1e-6):
        """One solve with stage-boundary events (no host syncs); returns (stats, {name: value})."""
        lib().spmv_amd_cg_slab_set_timeline(self.h, 1)
        try:
            st = self.solve(max_iters=max_iters, tol=tol)
        finally:
            lib().spmv_amd_cg_slab_set_timeline(self.h, 0)
        names = lib().spmv_amd_cg_slab_timeline_names().decode().split(",")
        v = np.zeros(len(names), dtype=np.float64)
        count = lib().spmv_amd_cg_slab_timeline(self.h, v.ctypes.data, len(v))
        return st, ({k: float(x) for k, x in zip(names, v)} if count == len(names) else {})

    def placement(self):
        """What the placement of the coefficient stream at creation did, or None if it did not run."""
        v = (C.c_double * 4)()
        lib().spmv_amd_cg_slab_placement.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        if lib().spmv_amd_cg_slab_placement(self.h, v, 4) != 4:
            return None
        return {"kind": "coefficient candidates", "candidates": int(v[1]), "spmv_ms_before": float(v[2]), "spmv_ms_kept": float(v[3])}

    def setup_ms(self):
        """Wall ms of creation's set-up phases (outside every timed region)."""
        v = (C.c_double * 5)()
        lib().spmv_amd_cg_slab_setup_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib().spmv_amd_cg_slab_setup_ms(self.h, v, 5)
        return dict(zip(("matrix_to_hbm", "streams_and_vectors", "verify_and_plans", "coefficient_placement", "tile_runs"), (float(x) for x in v)))

    def tile_runs(self):
        """Row-lds tiles per XCD and run as tuned at creation, or None if the trial did not run."""
        v = (C.c_double * 4)()
        lib().spmv_amd_cg_slab_tile_runs.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        if lib().spmv_amd_cg_slab_tile_runs(self.h, v, 4) != 4:
            return None
        return {"rule": int(v[0]), "kept": int(v[1]), "spmv_ms_rule": float(v[2]), "spmv_ms_kept": float(v[3])}

    def spmv_launch_ms(self):
        """The timed in-loop SpMV launches of the last solve, one by one (ms)."""
        out = (C.c_float * 1024)()
        lib().spmv_amd_cg_slab_spmv_launch_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        count = lib().spmv_amd_cg_slab_spmv_launch_ms(self.h, out, 1024)
        return np.array(out[:min(count, 1024)], dtype=np.float64)

    def set_option(self, name, value):
        """Option of this slab (LAB build only; include/spmv_amd/lab.h lists them): A/B runs on the same allocations."""
        if not is_lab():
            raise RuntimeError("slab options exist in the LAB build only (binding.use_lab(), lib/libspmv_amd_lab.so)")
        if lib().spmv_amd_cg_slab_set_option(self.h, name.encode(), int(value)) != 0:
            raise ValueError(f"unknown slab option {name!r}")

    def time_spmv(self, reps):
        ms = (C.c_float * reps)()
        lib().spmv_amd_cg_slab_time_spmv(self.h, reps, ms)
        return np.array(ms[:], dtype=np.float64)

    def destroy(self):
        if self.h:
            lib().spmv_amd_cg_slab_destroy(self.h)
            self.h = None
