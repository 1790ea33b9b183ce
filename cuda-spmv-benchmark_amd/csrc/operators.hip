// operators.hip -- the SpmvOperator tables behind get_operator().
//
//   "stencil5-csr"     <- reference SPMV_STENCIL5_CSR, src/spmv/spmv_stencil_csr_direct.cu:194-307
//   "cusparse-csr"     <- reference SPMV_CSR, src/spmv/spmv_cusparse_csr.cu:182-327 (the cuSPARSE
//                         call is replaced by this build's own CSR kernels; the name is kept so that
//                         calculate_spmv_metrics and the harness scripts keep working)
//   "ellpack",
//   "stencil5-ellpack" <- reference include/spmv_ellpack.h, include/spmv_stencil.h:25-42 (headers only)
//
// Lifecycle, ownership and error behaviour follow the reference: init builds the host CSR
// through build_csr_struct (kept in csr_mat for the other operators), uploads it, and allocates
// the x/y staging vectors; run_timed copies x in, times the kernel alone with events, copies y
// out; run_device enqueues on the default stream and returns; free drops device memory only.
// One live instance per operator per process (file-static state, as upstream).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "device_runtime.hpp"
#include "stencil_geometry.hpp"

using namespace spmv_amd;

namespace {

constexpr hipStream_t kDefaultStream = nullptr;  // the reference launches on the default stream

struct CsrBackedOperator {
    const char* tag;
    LaunchShape shape;        // the launch switches (kernels.hpp), read when a variant is picked
    Stencil5Plan plan;        // stencil5-csr: the launch plan of the whole matrix, made once per init / variant change
    DeviceCsr A;
    double* dX = nullptr;
    double* dY = nullptr;
    int rows = 0, cols = 0;
    bool ready = false;
    Stencil5Variant stencil_variant = Stencil5Variant::Auto;
    CsrVariant csr_variant = CsrVariant::Auto;
    const char* variant_name = "uninitialised";

    int tuned_run = 0;         // stencil5-csr: row-lds tiles per XCD and run kept by the set-up trial (0 = not tried yet, -1 = does not apply)
    double tuning[4] = {0, 0, 0, 0};  // {rule, kept, ms with the rule, ms kept}
    int y_candidates = 1;      // output placement: candidates timed for dY, and what the choice was worth (device_runtime.hpp)
    double y_gain = 1.0;
    void alloc_vectors() {
        dX = device_alloc<double>((size_t)cols);
        dY = nullptr;  // placed by place_output() once the kernel that writes it is known
    }
    // dY = the fastest of a few allocations for THIS operator's kernel reading dX (ones) and writing the candidate: the vector
    // run_timed's kernel writes (reference harness: main.cu:158-187 times run_timed). Vectors under 16 M rows are taken as they come.
    template <class Launch>
    void place_output(Launch&& launch) {
        device_release(dY);
        launch_fill(dX, (size_t)cols, 1.0, kDefaultStream);
        EventTimer t;
        dY = device_alloc_best_of<double>((size_t)rows, (size_t)16 << 20, [&](double* y) {
            float ms[3];
            launch(dX, y);
            for (float& m : ms) {
                t.begin(kDefaultStream);
                launch(dX, y);
                t.end(kDefaultStream);
                m = t.elapsed_ms();
            }
            std::sort(ms, ms + 3);
            return (double)ms[1];
        }, &y_candidates, &y_gain);
        HIP_CHECK(hipStreamSynchronize(kDefaultStream));
    }
    int init_from_host(MatrixData* mat) {
        if (build_csr_struct(mat) != EXIT_SUCCESS) return EXIT_FAILURE;
        drop();
        rows = csr_mat.nb_rows;
        cols = csr_mat.nb_cols;
        A.upload_slab(csr_mat, 0, rows, mat->grid_size);
        A.view.halo_after = cols - rows;  // x is readable on [0, cols)
        alloc_vectors();
        ready = true;
        return 0;
    }
    int init_synthetic(int n) {
        drop();
        if (n < 1 || (long long)n * n > 0x7fffffffLL || 5LL * n * n - 4LL * n > 0x7fffffffLL) {
            fprintf(stderr, "[%s] grid %d does not fit 32-bit CSR indices\n", tag, n);
            return EXIT_FAILURE;
        }
        rows = cols = n * n;
        A.generate_stencil5(n, 0, rows, 5.0, -1.0, kDefaultStream);
        // dimensions for calculate_spmv_metrics; the host arrays do not exist on this path
        spmv_amd_reset_host_matrices();
        csr_mat.nb_rows = rows;
        csr_mat.nb_cols = cols;
        csr_mat.nb_nonzeros = (int)A.view.nnz_local;
        alloc_vectors();
        ready = true;
        return 0;
    }
    void drop() {
        A.release();
        device_release(dX);
        device_release(dY);
        ready = false;
        tuned_run = 0;
        variant_name = "uninitialised";
    }
};

CsrBackedOperator g_stencil{"stencil5-csr"};
CsrBackedOperator g_csr{"cusparse-csr"};

// ---- stencil5-csr ----------------------------------------------------------------

void stencil_pick_variant() {
    g_stencil.shape = current_launch_shape();
    g_stencil.plan = plan_stencil5(g_stencil.A.view, 0, g_stencil.rows, g_stencil.stencil_variant, g_stencil.shape);
    g_stencil.variant_name = g_stencil.plan.name;
    if (g_stencil.dY == nullptr) {  // once per init: the class a good output vector lies in does not depend on the variant
        g_stencil.place_output([](const double* x, double* y) {
            (void)launch_stencil5_spmv(g_stencil.A.view, g_stencil.plan, x, y, 1.0, nullptr, nullptr, false, kDefaultStream);
        });
        g_stencil.tuned_run = 0;
    }
    // row-lds tiles per XCD and run: the rule's neighbours timed once per init on the operator's own vectors (device_runtime.hpp)
    if (g_stencil.stencil_variant == Stencil5Variant::Auto || g_stencil.stencil_variant == Stencil5Variant::RowLds) {
        if (g_stencil.tuned_run == 0 && g_stencil.shape.knobs.rowlds_group == 0) {
            const int run = tune_rowlds_xcd_run(g_stencil.A.view, g_stencil.shape, g_stencil.dX, g_stencil.dY, nullptr, kDefaultStream, g_stencil.tuning);
            g_stencil.tuned_run = run > 0 ? run : -1;
        }
        if (g_stencil.tuned_run > 0 && g_stencil.shape.knobs.rowlds_group == 0) {
            g_stencil.shape.knobs.rowlds_group = g_stencil.tuned_run;
            g_stencil.plan = plan_stencil5(g_stencil.A.view, 0, g_stencil.rows, g_stencil.stencil_variant, g_stencil.shape);
        }
    }
}

int stencil_init(MatrixData* mat) {
    printf("[stencil5-csr] Initializing (computed offsets, row-lds / row-direct kernels on gfx950)\n");
    if (g_stencil.init_from_host(mat) != 0) return EXIT_FAILURE;
    g_stencil.A.verify_stencil(kDefaultStream);
    stencil_pick_variant();
    printf("[stencil5-csr] %d rows, %d nnz, grid %dx%d, variant %s\n", csr_mat.nb_rows,
           csr_mat.nb_nonzeros, mat->grid_size, mat->grid_size, g_stencil.variant_name);
    if (g_stencil.tuned_run > 0)
        printf("[stencil5-csr] tiles per XCD and run: %d kept (rule %d: %.4f ms, kept %.4f ms)\n", g_stencil.tuned_run, (int)g_stencil.tuning[0],
               g_stencil.tuning[2], g_stencil.tuning[3]);
    return 0;
}

int stencil_run_device(const double* d_x, double* d_y) {
    if (!g_stencil.ready) {
        fprintf(stderr, "[stencil5-csr] run before init\n");
        return EXIT_FAILURE;
    }
    (void)launch_stencil5_spmv(g_stencil.A.view, g_stencil.plan, d_x, d_y, /*alpha=*/1.0, nullptr, nullptr,
                               /*reverse=*/false, kDefaultStream);
    return 0;
}

template <int (*RunDevice)(const double*, double*), CsrBackedOperator* Op>
int run_timed_generic(const double* x, double* y, double* kernel_time_ms) {
    if (!Op->ready) {
        fprintf(stderr, "[%s] run before init\n", Op->tag);
        return EXIT_FAILURE;
    }
    upload(Op->dX, x, (size_t)Op->cols);
    EventTimer t;
    t.begin(kDefaultStream);
    RunDevice(Op->dX, Op->dY);
    t.end(kDefaultStream);
    *kernel_time_ms = (double)t.elapsed_ms();
    HIP_CHECK(hipGetLastError());
    download(y, Op->dY, (size_t)Op->rows);
    return 0;
}

void stencil_free() {
    printf("[stencil5-csr] Cleaning up\n");
    release_cg_workspace();  // the single-GPU CG solver's vectors live as long as an operator does
    g_stencil.drop();
}

// The launch cg_solve_device uses when the operator it is handed is this one (device_runtime.hpp, FusedSpmv).
int stencil_fused_launch(const double* d_x, double* d_y, double* d_partials, const int* d_skip, bool reverse,
                         const ResidualOut* init, hipStream_t stream) {
    return launch_stencil5_spmv(g_stencil.A.view, g_stencil.plan, d_x, d_y, /*alpha=*/1.0, d_partials, d_skip, reverse, stream, init);
}

// ---- cusparse-csr ----------------------------------------------------------------

const char* csr_variant_name(CsrVariant v, const SlabCsr& m) {
    if (v == CsrVariant::Auto) v = csr_auto_variant(m);
    switch (v) {
        case CsrVariant::Stream: return "csr/stream";
        case CsrVariant::Adaptive: return "csr/adaptive";
        case CsrVariant::RowScalar: return "csr/row-scalar";
        default: return "csr/wavefront";
    }
}

void csr_place_output();

int csr_init(MatrixData* mat) {
    if (g_csr.init_from_host(mat) != 0) return EXIT_FAILURE;
    g_csr.variant_name = csr_variant_name(g_csr.csr_variant, g_csr.A.view);
    csr_place_output();
    printf("[cusparse-csr] %d rows, %d nnz, variant %s\n", csr_mat.nb_rows, csr_mat.nb_nonzeros,
           g_csr.variant_name);
    return EXIT_SUCCESS;
}

void csr_place_output() {
    if (g_csr.dY == nullptr)
        g_csr.place_output([](const double* x, double* y) {
            launch_csr_spmv(g_csr.A.view, x, y, 1.0, g_csr.csr_variant, kDefaultStream);
        });
}

int csr_run_device(const double* d_x, double* d_y) {
    if (!g_csr.ready) {
        fprintf(stderr, "[cusparse-csr] run before init\n");
        return EXIT_FAILURE;
    }
    launch_csr_spmv(g_csr.A.view, d_x, d_y, /*alpha=*/1.0, g_csr.csr_variant, kDefaultStream);
    return EXIT_SUCCESS;
}

// cg_solve_device's fused launches for the operators other than stencil5-csr: the SpMV kernel also writes the partial sums of
// x . (A x) (one per workgroup), so the loop needs no 16 B/row dot pass. No skip flag (the loop never enqueues a SpMV past
// convergence), no sweep direction, no fused initial residual.
int csr_fused_launch(const double* d_x, double* d_y, double* d_partials, const int*, bool, const ResidualOut* init, hipStream_t stream) {
    if (init != nullptr) return -1;
    launch_csr_spmv(g_csr.A.view, d_x, d_y, /*alpha=*/1.0, g_csr.csr_variant, stream, d_partials);
    return csr_fused_dot_partials(g_csr.A.view, g_csr.csr_variant);
}

void csr_free() {
    printf("[CSR] Cleaning up\n");
    release_cg_workspace();
    g_csr.drop();
}

// ---- ellpack / stencil5-ellpack -----------------------------------------------------

struct EllOperator {
    const char* tag;
    bool stencil_fast_path;
    // slot-major planes, both carved out of ONE allocation (val first): the kernel reads val[k][r] and idx[k][r] in lock step
    // (DeviceCsr::allocate has the reason)
    char* planes_block = nullptr;
    int* idx = nullptr;
    double* val = nullptr;
    void alloc_planes(size_t slots) {
        const size_t v_bytes = (slots * sizeof(double) + 4095) / 4096 * 4096;
        planes_block = device_alloc<char>(v_bytes + slots * sizeof(int));
        val = reinterpret_cast<double*>(planes_block);
        idx = reinterpret_cast<int*>(planes_block + v_bytes);
    }
    double* dX = nullptr;
    double* dY = nullptr;
    int rows = 0, cols = 0, width = 0, grid_size = -1;
    bool verified = false;
    bool ready = false;
    const char* variant_name = "uninitialised";
    int y_candidates = 1;  // output placement (device_runtime.hpp)
    double y_gain = 1.0;
    void drop() {
        device_release(planes_block);
        idx = nullptr;
        val = nullptr;
        device_release(dX);
        device_release(dY);
        ready = false;
        variant_name = "uninitialised";
    }
    void pick() {
        variant_name = (stencil_fast_path && verified && grid_size >= 3) ? "ell/stencil5-direct"
                                                                        : "ell/slot-major";
    }
};

EllOperator g_ell{"ellpack", false};
EllOperator g_ell_stencil{"stencil5-ellpack", true};

__global__ void csr_to_ell_slotmajor_kernel(SlabCsr m, int width, int* __restrict__ idx,
                                            double* __restrict__ val) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m.n_local) return;
    const int lo = m.row_ptr[r], len = m.row_ptr[r + 1] - lo;
    for (int k = 0; k < width; ++k) {
        const bool live = k < len;
        idx[(long long)k * m.n_local + r] = live ? m.col_idx[lo + k] : -1;
        val[(long long)k * m.n_local + r] = live ? m.values[lo + k] : 0.0;
    }
}

int ell_run(EllOperator& op, const double* d_x, double* d_y, double alpha, double beta);

// the vector the operator's kernel writes: the fastest of a few allocations (device_runtime.hpp, output placement)
void ell_place_output(EllOperator& op) {
    launch_fill(op.dX, (size_t)op.cols, 1.0, kDefaultStream);
    EventTimer t;
    op.dY = device_alloc_best_of<double>((size_t)op.rows, (size_t)16 << 20, [&](double* y) {
        float ms[3];
        ell_run(op, op.dX, y, 1.0, 0.0);
        for (float& m : ms) {
            t.begin(kDefaultStream);
            ell_run(op, op.dX, y, 1.0, 0.0);
            t.end(kDefaultStream);
            m = t.elapsed_ms();
        }
        std::sort(ms, ms + 3);
        return (double)ms[1];
    }, &op.y_candidates, &op.y_gain);
    HIP_CHECK(hipStreamSynchronize(kDefaultStream));
}

int ell_init_common(EllOperator& op, MatrixData* mat) {
    if (ensure_ellpack_structure_built(mat) != EXIT_SUCCESS) return EXIT_FAILURE;
    op.drop();
    const ELLPACKMatrix& E = ellpack_matrix;
    op.rows = E.nb_rows, op.cols = E.nb_cols, op.width = E.ell_width, op.grid_size = mat->grid_size;
    const size_t slots = (size_t)op.rows * op.width;
    int* idx_rm = device_alloc<int>(slots);
    double* val_rm = device_alloc<double>(slots);
    upload(idx_rm, E.indices, slots);
    upload(val_rm, E.values, slots);
    op.alloc_planes(slots);
    launch_ell_transpose(op.rows, op.width, idx_rm, val_rm, op.idx, op.val, kDefaultStream);
    HIP_CHECK(hipStreamSynchronize(kDefaultStream));
    device_release(idx_rm);
    device_release(val_rm);
    op.verified = false;
    if (op.stencil_fast_path && csr_mat.row_ptr != nullptr) {
        DeviceCsr probe;  // structure check on the CSR the ELL was built from
        probe.upload_slab(csr_mat, 0, op.rows, mat->grid_size);
        probe.verify_stencil(kDefaultStream);
        op.verified = probe.view.verified_stencil;
        probe.release();
    }
    op.dX = device_alloc<double>((size_t)op.cols);
    op.ready = true;
    op.pick();
    ell_place_output(op);
    printf("[%s] %d rows, width %d, variant %s\n", op.tag, op.rows, op.width, op.variant_name);
    return 0;
}

int ell_init_synthetic(EllOperator& op, int n) {
    op.drop();
    DeviceCsr src;
    if ((long long)n * n > 0x7fffffffLL || 5LL * n * n > 0x7fffffffLL) return EXIT_FAILURE;
    src.generate_stencil5(n, 0, n * n, 5.0, -1.0, kDefaultStream);
    op.rows = op.cols = n * n;
    op.width = n >= 3 ? 5 : (n == 2 ? 3 : 1);
    op.grid_size = n;
    const size_t slots = (size_t)op.rows * op.width;
    op.alloc_planes(slots);
    hipLaunchKernelGGL(csr_to_ell_slotmajor_kernel, dim3((op.rows + 255) / 256), dim3(256), 0,
                       kDefaultStream, src.view, op.width, op.idx, op.val);
    HIP_CHECK(hipStreamSynchronize(kDefaultStream));
    src.release();
    op.verified = true;
    spmv_amd_reset_host_matrices();
    csr_mat.nb_rows = csr_mat.nb_cols = op.rows;
    csr_mat.nb_nonzeros = (int)(5LL * n * n - 4LL * n);
    op.dX = device_alloc<double>((size_t)op.cols);
    op.ready = true;
    op.pick();
    ell_place_output(op);
    return 0;
}

int ell_run(EllOperator& op, const double* d_x, double* d_y, double alpha = 1.0, double beta = 0.0);
int ell_run(EllOperator& op, const double* d_x, double* d_y, double alpha, double beta) {
    if (!op.ready) {
        fprintf(stderr, "[%s] run before init\n", op.tag);
        return EXIT_FAILURE;
    }
    if (op.stencil_fast_path && op.verified)
        launch_ell_stencil5_spmv(op.rows, op.width, op.grid_size, op.idx, op.val, d_x, d_y, alpha, beta, kDefaultStream);
    else
        launch_ell_spmv(op.rows, op.width, op.idx, op.val, d_x, d_y, alpha, beta, kDefaultStream, op.verified ? op.grid_size : 0);
    return 0;
}

int ell_run_timed(EllOperator& op, const double* x, double* y, double* kernel_time_ms) {
    if (!op.ready) return EXIT_FAILURE;
    upload(op.dX, x, (size_t)op.cols);
    EventTimer t;
    t.begin(kDefaultStream);
    ell_run(op, op.dX, op.dY);
    t.end(kDefaultStream);
    *kernel_time_ms = (double)t.elapsed_ms();
    HIP_CHECK(hipGetLastError());
    download(y, op.dY, (size_t)op.rows);
    return 0;
}

template <EllOperator* Op>
int ell_fused_launch(const double* d_x, double* d_y, double* d_partials, const int*, bool, const ResidualOut* init, hipStream_t stream) {
    if (init != nullptr || !Op->ready) return -1;
    if (Op->stencil_fast_path && Op->verified)
        launch_ell_stencil5_spmv(Op->rows, Op->width, Op->grid_size, Op->idx, Op->val, d_x, d_y, 1.0, 0.0, stream, d_partials);
    else
        launch_ell_spmv(Op->rows, Op->width, Op->idx, Op->val, d_x, d_y, 1.0, 0.0, stream, Op->verified ? Op->grid_size : 0, d_partials);
    return ell_fused_dot_partials(Op->rows);
}

int ellg_init(MatrixData* m) { return ell_init_common(g_ell, m); }
int ellg_run_timed(const double* x, double* y, double* ms) { return ell_run_timed(g_ell, x, y, ms); }
int ellg_run_device(const double* x, double* y) { return ell_run(g_ell, x, y); }
void ellg_free() {
    release_cg_workspace();
    g_ell.drop();
}
int ells_init(MatrixData* m) { return ell_init_common(g_ell_stencil, m); }
int ells_run_timed(const double* x, double* y, double* ms) { return ell_run_timed(g_ell_stencil, x, y, ms); }
int ells_run_device(const double* x, double* y) { return ell_run(g_ell_stencil, x, y); }
void ells_free() {
    release_cg_workspace();
    g_ell_stencil.drop();
}

// ---- name table ------------------------------------------------------------------

enum class Which { None, Stencil, Csr, Ell, EllStencil };

Which which_operator(const char* mode) {
    if (!mode) return Which::None;
    if (!strcmp(mode, "stencil5-csr") || !strcmp(mode, "stencil5")) return Which::Stencil;
    if (!strcmp(mode, "cusparse-csr") || !strcmp(mode, "csr")) return Which::Csr;
    if (!strcmp(mode, "ellpack")) return Which::Ell;
    if (!strcmp(mode, "stencil5-ellpack")) return Which::EllStencil;
    return Which::None;
}

}  // namespace

SpmvOperator SPMV_STENCIL5_CSR = {"stencil5-csr", stencil_init,
                                  run_timed_generic<stencil_run_device, &g_stencil>,
                                  stencil_run_device, stencil_free};
SpmvOperator SPMV_CSR = {"cusparse-csr", csr_init, run_timed_generic<csr_run_device, &g_csr>,
                         csr_run_device, csr_free};
SpmvOperator SPMV_ELLPACK = {"ellpack", ellg_init, ellg_run_timed, ellg_run_device, ellg_free};
SpmvOperator SPMV_STENCIL5_ELLPACK = {"stencil5-ellpack", ells_init, ells_run_timed,
                                      ells_run_device, ells_free};
// Upstream declares this table under __has_include(<mpi.h>) and never defines it; here the
// multi-GPU SpMV lives in the slab solver (cg_slab.hip), so the name resolves to the
// single-GPU stencil operator.
SpmvOperator SPMV_STENCIL_HALO_MGPU = {"stencil5-halo-mgpu", stencil_init,
                                       run_timed_generic<stencil_run_device, &g_stencil>,
                                       stencil_run_device, stencil_free};

namespace spmv_amd {
FusedSpmv fused_spmv_of(const SpmvOperator* op) {
    FusedSpmv f;
    // (the row-generic kernel of small or unverified matrices keeps the plain dot kernel, as in the slab solver)
    if ((op == &SPMV_STENCIL5_CSR || op == &SPMV_STENCIL_HALO_MGPU) && g_stencil.ready && g_stencil.plan.partials > 0 &&
        g_stencil.plan.variant != Stencil5Variant::RowGeneric) {
        f.partials = g_stencil.plan.partials;
        f.can_init = g_stencil.plan.variant == Stencil5Variant::RowLds;
        f.launch = stencil_fused_launch;
    } else if (op == &SPMV_CSR && g_csr.ready && g_csr.rows == g_csr.cols) {
        f.partials = csr_fused_dot_partials(g_csr.A.view, g_csr.csr_variant);
        f.launch = f.partials > 0 ? csr_fused_launch : nullptr;
    } else if (op == &SPMV_ELLPACK && g_ell.ready && g_ell.rows == g_ell.cols) {
        f.partials = ell_fused_dot_partials(g_ell.rows);
        f.launch = ell_fused_launch<&g_ell>;
    } else if (op == &SPMV_STENCIL5_ELLPACK && g_ell_stencil.ready && g_ell_stencil.rows == g_ell_stencil.cols) {
        f.partials = ell_fused_dot_partials(g_ell_stencil.rows);
        f.launch = ell_fused_launch<&g_ell_stencil>;
    }
    return f;
}
}  // namespace spmv_amd

extern "C" SpmvOperator* get_operator(const char* mode) {
    switch (which_operator(mode)) {
        case Which::Stencil: return &SPMV_STENCIL5_CSR;
        case Which::Csr: return &SPMV_CSR;
        case Which::Ell: return &SPMV_ELLPACK;
        case Which::EllStencil: return &SPMV_STENCIL5_ELLPACK;
        default: break;
    }
    if (mode && !strcmp(mode, "stencil5-halo-mgpu")) return &SPMV_STENCIL_HALO_MGPU;
    return nullptr;
}

extern "C" int spmv_amd_init_stencil5_synthetic(const char* mode, int n) {
    switch (which_operator(mode)) {
        case Which::Stencil:
            if (g_stencil.init_synthetic(n) != 0) return EXIT_FAILURE;
            g_stencil.A.verify_stencil(kDefaultStream);
            stencil_pick_variant();
            return 0;
        case Which::Csr:
            if (g_csr.init_synthetic(n) != 0) return EXIT_FAILURE;
            HIP_CHECK(hipStreamSynchronize(kDefaultStream));
                    g_csr.variant_name = csr_variant_name(g_csr.csr_variant, g_csr.A.view);
            csr_place_output();
            return 0;
        case Which::Ell: return ell_init_synthetic(g_ell, n);
        case Which::EllStencil: return ell_init_synthetic(g_ell_stencil, n);
        default: return EXIT_FAILURE;
    }
}

extern "C" int spmv_amd_ellpack_run_device_scaled(const char* mode, const double* d_x, double* d_y,
                                                  double alpha, double beta) {
    switch (which_operator(mode)) {
        case Which::Ell: return ell_run(g_ell, d_x, d_y, alpha, beta);
        case Which::EllStencil: return ell_run(g_ell_stencil, d_x, d_y, alpha, beta);
        default: return EXIT_FAILURE;
    }
}

extern "C" int spmv_amd_download_device_csr(const char* mode, int* row_ptr, int* col_idx,
                                            double* values) {
    CsrBackedOperator* op = which_operator(mode) == Which::Stencil ? &g_stencil
                            : which_operator(mode) == Which::Csr   ? &g_csr
                                                                   : nullptr;
    if (!op || !op->ready) return EXIT_FAILURE;
    HIP_CHECK(hipDeviceSynchronize());
    if (row_ptr) download(row_ptr, op->A.row_ptr, (size_t)op->rows + 1);
    if (col_idx) download(col_idx, op->A.col_idx, (size_t)op->A.view.nnz_local);
    if (values) download(values, op->A.values, (size_t)op->A.view.nnz_local);
    return 0;
}

namespace {
// the operator's own staging vectors: what run_timed's kernel reads and writes
bool own_vectors(const char* mode, double** dX, double** dY, size_t* cols, int* candidates, double* gain) {
    switch (which_operator(mode)) {
        case Which::Stencil: *dX = g_stencil.dX, *dY = g_stencil.dY, *cols = (size_t)g_stencil.cols, *candidates = g_stencil.y_candidates, *gain = g_stencil.y_gain; return g_stencil.ready;
        case Which::Csr: *dX = g_csr.dX, *dY = g_csr.dY, *cols = (size_t)g_csr.cols, *candidates = g_csr.y_candidates, *gain = g_csr.y_gain; return g_csr.ready;
        case Which::Ell: *dX = g_ell.dX, *dY = g_ell.dY, *cols = (size_t)g_ell.cols, *candidates = g_ell.y_candidates, *gain = g_ell.y_gain; return g_ell.ready;
        case Which::EllStencil: *dX = g_ell_stencil.dX, *dY = g_ell_stencil.dY, *cols = (size_t)g_ell_stencil.cols, *candidates = g_ell_stencil.y_candidates, *gain = g_ell_stencil.y_gain; return g_ell_stencil.ready;
        default: return false;
    }
}
}  // namespace

// `reps` launches of the operator's run_device, each timed with events on the default stream (the kernel-only time run_timed
// reports, without its copies). d_x / d_y == NULL: the operator's OWN staging vectors -- the ones run_timed's kernel works on,
// x set to 1.0 (the reference benchmark's input, main.cu:141-144), y placed by the operator at init (device_runtime.hpp).
extern "C" int spmv_amd_time_run_device(const char* mode, const double* d_x, double* d_y, int reps,
                                        float* ms_each) {
    SpmvOperator* op = get_operator(mode);
    if (!op || reps <= 0) return EXIT_FAILURE;
    if (d_x == nullptr || d_y == nullptr) {
        double *ox = nullptr, *oy = nullptr;
        size_t cols = 0;
        int cand = 0;
        double gain = 0.0;
        if (!own_vectors(mode, &ox, &oy, &cols, &cand, &gain)) return EXIT_FAILURE;
        if (d_x == nullptr) {
            launch_fill(ox, cols, 1.0, kDefaultStream);
            d_x = ox;
        }
        if (d_y == nullptr) d_y = oy;
    }
    EventTimer t;
    for (int i = 0; i < reps; ++i) {
        t.begin(kDefaultStream);
        if (op->run_device(d_x, d_y) != 0) return EXIT_FAILURE;
        t.end(kDefaultStream);
        ms_each[i] = t.elapsed_ms();
    }
    HIP_CHECK(hipGetLastError());
    return 0;
}

// Output placement of an initialised operator: how many allocations were timed for its y vector and what the choice was worth
// (kernel time on the first candidate / on the one kept). 0 = unknown operator or not initialised.
extern "C" int spmv_amd_operator_placement(const char* mode, int* candidates, double* gain) {
    double *ox = nullptr, *oy = nullptr;
    size_t cols = 0;
    int cand = 0;
    double g = 0.0;
    if (!own_vectors(mode, &ox, &oy, &cols, &cand, &g)) return 0;
    if (candidates) *candidates = cand;
    if (gain) *gain = g;
    return 1;
}

extern "C" const char* spmv_amd_operator_variant(const char* mode) {
    switch (which_operator(mode)) {
        case Which::Stencil: return g_stencil.variant_name;
        case Which::Csr: return g_csr.variant_name;
        case Which::Ell: return g_ell.variant_name;
        case Which::EllStencil: return g_ell_stencil.variant_name;
        default: return "unknown-operator";
    }
}

extern "C" int spmv_amd_operator_select_variant(const char* mode, const char* variant) {
    const bool automatic = variant == nullptr || !strcmp(variant, "auto");
    switch (which_operator(mode)) {
        case Which::Stencil:
            if (automatic) g_stencil.stencil_variant = Stencil5Variant::Auto;
            else if (!strcmp(variant, "row-lds")) g_stencil.stencil_variant = Stencil5Variant::RowLds;
            else if (!strcmp(variant, "row-direct")) g_stencil.stencil_variant = Stencil5Variant::RowDirect;
            else if (!strcmp(variant, "row-generic")) g_stencil.stencil_variant = Stencil5Variant::RowGeneric;
            else return EXIT_FAILURE;
            if (g_stencil.ready) stencil_pick_variant();
            return 0;
        case Which::Csr: {
            CsrVariant v;
            if (automatic) v = CsrVariant::Auto;
            else if (!strcmp(variant, "stream")) v = CsrVariant::Stream;
            else if (!strcmp(variant, "adaptive")) v = CsrVariant::Adaptive;
            else if (!strcmp(variant, "row-scalar")) v = CsrVariant::RowScalar;
            else if (!strcmp(variant, "wavefront")) v = CsrVariant::Wavefront;
            else return EXIT_FAILURE;
            g_csr.csr_variant = v;
            if (g_csr.ready) {
                g_csr.variant_name = csr_variant_name(v, g_csr.A.view);
            }
            return 0;
        }
        case Which::Ell:
        case Which::EllStencil: {
            // one kernel each: "auto" only
            if (!automatic) return EXIT_FAILURE;
            EllOperator& e = which_operator(mode) == Which::Ell ? g_ell : g_ell_stencil;
            if (e.ready) e.pick();
            return 0;
        }
        default: return EXIT_FAILURE;
    }
}
