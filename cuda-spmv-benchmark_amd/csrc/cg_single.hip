// cg_single.hip -- single-GPU Conjugate Gradient over any SpmvOperator.
//
//   cg_solve        <- reference src/solvers/cg_solver.cu:154-378  (host scalars; upstream the SpMV goes through
//                      run_timed with the direction vector travelling host<->device each iteration; here only for
//                      operators that have no run_device, otherwise the vector stays where it is)
//   cg_solve_device <- reference src/solvers/cg_solver.cu:436-706  (device scalars; SpMV through
//                      run_device). Round 3: no longer a restatement of the reference's loop -- 10 kernels, a blocking
//                      4-byte read-back and a D2D copy per iteration, 152 B/row -- but the slab solver's loop
//                      (cg_slab.hip) around the caller's operator: fused r update + r.r partials, direction ring with the
//                      deferred x update, scalar step in the reduction's tail, status record in host-coherent memory.
//
// Same algebra, stopping rule (||r_k|| / ||r_0|| < tol, strict, the converging iteration is
// counted), statistics and verbose output as the reference. Differences: the BLAS1 kernels move
// 16 bytes per lane, and dot products use a fixed two-stage shape (cg_kernels.hip) instead of
// 256-wide blocks + host/strided final sum, which changes results at the 1e-15 level only.
#include <math.h>

#include <vector>

#include "device_runtime.hpp"

using namespace spmv_amd;

namespace spmv_amd {
std::vector<double>& last_cg_history() {
    static std::vector<double> h;
    return h;
}
// cg_slab.hip
int cg_solve_on_operator(SpmvOperator* op, int n, const double* b, double* x, const CGConfig& config, CGStats* stats,
                         std::vector<double>* history);
}  // namespace spmv_amd

namespace {

constexpr hipStream_t kStream = nullptr;  // default stream, shared with the operators

struct Vectors {
    double *x = nullptr, *b = nullptr, *r = nullptr, *p = nullptr, *Ap = nullptr;
    double* scratch = nullptr;
    void alloc(size_t n) {
        x = device_alloc<double>(n);
        b = device_alloc<double>(n);
        r = device_alloc<double>(n);
        p = device_alloc<double>(n);
        Ap = device_alloc<double>(n);
        scratch = device_alloc<double>(dot_scratch_doubles(n));
        HIP_CHECK(hipMemset(scratch, 0, dot_scratch_doubles(n) * sizeof(double)));  // ticket counter of the reduction
    }
    void release() {
        device_release(x);
        device_release(b);
        device_release(r);
        device_release(p);
        device_release(Ap);
        device_release(scratch);
    }
};

void fill_solution_checksums(const double* x, int n, double* sum, double* norm) {
    double s = 0.0, q = 0.0;
    for (int i = 0; i < n; i++) {
        s += x[i];
        q += x[i] * x[i];
    }
    *sum = s;
    *norm = sqrt(q);
}

void print_breakdown(const char* tag, const CGStats* st) {
    printf("[%s] Converged: %s\n", tag, st->converged ? "YES" : "NO");
    printf("[%s] Iterations: %d\n", tag, st->iterations);
    printf("[%s] Final residual: %e\n", tag, st->residual_norm);
    printf("[%s] Time breakdown:\n", tag);
    printf("     Total:      %.3f ms\n", st->time_total_ms);
    printf("     SpMV:       %.3f ms (%.1f%%)\n", st->time_spmv_ms, 100.0 * st->time_spmv_ms / st->time_total_ms);
    printf("     BLAS1:      %.3f ms (%.1f%%)\n", st->time_blas1_ms, 100.0 * st->time_blas1_ms / st->time_total_ms);
    printf("     Reductions: %.3f ms (%.1f%%)\n", st->time_reductions_ms,
           100.0 * st->time_reductions_ms / st->time_total_ms);
}

}  // namespace

int cg_solve(SpmvOperator* spmv_op, MatrixData* mat, const double* b, double* x, CGConfig config,
             CGStats* stats) {
    const int n = mat->rows;
    Vectors v;
    v.alloc((size_t)n);
    double* d_scalar = device_alloc<double>(1);
    upload(v.x, x, (size_t)n);
    upload(v.b, b, (size_t)n);
    // staging buffers of the host-interface SpMV: pinned, so that the two 8n-byte copies per iteration the interface
    // imposes (here and inside run_timed) run at the link's rate instead of through pageable bounce buffers
    double *h_in = nullptr, *h_out = nullptr;
    if (spmv_op->run_device == nullptr) {
        HIP_CHECK(hipHostMalloc((void**)&h_in, (size_t)n * sizeof(double), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&h_out, (size_t)n * sizeof(double), hipHostMallocDefault));
    }
    std::vector<double>& hist = last_cg_history();
    hist.clear();

    EventTimer total, part;
    double t_spmv = 0.0, t_blas = 0.0, t_red = 0.0;
    auto host_dot = [&](const double* a, const double* c) {
        part.begin(kStream);
        launch_dot((size_t)n, a, c, v.scratch, d_scalar, kStream);
        double h = 0.0;
        download(&h, d_scalar, 1);
        part.end(kStream);
        t_red += part.elapsed_ms();
        return h;
    };
    // The reference's host path hands the direction vector to run_timed as a HOST array: four PCIe transfers of the whole vector
    // per SpMV (3.5 s per solve at 20 000^2, all of it the link: profiles/r03_cg_entry_points_20k.txt). The solver's vectors live
    // on the device here as there, so when the operator offers run_device the product is taken where the vector already is; an
    // operator WITHOUT a device entry point (run_device == NULL is legal, reference include/spmv.h:125-134) is driven through
    // run_timed exactly as upstream. Scalars (alpha, beta, the norms) stay on the host either way: that is what this entry point is.
    const bool through_host = spmv_op->run_device == nullptr;
    bool op_failed = false;  // run_device returned non-zero: stop, release everything, return 1 (never exit() from a library)
    auto host_spmv = [&](const double* d_in, double* d_out) {
        double kernel_ms = 0.0;
        part.begin(kStream);
        if (through_host) {
            download(h_in, d_in, (size_t)n);
            spmv_op->run_timed(h_in, h_out, &kernel_ms);
            upload(d_out, h_out, (size_t)n);
        } else if (spmv_op->run_device(d_in, d_out) != 0) {
            fprintf(stderr, "[CG] operator '%s': run_device failed\n", spmv_op->name);
            op_failed = true;
        }
        part.end(kStream);
        t_spmv += part.elapsed_ms();
    };
    auto timed_blas = [&](auto&& launch) {
        part.begin(kStream);
        launch();
        part.end(kStream);
        t_blas += part.elapsed_ms();
    };

    total.begin(kStream);
    host_spmv(v.x, v.Ap);
    timed_blas([&] { launch_axpby((size_t)n, 1.0, v.b, -1.0, v.Ap, v.r, kStream); });
    HIP_CHECK(hipMemcpyAsync(v.p, v.r, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, kStream));
    double rr_old = host_dot(v.r, v.r);
    const double b_norm = sqrt(rr_old);
    hist.push_back(b_norm);
    if (config.verbose >= 1) printf("[CG] Initial residual: %e\n", b_norm);

    int iter;
    double residual_norm = b_norm;
    for (iter = 0; iter < config.max_iters && !op_failed; iter++) {
        host_spmv(v.p, v.Ap);
        if (op_failed) break;
        const double pAp = host_dot(v.Ap, v.p);
        const double alpha = rr_old / pAp;
        timed_blas([&] { launch_axpy((size_t)n, alpha, v.p, v.x, kStream); });
        timed_blas([&] { launch_axpy((size_t)n, -alpha, v.Ap, v.r, kStream); });
        const double rr_new = host_dot(v.r, v.r);
        residual_norm = sqrt(rr_new);
        hist.push_back(residual_norm);
        const double rel = residual_norm / b_norm;
        if (config.verbose >= 2)
            printf("[CG] Iter %3d: residual = %e (rel = %e)\n", iter + 1, residual_norm, rel);
        if (rel < config.tolerance) {
            iter++;
            break;
        }
        const double beta = rr_new / rr_old;
        timed_blas([&] { launch_axpby((size_t)n, 1.0, v.r, beta, v.p, v.p, kStream); });
        rr_old = rr_new;
    }
    total.end(kStream);
    const float total_ms = total.elapsed_ms();
    download(x, v.x, (size_t)n);

    stats->iterations = iter;
    stats->residual_norm = residual_norm;
    stats->time_total_ms = total_ms;
    stats->time_spmv_ms = t_spmv;
    stats->time_blas1_ms = t_blas;
    stats->time_reductions_ms = t_red;
    stats->converged = (residual_norm / b_norm < config.tolerance) ? 1 : 0;
    fill_solution_checksums(x, n, &stats->solution_sum, &stats->solution_norm);
    if (config.verbose >= 1 && !op_failed) print_breakdown("CG", stats);

    v.release();
    device_release(d_scalar);
    if (h_in) (void)hipHostFree(h_in);
    if (h_out) (void)hipHostFree(h_out);
    return op_failed ? 1 : 0;
}

int cg_solve_device(SpmvOperator* spmv_op, MatrixData* mat, const double* b, double* x,
                    CGConfig config, CGStats* stats) {
    if (!spmv_op->run_device) {
        fprintf(stderr, "[ERROR] Operator '%s' does not support device-native interface\n",
                spmv_op->name);
        return 1;
    }
    // the loop lives in cg_slab.hip (cg_solve_on_operator): the slab solver's fused kernels around this operator
    if (cg_solve_on_operator(spmv_op, mat->rows, b, x, config, stats, &last_cg_history()) != 0) return 1;
    fill_solution_checksums(x, mat->rows, &stats->solution_sum, &stats->solution_norm);
    if (config.verbose >= 1) print_breakdown("CG-DEVICE", stats);
    return 0;
}

extern "C" int spmv_amd_cg_solve(SpmvOperator* op, MatrixData* mat, const double* b, double* x,
                                 const CGConfig* config, CGStats* stats) {
    return cg_solve(op, mat, b, x, *config, stats);
}

extern "C" int spmv_amd_cg_solve_device(SpmvOperator* op, MatrixData* mat, const double* b,
                                        double* x, const CGConfig* config, CGStats* stats) {
    return cg_solve_device(op, mat, b, x, *config, stats);
}

extern "C" int spmv_amd_cg_last_history(double* out, int cap) {
    const std::vector<double>& h = last_cg_history();
    const int count = (int)h.size();
    for (int i = 0; i < count && i < cap; ++i) out[i] = h[i];
    return count;
}
