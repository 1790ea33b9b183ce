/* oracle/spmv_oracle.h -- CPU restatement of the reference's SpMV + CG arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under cuda-spmv-benchmark_amd/ links, loads
 * or calls this; it is used by tests/, by __graft_entry__.smoke() and by the
 * cpu_baseline leg of bench.py, always as the checker or the reported baseline,
 * never as the thing measured or shipped.
 *
 * Pinning: the reference contains no CPU SpMV/CG and its CUDA sources cannot be
 * compiled here (no nvcc), so each function below restates one reference kernel
 * or host routine (file:line given per function) and is pinned by
 *   - the known answers the reference's own tests hold (3x3 checksum -60,
 *     identity/diag/tridiag/upper-triangular checksums, stencil == CSR to 1e-12),
 *   - the reference's shipped data file matrix/example81x81.mtx (sum(y) = -52164),
 *   - the published "14 iterations" of CG on the generator stencil at n = 10^4,
 *   - the reference's own io.cu / generate_matrix.cu compiled with g++ into
 *     oracle/_ref (Matrix Market reader + stencil writer), see oracle/Makefile.
 * Results of the closed-source cuSPARSE / cuBLAS calls on the reference path are
 * pinned only up to summation order ("parity unpinned" beyond 1e-12 relative).
 *
 * Floating point: nvcc's default -fmad=true contracts a*b+c into one fused
 * multiply-add; this file writes those contractions out as fma() (left product
 * of a sum of two products fused last, as LLVM's combiner does), so that a GPU
 * kernel written with the same explicit fma() calls agrees bit for bit.
 */
#ifndef SPMV_ORACLE_H
#define SPMV_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int row;
    int col;
    double value;
} OracleEntry;

typedef struct {
    int iterations;
    int converged;
    double residual_norm;
    double b_norm; /* ||r_0||, the denominator of the stopping test */
    double solution_sum;
    double solution_norm;
} OracleCGResult;

typedef struct {
    double median_ms, mean_ms, std_dev_ms, min_ms, max_ms;
    int valid_runs, outliers_removed;
} OracleBenchStats;

/* nnz of the n x n 5-point stencil: 5n^2 - 4n (io.cu:324-340). */
long long oracle_stencil5_nnz(int n);

/* COO entries in the writer's order C,W,E,N,S per grid point (io.cu:374-391),
 * 0-based. center/off are 5.0/-1.0 for today's generator, -4.0/-1.0 for the
 * shipped example81x81.mtx. Returns nnz written. */
long long oracle_stencil5_coo(int n, double center, double off, OracleEntry* out);

/* Direct CSR of the same matrix (what build_csr of the COO above yields), for
 * sizes where the COO would not fit; also returns row_ptr. */
void oracle_stencil5_csr(int n, double center, double off, int* row_ptr, int* col_idx,
                         double* values);

/* COO -> CSR: count, exclusive prefix sum, scatter in input order, insertion
 * sort of each row by column (spmv_cusparse_csr.cu:62-170). */
int oracle_build_csr(const OracleEntry* entries, int rows, int nnz, int* row_ptr, int* col_idx,
                     double* values);

/* calculate_interior_csr_offset (spmv_stencil_csr_direct.cu:50-67). */
int oracle_interior_csr_offset(int row, int grid_size);

/* csr_spmv_kernel (cg_solver_mgpu_partitioned.cu:40-56): y[r] = sum_k v[k]*x[col[k]],
 * ascending k from sum = 0.0. */
void oracle_spmv_csr(int rows, const int* row_ptr, const int* col_idx, const double* values,
                     const double* x, double* y);

/* stencil5_csr_direct_kernel (spmv_stencil_csr_direct.cu:76-123): interior rows
 * W,C,E,N,S at offsets o+1,o+2,o+3,o+0,o+4 of the computed offset o; other rows the
 * CSR loop; y = alpha * sum. grid_size <= 0 sends every row to the CSR loop. */
void oracle_spmv_stencil5(int rows, const int* row_ptr, const int* col_idx, const double* values,
                          const double* x, double* y, int grid_size, double alpha);

/* stencil5_csr_partitioned_halo_kernel (spmv_stencil_partitioned_halo_kernel.cu:17-98).
 * row_ptr is slab-local (rebased), col_idx global; halos may be NULL. */
void oracle_spmv_halo(const int* row_ptr, const int* col_idx, const double* values,
                      const double* x_local, const double* x_halo_prev, const double* x_halo_next,
                      double* y, int n_local, int row_offset, int N, int grid_size);

/* ELLPACK (declared only upstream, include/spmv_ellpack.h:28-51): row-major,
 * width = longest row, padding value 0.0 / index -1; SpMV walks the slots in
 * order and skips padding, y = alpha*A*x + beta*y (include/spmv_stencil.h:25-42). */
int oracle_ell_width(int rows, const int* row_ptr);
void oracle_build_ell(int rows, const int* row_ptr, const int* col_idx, const double* values,
                      int width, int* ell_idx, double* ell_val);
void oracle_spmv_ell(int rows, int width, const int* ell_idx, const double* ell_val,
                     const double* x, double* y, double alpha, double beta);

/* The BLAS1 kernels element for element, nvcc's contraction written out as fma():
 * axpy_kernel (cg_solver.cu:38-43), axpby_kernel (:48-54), axpy_sub_kernel_device (:69-74),
 * update_p_kernel (:90-95). */
void oracle_axpy(int n, double alpha, const double* x, double* y);
void oracle_axpby(int n, double alpha, const double* x, double beta, const double* y, double* z);
void oracle_axpy_sub(int n, double alpha, const double* x, double* y);
void oracle_update_p(int n, const double* r, double beta, double* p);

/* dot_kernel + sum_block_results (cg_solver.cu:110-149): 256-wide tree per block,
 * then a left-to-right host sum over the blocks. */
double oracle_dot_host(int n, const double* x, const double* y);
/* dot_kernel + final_sum_kernel (cg_solver.cu:110-132,384-409): same blocks, then
 * one 256-thread block strides over the partials and tree-reduces. */
double oracle_dot_device(int n, const double* x, const double* y);

/* cg_solve (cg_solver.cu:154-378, host scalars) when device_form == 0,
 * cg_solve_device (:436-706) when 1. SpMV = oracle_spmv_stencil5 with the given
 * grid_size (<= 0: plain CSR semantics). history, if non-NULL, receives ||r_k||
 * for k = 0..iterations (capacity hist_cap). */
int oracle_cg(int n, const int* row_ptr, const int* col_idx, const double* values, int grid_size,
              const double* b, double* x, int max_iters, double tol, int device_form,
              double* history, int hist_cap, OracleCGResult* out);

/* cg_solve_mgpu_partitioned (cg_solver_mgpu_partitioned.cu:236-908) with `world`
 * ranks simulated one after another in this process: slabs of n/world rows
 * (last takes the remainder), halo rows of grid_size entries, local dots in the
 * dot_host shape summed over ranks in rank order (stand-in for cuBLAS ddot +
 * MPI_Allreduce, whose order is not specified). Returns 2 if a slab is shorter than
 * one grid row and 3 if a slab boundary falls inside a grid row: the reference
 * reads out of bounds there, so no reference result exists to compare with. */
int oracle_cg_partitioned(int n, const int* row_ptr, const int* col_idx, const double* values,
                          int grid_size, const double* b, double* x, int max_iters, double tol,
                          int world, double* history, int hist_cap, OracleCGResult* out);

/* Slab arithmetic of the multi-GPU solver (cg_solver_mgpu_partitioned.cu:261-268). */
void oracle_partition_rows(int n, int world, int rank, int* row_offset, int* n_local);

/* benchmark_with_stats' statistics (benchmark_stats.cu:8-89) on given times. */
int oracle_bench_stats(const double* times, int count, OracleBenchStats* out);

/* calculate_spmv_metrics (spmv_metrics.cu:46-102): returns GFLOP/s and the
 * "effective" GB/s for the two CSR-based operators. */
void oracle_spmv_metrics(double ms, int rows, int cols, int nnz, double* gflops, double* gbs);

#ifdef __cplusplus
}
#endif
#endif
