/* spmv_amd/types.h -- plain-data structs shared across the drop-in boundary.
 *
 * Every struct here is layout-compatible (same field order, same types) with the
 * struct of the same name in the reference, so objects compiled against the
 * reference headers can be linked against libspmv_amd.so unchanged:
 *
 *   Entry, MatrixData            <- reference include/io.h:43-59
 *   CSRMatrix                    <- reference include/spmv_csr.h:28-35
 *   ELLPACKMatrix                <- reference include/spmv_ellpack.h:28-36
 *   SpmvOperator                 <- reference include/spmv.h:125-134
 *   BenchmarkMetrics             <- reference include/spmv.h:76-113
 *   BenchmarkStats               <- reference include/benchmark_stats.h:12-20
 *   CGConfig, CGStats            <- reference include/solvers/cg_solver.h:21-43
 *   CGConfigMultiGPU,
 *   CGStatsMultiGPU              <- reference include/solvers/cg_solver_mgpu.h:38-71
 *
 * All indices are signed 32-bit, all values IEEE fp64.
 */
#ifndef SPMV_AMD_TYPES_H
#define SPMV_AMD_TYPES_H

#include <stdio.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAX_LINE_LENGTH 1024 /* io.h:33 -- longest header/comment line the .mtx reader accepts */
#define MAX_WIDTH 1000       /* spmv_ellpack.h:18 -- widest ELLPACK row the builder accepts */

/* One COO non-zero, 0-based after loading (16 bytes). */
typedef struct {
    int row;
    int col;
    double value;
} Entry;

/* A matrix as the Matrix Market loader hands it over. grid_size is n for an
 * n x n 5-point-stencil matrix announced by a "% STENCIL_GRID_SIZE n" comment,
 * -1 otherwise. The caller owns entries. */
typedef struct MatrixData {
    int rows;
    int cols;
    int nnz;
    int grid_size;
    Entry* entries;
} MatrixData;

/* Host CSR. row_ptr has nb_rows+1 entries; each row is sorted by column. */
struct CSRMatrix {
    int nb_rows;
    int nb_cols;
    int nb_nonzeros;
    int* row_ptr;
    int* col_indices;
    double* values;
};

/* Host ELLPACK, row-major: slot k of row r lives at [r * ell_width + k].
 * Unused slots hold value 0.0 and column index -1 (this build's choice: the
 * reference only declares the struct). */
struct ELLPACKMatrix {
    int nb_rows;
    int nb_cols;
    int ell_width;
    int grid_size;
    int* indices;
    int nb_nonzeros;
    double* values;
};

typedef struct CSRMatrix CSRMatrix;
typedef struct ELLPACKMatrix ELLPACKMatrix;

/* Operator table. run_timed takes HOST pointers and reports kernel-only
 * milliseconds; run_device takes DEVICE pointers, enqueues and returns without
 * synchronising; both return 0 on success. */
typedef struct {
    const char* name;
    int (*init)(MatrixData* mat);
    int (*run_timed)(const double* x, double* y, double* kernel_time_ms);
    int (*run_device)(const double* d_x, double* d_y);
    void (*free)();
} SpmvOperator;

typedef struct {
    double execution_time_ms;
    double gflops;
    double bandwidth_gb_s;
    int matrix_rows;
    int matrix_cols;
    int matrix_nnz;
    int grid_size;
    double sparsity_ratio;
    const char* operator_name;
    double sum_y;
    double norm2_y;
    struct {
        char name[128];
        int memory_mb;
        char compute_capability[16];
        int multiprocessor_count;
        int max_threads_per_block;
        int memory_clock_khz;
        int graphics_clock_mhz;
        int cuda_runtime_version; /* holds the HIP runtime version here */
        int cuda_driver_version;  /* holds the HIP driver version here */
        int cusparse_version;     /* 0: no vendor sparse library on this path */
        int current_temp_c;
        int max_temp_c;
        int power_draw_w;
        int power_limit_w;
        char persistence_mode[16];
        char cpu_model[128];
        int system_ram_gb;
        char pcie_generation[16];
        int pcie_link_width;
    } gpu_info;
} BenchmarkMetrics;

typedef struct {
    double median_ms;
    double mean_ms;
    double std_dev_ms;
    double min_ms;
    double max_ms;
    int valid_runs;
    int outliers_removed;
} BenchmarkStats;

typedef struct {
    int max_iters;
    double tolerance; /* on ||r_k|| / ||r_0||, strict < */
    int verbose;      /* 0 silent, 1 summary, 2 per iteration */
    int enable_detailed_timers;
} CGConfig;

typedef struct {
    int iterations;
    double residual_norm;
    double time_total_ms;
    double time_spmv_ms;
    double time_blas1_ms;
    double time_reductions_ms;
    int converged;
    double solution_sum;
    double solution_norm;
} CGStats;

typedef struct {
    int max_iters;
    double tolerance;
    int verbose;
    int enable_detailed_timers;
} CGConfigMultiGPU;

typedef struct {
    int iterations;
    double residual_norm;
    double time_total_ms;
    double time_spmv_ms;
    double time_blas1_ms;
    double time_reductions_ms;
    double time_allreduce_ms;
    double time_allgather_ms; /* halo exchange time (name kept from the reference) */
    int converged;
    double time_dot_rs_initial_ms;
    double time_dot_pAp_ms;
    double time_dot_rs_new_ms;
    double time_axpy_update_x_ms;
    double time_axpy_update_r_ms;
    double time_axpby_update_p_ms;
    double time_initial_r_ms;
    double solution_sum;
    double solution_norm;
} CGStatsMultiGPU;

#ifdef __cplusplus
}
#endif

#endif /* SPMV_AMD_TYPES_H */
