/* spmv_amd/api.h -- entry points of libspmv_amd.so.
 *
 * Part 1 re-declares, under the reference's own names and signatures, every
 * function the reference's harness calls on the SpMV + CG path, so that the
 * harness objects link against this library instead of the reference's
 * src/spmv, src/solvers and src/io objects. Each declaration cites the reference
 * declaration it replaces. Linkage (C vs C++) follows the reference header.
 *
 * Part 2 is the FFI surface: the same operations as plain `extern "C"`
 * functions taking only pointers, ints and doubles (structs by pointer), plus
 * the few things a caller without a HIP binding needs (device buffers, on-stream
 * timing, the multi-GPU communicator hook and a resident CG state so that a
 * benchmark can time solves with everything already in HBM).
 */
#ifndef SPMV_AMD_API_H
#define SPMV_AMD_API_H

#include <stddef.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif
#include "spmv_amd/types.h"

/* ===================================================================== *
 * Part 1 -- reference-named boundary                                     *
 * ===================================================================== */

#ifdef __cplusplus
extern "C" {
#endif

/* Process-wide host matrices. Populated by build_csr_struct / the operators'
 * init, read by calculate_spmv_metrics and by cg_solve_mgpu_partitioned.
 * reference: include/spmv.h:34,39 ; src/spmv/spmv_cusparse_csr.cu:23 */
extern CSRMatrix csr_mat;
extern ELLPACKMatrix ellpack_matrix;

/* reference: include/spmv.h:37-38 (declared there, defined nowhere in src/) */
int build_ellpack_from_csr_local(CSRMatrix* csr_matrix);
int ensure_ellpack_structure_built(MatrixData* mat);

/* Operator lookup. Canonical names: "stencil5-csr", "cusparse-csr", "ellpack",
 * "stencil5-ellpack"; accepted aliases: "stencil5", "csr". Unknown -> NULL.
 * reference: include/spmv.h:141-150 ; src/spmv/spmv.cu:11-23 */
SpmvOperator* get_operator(const char* mode);

/* reference: include/spmv.h:159-183 ; src/spmv/spmv_metrics.cu:46-102,190-324 ;
 * src/spmv/gpu_detection.cu */
void calculate_spmv_metrics(double execution_time_ms, const MatrixData* mat,
                            const char* operator_name, BenchmarkMetrics* metrics);
int get_gpu_properties(BenchmarkMetrics* metrics);
void print_benchmark_metrics(const BenchmarkMetrics* metrics, FILE* output_file);
void print_metrics_json(const BenchmarkMetrics* metrics, FILE* output_file);
void print_metrics_csv(const BenchmarkMetrics* metrics, FILE* output_file);

/* Matrix Market I/O. reference: include/io.h:71-134 ; src/io/io.cu */
int read_matrix_type(const char* filename);
void read_matrix_general(MatrixData* mat, const char* filename, int* rows, int* cols, int* nnz,
                         int** csr_rowptr, int** csr_colind, double** csr_val);
void read_matrix_symtogen(MatrixData* mat, const char* filename, int* rows, int* cols, int* nnz,
                          int** csr_rowptr, int** csr_colind, double** csr_val, int* nnz_general);
int load_matrix_market(const char* filename, MatrixData* mat);
void convert_csr_to_ellpack(const struct CSRMatrix* csr_matrix,
                            struct ELLPACKMatrix* ellpack_matrix, int* max_width);
int write_matrix_market_stencil5(int n, const char* filename);

/* N-run statistics with >2 sigma outlier removal and median.
 * reference: include/benchmark_stats.h:23-29 ; include/benchmark_stats_mgpu.h:12-15 ;
 * src/spmv/benchmark_stats.cu ; src/spmv/benchmark_stats_mgpu_partitioned.cu */
int benchmark_with_stats(int (*run_func)(const double*, double*, double*), const double* x,
                         double* y, int num_runs, BenchmarkStats* stats);
int cg_benchmark_with_stats_device(SpmvOperator* spmv_op, MatrixData* mat, double* b, double* x,
                                   CGConfig config, int num_runs, BenchmarkStats* bench_stats,
                                   CGStats* final_stats);
int cg_benchmark_with_stats_mgpu_partitioned(SpmvOperator* spmv_op, MatrixData* mat, double* b,
                                             double* x, CGConfigMultiGPU config, int num_runs,
                                             BenchmarkStats* bench_stats,
                                             CGStatsMultiGPU* final_stats);

/* reference: include/solvers/cg_metrics.h:23-37 ; src/solvers/cg_metrics.cu */
void export_cg_json(const char* filename, const char* mode, const MatrixData* mat,
                    const BenchmarkStats* bench_stats, const CGStats* cg_stats);
void export_cg_mgpu_json(const char* filename, const char* mode, const MatrixData* mat,
                         const BenchmarkStats* bench_stats, const CGStatsMultiGPU* cg_stats,
                         int num_gpus);
void export_cg_csv(const char* filename, const char* mode, const MatrixData* mat,
                   const BenchmarkStats* bench_stats, const CGStats* cg_stats, bool write_header);

#ifdef __cplusplus
} /* extern "C" */
#endif

#ifdef __cplusplus
/* C++ linkage, as in the reference headers. */

/* reference: include/spmv.h:137-139 */
extern SpmvOperator SPMV_CSR;               /* "cusparse-csr": own CSR kernels, no vendor library */
extern SpmvOperator SPMV_STENCIL5_CSR;      /* "stencil5-csr" */
extern SpmvOperator SPMV_STENCIL_HALO_MGPU; /* "stencil5-halo-mgpu": declared, never defined upstream */
extern SpmvOperator SPMV_ELLPACK;           /* "ellpack" (new: upstream ships headers only) */
extern SpmvOperator SPMV_STENCIL5_ELLPACK;  /* "stencil5-ellpack": ELL values, computed columns */

/* COO -> CSR with per-row insertion sort by column; fills csr_mat.
 * reference: include/spmv_csr.h:47 ; src/spmv/spmv_cusparse_csr.cu:62-170 */
int build_csr_struct(struct MatrixData* mat);

/* reference: include/spmv_ellpack.h:50-51 (declaration only upstream) */
int build_ellpack_from_csr_struct(const struct CSRMatrix* csr_matrix, ELLPACKMatrix* ellpack_matrix,
                                  int* max_width);

/* reference: include/solvers/cg_solver.h:59-77 ; src/solvers/cg_solver.cu:154-378,436-706 */
int cg_solve(SpmvOperator* spmv_op, MatrixData* mat, const double* b, double* x, CGConfig config,
             CGStats* stats);
int cg_solve_device(SpmvOperator* spmv_op, MatrixData* mat, const double* b, double* x,
                    CGConfig config, CGStats* stats);

/* 1-D row-slab CG over all ranks of the world communicator (see
 * spmv_amd_comm_set_world; none set = one rank). spmv_op is unused, callers pass
 * NULL. b and x are full-length host arrays on every rank; on return rank 0's x
 * holds the gathered solution.
 * reference: include/solvers/cg_solver_mgpu_partitioned.h:50-51 ;
 * src/solvers/cg_solver_mgpu_partitioned.cu:236-908 */
int cg_solve_mgpu_partitioned(SpmvOperator* spmv_op, MatrixData* mat, const double* b, double* x,
                              CGConfigMultiGPU config, CGStatsMultiGPU* stats);
#endif /* __cplusplus */

/* ===================================================================== *
 * Part 2 -- FFI surface (all extern "C", prefix spmv_amd_)               *
 * ===================================================================== */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- FFI aliases of the C++-linkage entry points above (configs by pointer) ---- */
int spmv_amd_build_csr_struct(MatrixData* mat);
int spmv_amd_build_ellpack_from_csr_struct(const CSRMatrix* csr, ELLPACKMatrix* ell,
                                           int* max_width);
int spmv_amd_cg_solve(SpmvOperator* op, MatrixData* mat, const double* b, double* x,
                      const CGConfig* config, CGStats* stats);
int spmv_amd_cg_solve_device(SpmvOperator* op, MatrixData* mat, const double* b, double* x,
                             const CGConfig* config, CGStats* stats);
int spmv_amd_cg_solve_mgpu_partitioned(MatrixData* mat, const double* b, double* x,
                                       const CGConfigMultiGPU* config, CGStatsMultiGPU* stats);

/* Drops the host CSR/ELL held in csr_mat / ellpack_matrix (the reference keeps
 * them for the life of the process; tests that load several matrices need this). */
void spmv_amd_reset_host_matrices(void);

/* ---- integer helpers that must match the reference bit for bit ---- */

/* CSR start of interior row `row` of an n x n 5-point stencil.
 * reference: calculate_interior_csr_offset, src/spmv/spmv_stencil_csr_direct.cu:50-67 */
int spmv_amd_interior_csr_offset(int row, int grid_size);

/* Row slab of `rank`: n / world rows each, the last rank takes the remainder.
 * reference: src/solvers/cg_solver_mgpu_partitioned.cu:261-268 */
void spmv_amd_partition_rows(int n, int world, int rank, int* row_offset, int* n_local);

/* ---- device plumbing for callers that have no HIP binding of their own ---- */
int spmv_amd_device_count(void);
int spmv_amd_set_device(int device);
/* The calling thread's current device; pci_bus_id (may be NULL) receives "0000:c1:00.0"-style text. */
int spmv_amd_current_device(char* pci_bus_id, int cap);
void* spmv_amd_device_alloc(size_t bytes);
void spmv_amd_device_free(void* d_ptr);
int spmv_amd_copy_to_device(void* d_dst, const void* h_src, size_t bytes);
int spmv_amd_copy_to_host(void* h_dst, const void* d_src, size_t bytes);
int spmv_amd_device_fill_f64(double* d_ptr, size_t count, double value);
int spmv_amd_device_synchronize(void);

/* Measurement aid: the rate this GPU sustains for the byte mix of one STENCIL5 row (48 B read : 8 B written,
 * SURVEY.md 8d) with ideal accesses -- coalesced 8-byte nontemporal loads / stores, no neighbours, no row
 * structure -- on the benchmark's own data. `warmup` untimed launches over `rows` rows, then `reps` launches
 * timed one by one with HIP events on the launch stream (ms_each[reps]). Returns bytes moved per launch. */
double spmv_amd_stream_ceiling(size_t rows, int warmup, int reps, float* ms_each);
/* The same probe for the byte mix of another format at five entries per row: mix 0 = STENCIL5 (56 B/row, as above),
 * 1 = CSR (80 B/row: values, column indices, row pointers, x, y), 2 = ELLPACK width 5 (76 B/row). The index
 * streams are read, nothing is gathered through them: what the streams alone cost on this GPU. */
double spmv_amd_stream_ceiling_mix(int mix, size_t rows, int warmup, int reps, float* ms_each);

/* ---- operators on synthetic, device-generated matrices ---- */

/* Initialises operator `mode` on the n x n 5-point stencil of
 * write_matrix_market_stencil5 (centre 5.0, neighbours -1.0) with the CSR/ELL
 * arrays produced directly in HBM by a generator kernel: same bytes as
 * load_matrix_market + init would upload, without the host COO/CSR (58 GB at
 * n = 20000). csr_mat gets the dimensions, its host arrays stay NULL. */
int spmv_amd_init_stencil5_synthetic(const char* mode, int n);

/* y = alpha*A*x + beta*y on an initialised "ellpack" / "stencil5-ellpack" operator: the full
 * contract of the kernel prototyped in reference include/spmv_stencil.h:25-42 (the operator table
 * itself always runs alpha = 1, beta = 0). Device pointers, default stream, no synchronisation. */
int spmv_amd_ellpack_run_device_scaled(const char* mode, const double* d_x, double* d_y, double alpha,
                                       double beta);

/* Downloads the device CSR of an initialised CSR-based operator (tests compare
 * it with build_csr_struct's host result). Any pointer may be NULL. */
int spmv_amd_download_device_csr(const char* mode, int* row_ptr, int* col_idx, double* values);

/* Launches run_device `reps` times back to back on the operator's stream and
 * returns each launch's duration from HIP events recorded on that stream.
 * d_x / d_y == NULL: the operator's own staging vectors, i.e. what run_timed's kernel reads and writes
 * (x set to 1.0, the reference benchmark's input; y placed by the operator at init, see below). */
int spmv_amd_time_run_device(const char* mode, const double* d_x, double* d_y, int reps,
                             float* ms_each);
/* Output placement. On MI355X a SpMV runs ~4.5 % faster when the vector it WRITES lies in another class of 32 GiB
 * address regions than the data it reads (csrc/device_runtime.hpp; profiles/r04_spmv_regions.txt), and only hipMalloc
 * decides the class: at init every operator times its kernel on up to SPMV_AMD_PLACEMENT_CANDIDATES (default 3, one region apart; 1 = off)
 * allocations for its y vector and keeps the fastest (vectors of >= 16 Mi rows). Reports how many were timed and
 * kernel time on the first / on the one kept. Returns 0 for an unknown or uninitialised operator. */
int spmv_amd_operator_placement(const char* mode, int* candidates, double* gain);

/* Which kernel variant the last init selected: a static string, one of
 * "stencil5/row-lds" (default on verified stencils with grid >= 512), "stencil5/row-direct"
 * (default on smaller verified stencils), "stencil5/row-generic", "stencil5/row-generic(csr-loop)";
 * "csr/stream", "csr/adaptive", "csr/row-scalar", "csr/wavefront"; "ell/slot-major", "ell/stencil5-direct". */
const char* spmv_amd_operator_variant(const char* mode);

/* Forces a kernel variant of `mode` ("row-lds", "row-direct", "row-generic" / "stream", "adaptive", "row-scalar",
 * "wavefront"; NULL or "auto" = automatic). A variant
 * whose preconditions the matrix does not meet falls back to the next applicable one. */
int spmv_amd_operator_select_variant(const char* mode, const char* variant);

/* ---- the CG building blocks one at a time (device pointers, each call synchronises) ----
 * reference kernels: axpy_kernel / axpby_kernel (cg_solver.cu:38-54), axpy_kernel_device /
 * axpy_sub_kernel_device (:59-74), update_p_kernel (:90-95), dot_kernel + final_sum_kernel (:110-132,384-409). */
int spmv_amd_blas1_axpy(size_t n, double a, const double* d_x, double* d_y);
int spmv_amd_blas1_axpby(size_t n, double a, const double* d_x, double b, const double* d_y, double* d_z);
int spmv_amd_blas1_axpy_dev(size_t n, double a, const double* d_x, double* d_y, int subtract);
int spmv_amd_blas1_update_p_dev(size_t n, const double* d_r, double b, double* d_p);
int spmv_amd_blas1_dot(size_t n, const double* d_x, const double* d_y, double* result);
/* The slab solver's fused steps on caller data. which = 0: r -= (rr_old/pAp)*Ap and r.r (d_a = Ap, d_b = r);
 * 1: p_out = r + beta*p_in (d_a = r, d_b = p_in, d_c = p_out); 2: r = b - Ap, p = r and r.r (d_a = b, d_b = Ap,
 * d_c = [r | p], 2n doubles). scalars = {rr_old, pAp, beta}. reverse: sweep direction (results must not depend on it). */
int spmv_amd_cg_fused_step(int which, size_t n, const double* scalars, const double* d_a, double* d_b, double* d_c,
                           int reverse, double* dot_out);

/* ---- residual history of the most recent CG solve in this process ---- */
/* Copies ||r_k||, k = 0..iterations, into out (at most cap values); returns how
 * many the solve recorded. */
int spmv_amd_cg_last_history(double* out, int cap);
/* cg_solve_device keeps its five vectors and direction ring between calls (the harness solves one system 13 times,
 * reference src/main/cg_solver.cu:154-178). They are released by the free() of any of this library's operators, or here --
 * the call for a caller that drives its own SpmvOperator table through cg_solve_device. Safe to call at any time. */
void spmv_amd_cg_release_workspace(void);

/* ---- multi-GPU communicator ---- */
typedef struct SpmvAmdComm SpmvAmdComm;

/* Host-side callbacks of a staged communicator: the library moves the halo rows
 * and the scalars through pinned host memory and calls these on host buffers
 * (the reference's D2H -> MPI -> H2D scheme, cg_solver_mgpu_partitioned.cu:173-231).
 * A NULL send/recv pointer means "no neighbour on that side". */
typedef int (*SpmvAmdHostHaloFn)(void* user, const double* send_prev, const double* send_next,
                                 double* recv_prev, double* recv_next, int count);
typedef int (*SpmvAmdHostAllreduceFn)(void* user, double* inout, int count);
/* root receives counts[r] doubles at displs[r] of recv; every rank sends send[0..n_send). */
typedef int (*SpmvAmdHostGatherFn)(void* user, const double* send, int n_send, double* recv,
                                   const int* counts, const int* displs);
typedef int (*SpmvAmdHostBarrierFn)(void* user);

/* 256-byte blob (two RCCL unique ids: one communicator for the halo send/recv, one for
 * the all-reduces), to be produced on rank 0 and handed to every rank. */
#define SPMV_AMD_COMM_ID_BYTES 256
int spmv_amd_comm_unique_id(void* out_id256);
/* One process per GPU: RCCL communicator over xGMI; halo rows travel by
 * ncclSend/ncclRecv on a side stream, dot products by ncclAllReduce. */
/* Returns NULL if RCCL cannot create the communicators. */
SpmvAmdComm* spmv_amd_comm_create_rccl(int rank, int world, const void* id256);
SpmvAmdComm* spmv_amd_comm_create_staged(int rank, int world, SpmvAmdHostHaloFn halo,
                                         SpmvAmdHostAllreduceFn allreduce,
                                         SpmvAmdHostGatherFn gather, SpmvAmdHostBarrierFn barrier,
                                         void* user);
void spmv_amd_comm_destroy(SpmvAmdComm* comm);
/* The communicator cg_solve_mgpu_partitioned runs over (MPI_COMM_WORLD's role). */
void spmv_amd_comm_set_world(SpmvAmdComm* comm);
int spmv_amd_comm_rank(const SpmvAmdComm* comm);
int spmv_amd_comm_size(const SpmvAmdComm* comm);
/* Runs one all-reduce (value rank+1 on every rank -> world*(world+1)/2), one neighbour exchange
 * (64 doubles carrying the sender's rank, both ways) and one barrier through the communicator and
 * checks what arrived; 0 on success. Collective: every rank must call it. */
int spmv_amd_comm_selftest(SpmvAmdComm* comm);
/* All ranks meet (reference: MPI_Barrier, cg_solver_mgpu_partitioned.cu:405). Like every wait on another
 * rank in this library it is bounded by the watchdog: after SPMV_AMD_WATCHDOG_S seconds (default 60,
 * 0 = unbounded) without progress the rank reports where it is stuck and exits with EXIT_FAILURE. */
int spmv_amd_comm_barrier(SpmvAmdComm* comm);
/* "rccl", "staged" or "self"; and the number of ranks the device transport itself reports
 * (ncclCommCount of both RCCL communicators; 0 for transports without such a notion). */
const char* spmv_amd_comm_transport(const SpmvAmdComm* comm);
int spmv_amd_comm_transport_ranks(const SpmvAmdComm* comm);

/* ---- peer-mailbox all-reduce (optional; csrc/comm.hpp) ----
 * The two dot-product all-reduces per CG iteration (reference: MPI_Allreduce on the host,
 * cg_solver_mgpu_partitioned.cu:583,645) as direct stores between the GPUs of one node: every rank owns a small
 * mailbox in uncached device memory that all ranks map through hipIpc. spmv_amd_comm_mailbox_enable() does the
 * whole set-up over the communicator's own transport (collective; returns 1 when EVERY rank has a mailbox that
 * passed a self-test of 2048 checked all-reduces, else 0 and the communicator keeps all-reducing through RCCL / the staged callbacks).
 * The three steps are also available one by one for callers that exchange the handles themselves. */
#define SPMV_AMD_MAILBOX_HANDLE_BYTES 64
int spmv_amd_comm_mailbox_enable(SpmvAmdComm* comm);
int spmv_amd_comm_mailbox_prepare(SpmvAmdComm* comm, void* handle_out64);
int spmv_amd_comm_mailbox_connect(SpmvAmdComm* comm, const void* all_handles, int count);
int spmv_amd_comm_mailbox_selftest(SpmvAmdComm* comm, int rounds);
void spmv_amd_comm_mailbox_disable(SpmvAmdComm* comm);
int spmv_amd_comm_mailbox_ready(const SpmvAmdComm* comm);

/* ---- resident multi-GPU CG (what cg_solve_mgpu_partitioned is built from) ---- */
typedef struct SpmvAmdCgSlab SpmvAmdCgSlab;

/* Slab of a host matrix: build_csr_struct + slice + upload, as the reference does. */
SpmvAmdCgSlab* spmv_amd_cg_slab_create(MatrixData* mat, SpmvAmdComm* comm);
/* Slab of the synthetic n x n stencil, generated in HBM; b = 1, x0 = 0. */
SpmvAmdCgSlab* spmv_amd_cg_slab_create_stencil5(int n, SpmvAmdComm* comm);
/* Full-length host vectors as in the reference (each rank uploads its slab);
 * NULL keeps b = 1 / x0 = 0. */
int spmv_amd_cg_slab_set_vectors(SpmvAmdCgSlab* s, const double* b_full, const double* x0_full);
/* One solve from the stored x0; the timed region is the reference's
 * (cg_solver_mgpu_partitioned.cu:405-413 -> 728-731).
 * The loop has two shapes (csrc/cg_slab.hip, LoopShape), bit-identical in their results. PIPELINE (default on a slab with
 * neighbours): the halo exchange runs on a side stream under the interior rows' SpMV; the boundary rows wait inside their
 * launch for a device flag the side stream raises behind the receive, and the side stream's exchange waits for a device
 * flag the direction update's first workgroups raise. CONCURRENCY REQUIREMENT: those two waits spin (bounded: 20 s, or half of
 * SPMV_AMD_WATCHDOG_S) for a kernel on the OTHER stream, so the two streams' kernels must be able to run at the same time --
 * true on the hardware queues of a normal process, NOT under a tool that serialises dispatches (rocprofv3 --pmc): run such
 * passes with SPMV_AMD_NO_OVERLAP=1. PLAIN (SPMV_AMD_NO_OVERLAP=1 read at creation, detailed timers, the in-place form, slabs too
 * thin to split): everything on the compute stream, the exchange behind the whole direction update, the reference's own order
 * (:680-703); no device flag, no spin. A wait that gives up ends the process with the watchdog's sentence on stderr. */
int spmv_amd_cg_slab_solve(SpmvAmdCgSlab* s, const CGConfigMultiGPU* config,
                           CGStatsMultiGPU* stats);
/* Which shape this slab's loop takes and who decided (a static string): "single rank"; "pipeline (verified against the plain
 * order at creation)"; "plain: SPMV_AMD_NO_OVERLAP=1"; "plain: the in-place form (ring 1) or a slab too thin to split"; "plain: the
 * pipeline's residual history differed from the plain order's in the creation check". The creation check: the first slab created on a communicator that exchanges halos solves four iterations
 * in each shape (default b = 1, x0 = 0); the shapes are bit-identical by construction, so a difference on any rank -- lost or stale
 * halo rows, a device flag that never comes -- makes every slab on that communicator run the plain order, with a line on
 * stderr. Set-up work, once per communicator, outside every timed region. */
const char* spmv_amd_cg_slab_loop_shape(const SpmvAmdCgSlab* s);
/* Gathers the solution into x_full on rank 0 (other ranks get their own slab). */
int spmv_amd_cg_slab_gather(SpmvAmdCgSlab* s, double* x_full);
int spmv_amd_cg_slab_history(SpmvAmdCgSlab* s, double* out, int cap);
/* Slab-local SpMV on caller data: uploads x_full's slab + halos, returns y slab. */
int spmv_amd_cg_slab_spmv(SpmvAmdCgSlab* s, const double* x_full, double* y_local);
void spmv_amd_cg_slab_info(const SpmvAmdCgSlab* s, int* row_offset, int* n_local, int* local_nnz);
/* The SpMV kernel variant the slab's solver launches ("stencil5/row-lds", ...): a static string. */
const char* spmv_amd_cg_slab_variant(const SpmvAmdCgSlab* s);
/* Average duration (HIP events on the solver's stream) of `reps` launches of the
 * slab SpMV kernel pair on the current direction vector. */
int spmv_amd_cg_slab_time_spmv(SpmvAmdCgSlab* s, int reps, float* ms_each);
/* Per-rank timeline of a solve: what the reference reports as max / min of six timers per rank
 * (cg_solver_mgpu_partitioned.cu:748-800), taken here WITHOUT the per-stage host syncs its detailed timers need:
 * with the timeline on, a solve records HIP events at the stage boundaries of every iteration (compute stream) and
 * around every halo exchange (side stream) and resolves them after the loop, so the overlap of the exchange with the
 * interior rows stays as it is in a timed solve. spmv_amd_cg_slab_timeline() returns the values of the last such
 * solve -- averages per counted iteration, microseconds -- in the order of the comma-separated names
 * spmv_amd_cg_slab_timeline_names() returns; 0 values if the last solve ran without the timeline. Each stage runs from
 * the end of the previous stage's last kernel to the end of its own, so queue gaps are inside the stage that waits.
 * (Since round 5 the scalar step of the RCCL path runs inside the direction update's launch: the stage named
 * "reduce_rr_allreduce_and_scalar_step" then holds the sum and the all-reduce only.)
 * Every event record is a barrier packet on the stream (~6 us on MI355X): a solve with the timeline on is ~40 us per iteration
 * slower than a plain one and its small stages consist mostly of that packet (profiles/r05_slab_timeline_p8.txt).
 * The direction-update stage is averaged over the launches that did work ("direction_updates": the converging
 * iteration's update is never needed -- the reference tests convergence before its p update, :652-676). */
void spmv_amd_cg_slab_set_timeline(SpmvAmdCgSlab* s, int on);
/* Placement at creation (csrc/cg_slab.hip). Slabs of >= 16 Mi rows place their coefficient stream by timing up to three
 * allocations one region apart: {0, candidates timed, SpMV ms before, SpMV ms kept}. Returns 4, or 0 if it did not run.
 * Addresses only: results never change; a candidate that does not fit ends the trial, never the process. */
int spmv_amd_cg_slab_placement(const SpmvAmdCgSlab* s, double* out, int cap);
/* Row-lds tiles per XCD and run, timed at creation on slabs of >= 16 Mi rows (csrc/device_runtime.hpp, tune_rowlds_xcd_run):
 * {rule, kept, SpMV ms with the rule, ms kept}. Returns 4, or 0 if the trial did not run. Workgroup -> tile mapping only. */
int spmv_amd_cg_slab_tile_runs(const SpmvAmdCgSlab* s, double* out, int cap);
/* Wall ms of creation's set-up phases, each closed by a device synchronisation; none of it lies inside a timed region (the
 * reference builds and uploads before its own, cg_solver_mgpu_partitioned.cu:303-413): {matrix to HBM (upload or generation),
 * streams + vectors, verification + launch plans, coefficient placement trial, tile-run trial}. Returns 5. */
int spmv_amd_cg_slab_setup_ms(const SpmvAmdCgSlab* s, double* out, int cap);
/* The timed in-loop SpMV launches of the last solve, one by one (ms, iteration order). Returns their number. */
int spmv_amd_cg_slab_spmv_launch_ms(const SpmvAmdCgSlab* s, float* out, int cap);
const char* spmv_amd_cg_slab_timeline_names(void);
int spmv_amd_cg_slab_timeline(const SpmvAmdCgSlab* s, double* out, int cap);
void spmv_amd_cg_slab_destroy(SpmvAmdCgSlab* s);

/* write_matrix_market_stencil5 with other value texts (e.g. "-4.0", "-1.0": the convention
 * of the shipped matrix/example81x81.mtx); used to regenerate test fixtures. */
int spmv_amd_write_stencil5_values(int n, const char* filename, const char* center_text,
                                   const char* off_text);

/* Library build string ("libspmv_amd <date> gfx950 ..."). */
const char* spmv_amd_version(void);

#ifdef __cplusplus
}
#endif

#endif /* SPMV_AMD_API_H */
