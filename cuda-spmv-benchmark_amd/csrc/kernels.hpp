// kernels.hpp -- host-callable launchers of the HIP kernels (gfx950 only).
// Every launcher enqueues on the given stream and returns; none synchronises.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

struct PeerMailbox;  // comm.hpp: the dot products' all-reduce as stores between the GPUs

namespace spmv_amd {

// A row slab of a CSR matrix resident in HBM. For a single-GPU operator the slab
// is the whole matrix (row_offset = 0, nnz_base = 0, no halos).
struct SlabCsr {
    const int* row_ptr = nullptr;    // n_local + 1 entries, rebased so row_ptr[0] == 0
    const int* col_idx = nullptr;    // GLOBAL column indices
    const double* values = nullptr;
    int n_local = 0;                 // rows in the slab
    int row_offset = 0;              // global index of local row 0
    long long nnz_local = 0;
    int max_row_nnz = 0;             // longest row of the slab (0 = not known): steers csr_auto_variant
    long long nnz_base = 0;          // global CSR position of the slab's first entry
    int n_global = 0;                // rows of the whole matrix
    int grid_size = -1;              // n of the n x n stencil, <= 0 if not a stencil
    bool verified_stencil = false;   // structure checked against the complete 5-point pattern
    // x is addressed as x[col - row_offset]; indices in [-halo_before, n_local + halo_after)
    // are readable, anything else contributes 0 (reference halo kernel semantics).
    int halo_before = 0;
    int halo_after = 0;
};

// Launch switches (SPMV_AMD_* environment variables, tools/README.md). They are read ONCE, when an operator is initialised or
// a solver slab is created (current_launch_shape()), into this struct; nothing on a launch path calls getenv.
struct Tunables {
    int rowlds_min_grid = 512;    // SPMV_AMD_ROWLDS_MIN_GRID: smallest grid that takes row-lds automatically (tests force it down)
    int rowlds_group = 0;         // SPMV_AMD_ROWLDS_GROUP: consecutive row-lds tiles per XCD; 0 = derived from the grid (xcd_run_group())
};

// How many consecutive logical blocks (each `block_columns` grid columns wide) one XCD takes of every run of
// 8 * group, for kernels that walk an n x n grid row-major. Measured on MI355X (profiles/r02_xcd_group.txt): the
// best run is one grid row plus ~1100 columns -- STENCIL5 row-lds at 20 000^2: 3.74-3.79 ms with runs of 4 tiles per
// XCD, 3.65-3.71 ms with 21; 17 000^2: 2.77 -> 2.70 ms; stencil-aware ELLPACK at 20 000^2: 4.01 ms in dispatch
// order, 3.6-3.7 ms so -- for grids of 8192 columns and more; smaller grids keep short runs (4096^2: 0.150 vs 0.165 ms).
inline int xcd_run_group(int n, int block_columns, int small_grid_group) {
    if (n < 8000) return small_grid_group;
    const int g = (n + 1100 + 8 * block_columns - 1) / (8 * block_columns);
    return g < 1 ? 1 : (g > 64 ? 64 : g);
}

struct LaunchShape {
    Tunables knobs;
    // walk the tiles from the last to the first (row-lds kernel): results are identical, only the order in
    // which addresses are touched changes (sweep direction alternation of the CG loop, cg_slab.hip)
    bool reverse = false;
};

// ---- structure ----
// Fills row_ptr/col_idx/values of the slab [row_offset, row_offset+n_local) of the
// n x n 5-point stencil (centre/off values as given), rows sorted by column.
void launch_generate_stencil5_csr(int n, int row_offset, int n_local, long long nnz_base,
                                  double center, double off, int* row_ptr, int* col_idx,
                                  double* values, hipStream_t stream);
// Sets *d_mismatch (int, zeroed by the caller) to non-zero if any row of the slab deviates
// from the complete 5-point pattern of an n x n grid.
void launch_verify_stencil5_csr(const SlabCsr& m, int* d_mismatch, hipStream_t stream);

// ---- STENCIL5 SpMV ----
// y[r] = alpha * (A x)[r]. d_dot_partials, if non-null, receives one partial of
// sum_r x[r]*y_unscaled[r] per launched wave (count: stencil5_partials_needed()).
enum class Stencil5Variant { Auto, RowDirect, RowGeneric, RowLds };
// Everything about one STENCIL5 launch over local rows [first_row, last_row) that does not depend on the vectors:
// which kernel, its grid, how many dot partials it writes. Computed once per (slab, row range) -- by an operator's
// init, by a solver slab's creation -- and reused for every launch.
struct Stencil5Plan {
    Stencil5Variant variant = Stencil5Variant::RowGeneric;
    int first_row = 0, last_row = 0;
    int gi_lo = 0, gi_hi = 0;  // row-direct / row-lds: the local grid rows of the range
    int row_blocks = 0;        // row-generic: workgroups of the launch; row-direct / row-lds: workgroups per grid row
    int xcd_run = 0;           // row-lds: consecutive tiles one XCD takes of every run of 8 * xcd_run
    int partials = 0;          // dot-partial slots one launch writes (a fixed function of the slab and the range)
    const char* name = "";     // "stencil5/row-lds", ...
};
Stencil5Plan plan_stencil5(const SlabCsr& m, int first_row, int last_row, Stencil5Variant variant,
                           const LaunchShape& shape);
int rowlds_xcd_run_rule(int n);  // the untuned run length of row-lds tiles per XCD for an n x n grid
// The first SpMV of a CG solve fused with the initial residual (row-lds plans only): instead of storing y = A x the
// launch writes r = b - A x and p = r and one partial of r.r per wave into d_dot_partials.
struct ResidualOut {
    const double* b;
    double* r;
    double* p;
};
// Launch by plan. reverse: walk the tiles from the last to the first (row-lds only; same results).
// init (may be null): see ResidualOut; y is then not written and may be null.
int launch_stencil5_spmv(const SlabCsr& m, const Stencil5Plan& plan, const double* x, double* y, double alpha,
                         double* d_dot_partials, const int* d_skip_flag, bool reverse, hipStream_t stream,
                         const ResidualOut* init = nullptr);
// first_row/last_row restrict the launch to local rows [first_row, last_row); used to split
// interior rows from halo-dependent rows. Returns the number of dot partials written (0 when
// d_dot_partials is null).
int launch_stencil5_spmv(const SlabCsr& m, const double* x, double* y, double alpha,
                          int first_row, int last_row, double* d_dot_partials,
                          const int* d_skip_flag, Stencil5Variant variant,
                          const LaunchShape& shape, hipStream_t stream);

// The first and the last grid row of a slab made of whole grid rows (the rows that wait for the halos), in one
// launch where the row-lds kernel applies, in two otherwise. Partial slots: first grid row's, then last grid row's.
// `head` / `tail` = the plans of the slab's first / last grid row.
int launch_stencil5_spmv_first_and_last_gridrow(const SlabCsr& m, const Stencil5Plan& head, const Stencil5Plan& tail, const double* x,
                                                double* y, double alpha, double* d_dot_partials, const int* d_skip_flag,
                                                hipStream_t stream, const ResidualOut* init = nullptr);

// ---- CSR SpMV ----
enum class CsrVariant { Auto, Stream, Adaptive, RowScalar, Wavefront };
CsrVariant csr_auto_variant(const SlabCsr& m);
// d_dot_partials (may be null; stream / adaptive variants of SQUARE matrices only): one partial of x . (A x) per logical block,
// csr_fused_dot_partials() of them (0 = the variant has no fused form).
void launch_csr_spmv(const SlabCsr& m, const double* x, double* y, double alpha,
                     CsrVariant variant, hipStream_t stream, double* d_dot_partials = nullptr);
int csr_fused_dot_partials(const SlabCsr& m, CsrVariant variant);

// ---- ELLPACK SpMV (device layout: slot-major, element (r,k) at [k * rows + r]) ----
void launch_ell_transpose(int rows, int width, const int* idx_rowmajor, const double* val_rowmajor,
                          int* idx_slotmajor, double* val_slotmajor, hipStream_t stream);
// grid_hint: n if the matrix is known to be an n x n stencil (block -> XCD relabelling is sized to a grid row), else 0.
void launch_ell_spmv(int rows, int width, const int* idx, const double* val, const double* x,
                     double* y, double alpha, double beta, hipStream_t stream,
                     int grid_hint = 0, double* d_dot_partials = nullptr);
// d_dot_partials (may be null; square matrices): one partial of x . (A x) per workgroup, ell_fused_dot_partials() of them.
int ell_fused_dot_partials(int rows);
// Interior rows take W,C,E,N,S from slots 1,2,3,0,4 with computed columns; others walk slots.
void launch_ell_stencil5_spmv(int rows, int width, int grid_size, const int* idx,
                              const double* val, const double* x, double* y, double alpha,
                              double beta, hipStream_t stream, double* d_dot_partials = nullptr);

// ---- BLAS1 + reductions for CG ----
// Device scalars of one CG solve, laid out in one small allocation.
struct CgScalars {
    double rr_old;
    double rr_new;
    double pAp;
    double b_norm;
    double residual;
    double alpha;
    double beta;
    int converged;
    int iterations;
    int max_history;
    int stop_at;  // LAB build only (read through cg_stop_at(), reduce_device.hpp; the product library ignores the field): the scalar
                  // step of this iteration declares convergence whatever the residual, so that a stand-in slab (whose mirrored
                  // system does not converge in 14 iterations) does a converging solve's work: 0 = off
    // r.r after step k in slot k & 1 (slot 0: the initial r.r): a launch that holds step k AND work that needs beta_k reads
    // beta_k = rr_new / rr_ring[(k - 1) & 1] in every workgroup without waiting for the step -- the step writes the OTHER slot
    // (and rr_old, which only later launches read)
    double rr_ring[2];
};

void launch_fill(double* d, size_t n, double value, hipStream_t stream);
// y = y + a*x   (reference axpy_kernel, cg_solver.cu:38-43)
void launch_axpy(size_t n, double a, const double* x, double* y, hipStream_t stream);
// z = a*x + b*y (reference axpby_kernel, cg_solver.cu:48-54)
void launch_axpby(size_t n, double a, const double* x, double b, const double* y, double* z,
                  hipStream_t stream);
// y += (*d_a)*x ; y -= (*d_a)*x ; p = r + (*d_b)*p (cg_solver.cu:59-95)
void launch_axpy_dev(size_t n, const double* d_a, const double* x, double* y, bool subtract,
                     hipStream_t stream);
void launch_update_p_dev(size_t n, const double* r, const double* d_b, double* p,
                         hipStream_t stream);
// result = sum x[i]*y[i], fixed reduction shape (deterministic run to run). scratch must
// hold dot_scratch_doubles(n) doubles and be zeroed once after allocation (ticket counter).
size_t dot_scratch_doubles(size_t n);
void launch_dot(size_t n, const double* x, const double* y, double* scratch, double* d_result,
                hipStream_t stream);
// *d_out = (*d_num) / (*d_den)   (scalar_divide_kernel, cg_solver.cu:414-419)
void launch_scalar_divide(const double* d_num, const double* d_den, double* d_out,
                          hipStream_t stream);
// check_convergence_kernel (cg_solver.cu:424-431)
void launch_check_convergence(const double* d_rr_new, double b_norm, double tol, int* d_converged,
                              double* d_residual, hipStream_t stream);

// ---- reductions of the CG loop ----
// Scratch of the reductions of ONE stream (slice sums, boundary-row extras, ticket; reduce_device.hpp): reduce_scratch_doubles()
// doubles, zero before the first use (every reduction leaves it ready for the next).
struct ReduceScratch {
    double* base = nullptr;  // null: a single workgroup sums everything (short lists only)
};
int reduce_scratch_doubles();
double* reduce_scratch_alloc();  // uncached device memory where the runtime offers it, zeroed; hipFree releases it

// Device-side arrival flag of a halo exchange (round 5): instead of a cross-stream event wait in front of the launch that reads
// the halo rows, the side stream raises *flag to the exchange's sequence number behind the receive (launch_halo_arrived) and the
// boundary waves of launch_stencil5_edges_and_reduce wait for it themselves -- bounded by timeout_ticks of the 100 MHz wall
// clock; a wave that gives up sets *late (host-coherent pinned memory), which ends the run on the host. flag == nullptr: none.
struct HaloArrival {
    const unsigned* flag = nullptr;
    unsigned expected = 0;
    long long timeout_ticks = 0;
    int* late = nullptr;
};
void launch_halo_arrived(unsigned* d_flag, unsigned sequence, hipStream_t stream);

// The first and / or last grid row of a solver slab (the rows that wait for the halos) AND the reduction of the SpMV's p.Ap
// partials -- the interior launch's d_interior_partials plus these rows' own -- into *d_out, in ONE launch (spmv_kernels.hip).
// `interior` = the plan of the launch over the other rows. Returns false, having launched nothing, where the fused form does
// not apply (not a row-lds slab, a slab of fewer than three grid rows): the caller then
// launches the rows and the reduction separately.
bool launch_stencil5_edges_and_reduce(const SlabCsr& m, const Stencil5Plan& interior, bool first_gridrow, bool last_gridrow, const double* x,
                                      double* y, double alpha, const double* d_interior_partials, double* d_out, const int* d_skip_flag,
                                      const ReduceScratch& scratch, int* host_progress, int progress_value, const PeerMailbox* mailbox,
                                      hipStream_t stream, const HaloArrival& halo = HaloArrival{});

// ---- fused CG steps of the slab solver (all skip their work when s->converged) ----
// r = b - Ap ; p = r ; partials of r.r
void launch_cg_init_residual(size_t n, const double* b, const double* Ap, double* r, double* p,
                             double* partials, hipStream_t stream);
// alpha = rr_old / pAp (per thread, from the scalars) ; r -= alpha Ap ; partials of r.r
// reverse: workgroups walk the vectors from the end (same results, same partial slots).
void launch_cg_update_r(size_t n, const CgScalars* s, const double* Ap, double* r, double* partials,
                        hipStream_t stream, bool reverse = false);
// x += alpha p (the update of iteration `iteration`), then p = 1.0*r + beta*p unless that iteration
// converged; one pass over p (axpy + axpby of cg_solver_mgpu_partitioned.cu:598,682 fused).
// x = x_in + alpha p: x_in is x, or the stored initial guess in the first iteration of a solve.
// fma_form: p = fma(beta, p, r) (the single-GPU device solver's update_p_kernel, cg_solver.cu:90-95) instead of
// fma(1.0, r, beta*p) (the multi-GPU solver's axpby_kernel, mgpu :136-140).
void launch_cg_update_px(size_t n, const CgScalars* s, const double* r, double* p, const double* x_in,
                         double* x, int iteration, hipStream_t stream, bool reverse = false, bool fma_form = false);
// ---- deferred x update (cg_slab.hip, "direction ring") ----
// p_out = 1.0*r + beta*p_in for iteration `iteration` unless it converged: the direction update written out
// of place, so that p_in stays available for a later x update. Same per-element arithmetic as above.
void launch_cg_update_p_ring(size_t n, const CgScalars* s, const double* r, const double* p_in, double* p_out,
                             int iteration, hipStream_t stream, bool reverse = false, bool fma_form = false);
// x = x_in + sum_j alpha[slot_j] * p[slot_j], slot_j = (first_slot + j) % slots for j = 0..count-1, added in that
// order with one fma each: element for element the x the per-iteration updates x += alpha_j p_j produce.
constexpr int kMaxRingSlots = 16;
struct RingSlots {
    const double* p[kMaxRingSlots];
};
// s (may be null) / window_start: only terms of iterations window_start + 1 ... s->iterations are added (a host that ran one
// iteration ahead may ask for one term too many: that iteration was never counted and its alpha slot holds an old value).
void launch_cg_flush_x(size_t n, const double* alphas, const RingSlots& ring, int slots, int first_slot, int count,
                       const double* x_in, double* x, hipStream_t stream, const CgScalars* s = nullptr, int window_start = 0);
int cg_partial_count(size_t n);  // partial slots written by the two reducing kernels above
// *d_out = sum of partials[0..count) and extra[0..extra_count) in a fixed order (reduce_device.hpp: slices of `partials`, then
// [slice sums | extra]), in ONE launch. extra: the partials of a split SpMV's boundary rows (may be null).
// host_progress (may be null): int in host-coherent pinned memory, set to progress_value once the sum is stored;
// read by the solver's watchdog report.
// mailbox (may be null; comm.hpp): the sum is completed across the ranks inside the same launch.
void launch_reduce_partials(const double* partials, int count, double* d_out,
                            const int* d_skip_flag, hipStream_t stream, const ReduceScratch& scratch = ReduceScratch{},
                            int* host_progress = nullptr, int progress_value = 0,
                            const PeerMailbox* mailbox = nullptr, const double* extra = nullptr, int extra_count = 0);
// The rows a slab's neighbours wait for: [0, count_a) and [second, second + count_b), all even (its first / last grid row,
// rounded outwards to 4 KiB).
struct EdgeRows {
    size_t count_a, second, count_b;
};
// The same reduction followed by launch_cg_scalars_step() in the same launch
// (only valid when no all-reduce has to happen between the sum and the step).
void launch_reduce_partials_and_step(const double* partials, int count, double* d_out, const int* d_skip_flag,
                                     hipStream_t stream, const ReduceScratch& scratch, CgScalars* s, double tol, double* history,
                                     int* host_record, int sequence, double* alpha_ring = nullptr, int ring_slots = 0,
                                     const PeerMailbox* mailbox = nullptr, int* host_progress = nullptr,
                                     int progress_value = 0);
// The scalar step (optional: step_host_record == nullptr means an earlier launch took it), the direction update p_out = r + beta p_in
// of the EdgeRows (optional) and of rows [bulk_lo, bulk_lo + bulk_rows) in ONE launch (round 5): no workgroup waits for the
// step -- each derives beta and the convergence verdict from the scalars the step does not touch (CgScalars::rr_ring) with
// the step's own expressions. The first workgroups take the edge rows, write them through and raise the scratch's edges_ready
// to `sequence` (the side stream's exchange is released by that flag, launch_edges_wait), the rest streams the bulk.
// iteration: as launch_cg_update_p_ring's. All row counts even. One launch instead of three on the RCCL path (step | edge rows |
// rest), instead of two elsewhere; no event between them.
struct DirectionLaunch {
    CgScalars* s;
    double tol;
    int iteration;
    double* history;       // the step's (may be null)
    int* step_host_record; // null: no step in this launch
    int sequence;
    double* alpha_ring;
    int ring_slots;
    const double* r;
    const double* p_in;
    double* p_out;
    size_t bulk_lo, bulk_rows;
    bool reverse, fma_form;
};
void launch_cg_direction(const DirectionLaunch& d, const EdgeRows* edge_rows, const ReduceScratch& scratch, hipStream_t stream);
// Side stream: one thread waits (bounded) until the step-and-edges launch `sequence` on the compute stream has written its edge
// rows through to memory; the halo exchange enqueued behind it then needs no cross-stream event. *late = 3 if it gave up.
void launch_edges_wait(const ReduceScratch& scratch, int sequence, long long timeout_ticks, int* late, hipStream_t stream);
// After the (all-reduced) r.r is known: b_norm (first call), residual, history, convergence
// flag, beta, rr_old <- rr_new, iteration counter.
void launch_cg_scalars_init(CgScalars* s, double* history, hipStream_t stream);
// host_record (may be null): three ints in host-coherent pinned memory, {sequence, converged, iterations};
// the kernel publishes the iteration's status there so the host needs no copy command on the stream.
// alpha_ring (may be null): alpha of the iteration is also stored at alpha_ring[(iterations - 1) % ring_slots].
void launch_cg_scalars_step(CgScalars* s, double tol, double* history, int* host_record, int sequence,
                            hipStream_t stream, double* alpha_ring = nullptr, int ring_slots = 0);

}  // namespace spmv_amd
