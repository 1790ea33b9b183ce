// cg_kernels.hip -- BLAS1, reductions and scalar steps of Conjugate Gradient on gfx950.
//
// Element-wise arithmetic follows the reference kernels with nvcc's contraction written
// out as explicit fma() (file compiled with -ffp-contract=off):
//   axpy_kernel          y = fma(a, x, y)               cg_solver.cu:38-43 ; mgpu :125-130
//   axpby_kernel         z = fma(a, x, b*y)             cg_solver.cu:48-54 ; mgpu :136-140
//   axpy_kernel_device   y = fma(*a, x, y)              cg_solver.cu:59-64
//   axpy_sub_kernel_dev  y = fma(-(*a), x, y)           cg_solver.cu:69-74
//   update_p_kernel      p = fma(*b, p, r)              cg_solver.cu:90-95
// Reductions use a fixed shape (grid-stride lanes -> wave shuffle tree -> one partial per
// wave -> single-block tree), so a solve is bit-reproducible run to run; the summation order
// differs from the reference's 256-wide blocks, which SURVEY.md section 8 allows (1e-10).
//
// All streaming kernels move 16 bytes per lane, one pair per thread on a one-shot grid of
// ONE-WAVE workgroups, with nontemporal loads and stores for every stream that is touched once per
// pass (everything except the store of p, whose lines the next SpMV's neighbour loads re-use).
// Measured on MI355X at 4e8 rows, non-zero data (tools/stream_probe4.hip, profiles/r01_stream_probe4.txt):
// r-update 1.71 ms (256-thread blocks, plain accesses) -> 1.47 ms, x/p-update 2.89 -> 2.68 ms; capped
// grid-stride shapes were slower still (tools/stream_probe.hip). Kernels that reduce write one
// partial per workgroup (= per wave: a shuffle tree, no LDS).
#include "kernels.hpp"

#include <stdlib.h>

#include "reduce_device.hpp"
#include "spmv_amd/hip_check.h"

namespace spmv_amd {
namespace {

constexpr int kBlock = kReduceBlock;  // the reduction kernels (reduce_device.hpp)
constexpr int kStream = 64;           // the streaming kernels: one wavefront per workgroup

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// Runs the body for this thread's pair (2i, 2i+1) with 16-byte accesses; the odd tail element is
// handled by thread 0 of block 0 in each kernel.
#define SPMV_AMD_STREAM_LOOP(n)                                                     \
    const size_t pairs = (n) >> 1;                                                  \
    const size_t i = (size_t)blockIdx.x * kStream + threadIdx.x;                    \
    if (i < pairs)

// Workgroup (= wave) partial of a per-thread value: written by lane 0 to partials[blockIdx.x].
__device__ __forceinline__ void block_partial(double acc, double* __restrict__ partials) {
    acc = wave_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// 16-byte accesses of streams that are read / written once per pass
__device__ __forceinline__ d2 load_once(const double* __restrict__ base, size_t pair) {
    return __builtin_nontemporal_load(reinterpret_cast<const d2*>(base) + pair);
}
__device__ __forceinline__ void store_once(double* __restrict__ base, size_t pair, d2 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<d2*>(base) + pair);
}

__global__ __launch_bounds__(kStream) void fill_kernel(double* __restrict__ d, size_t n, double value) {
    SPMV_AMD_STREAM_LOOP(n) {
        d2 v = {value, value};
        reinterpret_cast<d2*>(d)[i] = v;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) d[n - 1] = value;
}

__global__ __launch_bounds__(kStream) void axpy_kernel(size_t n, double a, const double* __restrict__ x,
                                                      double* __restrict__ y) {
    SPMV_AMD_STREAM_LOOP(n) {
        const d2 xv = load_once(x, i);
        d2 yv = load_once(y, i);
        yv.x = fma(a, xv.x, yv.x);
        yv.y = fma(a, xv.y, yv.y);
        store_once(y, i, yv);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = fma(a, x[n - 1], y[n - 1]);
}

__global__ __launch_bounds__(kStream) void axpby_kernel(size_t n, double a, const double* __restrict__ x,
                                                       double b, const double* y, double* z) {
    SPMV_AMD_STREAM_LOOP(n) {
        const d2 xv = load_once(x, i);
        const d2 yv = load_once(y, i);
        d2 zv;
        zv.x = fma(a, xv.x, b * yv.x);
        zv.y = fma(a, xv.y, b * yv.y);
        store_once(z, i, zv);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) z[n - 1] = fma(a, x[n - 1], b * y[n - 1]);
}

template <bool kSubtract>
__global__ __launch_bounds__(kStream) void axpy_dev_kernel(size_t n, const double* __restrict__ d_a,
                                                          const double* __restrict__ x,
                                                          double* __restrict__ y) {
    const double a = kSubtract ? -(*d_a) : *d_a;
    SPMV_AMD_STREAM_LOOP(n) {
        const d2 xv = load_once(x, i);
        d2 yv = load_once(y, i);
        yv.x = fma(a, xv.x, yv.x);
        yv.y = fma(a, xv.y, yv.y);
        store_once(y, i, yv);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = fma(a, x[n - 1], y[n - 1]);
}

__global__ __launch_bounds__(kStream) void update_p_dev_kernel(size_t n, const double* __restrict__ r,
                                                              const double* __restrict__ d_b,
                                                              double* __restrict__ p) {
    const double b = *d_b;
    SPMV_AMD_STREAM_LOOP(n) {
        const d2 rv = load_once(r, i);
        d2 pv = load_once(p, i);
        pv.x = fma(b, pv.x, rv.x);
        pv.y = fma(b, pv.y, rv.y);
        reinterpret_cast<d2*>(p)[i] = pv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = fma(b, p[n - 1], r[n - 1]);
}

// One partial per wave: lanes accumulate their grid-stride elements in order, then a tree.
__global__ __launch_bounds__(kStream) void dot_partials_kernel(size_t n, const double* __restrict__ x,
                                                              const double* __restrict__ y,
                                                              double* __restrict__ partials) {
    double acc = 0.0;
    SPMV_AMD_STREAM_LOOP(n) {
        const d2 xv = load_once(x, i);
        const d2 yv = load_once(y, i);
        acc = fma(xv.x, yv.x, acc);
        acc = fma(xv.y, yv.y, acc);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc = fma(x[n - 1], y[n - 1], acc);
    block_partial(acc, partials);
}

// ---- reductions (reduce_device.hpp has the shape and the one-launch hand-over) ----

// One launch: `blocks` slice workgroups, the one that finishes last runs the second stage and the tail.
__global__ __launch_bounds__(kBlock) void reduce_one_launch_kernel(const double* __restrict__ partials, int count, int slice,
                                                                   const double* extra, int extra_count, ReduceStage stage, ReduceTail tail) {
    __shared__ double s[kBlock];
    __shared__ int s_last;
    if (tail.skip_flag != nullptr && *tail.skip_flag != 0) {
        reduce_skipped(tail, blockIdx.x == 0);
        return;
    }
    reduce_slice_block(partials, count, slice, (int)blockIdx.x, (int)gridDim.x, reinterpret_cast<const unsigned long long*>(extra), extra_count,
                       (int)gridDim.x, stage, tail, s, &s_last);
}

// A short list of partials (count <= 1024): one workgroup does both stages -- the one slice, then [its sum | extra].
__global__ __launch_bounds__(kBlock) void reduce_single_kernel(const double* __restrict__ partials, int count, const double* __restrict__ extra,
                                                               int extra_count, ReduceTail tail) {
    __shared__ double s[kBlock];
    if (tail.skip_flag != nullptr && *tail.skip_flag != 0) {
        reduce_skipped(tail, true);
        return;
    }
    block_tree(strided_sum(partials, 0, count), s);
    const double slice_sum = s[0];
    __syncthreads();
    reduce_finish(1 + extra_count, [&](int i) { return i == 0 ? slice_sum : extra[i - 1]; }, s, tail);
}

__global__ void scalar_divide_kernel(const double* num, const double* den, double* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = (*num) / (*den);
}

__global__ void check_convergence_kernel(const double* rr_new, double b_norm, double tol,
                                         int* converged, double* residual) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *residual = sqrt(*rr_new);
        const double rel = (*residual) / b_norm;
        *converged = (rel < tol) ? 1 : 0;
    }
}

// ---- fused steps of the slab solver ----

__global__ __launch_bounds__(kStream) void cg_init_residual_kernel(size_t n, const double* __restrict__ b,
                                                                  const double* __restrict__ Ap,
                                                                  double* __restrict__ r,
                                                                  double* __restrict__ p,
                                                                  double* __restrict__ partials) {
    double acc = 0.0;
    SPMV_AMD_STREAM_LOOP(n) {
        const d2 bv = load_once(b, i);
        const d2 av = load_once(Ap, i);
        d2 rv;
        rv.x = fma(-1.0, av.x, bv.x);  // axpy_kernel(-1.0, Ap, b) then r = b (mgpu :475-476)
        rv.y = fma(-1.0, av.y, bv.y);
        store_once(r, i, rv);
        reinterpret_cast<d2*>(p)[i] = rv;
        acc = fma(rv.x, rv.x, acc);
        acc = fma(rv.y, rv.y, acc);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const double rv = fma(-1.0, Ap[n - 1], b[n - 1]);
        r[n - 1] = rv;
        p[n - 1] = rv;
        acc = fma(rv, rv, acc);
    }
    block_partial(acc, partials);
}

// r -= alpha*Ap with the r.r partials (axpy_kernel(-alpha, Ap, r) + the dot, mgpu :612,627), in place like the reference's
// (an out-of-place form measured slower in round 4: profiles/r04_ab_r_pingpong.txt).
__global__ __launch_bounds__(kStream) void cg_update_r_kernel(size_t n, const CgScalars* __restrict__ s,
                                                             const double* __restrict__ Ap,
                                                             double* __restrict__ r,
                                                             double* __restrict__ partials, int reverse) {
    // The vector loads are issued BEFORE the scalars are looked at: a wave does not wait for the scalar loads, the
    // convergence test and an fp64 division ahead of its first memory request (a launch enqueued past convergence
    // only reads). alpha = rr_old / pAp from the (all-reduced) dot product: one IEEE division of two wave-uniform
    // scalars per thread, the same value the scalar step stores for the x update further down.
    const unsigned block = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;  // logical workgroup
    const size_t pairs = n >> 1;
    const size_t i = (size_t)block * kStream + threadIdx.x;
    d2 av = {0.0, 0.0}, rv = {0.0, 0.0};
    if (i < pairs) {
        av = load_once(Ap, i);
        rv = load_once(r, i);
    }
    const int converged = s->converged;
    const double rr_old = s->rr_old, pAp = s->pAp;
    if (converged) return;
    const double alpha = rr_old / pAp;
    double acc = 0.0;
    if (i < pairs) {
        rv.x = fma(-alpha, av.x, rv.x);
        rv.y = fma(-alpha, av.y, rv.y);
        store_once(r, i, rv);
        acc = fma(rv.x, rv.x, acc);
        acc = fma(rv.y, rv.y, acc);
    }
    if ((n & 1) && block == 0 && threadIdx.x == 0) {
        const double rl = fma(-alpha, Ap[n - 1], r[n - 1]);
        r[n - 1] = rl;
        acc = fma(rl, rl, acc);
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) partials[block] = acc;
}

// The direction update p' = r + beta p exists in two roundings upstream: the multi-GPU solver's axpby_kernel
// evaluates 1.0*r + beta*p, i.e. fma(1.0, r, beta*p) with beta*p rounded first (cg_solver_mgpu_partitioned.cu:136-140,
// :682), the single-GPU device solver's update_p_kernel evaluates r + beta*p as one fma(beta, p, r)
// (cg_solver.cu:90-95). fma_form selects the second; each solver keeps its own reference's arithmetic.
__device__ __forceinline__ double direction(double r, double beta, double p, int fma_form) {
    return fma_form ? fma(beta, p, r) : fma(1.0, r, beta * p);
}

// x += alpha*p of iteration `iteration` and, unless that iteration converged, p = 1.0*r + beta*p, in
// one pass over p (the reference reads p twice: axpy_kernel(alpha, p, x) :598 and axpby_kernel :682).
// Same per-element arithmetic, so results are unchanged. A launch enqueued for an iteration beyond
// the converging one finds s->iterations != iteration and does nothing.
// x_in is x itself except in the first iteration of a solve, where it is the stored initial guess:
// the solve never has to copy x0 into x first.
__global__ __launch_bounds__(kStream) void cg_update_px_kernel(size_t n, const CgScalars* __restrict__ s,
                                                              const double* __restrict__ r,
                                                              double* __restrict__ p,
                                                              const double* x_in, double* x, int iteration,
                                                              int reverse, int fma_form) {
    if (s->iterations != iteration) return;
    const bool advance = s->converged == 0;
    const double alpha = s->alpha, beta = s->beta;
    const unsigned block = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const size_t pairs = n >> 1;
    const size_t i = (size_t)block * kStream + threadIdx.x;
    if (i < pairs) {
        d2 pv = load_once(p, i);
        d2 xv = load_once(x_in, i);
        xv.x = fma(alpha, pv.x, xv.x);
        xv.y = fma(alpha, pv.y, xv.y);
        store_once(x, i, xv);
        if (advance) {
            const d2 rv = load_once(r, i);
            pv.x = direction(rv.x, beta, pv.x, fma_form);
            pv.y = direction(rv.y, beta, pv.y, fma_form);
            // plain store: a nontemporal one measured the same (15.89-15.97 ms per solve at 50 M rows either way)
            reinterpret_cast<d2*>(p)[i] = pv;
        }
    }
    if ((n & 1) && block == 0 && threadIdx.x == 0) {
        const double pv = p[n - 1];
        x[n - 1] = fma(alpha, pv, x_in[n - 1]);
        if (advance) p[n - 1] = direction(r[n - 1], beta, pv, fma_form);
    }
}

// rr_new holds the (all-reduced) initial r.r: b_norm = sqrt, history[0], rr_old.
__global__ void cg_scalars_init_kernel(CgScalars* s, double* history) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    s->rr_old = s->rr_new;
    s->rr_ring[0] = s->rr_new;
    s->b_norm = sqrt(s->rr_new);
    s->residual = s->b_norm;
    s->converged = 0;
    s->iterations = 0;
    if (history != nullptr && s->max_history > 0) history[0] = s->b_norm;
}

// rr_new holds the (all-reduced) new r.r: stopping test (strict <, on ||r||/||r0||), iteration
// count including the converging iteration, beta, rr_old <- rr_new (mgpu :652-676,716).
__global__ void cg_scalars_step_kernel(CgScalars* s, double tol, double* history, int* host_record,
                                       int sequence, double* alpha_ring, int ring_slots) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    cg_scalars_step(s, tol, history, host_record, sequence, alpha_ring, ring_slots);
}

// Step k (optional), the edge rows (optional) and a range of the direction update in one launch; see kernels.hpp, DirectionLaunch.
// Workgroup 0's first thread takes the step; NOBODY waits for it: beta_k and the verdict come from scalars the step does not
// write. A reader that sees `converged` set -- before this launch, or by this launch's step -- skips, as its own verdict would.
struct DirectionArgs {
    StepArgs step;  // step.host_record == nullptr: no step here (step.scalars, step.tol are always set)
    int iteration;
    const double* r;
    const double* p_in;
    double* p_out;
    size_t bulk_lo, bulk_pairs;
    unsigned edge_blocks;
    int reverse, fma_form;
};
__global__ __launch_bounds__(kStream) void cg_direction_kernel(DirectionArgs a, EdgeRows e, ReduceStage stage) {
    CgScalars* s = a.step.scalars;
    // (scalars first, vectors second, as in cg_update_p_ring_kernel: the launch of a converging iteration reads no vector)
    const bool edge_block = blockIdx.x < a.edge_blocks;
    const size_t pairs_a = e.count_a >> 1, edge_pairs = (e.count_a + e.count_b) >> 1, shift = (e.second - e.count_a) >> 1;
    size_t at = 0;
    bool live_lane = false;
    if (edge_block) {
        const size_t i = (size_t)blockIdx.x * kStream + threadIdx.x;
        live_lane = i < edge_pairs;
        at = i < pairs_a ? i : i + shift;
    } else {
        const unsigned bulk_blocks = gridDim.x - a.edge_blocks, logical = blockIdx.x - a.edge_blocks;
        const size_t i = (size_t)(a.reverse ? bulk_blocks - 1 - logical : logical) * kStream + threadIdx.x;
        live_lane = i < a.bulk_pairs;
        at = (a.bulk_lo >> 1) + i;
    }
    // (atomic: when the step rides in this launch, workgroup 0 may be storing `converged` while the others load it -- either
    // value leads to the verdict every workgroup derives below, but the load must not be one the compiler may assume unraced)
    const int was_converged = __hip_atomic_load(&s->converged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double rr_new = s->rr_new, rr_prev = s->rr_ring[(a.iteration - 1) & 1], b_norm = s->b_norm;
    const int stop_at = cg_stop_at(s);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.step.host_record != nullptr)
        cg_scalars_step(s, a.step.tol, a.step.history, a.step.host_record, a.step.sequence, a.step.alpha_ring, a.step.ring_slots);
    const bool live = was_converged == 0 && !cg_converging(rr_new, b_norm, a.step.tol, stop_at, a.iteration);
    if (live && live_lane) {
        const double beta = rr_new / rr_prev;
        const d2 rv = load_once(a.r, at);
        d2 pv = load_once(a.p_in, at);
        pv.x = direction(rv.x, beta, pv.x, a.fma_form);
        pv.y = direction(rv.y, beta, pv.y, a.fma_form);
        if (edge_block) {
            // written through (agent-scope stores): the exchange is released by edges_ready, not by this launch's end
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.p_out + 2 * at), (unsigned long long)__double_as_longlong(pv.x),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.p_out + 2 * at + 1), (unsigned long long)__double_as_longlong(pv.y),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            reinterpret_cast<d2*>(a.p_out)[at] = pv;  // plain: the next SpMV's neighbour loads re-use these lines
        }
    }
    if (edge_block) {  // one wave per workgroup: no barrier needed between the stores' acknowledgement and the ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            const unsigned drawn = __hip_atomic_fetch_add(stage.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (drawn == a.edge_blocks - 1) {
                __hip_atomic_store(stage.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(stage.edges_ready, (unsigned)a.step.sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Side stream, in front of the halo exchange: waits until the direction update's launch `sequence` has raised edges_ready
// (bounded; a wait that gives up sets *late = 3 and lets the exchange go). Stands where a cross-stream event stood: an event
// record cost the compute stream a barrier packet per iteration and could only follow a whole launch.
__global__ void edges_wait_kernel(const unsigned* ready, unsigned sequence, long long timeout_ticks, int* late) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - sequence) < 0) {
        if (wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(late, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

// Direction update written out of place (deferred x update, cg_slab.hip): p_out = 1.0*r + beta*p_in.
__global__ __launch_bounds__(kStream) void cg_update_p_ring_kernel(size_t n, const CgScalars* __restrict__ s,
                                                                   const double* __restrict__ r,
                                                                   const double* __restrict__ p_in,
                                                                   double* __restrict__ p_out, int iteration,
                                                                   int reverse, int fma_form) {
    // Scalars FIRST here, unlike cg_update_r_kernel: this launch is enqueued before the host knows whether the
    // iteration converged, and the launch of the converging iteration must cost nothing (loading first would read
    // 16 B/row for nothing once per solve).
    if (s->iterations != iteration || s->converged != 0) return;
    const double beta = s->beta;
    const unsigned block = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const size_t pairs = n >> 1;
    const size_t i = (size_t)block * kStream + threadIdx.x;
    if (i < pairs) {
        const d2 rv = load_once(r, i);
        d2 pv = load_once(p_in, i);
        pv.x = direction(rv.x, beta, pv.x, fma_form);
        pv.y = direction(rv.y, beta, pv.y, fma_form);
        reinterpret_cast<d2*>(p_out)[i] = pv;  // plain: the next SpMV's neighbour loads re-use these lines
    }
    if ((n & 1) && block == 0 && threadIdx.x == 0) p_out[n - 1] = direction(r[n - 1], beta, p_in[n - 1], fma_form);
}

// K directions of the ring window for one 16-byte pair: all K loads first, then the fmas in iteration order.
template <int K>
__device__ __forceinline__ void flush_chunk(d2& xv, const RingSlots& ring, const double* __restrict__ alphas, int slots,
                                            int& slot, size_t i) {
    d2 pv[K];
    double a[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const int sl = (slot + u) % slots;
        pv[u] = load_once(ring.p[sl], i);
        a[u] = alphas[sl];
    }
#pragma unroll
    for (int u = 0; u < K; ++u) {
        xv.x = fma(a[u], pv[u].x, xv.x);
        xv.y = fma(a[u], pv[u].y, xv.y);
    }
    slot = (slot + K) % slots;
}

// x = x_in + sum of alpha_j p_j over the ring window, one fma per term in iteration order: the chain
// x <- fma(alpha_j, p_j, x) the per-iteration x updates evaluate (axpy_kernel, mgpu :598), read in one pass
// with up to eight directions in flight per lane (all fourteen requested before the first fma measured slower in round 3:
// 8.56-8.59 ms against 8.03-8.36 ms at 4e8 rows, same bits).
__global__ __launch_bounds__(kStream) void cg_flush_x_kernel(size_t n, const double* __restrict__ alphas, RingSlots ring,
                                                             int slots, int first_slot, int count,
                                                             const double* x_in, double* x, const CgScalars* __restrict__ s, int window_start) {
    // Only directions of iterations that were COUNTED enter x: the host may have enqueued one iteration more than the solve needed
    // (it runs one iteration ahead of the status records while convergence looks far, cg_slab.hip), and that iteration's kernels
    // -- the step included -- did nothing: its alpha slot holds an old value. The terms are iterations window_start + 1 ...
    if (s != nullptr) {
        const int counted = s->iterations - window_start;
        if (count > counted) count = counted > 0 ? counted : 0;
    }
    const size_t pairs = n >> 1;
    const size_t i = (size_t)blockIdx.x * kStream + threadIdx.x;
    if (i < pairs) {
        d2 xv = load_once(x_in, i);
        int slot = first_slot;
        int left = count;
        for (; left >= 8; left -= 8) flush_chunk<8>(xv, ring, alphas, slots, slot, i);
        if (left >= 4) { flush_chunk<4>(xv, ring, alphas, slots, slot, i); left -= 4; }
        if (left >= 2) { flush_chunk<2>(xv, ring, alphas, slots, slot, i); left -= 2; }
        if (left >= 1) flush_chunk<1>(xv, ring, alphas, slots, slot, i);
        store_once(x, i, xv);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double xv = x_in[n - 1];
        int slot = first_slot;
        for (int j = 0; j < count; ++j) {
            xv = fma(alphas[slot], ring.p[slot][n - 1], xv);
            slot = (slot + 1) % slots;
        }
        x[n - 1] = xv;
    }
}

inline unsigned stream_grid(size_t n) {
    const size_t want = ((n >> 1) + kStream - 1) / kStream;
    return (unsigned)(want < 1 ? 1 : want);
}

}  // namespace

void launch_fill(double* d, size_t n, double value, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(fill_kernel, dim3(stream_grid(n)), dim3(kStream), 0, stream, d, n, value);
}

void launch_axpy(size_t n, double a, const double* x, double* y, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(axpy_kernel, dim3(stream_grid(n)), dim3(kStream), 0, stream, n, a, x, y);
}

void launch_axpby(size_t n, double a, const double* x, double b, const double* y, double* z,
                  hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(axpby_kernel, dim3(stream_grid(n)), dim3(kStream), 0, stream, n, a, x, b, y, z);
}

void launch_axpy_dev(size_t n, const double* d_a, const double* x, double* y, bool subtract,
                     hipStream_t stream) {
    if (n == 0) return;
    if (subtract)
        hipLaunchKernelGGL(axpy_dev_kernel<true>, dim3(stream_grid(n)), dim3(kStream), 0, stream, n, d_a, x, y);
    else
        hipLaunchKernelGGL(axpy_dev_kernel<false>, dim3(stream_grid(n)), dim3(kStream), 0, stream, n, d_a, x, y);
}

void launch_update_p_dev(size_t n, const double* r, const double* d_b, double* p, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(update_p_dev_kernel, dim3(stream_grid(n)), dim3(kStream), 0, stream, n, r, d_b, p);
}

size_t dot_scratch_doubles(size_t n) { return (size_t)stream_grid(n) + (size_t)reduce_scratch_doubles(); }
int cg_partial_count(size_t n) { return (int)stream_grid(n); }

void launch_dot(size_t n, const double* x, const double* y, double* scratch, double* d_result,
                hipStream_t stream) {
    // scratch = [one partial per block | the reduction's own scratch]
    const unsigned blocks = stream_grid(n);
    hipLaunchKernelGGL(dot_partials_kernel, dim3(blocks), dim3(kStream), 0, stream, n, x, y, scratch);
    launch_reduce_partials(scratch, (int)blocks, d_result, nullptr, stream, ReduceScratch{scratch + blocks});
}

void launch_scalar_divide(const double* d_num, const double* d_den, double* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(scalar_divide_kernel, dim3(1), dim3(1), 0, stream, d_num, d_den, d_out);
}

void launch_check_convergence(const double* d_rr_new, double b_norm, double tol, int* d_converged,
                              double* d_residual, hipStream_t stream) {
    hipLaunchKernelGGL(check_convergence_kernel, dim3(1), dim3(1), 0, stream, d_rr_new, b_norm, tol,
                       d_converged, d_residual);
}

void launch_cg_init_residual(size_t n, const double* b, const double* Ap, double* r, double* p,
                             double* partials, hipStream_t stream) {
    hipLaunchKernelGGL(cg_init_residual_kernel, dim3(stream_grid(n)), dim3(kStream), 0, stream, n, b, Ap,
                       r, p, partials);
}

void launch_cg_update_r(size_t n, const CgScalars* s, const double* Ap, double* r, double* partials,
                        hipStream_t stream, bool reverse) {
    hipLaunchKernelGGL(cg_update_r_kernel, dim3(stream_grid(n)), dim3(kStream), 0, stream, n, s, Ap, r, partials,
                       reverse ? 1 : 0);
}

void launch_cg_update_px(size_t n, const CgScalars* s, const double* r, double* p, const double* x_in,
                         double* x, int iteration, hipStream_t stream, bool reverse, bool fma_form) {
    hipLaunchKernelGGL(cg_update_px_kernel, dim3(stream_grid(n)), dim3(kStream), 0, stream, n, s, r, p, x_in, x,
                       iteration, reverse ? 1 : 0, fma_form ? 1 : 0);
}

namespace {
const StepArgs kNoStep{nullptr, 0.0, nullptr, nullptr, 0, nullptr, 0};

void reduce_impl(const double* partials, int count, const double* extra, int extra_count, double* d_out, const int* d_skip_flag,
                 hipStream_t stream, const ReduceScratch& scratch, int* host_progress, int progress_value, const PeerMailbox* mailbox,
                 const StepArgs& step) {
    const ReduceTail tail{d_out, d_skip_flag, host_progress, progress_value, mailbox, step};
    // One block walking tens of thousands of partials is latency-bound (0.3 ms for 200k on MI355X): beyond 1024 partials
    // the sum is split over up to kReduceStageBlocks slice workgroups first. Both shapes are fixed.
    int slice = 0, blocks = 0;
    reduce_geometry(count, &slice, &blocks);
    if (blocks > 1 && scratch.base == nullptr) {
        fprintf(stderr, "[reduce] %d partials need a scratch buffer\n", count);
        exit(EXIT_FAILURE);
    }
    if (blocks <= 1) {
        hipLaunchKernelGGL(reduce_single_kernel, dim3(1), dim3(kBlock), 0, stream, partials, count, extra, extra_count, tail);
    } else {
        hipLaunchKernelGGL(reduce_one_launch_kernel, dim3(blocks), dim3(kBlock), 0, stream, partials, count, slice, extra, extra_count,
                           reduce_stage_of(scratch.base), tail);
    }
}
}  // namespace

void launch_reduce_partials(const double* partials, int count, double* d_out, const int* d_skip_flag,
                            hipStream_t stream, const ReduceScratch& scratch, int* host_progress, int progress_value,
                            const PeerMailbox* mailbox, const double* extra, int extra_count) {
    reduce_impl(partials, count, extra, extra_count, d_out, d_skip_flag, stream, scratch, host_progress, progress_value, mailbox, kNoStep);
}

void launch_reduce_partials_and_step(const double* partials, int count, double* d_out, const int* d_skip_flag,
                                     hipStream_t stream, const ReduceScratch& scratch, CgScalars* s, double tol, double* history,
                                     int* host_record, int sequence, double* alpha_ring, int ring_slots,
                                     const PeerMailbox* mailbox, int* host_progress, int progress_value) {
    reduce_impl(partials, count, nullptr, 0, d_out, d_skip_flag, stream, scratch, host_progress, progress_value, mailbox,
                StepArgs{s, tol, history, host_record, sequence, alpha_ring, ring_slots});
}

void launch_cg_direction(const DirectionLaunch& d, const EdgeRows* edge_rows, const ReduceScratch& scratch, hipStream_t stream) {
    const bool edges = edge_rows != nullptr && scratch.base != nullptr && edge_rows->count_a + edge_rows->count_b > 0;
    const EdgeRows e = edges ? *edge_rows : EdgeRows{0, 0, 0};
    const unsigned edge_blocks = (unsigned)((((e.count_a + e.count_b) >> 1) + kStream - 1) / kStream);
    const size_t bulk_pairs = d.bulk_rows >> 1;
    const unsigned bulk_blocks = (unsigned)((bulk_pairs + kStream - 1) / kStream);
    const unsigned blocks = edge_blocks + bulk_blocks;
    hipLaunchKernelGGL(cg_direction_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(kStream), 0, stream,
                       DirectionArgs{StepArgs{d.s, d.tol, d.history, d.step_host_record, d.sequence, d.alpha_ring, d.ring_slots}, d.iteration, d.r,
                                     d.p_in, d.p_out, d.bulk_lo, bulk_pairs, edge_blocks, d.reverse ? 1 : 0, d.fma_form ? 1 : 0},
                       e, scratch.base != nullptr ? reduce_stage_of(scratch.base) : ReduceStage{});
}

void launch_edges_wait(const ReduceScratch& scratch, int sequence, long long timeout_ticks, int* late, hipStream_t stream) {
    hipLaunchKernelGGL(edges_wait_kernel, dim3(1), dim3(64), 0, stream, reduce_stage_of(scratch.base).edges_ready, (unsigned)sequence, timeout_ticks,
                       late);
}

int reduce_scratch_doubles() { return kReduceStageBlocks + kReduceExtraMax + 2; }  // sums | extras | ticket | edges_ready

double* reduce_scratch_alloc() {
    // uncached device memory where the runtime offers it: the slice sums and the ticket are handed between workgroups on
    // different XCDs inside one launch (reduce_device.hpp); plain device memory works through the same sc1 accesses
    void* p = nullptr;
    const size_t bytes = (size_t)reduce_scratch_doubles() * sizeof(double);
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        HIP_CHECK(hipMalloc(&p, bytes));
    }
    HIP_CHECK(hipMemset(p, 0, bytes));
    return static_cast<double*>(p);
}

void launch_cg_scalars_init(CgScalars* s, double* history, hipStream_t stream) {
    hipLaunchKernelGGL(cg_scalars_init_kernel, dim3(1), dim3(1), 0, stream, s, history);
}

void launch_cg_scalars_step(CgScalars* s, double tol, double* history, int* host_record, int sequence,
                            hipStream_t stream, double* alpha_ring, int ring_slots) {
    hipLaunchKernelGGL(cg_scalars_step_kernel, dim3(1), dim3(1), 0, stream, s, tol, history, host_record,
                       sequence, alpha_ring, ring_slots);
}

void launch_cg_update_p_ring(size_t n, const CgScalars* s, const double* r, const double* p_in, double* p_out,
                             int iteration, hipStream_t stream, bool reverse, bool fma_form) {
    hipLaunchKernelGGL(cg_update_p_ring_kernel, dim3(stream_grid(n)), dim3(kStream), 0, stream, n, s, r, p_in, p_out,
                       iteration, reverse ? 1 : 0, fma_form ? 1 : 0);
}

void launch_cg_flush_x(size_t n, const double* alphas, const RingSlots& ring, int slots, int first_slot, int count,
                       const double* x_in, double* x, hipStream_t stream, const CgScalars* s, int window_start) {
    if (n == 0 || count <= 0) return;
    hipLaunchKernelGGL(cg_flush_x_kernel, dim3(stream_grid(n)), dim3(kStream), 0, stream, n, alphas, ring, slots,
                       first_slot, count, x_in, x, s, window_start);
}

}  // namespace spmv_amd
