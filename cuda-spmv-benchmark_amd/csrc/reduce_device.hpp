// reduce_device.hpp -- device side of the CG loop's dot-product reductions: ONE launch per dot product (round 5).
//
// Shape: up to kReduceStageBlocks workgroups of 256 threads each sum one contiguous slice of the partials (thread t takes
// t, t + 256, ... in ascending order, then a 256-wide LDS tree); ONE workgroup then sums [slice sums | extra values] the same
// way, completes the sum across the ranks (peer mailbox, optional) and runs the CG scalar step (optional). The slice stage is
// round 2's, so every sum without extras keeps its bits; the EXTRA values are the partials of a split SpMV's boundary rows
// (the slab's first / last grid row), which enter in the second stage whichever launch computed them -- the same sum, bit
// for bit, whether the rows ran in a launch of their own or inside the reducing launch (spmv_kernels.hip). Rounds 2-4 issued
// these as two launches (reduce_slices_kernel, reduce_final_kernel): 13-17 us per dot product on the P = 8 slab of the headline
// grid and 29 us at 4e8 rows, two of them per iteration -- half of the fixed cost that keeps a 1/P slab from costing T1/P
// (profiles/r05_slab_attribution.txt). Two changes:
//  * the slice loop issues sixteen independent loads before it adds them (same order of additions): the two-launch kernel walked
//    its 6-48 partials per thread as a chain of dependent load -> add steps, ~0.6 us each;
//  * the workgroup that finishes LAST does the second stage in the same launch. Hand-over without cache maintenance: a slice
//    sum is published with an agent-scope relaxed atomic store (sc1: written through to the memory side, past the XCD-private
//    L2), s_waitcnt vmcnt(0) waits for its acknowledgement, then an agent-scope relaxed fetch-add draws a ticket; the workgroup
//    that draws the last ticket reads the slice sums with agent-scope atomic loads (sc1: not served from its own L2). This is
//    the instruction sequence of a release / acquire pair WITHOUT buffer_wbl2 / buffer_inv: those write back and invalidate a
//    whole L2, which is what made round 2's one-launch form (agent-scope fences in 256 workgroups) slower than two launches.
//    The stage buffer is uncached device memory where the runtime offers it (reduce_stage_alloc()).
// A launch enqueued past convergence (skip flag set; identical on all ranks) sums nothing; its first workgroup still publishes
// the pending status record.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "mailbox_device.hpp"

namespace spmv_amd {

constexpr int kReduceBlock = 256;
constexpr int kReduceStageBlocks = 256;

// The CG scalar step after the (all-reduced) r.r is known: stopping test (strict <, on ||r|| / ||r0||), iteration count
// including the converging iteration, beta, rr_old <- rr_new (reference cg_solver_mgpu_partitioned.cu:652-676,716), and the
// status record for the host.
// The stopping test of step k on the all-reduced r.r (one expression for the step and for every workgroup that derives the
// verdict by itself, cg_direction_kernel): strict <, on ||r|| / ||r0||.
__device__ __forceinline__ bool cg_converging(double rr_new, double b_norm, double tol, int stop_at, int k) {
    return sqrt(rr_new) / b_norm < tol || (stop_at > 0 && k == stop_at);
}
// stop_at: measurement hook of the LAB build only (kernels.hpp, CgScalars); the product library's test is the reference's alone
__device__ __forceinline__ int cg_stop_at(const CgScalars* s) {
#ifdef SPMV_AMD_LAB
    return s->stop_at;
#else
    (void)s;
    return 0;
#endif
}

__device__ __forceinline__ void cg_scalars_step(CgScalars* s, double tol, double* history, int* host_record,
                                                int sequence, double* alpha_ring, int ring_slots) {
    if (!s->converged) {
        s->alpha = s->rr_old / s->pAp;  // the alpha update_r used (same division), kept for the x update
        const double res = sqrt(s->rr_new);
        s->residual = res;
        s->iterations += 1;
        s->rr_ring[s->iterations & 1] = s->rr_new;
        if (alpha_ring != nullptr) alpha_ring[(s->iterations - 1) % ring_slots] = s->alpha;
        if (history != nullptr && s->iterations < s->max_history) history[s->iterations] = res;
        if (cg_converging(s->rr_new, s->b_norm, tol, cg_stop_at(s), s->iterations)) {
            s->converged = 1;
        } else {
            s->beta = s->rr_new / s->rr_old;
            s->rr_old = s->rr_new;
        }
    }
    if (host_record != nullptr) {
        // status record in host-coherent pinned memory: payload first, then the sequence number with
        // system-scope release, so a host that sees `sequence` sees this iteration's payload
        host_record[1] = s->converged;
        host_record[2] = s->iterations;
        __hip_atomic_store(&host_record[0], sequence, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The CG scalar step, optional, in the tail of a reduction (scalars == nullptr: none).
struct StepArgs {
    CgScalars* scalars;
    double tol;
    double* history;
    int* host_record;
    int sequence;
    double* alpha_ring;
    int ring_slots;
};

// What happens to the finished sum.
//  * mailbox (may be null): the sum is completed ACROSS THE RANKS by the finishing workgroup's first wave (comm.hpp);
//  * step.scalars (may be null): the CG scalar step runs on the finished sum in the same launch;
//  * host_progress (may be null): an int in host-coherent pinned memory that receives progress_value once the local
//    sum is known -- the solver's watchdog reads it to say how far the GPU got when a rank stops making progress.
struct ReduceTail {
    double* out;
    const int* skip_flag;
    int* host_progress;
    int progress_value;
    const PeerMailbox* mailbox;
    StepArgs step;
};

// Slice sums, ticket and extra values of one reduction in flight (reduce_stage_alloc(): zeroed once; the ticket returns to
// zero at the end of every reduction).
constexpr int kReduceExtraMax = 1024;  // boundary-row partials a fused launch may add (2 grid rows of <= 46 340 columns: 726)
struct ReduceStage {
    unsigned long long* sums;   // kReduceStageBlocks doubles, as bits
    unsigned long long* extra;  // kReduceExtraMax doubles, as bits
    unsigned* ticket;
    unsigned* edges_ready;      // cg_direction_kernel: sequence, raised once every workgroup's share of the edge rows has reached memory
};

// scratch layout (kernels.hpp, ReduceScratch): [kReduceStageBlocks sums | kReduceExtraMax extras | ticket | edges_ready]
__host__ __device__ inline ReduceStage reduce_stage_of(double* base) {
    unsigned long long* b = reinterpret_cast<unsigned long long*>(base);
    unsigned long long* tail = b + kReduceStageBlocks + kReduceExtraMax;
    return ReduceStage{b, b + kReduceStageBlocks, reinterpret_cast<unsigned*>(tail), reinterpret_cast<unsigned*>(tail + 1)};
}
// Slice workgroups of a reduction over `count` partials: one up to 1024 partials, else ceil(count / slice) with
// slice = ceil(count / kReduceStageBlocks).
inline void reduce_geometry(int count, int* slice, int* blocks) {
    if (count <= 4 * kReduceBlock) {
        *slice = count;
        *blocks = 1;
        return;
    }
    *slice = (count + kReduceStageBlocks - 1) / kReduceStageBlocks;
    *blocks = (count + *slice - 1) / *slice;
}

// 256-wide tree over s[] (every thread of the workgroup holds its value in `acc`); the sum is in s[0] afterwards.
__device__ __forceinline__ void block_tree(double acc, double* __restrict__ s) {
    s[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int stride = kReduceBlock / 2; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride) s[threadIdx.x] += s[threadIdx.x + stride];
        __syncthreads();
    }
}

// Thread t's share of partials[lo, hi): elements lo + t, lo + t + 256, ... added in ascending order, sixteen loads in flight
// (4e8 rows: 48 partials per thread; eight in flight took 12.3 us per reduction, profiles/r05_bench_kernel_stats.csv).
__device__ __forceinline__ double strided_sum(const double* __restrict__ partials, int lo, int hi) {
    constexpr int kInFlight = 16;
    double acc = 0.0;
    for (int base = lo + (int)threadIdx.x; base < hi; base += kInFlight * kReduceBlock) {
        double v[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const int i = base + u * kReduceBlock;
            v[u] = i < hi ? partials[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u)
            if (base + u * kReduceBlock < hi) acc += v[u];
    }
    return acc;
}

// Second stage + tail, run by ALL threads of one workgroup: the sum of `count` stage values read by `load(i)`, then the
// first wave finishes (mailbox, out, scalar step).
template <class Load>
__device__ __forceinline__ void reduce_finish(int count, Load&& load, double* __restrict__ s, const ReduceTail& tail) {
    double acc = 0.0;
    for (int i = (int)threadIdx.x; i < count; i += kReduceBlock) acc += load(i);
    block_tree(acc, s);
    if (threadIdx.x >= 64) return;  // the first wave finishes
    double total = s[0];
    if (threadIdx.x == 0 && tail.host_progress != nullptr)
        __hip_atomic_store(tail.host_progress, tail.progress_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tail.mailbox != nullptr) total = mailbox_allreduce_wave(*tail.mailbox, total);
    if (threadIdx.x == 0) {
        *tail.out = total;
        if (tail.step.scalars != nullptr)
            cg_scalars_step(tail.step.scalars, tail.step.tol, tail.step.history, tail.step.host_record, tail.step.sequence,
                            tail.step.alpha_ring, tail.step.ring_slots);
    }
}

// A launch enqueued past convergence: nothing is summed, the pending status record is still published (by `publisher`).
__device__ __forceinline__ void reduce_skipped(const ReduceTail& tail, bool publisher) {
    if (publisher && tail.step.scalars != nullptr && threadIdx.x == 0)
        cg_scalars_step(tail.step.scalars, tail.step.tol, tail.step.history, tail.step.host_record, tail.step.sequence,
                        tail.step.alpha_ring, tail.step.ring_slots);
}

// Publishes one value of this reduction (a slice sum, a boundary tile's partial): an agent-scope relaxed atomic store, then
// the wait for its acknowledgement. See the header for why there is no fence here.
__device__ __forceinline__ void publish(unsigned long long* slot, double value) {
    __hip_atomic_store(slot, (unsigned long long)__double_as_longlong(value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the store has been acknowledged by the memory side
}
__device__ __forceinline__ double published(const unsigned long long* slot) {
    return __longlong_as_double((long long)__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// A value ANOTHER AGENT may have written (a peer GPU's store over xGMI, a DMA engine, the RCCL receive kernel on behalf of a
// peer): system scope, so that no cache level of this device may answer from a line it held before the writer's data arrived.
__device__ __forceinline__ double published_by_any_agent(const unsigned long long* slot) {
    return __longlong_as_double((long long)__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

// Called by ALL threads of a workgroup once everything it contributes has been published (by any of its threads, each
// followed by publish()'s wait). Thread 0 draws a ticket; the workgroup that draws the last of `tickets` sums the `blocks`
// slice sums, then the `extra_count` values at `extra`, and runs the tail. s: kReduceBlock doubles of LDS, s_last: one int.
__device__ __forceinline__ void draw_ticket_and_finish_if_last(const ReduceStage& stage, int blocks, const unsigned long long* extra,
                                                               int extra_count, int tickets, const ReduceTail& tail,
                                                               double* __restrict__ s, int* __restrict__ s_last) {
    __syncthreads();  // every wave's publish() has returned
    if (threadIdx.x == 0) {
        const unsigned drawn = __hip_atomic_fetch_add(stage.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = drawn == (unsigned)(tickets - 1) ? 1 : 0;
        if (*s_last) __hip_atomic_store(stage.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next reduction on this stream
    }
    __syncthreads();
    if (*s_last == 0) return;
    reduce_finish(blocks + extra_count, [&](int i) { return i < blocks ? published(stage.sums + i) : published(extra + (i - blocks)); }, s, tail);
}

// One slice workgroup: slot `slot` of `blocks` (slice = ceil(count / blocks) partials each).
__device__ __forceinline__ void reduce_slice_block(const double* __restrict__ partials, int count, int slice, int slot, int blocks,
                                                   const unsigned long long* extra, int extra_count, int tickets, const ReduceStage& stage,
                                                   const ReduceTail& tail, double* __restrict__ s, int* __restrict__ s_last) {
    const int lo = slot * slice;
    const int hi = min(lo + slice, count);
    block_tree(strided_sum(partials, lo, hi), s);
    if (threadIdx.x == 0) publish(stage.sums + slot, s[0]);
    draw_ticket_and_finish_if_last(stage, blocks, extra, extra_count, tickets, tail, s, s_last);
}

}  // namespace spmv_amd
