// mailbox_device.hpp -- device side of the peer-mailbox all-reduce (comm.hpp): callable from any kernel whose
// first wave has the locally reduced value, so a reduction kernel can finish its sum across the ranks in its own tail.
#pragma once

#include <hip/hip_runtime.h>

#include "comm.hpp"

namespace spmv_amd {

// Called by ALL 64 lanes of one wave with the same `mine`; returns the sum over the ranks (same bits on every
// rank: contributions are added in rank order). Exactly one wave per rank may be inside at a time; calls are
// ordered by the stream(s) the caller uses, identically on every rank.
__device__ __forceinline__ double mailbox_allreduce_wave(const PeerMailbox& mb, double mine) {
    const int lane = (int)(threadIdx.x & 63);
    const int world = mb.world;
    const unsigned long long seq = *mb.seq + 1;  // wave-uniform
    const int set = (int)(seq & 1) * world;
    double got = 0.0;
    int late = 0;
    // once a wait has timed out the mailbox is dead: later calls return at once (the host ends the solve, or the
    // self-test reports failure, without paying the limit again per call)
    if (__hip_atomic_load(mb.host_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return mine;
    if (lane < world) {
        MailboxSlot* dst = mb.peer_inbox[lane] + set + mb.rank;
        __hip_atomic_store(&dst->value_bits, (unsigned long long)__double_as_longlong(mine), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        // release: the value is visible at system scope before the sequence number that announces it
        __hip_atomic_store(&dst->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        MailboxSlot* src = mb.inbox + set + lane;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&src->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (wall_clock64() - t0 > mb.timeout_ticks) {
                late = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        got = __longlong_as_double((long long)__hip_atomic_load(&src->value_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
    double sum = 0.0;
    for (int r = 0; r < world; ++r) sum += __shfl(got, r);  // rank order on every rank
    if (late) __hip_atomic_store(mb.host_error, lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lane == 0) *mb.seq = seq;
    return sum;
}

}  // namespace spmv_amd
