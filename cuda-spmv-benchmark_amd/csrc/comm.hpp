// comm.hpp -- the communicator the multi-GPU CG runs over (the role MPI_COMM_WORLD plays in
// reference src/solvers/cg_solver_mgpu_partitioned.cu). One process per GPU.
//
// Two transports behind one interface:
//  * RcclComm   -- device-native: neighbour halo rows by ncclSend/ncclRecv (one group per
//                  exchange, xGMI point-to-point links), dot products by ncclAllReduce on device
//                  scalars. Two RCCL communicators are created so that the halo exchange (side
//                  stream) and the all-reduce (compute stream) never queue behind one another.
//  * StagedComm -- the reference's own scheme (D2H -> host exchange -> H2D,
//                  cg_solver_mgpu_partitioned.cu:173-231) with the host exchange supplied by the
//                  caller as callbacks (e.g. an MPI or a gloo binding). Used where RCCL cannot
//                  run (several ranks sharing one GPU in tests).
//  * no communicator / world == 1 -- SelfComm, every collective is the identity.
#pragma once

#include <hip/hip_runtime.h>
#include <stdio.h>

#include "spmv_amd.h"

// ---- peer mailbox: the dot products' all-reduce as plain stores over xGMI ----
// An 8-byte all-reduce is pure latency: an RCCL launch per reduction costs tens of microseconds on a loop whose
// whole iteration is ~1 ms at 8 GPUs, and there are two per iteration. Each rank therefore owns a small mailbox in
// uncached device memory, mapped into every other rank's address space (hipIpc over the node's xGMI fabric). To
// all-reduce, one wave writes {value, sequence} into its slot of EVERY rank's mailbox (system-scope stores, the
// sequence number released after the value), waits until its own mailbox holds this sequence number from every rank,
// and adds the values in rank order -- so all ranks obtain the same bits, independent of arrival order. Two slot sets
// alternate by sequence parity: a rank can only be one all-reduce ahead of the slowest reader, never two.
// Every wait is bounded (timeout_ticks of the 100 MHz wall clock); a timeout raises *host_error and the solver ends.
// Optional: a communicator without a (working) mailbox all-reduces through its transport (ncclAllReduce / staged).
constexpr int kMailboxMaxRanks = 16;
struct MailboxSlot {
    unsigned long long value_bits;  // the double, as bits: written and read with 8-byte system-scope atomics
    unsigned long long seq;
};
struct PeerMailbox {
    MailboxSlot* inbox;                          // this rank's mailbox: [2][world] slots
    MailboxSlot* peer_inbox[kMailboxMaxRanks];   // every rank's mailbox as mapped into this process (own one included)
    unsigned long long* seq;                     // all-reduces completed so far (device memory, advanced by the kernels)
    int* host_error;                             // pinned host memory: non-zero once a wait has timed out
    int rank, world;
    long long timeout_ticks;
};

struct SpmvAmdComm {
    int rank = 0;
    int world = 1;
    // device copy of the connected mailbox, or null (see above; mailbox.hip)
    PeerMailbox* d_mailbox = nullptr;
    struct MailboxHost* mailbox_host = nullptr;  // owner of the allocations behind d_mailbox
    bool mailbox_ready() const { return d_mailbox != nullptr; }
    // Test hook, LAB build only (SPMV_AMD_FORCE_COLLECTIVES=1): issue the all-reduces even with one rank, so that a
    // 1-GPU box drives the RCCL calls of the CG loop; a 1-rank all-reduce is the identity.
    bool force_collectives = false;
    bool collective() const { return world > 1 || force_collectives; }
    // Test hook, LAB build only (SPMV_AMD_SELF_NEIGHBOUR=1, one RCCL rank): the rank is its own previous and next
    // neighbour. The solver then runs the whole multi-rank pipeline -- halo-carrying buffers, send / recv on the side
    // stream under the interior SpMV, the arrival flag, split SpMV launches -- on one GPU. On the WHOLE grid the rows that
    // would read the halos are the first and last grid row of the global grid, which have no north / south entry: the
    // received values are never used and the solve must reproduce the plain single-rank result. On a stand-in slab
    // (spmv_amd_cg_slab_create_stencil5_as) they ARE used: the slab mirrored at its cuts, checked against the oracle.
    bool self_neighbour = false;
    bool exchanges_halos() const { return world > 1 || self_neighbour; }
    // What the first slab created on this communicator found when it solved a few iterations in both loop shapes
    // (cg_slab.hip, verify_pipeline): 0 = not checked yet, 1 = the overlapped pipeline reproduced the plain order's residual
    // history bit for bit on every rank, -1 = it did not on some rank: every slab on this communicator runs the plain order.
    int pipeline_verdict = 0;
    virtual ~SpmvAmdComm() {}
    // Exchanges `count` doubles with rank-1 (send_prev/recv_prev) and rank+1 (send_next/recv_next);
    // pointers are device pointers, NULL where the slab has no neighbour on that side (the pointers, not the
    // rank, decide: a self-neighbour probe can stand in for any rank of a larger job). Ordered on `stream`.
    virtual void halo_exchange(const double* d_send_prev, const double* d_send_next,
                               double* d_recv_prev, double* d_recv_next, int count,
                               hipStream_t stream) = 0;
    // In-place sum over all ranks of `count` device doubles, ordered on `stream`.
    virtual void allreduce_sum(double* d_buf, int count, hipStream_t stream) = 0;
    // Every rank contributes n_local device doubles; rank 0 receives them in h_full at displs[r].
    virtual void gather_to_root(const double* d_local, int n_local, double* h_full,
                                const int* counts, const int* displs) = 0;
    // All ranks meet here. `stream` is the stream the caller's device work is ordered on (the solver's compute
    // stream): a device-side transport enqueues its barrier there, so that each RCCL communicator is only ever
    // driven from one stream, and returns once that stream has drained.
    virtual void barrier(hipStream_t stream) = 0;
    virtual const char* transport() const = 0;
    // Ranks the device transport itself reports (ncclCommCount of both communicators, 0 if they disagree or the
    // transport has no such notion): lets a benchmark line prove how many devices RCCL really spans.
    virtual int transport_ranks() const { return 0; }
    // One or two lines for the watchdog report (e.g. ncclCommGetAsyncError of each communicator). Must not block.
    virtual void describe(FILE*) const {}
    // Self-test aid: moves `count` device doubles from d_send to d_recv through the transport's own
    // point-to-point path with this rank as its own peer (RCCL: ncclSend + ncclRecv to self in one group),
    // so a one-GPU box runs the send / recv calls the halo exchange is made of. False = not supported.
    virtual bool loopback(const double*, double*, int, hipStream_t) { return false; }
};

namespace spmv_amd {
// In-place sum of d_value over the ranks through the mailbox (comm->mailbox_ready() must hold); one tiny launch.
void launch_mailbox_allreduce(const SpmvAmdComm* comm, double* d_value, hipStream_t stream);
// Aborts with a message if a mailbox wait has timed out since the last call (cheap: reads one pinned int).
void mailbox_check(const SpmvAmdComm* comm);
void mailbox_release(SpmvAmdComm* comm);
// The communicator cg_solve_mgpu_partitioned uses; never NULL (SelfComm by default).
SpmvAmdComm* world_comm();
SpmvAmdComm* self_comm();
}  // namespace spmv_amd
