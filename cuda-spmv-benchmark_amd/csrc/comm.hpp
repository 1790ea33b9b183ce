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

struct SpmvAmdComm {
    int rank = 0;
    int world = 1;
    // Test hook (SPMV_AMD_FORCE_COLLECTIVES=1): issue the all-reduces even with one rank, so that a
    // 1-GPU box drives the RCCL calls of the CG loop; a 1-rank all-reduce is the identity.
    bool force_collectives = false;
    bool collective() const { return world > 1 || force_collectives; }
    // Test hook (SPMV_AMD_SELF_NEIGHBOUR=1, one RCCL rank): the rank is its own previous and next neighbour.
    // The solver then runs the whole multi-rank pipeline -- halo-carrying buffers, send / recv on the side
    // stream under the interior SpMV, event waits, split SpMV launches -- on one GPU. The rows that would
    // read the halos are the first and last grid row of the GLOBAL grid, which have no north / south entry,
    // so the received values are never used and the solve must reproduce the plain single-rank result.
    bool self_neighbour = false;
    bool exchanges_halos() const { return world > 1 || self_neighbour; }
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
// The communicator cg_solve_mgpu_partitioned uses; never NULL (SelfComm by default).
SpmvAmdComm* world_comm();
SpmvAmdComm* self_comm();
}  // namespace spmv_amd
