/* lab.h -- entry points of the LAB build only (cuda-spmv-benchmark_amd/lib/libspmv_amd_lab.so: the product's sources compiled
 * with -DSPMV_AMD_LAB). Test and measurement hooks: tests/, tools/ and bench.py's one-GPU scaling probe load that library;
 * the product library (libspmv_amd.so, include/spmv_amd/api.h) neither exports these symbols nor reads the environment
 * switches listed here -- the reference has no such switches either (include/spmv.h:46-64: errors exit, nothing else).
 *
 * Environment switches the LAB build reads in addition to the product's (each ONCE, at creation):
 *   SPMV_AMD_SELF_NEIGHBOUR=1, SPMV_AMD_FORCE_COLLECTIVES=1   a single RCCL rank as its own neighbour / issuing its all-reduces
 *   SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE=1 | 2 | 3 | 4     fault injection: the side-stream exchange never returns on the
 *                                                             host | the overlapped pipeline delivers a nudged system | the rows
 *                                                             of a side-stream exchange never travel | its arrival flag never comes
 *   SPMV_AMD_PLACEMENT_FAIL_AFTER=k                           the k-th further placement candidate "does not fit"
 */
#ifndef SPMV_AMD_LAB_H
#define SPMV_AMD_LAB_H

#include "spmv_amd/api.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Stand-in slab: the slab rank `as_rank` of an `as_world`-GPU run would own (same rows, CSR bytes, halo length and
 * neighbour sides; reference partition, cg_solver_mgpu_partitioned.cu:259-268,306-331), carried by ONE rank whose
 * communicator was created under SPMV_AMD_SELF_NEIGHBOUR=1 and exchanges the halo rows with itself through the transport's
 * own send / recv path: the previous-rank halo receives the slab's own first grid row, the next-rank halo its last (the slab
 * mirrored at its cuts, not the global system). Timing of the real per-rank shapes on one GPU, and the oracle test of the
 * RCCL hand-overs (tests/test_distributed.py). */
SpmvAmdCgSlab* spmv_amd_cg_slab_create_stencil5_as(int n, int as_rank, int as_world, SpmvAmdComm* comm);

/* Options of an existing slab (A/B runs on the same allocations): "no_overlap" 0/1 (1 = the PLAIN loop shape: halo exchange on
 * the compute stream behind the whole direction update), "late_bulk" 0 / 1 / 2 (off / the lead-status-rest protocol in EVERY
 * iteration / only where the known residual says convergence is near: the default rule on slabs of >= 1e8 rows), "lead_rows" N, "run_ahead" 0 / 1 / 2
 * (1, the default: the host runs one iteration ahead of the status records while the known residual is far from the tolerance;
 * 2: test hook, every iteration is guessed "far", so each solve learns of its convergence one iteration late) --
 * results are bit-identical under each --, "spmv_event_stride" N (time every N-th in-loop SpMV launch; default 7, phase advancing with every solve; 0 = none),
 * and the one option that is NOT result-neutral, a timing aid for stand-in slabs: "stop_at" K (iteration K counts as the
 * converging one whatever its residual; 0 = off). Returns 0, or -1 for an unknown name. */
int spmv_amd_cg_slab_set_option(SpmvAmdCgSlab* s, const char* name, long long value);

#ifdef __cplusplus
}
#endif

#endif /* SPMV_AMD_LAB_H */
