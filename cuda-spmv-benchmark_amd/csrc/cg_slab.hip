// cg_slab.hip -- Conjugate Gradient on a 1-D row slab per GPU: the MI355X counterpart of
// reference src/solvers/cg_solver_mgpu_partitioned.cu:236-908 (the binary behind every
// published CG number, "1 GPU" included).
//
// Same partition (n / P rows, last rank takes the remainder, :261-268), same local CSR
// (row_ptr rebased, global col_idx, :306-329), same algebra and stopping rule (:450-716), same
// timed region (after setup + barrier, through the loop, before the gather, :405-413 -> :728-731).
//
// What is organised differently, for the hardware:
//  * the direction vector p lives in ONE allocation [prev halo | local rows | next halo], so the
//    +-grid_size neighbours of the first/last grid row are ordinary addresses and the slab kernel
//    is the single-GPU row-lds kernel with a row offset; RCCL receives straight into the halos;
//  * the first SpMV of a solve writes r0 = b - A x0, p0 and the r0.r0 partials itself; every later SpMV is fused
//    with the p.Ap partial sums; r -= a Ap carries the r.r partials; the direction update
//    p' = r + b p is written out of place into a ring of direction buffers and x = x0 + sum a_k p_k is
//    evaluated by one flush pass per ring length (deferred x update; the in-place form, where x += a p
//    rides with the direction update, remains for short rings);
//    152 -> 113 bytes per row per iteration, element-wise results unchanged;
//  * consecutive streaming kernels sweep the vectors in alternating directions (Infinity-Cache reuse);
//  * scalars (alpha, beta, the norms, the convergence flag, the iteration counter) stay in HBM;
//    kernels of iterations enqueued past convergence see the flag and return, so the host reads
//    one 16-byte record per iteration while the GPU is already busy with the next SpMV; the scalar step -- and,
//    with a peer mailbox (comm.hpp), the all-reduce across the ranks -- runs in the tail of the reduction's
//    single-block last stage, so a multi-rank iteration issues the launches of a single-rank one;
//  * the halo exchange of iteration k+1 runs on a side stream under the interior rows' SpMV; the
//    first and last grid row of the slab are launched once the halo has landed. Each row is
//    computed by the same code whichever launch it falls in, so overlap cannot change results;
//  * (round 3) the direction update runs on the slab's first / last grid row FIRST and the exchange starts behind them:
//    the RCCL send / recv kernel, which otherwise competes for CUs with a SpMV that fills the chip and ends after
//    it, is over long before the interior rows are (early halo);
//  * every wait on another rank is bounded (watchdog.hpp): a wedged peer becomes a report and a non-zero exit;
//  * (round 4) WHERE the vectors lie is part of the design: on MI355X kernels that walk several vectors in lock step lose 6.5 %
//    when the vectors lie in different classes of 32 GiB address regions, and only hipMalloc decides the class. r, Ap and the
//    direction ring are therefore carved out of ONE allocation (vector arena) and the coefficient stream is placed by timing
//    three allocations one region apart (place_coefficients). All of it at creation, outside the timed region. (The class-aware
//    allocator on HIP's virtual-memory API of round 4 -- -0.7 % in a clean process, 17.8 s of set-up for a second slab -- was
//    removed in round 5: profiles/r04_class_pool_*.txt, history up to commit 258dcef.);
//  * (round 4) the direction update of an iteration is enqueued as a lead piece + -- once the status record says the loop goes
//    on -- the rest (late bulk, slabs of >= 1e8 rows): the converging iteration no longer dispatches 3 M workgroups that only
//    read a flag;
//  * (round 5) no event and no cross-stream wait inside an iteration: an event record is a barrier packet (~6 us), a small
//    launch 7-10 us. Every dot product is ONE launch (the workgroup that finishes last sums the slice sums); a slab's boundary
//    rows ride in the launch that sums p.Ap and wait for the halo's device-side arrival flag themselves; the direction update
//    is ONE launch whose first workgroups write the rows the neighbours need through to memory and raise the flag the side
//    stream's exchange waits for -- on the RCCL path it also takes the scalar step, and nobody waits for it (every workgroup
//    derives beta and the verdict from scalars the step does not write). Five launches per iteration on a slab with neighbours
//    (+ two ncclAllReduce).
//  * (round 6) TWO shapes of the loop, chosen once per solve (LoopShape): the PIPELINE above, and the PLAIN order -- halo
//    exchange on the compute stream behind the whole direction update, the reference's own (:680-703) -- for detailed timers,
//    SPMV_AMD_NO_OVERLAP=1 (bench.py's fallback), the in-place form and slabs too thin to split. The A/B switches that kept the
//    shapes of rounds 2-5 alive (two-launch reductions, event-ordered hand-overs, the direction update in three launches, no
//    sweep alternation) are gone with their code; both shapes give the same bits (tests/test_cg_gpu.py). Test and measurement
//    hooks (stand-in slabs, stop_at, loop options, fault injection) exist in the LAB build only (-DSPMV_AMD_LAB,
//    lib/libspmv_amd_lab.so): the product library has no switch that can change a result.
//  * (round 6) the host off the critical path where the residual allows: while the known residual is more than 16 x the tolerance
//    away the host neither uses the late bulk's lead / status / rest protocol nor waits for the iteration's status record before
//    it enqueues the next iteration (it runs ONE iteration ahead; the rule reads the residual history only, so every rank takes
//    the same decision). A container that is being CPU-throttled answers late (profiles/r06_throttle_probe.txt).
//
// The same loop also serves the reference's SINGLE-GPU entry point, cg_solve_device (cg_solver.cu:436-706): a slab that
// borrows the caller's SpmvOperator instead of owning a CSR (cg_solve_on_operator, near the end of this file), and it can
// record a per-stage timeline of a solve with HIP events and no host syncs (spmv_amd_cg_slab_set_timeline): what the
// reference's six MAX / MIN-reduced timers are (:748-800), without serialising the pipeline to take them.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "comm.hpp"
#include "device_runtime.hpp"
#include "stencil_geometry.hpp"
#include "trace_ranges.hpp"
#include "watchdog.hpp"

using namespace spmv_amd;

namespace spmv_amd {
std::vector<double>& last_cg_history();
}

struct SpmvAmdCgSlab {
    SpmvAmdComm* comm = nullptr;
    // Which slab of how many this is. Equal to the communicator's rank / world, except for a stand-in slab
    // (spmv_amd_cg_slab_create_stencil5_as): one self-neighbour rank carrying the slab of rank `part_rank` of a
    // `part_world`-GPU job, so that one GPU can time the real per-rank slab shapes.
    int part_rank = 0, part_world = 1;
    // Borrowed operator (cg_solve_device, reference src/solvers/cg_solver.cu:436-706): the matrix stays with the caller's
    // SpmvOperator and every SpMV goes through its vtable's run_device -- or, when the operator is this library's own
    // stencil5-csr, through its fused launch (FusedSpmv: p.Ap partials / initial residual written by the SpMV itself).
    // Such a slab is the whole matrix on one rank, has no halos, and runs on the DEFAULT stream, where run_device enqueues.
    SpmvOperator* op = nullptr;
    bool op_failed = false;  // the borrowed operator's run_device returned non-zero: the solve stops enqueuing and returns 1
    FusedSpmv fused;
    const char* label = nullptr;  // verbose prefix of the reference entry point that owns the solve ("CG-DEVICE")
    bool device_form = false;     // direction update rounded as update_p_kernel does (cg_solver.cu:90-95), see cg_kernels.hip
    int n = 0, grid = -1, row_offset = 0, n_local = 0, halo = 0;
    bool has_prev = false, has_next = false;
    DeviceCsr A;
    double *x = nullptr, *x0 = nullptr, *r = nullptr, *Ap = nullptr, *b = nullptr;
    double* x0_alloc = nullptr;  // x0 carries halo rows too ([pad | prev halo | local | next halo]): the initial SpMV reads it in place
    // Vector arena (round 4): r, Ap and every direction buffer are carved out of ONE allocation, in that order, r and Ap
    // first. On MI355X a kernel that walks two or three vectors at the same index runs 6.5 % slower when the vectors lie in
    // different 32 GiB regions of the physical address space (regions of the same class repeat every 96 GiB: 288 GiB = 3 classes
    // x 3; all pairs of 40 vectors in one 143 GB allocation: profiles/r04_arena_probe40.txt) -- the price of alternating
    // between two stack-level ranks on the same channels, presumably. Separate hipMallocs land in a class at random (the fast /
    // slow lottery behind the "+-1.3 % between processes" of rounds 2-3: profiles/r04_placement_*.txt, r04_offset_probe.txt);
    // neighbours inside one allocation share a region except where it crosses a boundary. The SpMV's coefficient stream does
    // not take part (40 B/row against 8 B/row: not in lock step; r04_arena_spmv_mix.txt), so the CSR arrays stay where they are.
    double* vec_arena = nullptr;
    // place_coefficients: {0, candidates timed, SpMV ms before, SpMV ms kept}
    std::vector<double> placement;
    std::vector<double> tile_runs;  // tune_tile_runs: {rule, kept, SpMV ms with the rule, ms kept}; empty = did not run
    // wall ms of the set-up phases of creation, each closed by a device synchronisation (spmv_amd_cg_slab_setup_ms):
    // matrix to HBM (upload or generation), streams + vectors, verification + launch plans, coefficient placement, tile runs
    double setup_ms[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    // false for the slab a one-shot entry point creates, solves on once and destroys (cg_solve_mgpu_partitioned): the two timed
    // trials of creation cost 0.05-5 s (the placement trial allocates and frees two spacers of up to 28 GiB) and can win
    // back 2 % of ONE solve at most; a slab the caller keeps (spmv_amd_cg_slab_create*) runs them
    bool setup_trials = true;
    double* p_alloc = nullptr;  // [pad | prev halo | local | next halo]
    double* p = nullptr;        // local part of the CURRENT direction vector, 16-byte aligned
    // Direction ring (deferred x update). With ring_slots > 1 the direction update is written out of place into
    // the next of ring_slots halo-carrying buffers, and x = x0 + sum alpha_k p_k is evaluated by one flush pass per
    // ring_slots iterations (and at the end) instead of a read-modify-write of x in every iteration: the loop
    // moves 56 + 24 + 24 B/row plus 8 B/row for the deferred re-read of p_k (+ 16/ring_slots for x) instead of
    // 56 + 24 + 40. The fma chain per element of x is the same, in the same order. Costs ring_slots - 1 extra
    // vectors of HBM (48 GB at 400 M rows and 16 slots, of 288 GB); ring_slots = 1 is the in-place form.
    size_t slot_doubles = 0, slot_lead = 0;  // size of one halo-carrying buffer and where its local part starts
    std::vector<double*> ring_alloc;  // allocations, ring_alloc[0] == p_alloc
    std::vector<double*> ring;        // local parts
    int ring_slots = 1;
    double* d_alpha_ring = nullptr;
    double* partials_spmv = nullptr;  // dot partials of the SpMV launches (interior, head, tail back to back)
    double* partials_blas = nullptr;
    double* reduce_stage = nullptr;  // scratch of this slab's reductions (kernels.hpp, ReduceScratch)
    // One launch per dot product (round 5, reduce_device.hpp); the boundary rows of a split SpMV ride in the launch that
    // reduces the SpMV's partials (launch_stencil5_edges_and_reduce).
    ReduceScratch scratch() const { return ReduceScratch{reduce_stage}; }
    CgScalars* d_s = nullptr;
    double* d_hist = nullptr;
    int hist_cap = 0;
    // pinned, host-coherent. progress = 4 * sequence + k, written by the last block of the local reductions of the
    // iteration that will publish `sequence`: k = 1 local p.Ap summed, k = 2 local r.r summed (watchdog report only)
    // halo_late: set by a boundary wave that gave up waiting for the halo's arrival flag (kernels.hpp, HaloArrival)
    struct Poll { int sequence; int converged; int iterations; int progress; int halo_late; }* h_poll = nullptr;
    // Device-side arrival flag of the halo exchange (round 5): behind every exchange on the SIDE stream a one-thread launch
    // raises *d_halo_flag to that exchange's sequence number, and the boundary waves of the launch that needs the halo rows
    // wait for it themselves -- no cross-stream event wait (a barrier packet, ~10 us in front of the launch it guards:
    // profiles/r05_ab_reduce_one_launch.txt, 17 us idle against 7) on the path between a SpMV and its dot product.
    unsigned* d_halo_flag = nullptr;
    unsigned halo_sequence = 0;
#ifdef SPMV_AMD_LAB
    // Measurement hook, set_option("stop_at", k): iteration k counts as the converging one whatever its residual (kernels.hpp,
    // CgScalars::stop_at). A stand-in slab's mirrored system does not converge in 14 iterations; with max_iters alone it would run
    // one direction update + halo exchange more than the rank of a real job, whose 14th iteration converges. Timing only.
    int stop_at = 0;
    int test_wedge_overlapped_exchange = 0;  // SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE=1 / 2, see exchange_halo
    bool guess_far_always = false;  // set_option("run_ahead", 2): every iteration is guessed "cannot converge": each solve learns of its convergence one iteration late
#endif
    // non-null while a solve runs on a communicator with a working peer mailbox: the last stage of every dot
    // product then completes the sum across the ranks itself (no all-reduce launch)
    const PeerMailbox* reduce_mailbox = nullptr;
    int* spmv_progress = nullptr;  // where the p.Ap reduction of the SpMV being enqueued reports (in-loop SpMVs only)
    int spmv_progress_value = 0;
    const char* enqueued_stage = "";  // the last piece of work the host put on the streams
    int enqueued_iteration = -1;
    hipStream_t compute = nullptr, side = nullptr;
    bool owns_streams = true;
    hipEvent_t ev_p_ready = nullptr, ev_halo_done = nullptr;
    // Timeline of a solve (spmv_amd_cg_slab_set_timeline): events at the stage boundaries of every iteration, recorded
    // without any host sync and resolved after the loop. kTimelineMarks per iteration on the compute stream, two on the
    // side stream around the halo exchange.
    bool timeline_on = false;
    std::vector<hipEvent_t> tl_compute, tl_side;
    hipEvent_t tl_after_interior = nullptr;  // where slab_spmv marks the end of the interior launch (split SpMV)
    std::vector<double> timeline_us;         // result of the last timeline solve, order = kTimelineNames
    LaunchShape shape;
    // launch plans, made once at creation: the whole slab, the rows that need no halo, the first / last grid row
    Stencil5Plan plan_whole, plan_interior, plan_head, plan_tail;
    int partials_cap = 0;
    int spmv_split_at = -1;  // set by slab_spmv: partials written by the interior launch of a split SpMV (-1: one launch)
    const char* variant_name = "";
    bool fused_dot = false;
    bool fuse_init_residual = false;  // r0 = b - A x0, p0, r0.r0 written by the first SpMV's launches (row-lds slabs)
    std::vector<double> history;
    // event pairs around every spmv_event_stride-th in-loop SpMV (recorded without any host sync,
    // resolved after the loop): time_spmv_ms with timers off is the live average of those launches
    // times the iteration count. Each pair puts two barrier packets (~7 us each, profiles/r02_slab_timeline.txt) next to the
    // launch it times, so every 7th launch is timed and the phase moves on by one with every solve: a run of solves covers
    // every iteration of the loop (later iterations work on other ring slots and are up to 2 % faster or slower).
    // (LAB build: set_option("spmv_event_stride", 1) times every launch, 0 none.)
    std::vector<hipEvent_t> spmv_ev;
    int spmv_event_stride = 7;
    int spmv_event_phase = 0;  // solves so far
    // Sweep-direction alternation: consecutive streaming kernels of the loop walk the vectors in opposite
    // directions, so each starts on the addresses its predecessor touched last and finds part of them in
    // the 256 MiB Infinity Cache. Results and partial slots are independent of the direction. Measured on
    // MI355X, whole solve: 50 M rows (the per-GPU slab of an 8-GPU run) 15.90 -> 15.60 ms, SpMV launches
    // 0.520 -> 0.496 ms; 100 M rows -0.5 %, 200 M rows -0.8 %, 400 M rows unchanged. Making the producers'
    // stores / consumers' loads plain instead of nontemporal did not raise the hit share.
    bool roctx_always = false;  // SPMV_AMD_ROCTX=1: roctx ranges even without detailed timers
    bool no_overlap = false;  // SPMV_AMD_NO_OVERLAP=1: the PLAIN loop shape (halo exchange on the compute stream: the reference's; bench.py's fallback)
    // verify_pipeline's two short solves: waits on the other stream's flags give up after wait_limit_s instead of 20 s, and a
    // wait that gave up ends that solve (selfcheck_late) instead of the process
    bool selfcheck = false, selfcheck_late = false;
    double wait_limit_s = 0.0;  // 0 = the default bound (halo_wait_limit_s)
    // Late bulk (round 4): the direction update of iteration k is enqueued before the host knows whether k converged, and on
    // the converging iteration that launch only reads a flag -- 3.1 M one-wave workgroups at 4e8 rows, 0.65 ms of pure
    // dispatch per solve. On large slabs the host enqueues a LEAD piece of lead_rows rows (long enough to cover one host
    // wake-up and a launch), reads the status record, and enqueues the rest only if the loop goes on: the reference tests
    // convergence before its p update too (cg_solver_mgpu_partitioned.cu:652-676). Same kernel over disjoint row ranges:
    // same bits. Measured at 4e8 rows on one slab, settings alternated between solves: 108.04 -> 107.34 ms per solve, -0.65 %
    // (profiles/r04_ab_late_bulk.txt). Small slabs (a whole update is ~0.2 ms at 5e7 rows) keep the single launch. Ring mode only.
    bool late_bulk = false;
    // The lead / status / rest protocol puts the host on the critical path with only the lead piece to hide behind, so it is
    // used only where it can pay: in iterations that MAY converge, judged by the residual the host already knows (the previous
    // iteration's, in the host-coherent history): still more than 16 x the tolerance away -> this iteration will not converge
    // (a drop of 16 x in one iteration would be needed) and the whole update is enqueued at once, as on small slabs. A wrong
    // guess costs the empty dispatch once, never a result: the kernels test the flag themselves. On the 20 000^2 solve the
    // protocol runs in iterations 11-14 of 14. (LAB build: set_option("late_bulk", 1) forces the protocol in every iteration.)
    bool late_predict = true;
    // The host runs one iteration ahead of the status records while convergence is far (SolveRun::read_status): an iteration's
    // worth of work is always queued, so a host that answers late does not idle the GPU. (LAB build: set_option("run_ahead", 0).)
    bool run_ahead = true;
    // 2^25 rows = ~125 us of streaming: the host's read + launch take ~20 us on a quiet box, but a container that is being CPU-
    // throttled answers later (tools/throttle_probe.sh: the direction stage is where the host sits on the critical path). On one
    // slab, settings alternated (profiles/r06_ab_lead_rows.txt): 2^22 ... 2^26 rows all within 0.1 % (104.08-104.19 ms per solve
    // at 4e8 rows), late bulk off 104.67, 2^27 104.69.
    size_t lead_rows = (size_t)1 << 25;
    int poll_sequence = 0;
    double last_spmv_ms = 0.0;
    int last_spmv_launches = 0;
    std::vector<float> last_spmv_each;  // the timed in-loop launches of the last solve, in iteration order
};

namespace {
// marks per iteration on the compute stream: iteration start | interior SpMV enqueued-and-done | boundary rows |
// p.Ap sum (+ all-reduce) | r update | r.r sum (+ all-reduce) + scalar step | direction update
constexpr int kTimelineMarks = 7;
const char* const kTimelineNames =
    "iterations,solve_ms,initial_residual_us,spmv_interior_us,halo_wait_and_boundary_rows_us,reduce_pAp_and_allreduce_us,"
    "update_r_us,reduce_rr_allreduce_and_scalar_step_us,direction_update_us,gap_before_next_iteration_us,iteration_us,"
    "halo_exchange_on_side_stream_us,final_x_flush_us,direction_updates";
}  // namespace

namespace {

// (Re)binds a borrowed-operator slab to `op`: asks for the operator's fused launch and sizes the partial buffer for it.
// Called at creation and at the start of every cg_solve_device (the operator may have been re-initialised in between).
void adopt_operator(SpmvAmdCgSlab* s, SpmvOperator* op) {
    s->op = op;
    s->fused = fused_spmv_of(op);
    s->fused_dot = s->fused.partials > 0;
    s->fuse_init_residual = s->fused_dot && s->fused.can_init;
    s->variant_name = s->fused_dot ? "operator/fused-launch" : "operator/run_device+dot";
    if (s->fused.partials > s->partials_cap) {
        device_release(s->partials_spmv);
        s->partials_cap = s->fused.partials;
        s->partials_spmv = device_alloc<double>((size_t)s->partials_cap);
        HIP_CHECK(hipMemset(s->partials_spmv, 0, (size_t)s->partials_cap * sizeof(double)));
    }
}

void place_coefficients(SpmvAmdCgSlab* s);  // below
void tune_tile_runs(SpmvAmdCgSlab* s);

// Set-up phase clock: wall time since `from`, after everything enqueued so far has finished.
double setup_phase_ms(std::chrono::steady_clock::time_point& from) {
    HIP_CHECK(hipDeviceSynchronize());
    const auto now = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(now - from).count();
    from = now;
    return ms;
}

void make_common(SpmvAmdCgSlab* s) {
    const size_t nl = (size_t)s->n_local;
    auto phase = std::chrono::steady_clock::now();
    // A self-neighbour rank that owns the WHOLE grid (part_world == 1) keeps halos on both sides: the rows that would
    // read them are the first and last grid row of the global grid, which have no north / south entry, so the
    // full pipeline runs and the solve must still reproduce the plain one (tests/test_distributed.py).
    const bool whole_grid_probe = s->comm->self_neighbour && s->part_world == 1;
    s->has_prev = s->comm->exchanges_halos() && (s->part_rank > 0 || whole_grid_probe);
    s->has_next = s->comm->exchanges_halos() && (s->part_rank < s->part_world - 1 || whole_grid_probe);
    s->halo = s->comm->exchanges_halos() ? s->grid : 0;
    s->A.view.halo_before = s->has_prev ? s->halo : 0;
    s->A.view.halo_after = s->has_next ? s->halo : 0;
    if (s->op != nullptr) {
        // borrowed operator: its run_device enqueues on the default stream (reference spmv_stencil_csr_direct.cu:267-271),
        // so the whole solve runs there; one rank, no halo exchange, no side stream
        s->compute = nullptr;
        s->side = nullptr;
        s->owns_streams = false;
    } else {
        HIP_CHECK(hipStreamCreateWithFlags(&s->compute, hipStreamNonBlocking));
        // The side stream has the DEFAULT priority. A highest-priority side stream (so that the halo exchange
        // gets CUs at once while the interior SpMV saturates the chip) was measured with the rank as its own
        // neighbour (SPMV_AMD_SELF_NEIGHBOUR=1, 50 M rows; profiles/r01_multirank_pipeline.txt): every kernel of the
        // normal-priority compute stream slows down while such a queue exists -- SpMV 0.50 -> 0.66 ms, 5 us kernels ->
        // 50 us, a solve 15.7 -> 25.3 ms -- whether the exchange is RCCL send/recv or a plain copy. At equal priority
        // the exchange kernel starts 50-90 us into the interior SpMV and ends long before it (rocprofv3 trace).
        HIP_CHECK(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
    }
    HIP_CHECK(hipEventCreateWithFlags(&s->ev_p_ready, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&s->ev_halo_done, hipEventDisableTiming));
    // The local part of every halo-carrying buffer starts on a 4 KiB boundary whatever the halo length, like the
    // plain allocations of r, Ap, x. Measured with the rank as its own neighbour: a halo of 14 142 doubles put
    // every access of the direction buffers across two 128-byte lines (direction update 0.88 ms against 0.71 ms
    // for the r update at 200 M rows); line-aligned but not 4 KiB-aligned local parts (halo 20 000) still cost
    // 4 % in the direction update (1.52 vs 1.46 ms at 400 M rows, 110.7 vs 109.1 ms per solve).
    constexpr size_t kLeadUnit = 512;  // doubles
    const size_t lead = s->halo == 0 ? 0 : ((size_t)s->halo + kLeadUnit - 1) / kLeadUnit * kLeadUnit;
    const size_t slot_doubles = lead + nl + (size_t)s->halo + 2;
    s->slot_doubles = slot_doubles;
    s->slot_lead = lead;
    s->x = device_alloc<double>(nl);
    s->b = device_alloc<double>(nl);
    s->x0_alloc = device_alloc<double>(slot_doubles);
    s->x0 = s->x0_alloc + lead;
    HIP_CHECK(hipMemset(s->x0_alloc, 0, slot_doubles * sizeof(double)));
    {
        // as many direction buffers as fit comfortably (default 16, SPMV_AMD_P_RING=1 keeps the in-place update)
        int want = kMaxRingSlots;
        const char* forced = getenv("SPMV_AMD_P_RING");
        if (forced) want = atoi(forced);
        want = want < 1 ? 1 : (want > kMaxRingSlots ? kMaxRingSlots : want);
        // One vector every `pitch` doubles: the vector rounded up to 2 MiB, plus 4 KiB -- consecutive vectors then differ in
        // their phase inside the memory system's interleave as well (~1 % on both BLAS1 kernels against a pitch of whole
        // 2 MiB units, profiles/r04_arena_probe.txt). r and Ap are slots 0 and 1, the direction buffers follow.
        constexpr size_t k2MiB = (size_t)2 << 20, k4KiB = 4096;
        const size_t pitch = ((slot_doubles * sizeof(double) + k2MiB - 1) / k2MiB * k2MiB + k4KiB) / sizeof(double);
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        const size_t keep_free = (size_t)4 << 30;  // leave room for the caller's own buffers
        const size_t per_slot = pitch * sizeof(double);
        const int asked = want;
        // A slab that owns its matrix may take what is free. The workspace of cg_solve_device outlives the call (until an
        // operator's free() or spmv_amd_cg_release_workspace()) next to a caller who goes on allocating -- a second operator,
        // say -- so its ring is held to a quarter of what is free now: 16 slots of 3.2 GB at 4e8 rows on an otherwise idle
        // MI355X, fewer on a fuller device, the in-place form when even four do not fit.
        const size_t fixed = 3 * per_slot;  // r, Ap, the first direction buffer
        const size_t budget = s->op != nullptr ? free_b / 4 : (free_b > keep_free + fixed ? free_b - keep_free - fixed : 0);
        while (want > 1 && (size_t)(want - 1) * per_slot > budget) --want;
        if (want < asked && want < 4) want = 1;  // a ring cut short by memory flushes too often to pay
        s->vec_arena = device_alloc<double>((size_t)(2 + want) * pitch);
        HIP_CHECK(hipMemset(s->vec_arena, 0, (size_t)(2 + want) * pitch * sizeof(double)));
        s->r = s->vec_arena;
        s->Ap = s->vec_arena + pitch;
        s->ring_alloc.clear();
        s->ring.clear();
        for (int k = 0; k < want; ++k) {
            double* a = s->vec_arena + (size_t)(2 + k) * pitch;
            s->ring_alloc.push_back(a);
            s->ring.push_back(a + lead);
        }
        s->p_alloc = s->ring_alloc[0];
        s->p = s->ring[0];
        s->ring_slots = want;
        s->d_alpha_ring = device_alloc<double>(kMaxRingSlots);
        HIP_CHECK(hipMemset(s->d_alpha_ring, 0, kMaxRingSlots * sizeof(double)));
    }
    s->shape = current_launch_shape();
    if (const char* v = getenv("SPMV_AMD_NO_OVERLAP")) s->no_overlap = v[0] == '1';
#ifdef SPMV_AMD_LAB
    if (const char* v = getenv("SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE")) s->test_wedge_overlapped_exchange = atoi(v);
#endif
    if (const char* v = getenv("SPMV_AMD_ROCTX")) s->roctx_always = v[0] == '1';
    s->late_bulk = nl >= 100000000;
    s->partials_blas = device_alloc<double>(dot_scratch_doubles(nl));
    HIP_CHECK(hipMemset(s->partials_blas, 0, dot_scratch_doubles(nl) * sizeof(double)));
    s->reduce_stage = reduce_scratch_alloc();
    s->d_halo_flag = device_alloc<unsigned>(1);
    HIP_CHECK(hipMemset(s->d_halo_flag, 0, sizeof(unsigned)));
    s->d_s = device_alloc<CgScalars>(1);
    HIP_CHECK(hipMemset(s->d_s, 0, sizeof(CgScalars)));
    HIP_CHECK(hipHostMalloc((void**)&s->h_poll, sizeof(*s->h_poll), hipHostMallocCoherent | hipHostMallocMapped));
    memset(s->h_poll, 0, sizeof(*s->h_poll));
    if (s->op != nullptr) {
        adopt_operator(s, s->op);
        launch_fill(s->b, nl, 1.0, s->compute);
        launch_fill(s->x0, nl, 0.0, s->compute);
        HIP_CHECK(hipDeviceSynchronize());
        return;
    }
    s->setup_ms[1] = setup_phase_ms(phase);
    s->A.verify_stencil(s->compute);
    {
        // dot partials: one slot per launched wave; the launch geometry is a fixed function of the
        // slab and the row range, so size for the larger of the two ways a SpMV is issued
        const SlabCsr& A = s->A.view;
        const int lo = s->has_prev ? s->halo : 0, hi = s->n_local - (s->has_next ? s->halo : 0);
        const auto plan = [&](int a, int b) { return plan_stencil5(A, a, b > a ? b : a, Stencil5Variant::Auto, s->shape); };
        s->plan_whole = plan(0, s->n_local);
        s->plan_interior = plan(lo, hi);
        s->plan_head = plan(0, lo);
        s->plan_tail = plan(hi, s->n_local);
        const int whole = s->plan_whole.partials;
        const int split = (hi > lo ? s->plan_interior.partials : 0) + (lo > 0 ? s->plan_head.partials : 0) +
                          (hi < s->n_local ? s->plan_tail.partials : 0);
        s->partials_cap = whole > split ? whole : split;
        s->partials_spmv = device_alloc<double>((size_t)s->partials_cap);
        HIP_CHECK(hipMemset(s->partials_spmv, 0, (size_t)s->partials_cap * sizeof(double)));
        s->variant_name = s->plan_whole.name;
        // unverified / unaligned slabs run the row-generic kernel and use the plain dot kernel
        s->fused_dot = s->plan_whole.variant != Stencil5Variant::RowGeneric;
        // the initial residual rides in the first SpMV where every launch of the slab is a row-lds launch
        const auto rowlds = [&](const Stencil5Plan& p) { return p.last_row <= p.first_row || p.variant == Stencil5Variant::RowLds; };
        s->fuse_init_residual = s->plan_whole.variant == Stencil5Variant::RowLds && rowlds(s->plan_interior) && rowlds(s->plan_head) && rowlds(s->plan_tail);
    }
    s->setup_ms[2] = setup_phase_ms(phase);
    if (s->setup_trials) place_coefficients(s);
    s->setup_ms[3] = setup_phase_ms(phase);
    if (s->setup_trials) tune_tile_runs(s);
    launch_fill(s->b, nl, 1.0, s->compute);   // default right-hand side b = 1
    launch_fill(s->x0, nl, 0.0, s->compute);  // default initial guess x0 = 0
    s->setup_ms[4] = setup_phase_ms(phase);
}

// The coefficient stream's place (round 4). The in-loop SpMV reads the coefficients V and a direction buffer x and writes Ap; by
// class of address regions (device_runtime.hpp; whole solves with every vector bound to a slot of a chosen class:
// profiles/r04_loop_regions.txt): Ap in a class of its own 3.55 ms, Ap with x 3.70 ms, Ap with V but not with x 3.88 ms -- and
// a solve between 103.3 and 109.4 ms. In the vector arena Ap shares its region with the first direction buffers and not with the
// later ones, so the one thing left to chance is whether V lies in Ap's class: if it does, every iteration on a later buffer runs
// in the slow mode (in-loop average 3.75-3.80 ms instead of 3.60-3.65: the process that is 2 % slow). So the coefficients are
// offered up to two MORE allocations one region apart, each timed on the SpMV from an early and from a late direction buffer,
// and the fastest is kept. Set-up work (the reference builds and uploads its CSR before its timed region,
// cg_solver_mgpu_partitioned.cu:303-413); the values are copied, never changed. Slabs of >= 16 Mi rows that own their matrix.
// The trial is optional in every respect: a candidate the device cannot provide ends it with the array as it was, and the
// copy that loses is freed (DeviceCsr::separate_values).
bool wants_coefficient_placement(size_t n_local) { return n_local >= ((size_t)16 << 20) && placement_candidates() > 1; }

void place_coefficients(SpmvAmdCgSlab* s) {
    const size_t nl = (size_t)s->n_local;
    if (s->op != nullptr || !wants_coefficient_placement(nl) || s->ring.size() < 4 || s->A.values == nullptr) return;
    hipStream_t q = s->compute;
    const size_t count = (size_t)s->A.view.nnz_local;
    const double* early = s->ring[1];
    const double* late = s->ring[s->ring.size() - 3];
    launch_fill(s->ring[1], nl, 1.0, q);
    launch_fill(s->ring[s->ring.size() - 3], nl, 1.0, q);
    HIP_CHECK(hipMemsetAsync(&s->d_s->converged, 0, sizeof(int), q));
    EventTimer timer;
    double* const original = s->A.values;
    auto cost = [&](double* values) {
        s->A.view.values = values;
        double total = 0.0;
        for (const double* x : {early, late}) {
            float ms[3];
            for (int i = 0; i < 4; ++i) {
                timer.begin(q);
                (void)launch_stencil5_spmv(s->A.view, s->plan_whole, x, s->Ap, 1.0, s->fused_dot ? s->partials_spmv : nullptr, nullptr, false, q);
                timer.end(q);
                const float t = timer.elapsed_ms();
                if (i > 0) ms[i - 1] = t;
            }
            std::sort(ms, ms + 3);
            total += ms[1];
        }
        s->A.view.values = original;
        return 0.5 * total;
    };
    int tried = 0;
    double before = 0.0, after = 0.0;
    double* best = device_alloc_best_of<double>(count, 0, [&](double* cand) {
        if (cand != original) HIP_CHECK(hipMemcpyAsync(cand, original, count * sizeof(double), hipMemcpyDeviceToDevice, q));
        const double ms = cost(cand);
        if (cand == original) before = ms;
        return ms;
    }, &tried, nullptr, original, /*release_first=*/false);
    after = best == original ? before : cost(best);
    HIP_CHECK(hipStreamSynchronize(q));
    if (best != original) {
        if (after < 0.99 * before) s->A.replace_values(best);  // frees `original` when it is an allocation of its own
        else device_release(best);
    }
    s->A.view.values = s->A.values;
    for (double* a : s->ring_alloc) HIP_CHECK(hipMemsetAsync(a, 0, s->slot_doubles * sizeof(double), q));
    HIP_CHECK(hipStreamSynchronize(q));
    s->placement = {0.0, (double)tried, before, s->A.values == original ? before : after};
}

// Row-lds tiles per XCD and run, by measurement on the slab's own vectors (device_runtime.hpp, tune_rowlds_xcd_run): the four
// launch plans are then re-made with the run length kept. Their partial counts do not depend on it.
void tune_tile_runs(SpmvAmdCgSlab* s) {
    if (s->op != nullptr || s->ring.size() < 2) return;
    const size_t nl = (size_t)s->n_local;
    hipStream_t q = s->compute;
    double* x = s->ring[1];
    launch_fill(x, nl, 1.0, q);
    HIP_CHECK(hipMemsetAsync(&s->d_s->converged, 0, sizeof(int), q));
    double rec[4] = {0, 0, 0, 0};
    const int run = tune_rowlds_xcd_run(s->A.view, s->shape, x, s->Ap, s->fused_dot ? s->partials_spmv : nullptr, q, rec);
    HIP_CHECK(hipMemsetAsync(s->ring_alloc[1], 0, s->slot_doubles * sizeof(double), q));
    HIP_CHECK(hipStreamSynchronize(q));
    if (run <= 0) return;
    s->shape.knobs.rowlds_group = run;
    const int lo = s->has_prev ? s->halo : 0, hi = s->n_local - (s->has_next ? s->halo : 0);
    const auto plan = [&](int a, int b) { return plan_stencil5(s->A.view, a, b > a ? b : a, Stencil5Variant::Auto, s->shape); };
    s->plan_whole = plan(0, s->n_local);
    s->plan_interior = plan(lo, hi);
    s->plan_head = plan(0, lo);
    s->plan_tail = plan(hi, s->n_local);
    s->tile_runs.assign(rec, rec + 4);
}

// In-kernel / side-stream waits for the other stream's flags give up well inside the host's watchdog.
double halo_wait_limit_s(const SpmvAmdCgSlab* s) {
    if (s->wait_limit_s > 0.0) return s->wait_limit_s;
    const double w = watchdog_limit_seconds();
    return w > 0.0 && w < 40.0 ? 0.5 * w : 20.0;
}

bool partition_ok(const SpmvAmdComm* comm, int n, int grid, int n_local) {
    if (!comm->exchanges_halos()) return true;
    if (grid <= 0) {
        fprintf(stderr, "[cg-slab] multi-GPU solve needs a stencil matrix (grid_size > 0)\n");
        return false;
    }
    if (n_local < grid) {
        fprintf(stderr, "[cg-slab] slab of %d rows is thinner than one grid row (%d)\n", n_local, grid);
        return false;
    }
    (void)n;
    return true;
}

// The slab's first and last grid row (the rows that read the halos) on `stream`; partial slots follow the
// interior launch's. Returns the number of partials written (0 without partials).
int slab_boundary_spmv(SpmvAmdCgSlab* s, const double* in, double* part, const int* skip, hipStream_t stream,
                       const ResidualOut* init = nullptr) {
    const SlabCsr& A = s->A.view;
    const int lo = s->has_prev ? s->halo : 0;
    const int hi = s->n_local - (s->has_next ? s->halo : 0);
    double* at = part ? part + (hi > lo ? s->plan_interior.partials : 0) : nullptr;
    int used = 0;
    if (lo > 0 && hi < s->n_local && lo == A.grid_size && s->n_local - hi == A.grid_size) {
        // a rank with two neighbours: its first and last grid row in one launch
        used += launch_stencil5_spmv_first_and_last_gridrow(A, s->plan_head, s->plan_tail, in, s->Ap, 1.0, at, skip, stream, init);
    } else {
        if (lo > 0) used += launch_stencil5_spmv(A, s->plan_head, in, s->Ap, 1.0, at, skip, false, stream, init);
        if (hi < s->n_local)
            used += launch_stencil5_spmv(A, s->plan_tail, in, s->Ap, 1.0, at ? at + used : nullptr, skip, false, stream, init);
    }
    return part ? used : 0;
}

// Sum of the partials the last slab_spmv wrote (`used` of them): a split SpMV's boundary-row partials as extra values.
void reduce_spmv_partials(SpmvAmdCgSlab* s, int used, double* d_out, const int* skip, int* progress, int progress_value, const PeerMailbox* mailbox) {
    const int first = s->spmv_split_at >= 0 && s->spmv_split_at < used ? s->spmv_split_at : used;
    launch_reduce_partials(s->partials_spmv, first, d_out, skip, s->compute, s->scratch(), progress, progress_value, mailbox,
                           s->partials_spmv + first, used - first);
}

// SpMV of the slab on p (halos must be current or in flight on the side stream).
// overlap = the halo exchange was started on the side stream and ev_halo_done marks its end.
// spmv_done (optional): recorded behind the last SpMV launch, before the reduction of its partials.
// init (may be null): the launches write r = b - A x, p = r and r.r partials instead of A x (fused initial residual);
// the caller reduces the partials. Returns the number of partial slots the launches wrote.
int slab_spmv(SpmvAmdCgSlab* s, bool with_dot, bool overlap, const int* skip,
              const double* input = nullptr, hipEvent_t spmv_done = nullptr, const ResidualOut* init = nullptr) {
    const SlabCsr& A = s->A.view;
    const double* in = input ? input : s->p;
    double* part = ((with_dot && s->fused_dot) || init) ? s->partials_spmv : nullptr;
    const int lo = s->has_prev ? s->halo : 0;
    const int hi = s->n_local - (s->has_next ? s->halo : 0);
    int used = 0;
    s->spmv_split_at = -1;
    if (s->op != nullptr) {
        // the caller's operator: its fused launch when it has one (this library's stencil5-csr), else the vtable
        if (part != nullptr) {
            used = s->fused.launch(in, s->Ap, part, skip, s->shape.reverse, init, s->compute);
        } else if (s->op->run_device(in, s->Ap) != 0) {
            // The reference ignores this return value (cg_solver.cu:498,541); a library must not end its caller's process
            // over it either: remember it, let the solve wind down and hand the failure back as a status.
            fprintf(stderr, "[cg] operator '%s': run_device failed\n", s->op->name);
            s->op_failed = true;
        }
        if (s->tl_after_interior) HIP_CHECK(hipEventRecord(s->tl_after_interior, s->compute));
    } else if (hi <= lo || (lo == 0 && hi == s->n_local) || (part == nullptr && !overlap)) {
        // one launch: a slab without halo rows, or of one or two grid rows, or a plain y = A x with the halos already in place
        if (overlap) HIP_CHECK(hipStreamWaitEvent(s->compute, s->ev_halo_done, 0));
        used = launch_stencil5_spmv(A, s->plan_whole, in, s->Ap, 1.0, part, skip, s->shape.reverse, s->compute, init);
        if (s->tl_after_interior) HIP_CHECK(hipEventRecord(s->tl_after_interior, s->compute));
    } else {
        // rows whose north and south neighbours are local run under the halo exchange; the first / last grid
        // row of the slab once the halo rows have landed. (Launching those two rows behind the exchange on the
        // side stream instead, so that this stream only waits for an event, measured slower: 15.77 vs 15.59 ms
        // per solve at 50 M rows with the rank as its own neighbour.) A launch that writes dot partials is split in this way
        // even when the exchange is NOT overlapped (detailed timers, SPMV_AMD_NO_OVERLAP): the sum's shape -- slices of the
        // interior partials, then [slice sums | boundary rows' partials] -- must not depend on how the halos travelled.
        // (The wait for the halo rows in FRONT of the interior launch -- the exchange then overlaps only the direction update
        // it was started under, and the cross-stream wait leaves the path between the SpMV and its dot product -- measured
        // slower on every stand-in slab: +0.1-0.2 % at 20 000^2, +2.4 % on the P = 8 slab of 10 000^2,
        // profiles/r05_ab_halo_wait_first.txt.)
        used = launch_stencil5_spmv(A, s->plan_interior, in, s->Ap, 1.0, part, skip, s->shape.reverse, s->compute, init);
        s->spmv_split_at = used;  // the boundary rows' partials follow: they enter the sum as extra values (reduce_device.hpp)
        if (s->tl_after_interior) HIP_CHECK(hipEventRecord(s->tl_after_interior, s->compute));
        // in-loop SpMV: the boundary rows ride in the launch that reduces the partials (one launch instead of three); the
        // launch timer and the timeline's boundary-row mark then stop behind the interior rows (all but one or two grid rows
        // of the slab), and the stage "reduce_pAp" holds the halo wait, the boundary rows and the sum
        const bool fused_tail = with_dot && part != nullptr && init == nullptr && lo % A.grid_size == 0 &&
                                (s->n_local - hi) % A.grid_size == 0 && lo <= A.grid_size && s->n_local - hi <= A.grid_size;
        if (fused_tail && spmv_done) HIP_CHECK(hipEventRecord(spmv_done, s->compute));
        HaloArrival arrival;
        if (fused_tail && overlap) {
            // the boundary waves wait for the exchange's arrival flag themselves; bounded well inside the host's watchdog
            arrival = HaloArrival{s->d_halo_flag, s->halo_sequence, (long long)(halo_wait_limit_s(s) * 1e8), &s->h_poll->halo_late};
        } else if (overlap) {
            HIP_CHECK(hipStreamWaitEvent(s->compute, s->ev_halo_done, 0));
        }
        if (fused_tail && launch_stencil5_edges_and_reduce(A, s->plan_interior, lo > 0, hi < s->n_local, in, s->Ap, 1.0, part, &s->d_s->pAp, skip,
                                                           s->scratch(), s->spmv_progress, s->spmv_progress_value, s->reduce_mailbox, s->compute,
                                                           arrival)) {
            return used;
        }
        if (arrival.flag != nullptr) HIP_CHECK(hipStreamWaitEvent(s->compute, s->ev_halo_done, 0));  // the fused launch did not apply
        if (fused_tail) spmv_done = nullptr;  // already recorded
        used += slab_boundary_spmv(s, in, part, skip, s->compute, init);
    }
    if (spmv_done) HIP_CHECK(hipEventRecord(spmv_done, s->compute));
    if (with_dot) {
        if (part)
            reduce_spmv_partials(s, used, &s->d_s->pAp, skip, s->spmv_progress, s->spmv_progress_value, s->reduce_mailbox);
        else
            launch_dot((size_t)s->n_local, s->p, s->Ap, s->partials_blas, &s->d_s->pAp, s->compute);
    }
    return used;
}

const char* query_name(hipError_t e) {
    return e == hipSuccess ? "idle (all work done)" : e == hipErrorNotReady ? "busy (work pending)" : hipGetErrorString(e);
}

// What the watchdog prints about a slab whose rank stopped making progress. Runs on the watchdog's thread;
// queries only.
void report_slab_state(void* user, FILE* out) {
    const SpmvAmdCgSlab* s = static_cast<const SpmvAmdCgSlab*>(user);
    const int seq = __atomic_load_n(&s->h_poll->sequence, __ATOMIC_ACQUIRE);
    const int prog = __atomic_load_n(&s->h_poll->progress, __ATOMIC_ACQUIRE);
    fprintf(out, "[cg-slab] rank %d (slab %d of %d, rows [%d, %d)): last work enqueued by the host: %s of iteration %d\n",
            s->comm->rank, s->part_rank, s->part_world, s->row_offset, s->row_offset + s->n_local, s->enqueued_stage,
            s->enqueued_iteration);
    fprintf(out, "[cg-slab] status records: waiting for #%d, GPU has published #%d (iterations counted %d, converged %d)\n",
            s->poll_sequence, seq, s->h_poll->iterations, s->h_poll->converged);
    const char* where = "before the local p.Ap sum: in the SpMV, or waiting for the halo rows (see the events below)";
    if (prog == 4 * s->poll_sequence + 1)
        where = "local p.Ap summed: in the all-reduce of p.Ap, the r update or the local r.r sum";
    else if (prog == 4 * s->poll_sequence + 2)
        where = "local r.r summed: in the all-reduce of r.r or the scalar step";
    if (seq != s->poll_sequence) fprintf(out, "[cg-slab] GPU progress inside that iteration: %s\n", where);
    fprintf(out, "[cg-slab] compute stream: %s; side (halo) stream: %s\n", query_name(hipStreamQuery(s->compute)),
            query_name(hipStreamQuery(s->side)));
    fprintf(out, "[cg-slab] event 'direction vector ready for the halo exchange': %s; event 'halo rows received': %s\n",
            query_name(hipEventQuery(s->ev_p_ready)), query_name(hipEventQuery(s->ev_halo_done)));
    s->comm->describe(out);
}

// Has the scalar step that publishes record `sequence` run? (Records carry increasing sequence numbers, one per iteration, over
// the life of the slab; a later record implies the earlier ones.)
bool record_arrived(const SpmvAmdCgSlab* s, int sequence) {
    return (int)(__atomic_load_n(&s->h_poll->sequence, __ATOMIC_ACQUIRE) - sequence) >= 0;
}

// Blocks the host until record `sequence` has been published. Bounded by the watchdog (SPMV_AMD_WATCHDOG_S): a rank whose
// peers never answer ends with a report, not a hang.
void wait_for_record(SpmvAmdCgSlab* s, int sequence) {
    if (record_arrived(s, sequence)) return;
    WatchdogScope guard("waiting for the iteration's status record", s->comm->rank, s->enqueued_iteration,
                        report_slab_state, s);
    long spins = 0;
    while (!record_arrived(s, sequence)) {
        __builtin_ia32_pause();  // the record arrives within one iteration; the spinning core at least yields its pipeline
        if (++spins % (1L << 20) == 0) {
            // surface a faulted stream at once instead of waiting for the watchdog
            const hipError_t e = hipStreamQuery(s->compute);
            if (e != hipSuccess && e != hipErrorNotReady) HIP_CHECK(e);
            if (e == hipSuccess && !record_arrived(s, sequence)) {
                fprintf(stderr, "[cg-slab] status record missing after the stream drained\n");
                exit(EXIT_FAILURE);
            }
        }
    }
}

// Halo rows of a halo-carrying vector v (local part at v): first / last grid row to the neighbours, theirs
// into [v - halo, v) and [v + n_local, v + n_local + halo).
void exchange_halo(SpmvAmdCgSlab* s, double* v, hipStream_t stream) {
    if (!s->comm->exchanges_halos()) return;
    // a staged transport blocks in here on its host exchange, RCCL may block while it connects peers
    WatchdogScope guard("halo exchange (send/recv of the first and last grid row)", s->comm->rank, s->enqueued_iteration,
                        report_slab_state, s);
    // LAB build, SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE=1 (test hook): an exchange issued on the SIDE stream never returns -- the host
    // thread stays in here, as it would behind an RCCL call that blocks -- so that the watchdog ends the process and a
    // supervisor can be shown to restart the ranks without the overlap (tests/test_distributed.py). Nothing is wedged on the GPU.
    // = 2: the exchange returns, but a pipeline that overlaps delivers WRONG NUMBERS (the right-hand side is nudged once): what a
    // hand-over between the streams that loses rows between two devices would look like to the supervisor.
#ifdef SPMV_AMD_LAB
    while (s->test_wedge_overlapped_exchange == 1 && stream == s->side && s->side != nullptr) std::this_thread::sleep_for(std::chrono::milliseconds(200));
    if (s->test_wedge_overlapped_exchange == 2 && stream == s->side && s->side != nullptr) {
        static const double nudged = 1.001;
        HIP_CHECK(hipMemcpyAsync(s->b, &nudged, sizeof nudged, hipMemcpyHostToDevice, stream));
        s->test_wedge_overlapped_exchange = 0;
    }
    // = 3: the rows of a side-stream exchange never travel (the halos keep what they held) -- what the creation check
    // (verify_pipeline) must notice; = 4: the exchange runs but its arrival flag is never raised (SolveRun::start_halo)
    if (s->test_wedge_overlapped_exchange == 3 && stream == s->side && s->side != nullptr) return;
#endif
    s->comm->halo_exchange(s->has_prev ? v : nullptr, s->has_next ? v + (s->n_local - s->halo) : nullptr,
                           s->has_prev ? v - s->halo : nullptr, s->has_next ? v + s->n_local : nullptr, s->halo, stream);
}
void allreduce_scalar(SpmvAmdCgSlab* s, double* d_value, const char* stage) {
    if (s->comm->mailbox_ready()) {  // stores between the GPUs, one small launch, nothing for the host to wait on
        launch_mailbox_allreduce(s->comm, d_value, s->compute);
        return;
    }
    WatchdogScope guard(stage, s->comm->rank, s->enqueued_iteration, report_slab_state, s);
    s->comm->allreduce_sum(d_value, 1, s->compute);
}

}  // namespace

namespace {
struct LoopShape {
    bool multi = false;      // halo exchange needed
    bool reduce = false;     // all-reduce of the dot products needed
    bool separate = false;   // ... as a call of its own (RCCL / staged); false: inside the sums' last stage (peer mailbox) or not at all
    const PeerMailbox* mailbox = nullptr;
    bool detail = false;     // the reference's per-category timers: a host sync per stage, everything on the compute stream
    // PIPELINE: the exchange on the side stream under the interior rows, the boundary rows behind its arrival flag, and the rows
    // the neighbours wait for + the first piece of the rest + (RCCL path) the scalar step in ONE direction launch whose device
    // flag releases the exchange. false = PLAIN: everything on the compute stream, the exchange behind the whole direction update.
    bool pipeline = false;
    bool late = false;       // direction update as lead piece + rest (late bulk)
    int slots = 1;           // direction ring length; 1 = the in-place x / p update
    EdgeRows edges{0, 0, 0}; // pipeline: [0, count_a) and [second, second + count_b)
    size_t bulk_lo = 0, bulk_hi = 0;  // the rows of the direction update that are not edge rows
};


LoopShape loop_shape(const SpmvAmdCgSlab* s, const CGConfigMultiGPU* config) {
    LoopShape L;
    const SpmvAmdComm* comm = s->comm;
    const size_t nl = (size_t)s->n_local;
    L.multi = comm->exchanges_halos();
    L.reduce = comm->collective();
    L.mailbox = (L.reduce && comm->mailbox_ready()) ? comm->d_mailbox : nullptr;
    L.separate = L.reduce && L.mailbox == nullptr;
    L.detail = config->enable_detailed_timers != 0;
    L.slots = s->ring_slots;
    // (ring mode only: with the in-place form the x update of the converging iteration rides in that very launch)
    L.late = s->late_bulk && !L.detail && L.slots > 1;
    // Early halo (round 3): the two edge ranges are rounded OUTWARDS to 4 KiB (512 doubles), so that the launch over the rest
    // starts on a 4 KiB boundary like every whole-vector launch does: with the ranges cut exactly at the grid row, the rest of a
    // 15 000-column slab started 64 bytes off a 128-byte line and its direction update ran 20-30 % slower (485 vs 386 us at
    // 112.5 M rows; at 20 000 columns, 128-byte but not 4 KiB aligned, 8 %). A few rows beyond the grid row updated early is harmless.
    constexpr size_t kAlign = 512;
    const size_t head_rows = s->has_prev ? ((size_t)s->halo + kAlign - 1) / kAlign * kAlign : 0;
    const size_t tail_start = s->has_next ? (nl - (size_t)s->halo) / kAlign * kAlign : nl;
    // the pipeline needs the direction ring (its one-launch direction update writes out of place) and a slab thick enough to
    // have rows between its edge ranges; the in-place form and thinner slabs take the plain order
    L.pipeline = L.multi && !L.detail && !s->no_overlap && L.slots > 1 && s->reduce_stage != nullptr && (nl % 2) == 0 &&
                 nl >= 4 * (size_t)s->halo + 4 * kAlign && head_rows < tail_start;
    if (L.pipeline) {
        L.edges = EdgeRows{head_rows, tail_start, nl - tail_start};
        L.bulk_lo = head_rows, L.bulk_hi = tail_start;
    } else {
        L.bulk_lo = 0, L.bulk_hi = nl;
    }
    return L;
}

// Every halo row of every halo-carrying buffer (x0's, the direction ring's) set to NaN (all bits one). A correct loop receives
// each of them before it reads it. Repeated solves of one system write the SAME values into the same slots every time: rows
// that were lost, or read before they arrived, would otherwise be indistinguishable from rows that travelled.
void poison_halos(SpmvAmdCgSlab* s) {
    if (s->halo == 0 || s->op != nullptr) return;
    const size_t bytes = (size_t)s->halo * sizeof(double);
    auto both_sides = [&](double* local) {
        if (s->has_prev) HIP_CHECK(hipMemsetAsync(local - s->halo, 0xFF, bytes, s->compute));
        if (s->has_next) HIP_CHECK(hipMemsetAsync(local + s->n_local, 0xFF, bytes, s->compute));
    };
    both_sides(s->x0);
    for (double* local : s->ring) both_sides(local);
}

// The pipeline hands rows between its two streams through device flags and reads the received halo rows inside a running
// kernel -- constructions that were proven on ONE device (with the rank as its own neighbour) and cannot be proven between
// devices on a one-GPU box. So the FIRST slab created on a communicator that exchanges halos checks them where it runs:
// four iterations in the plain order (everything on the compute stream, no flag), four in the pipeline; the two shapes give
// the same bits by construction, so any difference in the residual history on any rank -- lost or stale halo rows (the halos
// are set to NaN before each of the two solves), a flag that never comes (the waits give up after 2 s here) -- refuses the
// pipeline for every slab on that communicator: they run the plain order, say so on stderr and in spmv_amd_cg_slab_loop_shape(). ~10 iterations' worth of set-up, once per
// communicator, outside every timed region (the library needed the check bench.py had).
void verify_pipeline(SpmvAmdCgSlab* s) {
    SpmvAmdComm* comm = s->comm;
    if (s->op != nullptr || !comm->exchanges_halos()) return;
    if (s->no_overlap) return;  // the caller asked for the plain order: nothing to verify, nothing recorded
    const CGConfigMultiGPU few = {4, 0.0, 0, 0};
    // Whether THIS rank's slab takes the pipeline is a local fact (an odd row count, a thin last slab, the in-place form make it
    // plain) while the check is collective: every rank of the communicator runs the two solves -- a plain-shaped rank solves
    // the plain order twice -- and the verdict counts only if at least one rank really ran the pipeline.
    const bool mine = loop_shape(s, &few).pipeline;
    if (comm->world == 1 && !mine) return;
    if (comm->pipeline_verdict == 0) {
        CGStatsMultiGPU st;
        s->selfcheck = true, s->selfcheck_late = false, s->wait_limit_s = 2.0;
        s->no_overlap = true;
        poison_halos(s);
        spmv_amd_cg_slab_solve(s, &few, &st);
        const std::vector<double> plain = s->history;
        s->no_overlap = false;
        poison_halos(s);  // what the plain solve left in the halos is exactly what the pipeline should receive: wipe it
        spmv_amd_cg_slab_solve(s, &few, &st);
        poison_halos(s);
        HIP_CHECK(hipStreamSynchronize(s->compute));
        if (s->side) HIP_CHECK(hipStreamSynchronize(s->side));
        // everything the check enqueued has run: a wait that gave up in an iteration the host had already enqueued ahead may have
        // raised the flag again after read_status cleared it
        __atomic_store_n(&s->h_poll->halo_late, 0, __ATOMIC_RELEASE);
        bool finite = true;  // NaN from a poisoned halo in the PLAIN order: the transport itself does not deliver the rows
        for (double v : plain) finite = finite && std::isfinite(v);
        if (!finite)
            fprintf(stderr, "[cg-slab] rank %d: the plain order's residual history is not finite in the creation check: the communicator's halo "
                            "exchange does not deliver this rank's neighbour rows (transport: %s)\n", comm->rank, comm->transport());
        const bool same = finite && !s->selfcheck_late && plain.size() == s->history.size() && plain.size() == 5 &&
                          memcmp(plain.data(), s->history.data(), plain.size() * sizeof(double)) == 0;
        s->selfcheck = false, s->selfcheck_late = false, s->wait_limit_s = 0.0;
        double votes[2] = {same ? 0.0 : 1.0, mine ? 1.0 : 0.0};  // {ranks whose histories differ, ranks that ran the pipeline}
        if (!same)
            fprintf(stderr, "[cg-slab] rank %d: the overlapped pipeline did NOT reproduce the plain order's residual history in the creation check\n", comm->rank);
        if (comm->world > 1) {  // every rank learns what every rank found
            double* d = device_alloc<double>(2);
            upload(d, votes, 2);
            WatchdogScope guard("creation check: all-reduce of the ranks' verdicts", comm->rank, -1, report_slab_state, s);
            comm->allreduce_sum(d, 2, s->compute);
            HIP_CHECK(hipStreamSynchronize(s->compute));
            download(votes, d, 2);
            device_release(d);
        }
        // no rank ran the pipeline (every slab thin or in place): nothing was verified, a later slab on this communicator checks again
        if (votes[0] != 0.0 || votes[1] != 0.0) comm->pipeline_verdict = votes[0] == 0.0 ? 1 : -1;
        if (comm->pipeline_verdict < 0 && comm->rank == 0)
            fprintf(stderr, "[cg-slab] %g rank(s) refused the overlapped pipeline: every solve on this communicator runs the plain order "
                            "(halo exchange on the compute stream, the reference's own)\n", votes[0]);
    }
    if (comm->pipeline_verdict < 0) s->no_overlap = true;
}

SpmvAmdCgSlab* create_from_matrix(MatrixData* mat, SpmvAmdComm* comm, bool setup_trials) {
    if (comm == nullptr) comm = self_comm();
    if (mat->rows != mat->cols) {
        fprintf(stderr, "[cg-slab] CG needs a square matrix\n");
        return nullptr;
    }
    int row_offset = 0, n_local = 0;
    spmv_amd_partition_rows(mat->rows, comm->world, comm->rank, &row_offset, &n_local);
    if (!partition_ok(comm, mat->rows, mat->grid_size, n_local)) return nullptr;
    // every rank builds the CSR of the whole matrix, then keeps its slab (reference :303-331)
    if (build_csr_struct(mat) != EXIT_SUCCESS) return nullptr;
    SpmvAmdCgSlab* s = new SpmvAmdCgSlab();
    s->comm = comm;
    s->part_rank = comm->rank;
    s->part_world = comm->world;
    s->n = mat->rows;
    s->grid = mat->grid_size;
    s->row_offset = row_offset;
    s->n_local = n_local;
    s->setup_trials = setup_trials;
    s->A.separate_values = setup_trials && wants_coefficient_placement((size_t)n_local);  // the copy that loses the placement trial can then be freed
    auto phase = std::chrono::steady_clock::now();
    s->A.upload_slab(csr_mat, row_offset, n_local, mat->grid_size);
    s->setup_ms[0] = setup_phase_ms(phase);
    make_common(s);
    verify_pipeline(s);
    return s;
}
}  // namespace

extern "C" SpmvAmdCgSlab* spmv_amd_cg_slab_create(MatrixData* mat, SpmvAmdComm* comm) { return create_from_matrix(mat, comm, true); }

// Which shape this slab's loop takes and who decided: "single rank", "pipeline", "plain: <why>". A static string.
extern "C" const char* spmv_amd_cg_slab_loop_shape(const SpmvAmdCgSlab* s) {
    if (!s->comm->exchanges_halos()) return "single rank";
    if (!s->no_overlap) {
        const CGConfigMultiGPU any = {1, 0.0, 0, 0};
        if (!loop_shape(s, &any).pipeline) return "plain: the in-place form (ring 1) or a slab too thin to split";
        return s->comm->pipeline_verdict > 0 ? "pipeline (verified against the plain order at creation)" : "pipeline";
    }
    return s->comm->pipeline_verdict < 0 ? "plain: the pipeline's residual history differed from the plain order's in the creation check"
                                         : "plain: SPMV_AMD_NO_OVERLAP=1";
}

namespace {
SpmvAmdCgSlab* create_stencil5_slab(int n, int part_rank, int part_world, SpmvAmdComm* comm) {
    if (n < 2 || (long long)n * n > 0x7fffffffLL || 5LL * n * n - 4LL * n > 0x7fffffffLL) {
        fprintf(stderr, "[cg-slab] grid %d does not fit 32-bit CSR indices\n", n);
        return nullptr;
    }
    int row_offset = 0, n_local = 0;
    spmv_amd_partition_rows(n * n, part_world, part_rank, &row_offset, &n_local);
    if (!partition_ok(comm, n * n, n, n_local)) return nullptr;
    SpmvAmdCgSlab* s = new SpmvAmdCgSlab();
    s->comm = comm;
    s->part_rank = part_rank;
    s->part_world = part_world;
    s->n = n * n;
    s->grid = n;
    s->row_offset = row_offset;
    s->n_local = n_local;
    s->A.separate_values = wants_coefficient_placement((size_t)n_local);
    auto phase = std::chrono::steady_clock::now();
    s->A.generate_stencil5(n, row_offset, n_local, 5.0, -1.0, nullptr);
    s->setup_ms[0] = setup_phase_ms(phase);
    make_common(s);
    verify_pipeline(s);
    return s;
}
}  // namespace

extern "C" SpmvAmdCgSlab* spmv_amd_cg_slab_create_stencil5(int n, SpmvAmdComm* comm) {
    if (comm == nullptr) comm = self_comm();
    return create_stencil5_slab(n, comm->rank, comm->world, comm);
}

#ifdef SPMV_AMD_LAB
// (LAB build only.) Stand-in slab for measurements on one GPU: the slab rank `as_rank` of an `as_world`-GPU job would own (same
// rows, same CSR bytes, same halo length, neighbours on the same sides), carried by a single-rank communicator
// created under SPMV_AMD_SELF_NEIGHBOUR=1, which exchanges the halo rows with itself through the transport's own
// send / recv path. The numbers it produces are those of a slab whose north / south neighbours are its own first / last grid
// row (the slab mirrored at both cuts; tests/test_distributed.py, stand_in_system, writes that matrix out and checks the solve
// against the oracle): a different linear system with the same work per iteration. Timing, and that test.
extern "C" SpmvAmdCgSlab* spmv_amd_cg_slab_create_stencil5_as(int n, int as_rank, int as_world, SpmvAmdComm* comm) {
    if (comm == nullptr || comm->world != 1 || (as_world > 1 && !comm->self_neighbour)) {
        fprintf(stderr, "[cg-slab] a stand-in slab needs a single-rank self-neighbour communicator\n");
        return nullptr;
    }
    if (as_world < 1 || as_rank < 0 || as_rank >= as_world) return nullptr;
    return create_stencil5_slab(n, as_rank, as_world, comm);
}
#endif  // SPMV_AMD_LAB

extern "C" int spmv_amd_cg_slab_set_vectors(SpmvAmdCgSlab* s, const double* b_full,
                                            const double* x0_full) {
    if (b_full) upload(s->b, b_full + s->row_offset, (size_t)s->n_local);
    if (x0_full) upload(s->x0, x0_full + s->row_offset, (size_t)s->n_local);
    return 0;
}

// ---------------------------------------------------------------------------------------
// One solve. The loop has TWO shapes, fixed per solve (LoopShape): the pipeline and the plain order; an iteration is the
// same five stages in both -- SpMV (+ p.Ap sum) | all-reduce | r update | r.r sum (+ all-reduce) + scalar step | direction
// update + halo exchange -- and each stage is one function below. What a stage enqueues depends on the shape only.
// ---------------------------------------------------------------------------------------
namespace {

// State of one solve in flight: what the stages share.
struct SolveRun {
    SpmvAmdCgSlab* s;
    const CGConfigMultiGPU* config;
    CGStatsMultiGPU* stats;
    const LoopShape L;
    const TraceRanges trace;
    EventTimer total, part;
    const size_t nl;
    const int* skip;
    const bool timeline;
    RingSlots ring_view;
    int window_start = 0;       // first iteration whose alpha_k p_k is not in x yet (ring mode)
    int tl_exchanges = 0;       // halo exchanges marked on the side stream (exchange j precedes the SpMV of iteration j)
    bool halo_in_flight = false;
    bool step_in_direction = false;  // this iteration's scalar step rides in the direction update's launch
    // Status records: the record of iteration j (1-based) of this solve carries sequence0 + j. records_read = the latest
    // iteration whose record the host has seen; the host runs at most ONE iteration ahead of it (read_status).
    int sequence0 = 0, records_read = 0;
    bool backward = false;      // sweep direction of this iteration's SpMV and direction update (the r update walks the other way)
    int enqueued = 0, sampled = 0;
    std::vector<int> sampled_iteration;  // 0-based loop index of each timed SpMV

    SolveRun(SpmvAmdCgSlab* slab, const CGConfigMultiGPU* cfg, CGStatsMultiGPU* st)
        : s(slab), config(cfg), stats(st), L(loop_shape(slab, cfg)), trace(L.detail || slab->roctx_always), nl((size_t)slab->n_local),
          skip(&slab->d_s->converged), timeline(slab->timeline_on && !L.detail) {
        for (int k = 0; k < kMaxRingSlots; ++k) ring_view.p[k] = s->ring[(size_t)k % s->ring.size()];
    }

    // detailed timers (the reference's, :770-800): a stage between two events and a host sync; otherwise just the stage
    template <class Work>
    void timed(double* bucket, double* bucket2, Work&& work) {
        if (!L.detail) {
            work();
            return;
        }
        part.begin(s->compute);
        work();
        part.end(s->compute);
        WatchdogScope guard("detailed timers: waiting for the stage just enqueued", s->comm->rank, s->enqueued_iteration, report_slab_state, s);
        const double ms = part.elapsed_ms();
        if (bucket) *bucket += ms;
        if (bucket2) *bucket2 += ms;
    }
    static hipEvent_t tl_event(std::vector<hipEvent_t>& pool, size_t index) {
        while (pool.size() <= index) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreate(&e));
            pool.push_back(e);
        }
        return pool[index];
    }
    // compute-stream marks: [0] solve start, [1] initial residual done, then kTimelineMarks per iteration, then the flush
    size_t mark_index(int iteration, int k) const { return 2 + (size_t)iteration * kTimelineMarks + k; }
    void mark(int iteration, int k) {
        if (timeline) HIP_CHECK(hipEventRecord(tl_event(s->tl_compute, mark_index(iteration, k)), s->compute));
    }

    void begin();
    void initial_residual();
    void start_halo(double* v, bool released_by_flag);
    void stage_spmv();
    void stage_allreduce_pAp();
    void stage_update_r();
    void stage_sum_rr_and_step();
    void stage_direction_and_halo();
    bool far_from_convergence(int iteration, int basis) const;
    bool converged_at(int k) const;
    void await_record(int iteration);
    bool read_status();
    void finish();
    void resolve_timeline(const CgScalars& fin, float total_ms);
};

void SolveRun::begin() {
    s->reduce_mailbox = nullptr;  // the initial SpMV has no dot product
    s->op_failed = false;
    memset(stats, 0, sizeof(*stats));
    // residual history: one slot per iteration, capped at 2^20 entries (later iterations go unrecorded)
    const int want_hist = config->max_iters < (1 << 20) ? config->max_iters + 1 : (1 << 20);
    if (s->hist_cap < want_hist) {
        // host-coherent pinned memory, written by the scalar kernels one 8-byte store per iteration: the host reads the history
        // (and with it the final residual) where it lies, no copy command at the end of a solve
        if (s->d_hist) HIP_CHECK(hipHostFree(s->d_hist));
        s->hist_cap = want_hist;
        HIP_CHECK(hipHostMalloc((void**)&s->d_hist, (size_t)s->hist_cap * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
    }
    CgScalars init;
    memset(&init, 0, sizeof init);
    init.max_history = s->hist_cap;
#ifdef SPMV_AMD_LAB
    init.stop_at = s->stop_at;
#endif
    HIP_CHECK(hipMemcpyAsync(s->d_s, &init, sizeof init, hipMemcpyHostToDevice, s->compute));
#ifdef SPMV_AMD_LAB
    poison_halos(s);  // the lab's tests solve one system again and again: a row that did not travel must not find last solve's copy
#endif
    // Every solve starts from the stored x0 (the reference benchmark wrapper restores x on the host before each run). x is not
    // overwritten with x0 first: the initial SpMV reads x0 and the first x update computes x = x0 + alpha p.
    HIP_CHECK(hipStreamSynchronize(s->compute));
    s->timeline_us.clear();
    s->p = s->ring[0];
    sequence0 = s->poll_sequence;
    records_read = 0;
    s->h_poll->converged = 0, s->h_poll->iterations = 0;  // (the stream is idle: nothing of the previous solve can still write here)
    s->enqueued_stage = "barrier before the timed region";
    s->enqueued_iteration = -1;
    {
        WatchdogScope guard("barrier before the timed region", s->comm->rank, -1, report_slab_state, s);
        s->comm->barrier(s->compute);
    }
    total.begin(s->compute);
    if (timeline) HIP_CHECK(hipEventRecord(tl_event(s->tl_compute, 0), s->compute));
}

// Halo rows of the halo-carrying vector v to / from the neighbours. PIPELINE: on the side stream, behind the rows to send --
// released by the direction launch's device flag (released_by_flag: the edge rows were written through by that launch's first
// workgroups; no event on the compute stream, and not as late as the launch's end) or by an event (x0's rows at the start of
// a solve) -- and followed by the arrival flag the boundary waves wait for and the event everything else waits for.
// PLAIN: on the compute stream, in order.
void SolveRun::start_halo(double* v, bool released_by_flag) {
    if (!L.multi) return;
    if (!L.pipeline) {
        timed(&stats->time_allgather_ms, nullptr, [&] { exchange_halo(s, v, s->compute); });
        halo_in_flight = false;
        return;
    }
    if (released_by_flag) {
        launch_edges_wait(s->scratch(), s->poll_sequence, (long long)(halo_wait_limit_s(s) * 1e8), &s->h_poll->halo_late, s->side);
    } else {
        HIP_CHECK(hipEventRecord(s->ev_p_ready, s->compute));
        HIP_CHECK(hipStreamWaitEvent(s->side, s->ev_p_ready, 0));
    }
    const bool in_loop = v != s->x0;
    if (timeline && in_loop) HIP_CHECK(hipEventRecord(tl_event(s->tl_side, 2 * (size_t)tl_exchanges), s->side));
    exchange_halo(s, v, s->side);
    ++s->halo_sequence;
#ifdef SPMV_AMD_LAB
    if (s->test_wedge_overlapped_exchange != 4)  // fault injection: an arrival flag that never comes
#endif
        launch_halo_arrived(s->d_halo_flag, s->halo_sequence, s->side);
    if (timeline && in_loop) HIP_CHECK(hipEventRecord(tl_event(s->tl_side, 2 * (size_t)tl_exchanges++ + 1), s->side));
    HIP_CHECK(hipEventRecord(s->ev_halo_done, s->side));
    halo_in_flight = true;
}

// ---- r0 = b - A x0 ; p0 = r0 ; rr0 ----
// x0 is read where it lies: its allocation carries the halo rows the first / last grid row need (its halo rows travel like
// p's, so that the point-to-point communicator is driven from one stream only).
void SolveRun::initial_residual() {
    start_halo(s->x0, /*released_by_flag=*/false);
    if (s->fuse_init_residual) {
        // row-lds slabs: r0 = b - A x0, p0 = r0 and the r0.r0 partials come out of the SpMV launch itself (A x0 is
        // never written out and read back: 16 B/row less, once per solve)
        const ResidualOut init{s->b, s->r, s->p};
        int used = 0;
        timed(&stats->time_initial_r_ms, nullptr, [&] { used = slab_spmv(s, /*with_dot=*/false, halo_in_flight, nullptr, s->x0, nullptr, &init); });
        timed(&stats->time_dot_rs_initial_ms, nullptr, [&] { reduce_spmv_partials(s, used, &s->d_s->rr_new, nullptr, nullptr, 0, L.mailbox); });
    } else {
        slab_spmv(s, /*with_dot=*/false, halo_in_flight, nullptr, s->x0);
        timed(&stats->time_initial_r_ms, nullptr, [&] { launch_cg_init_residual(nl, s->b, s->Ap, s->r, s->p, s->partials_blas, s->compute); });
        timed(&stats->time_dot_rs_initial_ms, nullptr, [&] {
            launch_reduce_partials(s->partials_blas, cg_partial_count(nl), &s->d_s->rr_new, nullptr, s->compute, s->scratch(), nullptr, 0, L.mailbox);
        });
    }
    s->enqueued_stage = "initial residual";
    if (L.separate) allreduce_scalar(s, &s->d_s->rr_new, "all-reduce of the initial r.r");
    s->reduce_mailbox = s->fused_dot ? L.mailbox : nullptr;  // the in-loop SpMVs' p.Ap reduction
    launch_cg_scalars_init(s->d_s, s->d_hist, s->compute);
    if (timeline) HIP_CHECK(hipEventRecord(tl_event(s->tl_compute, 1), s->compute));
    if (s->label && config->verbose >= 1) {  // the reference prints ||r0|| before its loop (cg_solver.cu:529-531); verbose runs only
        CgScalars now;
        HIP_CHECK(hipStreamSynchronize(s->compute));
        HIP_CHECK(hipMemcpy(&now, s->d_s, sizeof now, hipMemcpyDeviceToHost));
        printf("[%s] Initial residual: %e\n", s->label, now.b_norm);
    }
    start_halo(s->p, /*released_by_flag=*/false);  // halo rows of p0: under the first interior SpMV
}

// Stage 1: Ap = A p with the p.Ap partials and their sum (slab_spmv: interior rows | boundary rows + sum on a split slab).
void SolveRun::stage_spmv() {
    // SpMV and direction update of iteration k walk one way, the r update between them the other way; the direction flips
    // every iteration, so every kernel starts where the previous one ended (iteration 0 follows the forward initial pass)
    backward = (enqueued & 1) == 0;
    s->shape.reverse = backward;
    s->enqueued_iteration = enqueued;
    s->enqueued_stage = "SpMV";
    TraceScope range(trace, "SpMV");
    s->spmv_progress = &s->h_poll->progress;
    s->spmv_progress_value = 4 * (s->poll_sequence + 1) + 1;
    mark(enqueued, 0);
    if (timeline) {
        // [1] is recorded inside slab_spmv behind the interior launch, [2] behind the boundary rows; the reduction of the
        // partials is issued by slab_spmv too, so [3] follows directly
        s->tl_after_interior = tl_event(s->tl_compute, mark_index(enqueued, 1));
        slab_spmv(s, true, halo_in_flight, skip, nullptr, tl_event(s->tl_compute, mark_index(enqueued, 2)));
        s->tl_after_interior = nullptr;
    } else if (L.detail) {
        timed(&stats->time_spmv_ms, nullptr, [&] { slab_spmv(s, true, halo_in_flight, skip); });
    } else if (s->spmv_event_stride > 0 && (enqueued + s->spmv_event_phase) % s->spmv_event_stride == 0) {
        while (s->spmv_ev.size() < 2 * (size_t)(sampled + 1)) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreate(&e));
            s->spmv_ev.push_back(e);
        }
        HIP_CHECK(hipEventRecord(s->spmv_ev[2 * sampled], s->compute));
        slab_spmv(s, true, halo_in_flight, skip, nullptr, s->spmv_ev[2 * sampled + 1]);
        sampled_iteration.push_back(enqueued);
        ++sampled;
    } else {
        slab_spmv(s, true, halo_in_flight, skip);
    }
    s->spmv_progress = nullptr;
}

// Stage 2 (only where the sum is not completed across the ranks inside the reduction's launch).
void SolveRun::stage_allreduce_pAp() {
    s->enqueued_stage = "all-reduce of p.Ap";
    if (L.separate || (L.reduce && !s->fused_dot)) {
        TraceScope r(trace, "AllReduce");
        timed(&stats->time_allreduce_ms, nullptr, [&] { allreduce_scalar(s, &s->d_s->pAp, "all-reduce of p.Ap"); });
    }
    mark(enqueued, 3);
}

// Stage 3: r -= alpha Ap with the r.r partials (alpha = rr_old / pAp in every thread).
void SolveRun::stage_update_r() {
    s->enqueued_stage = "r update";
    TraceScope range(trace, "BLAS_AXPY");
    timed(&stats->time_blas1_ms, &stats->time_axpy_update_r_ms,
          [&] { launch_cg_update_r(nl, s->d_s, s->Ap, s->r, s->partials_blas, s->compute, !backward); });
    mark(enqueued, 4);
}

// Stage 4: sum of the r.r partials, across the ranks, and the scalar step (residual, history, verdict, beta, alpha into the
// ring's slot, status record into host-coherent memory). One launch without a separate all-reduce; with one (RCCL) the
// step follows the all-reduce -- as a launch of its own, or inside the direction update's launch (pipeline), except in
// the iteration whose new direction re-uses a slot the pending x flush still has to read: the flush needs this step's alpha
// and must run before the slot is overwritten.
void SolveRun::stage_sum_rr_and_step() {
    step_in_direction = L.pipeline && L.separate && (enqueued + 1) - window_start < L.slots;
    ++s->poll_sequence;
    TraceScope range(trace, "Dot_Product");
    if (L.separate) {
        timed(&stats->time_reductions_ms, &stats->time_dot_rs_new_ms, [&] {
            launch_reduce_partials(s->partials_blas, cg_partial_count(nl), &s->d_s->rr_new, skip, s->compute, s->scratch(), &s->h_poll->progress,
                                   4 * s->poll_sequence + 2);
        });
        s->enqueued_stage = "all-reduce of r.r";
        {
            TraceScope r(trace, "AllReduce");
            timed(&stats->time_allreduce_ms, nullptr, [&] { allreduce_scalar(s, &s->d_s->rr_new, "all-reduce of r.r"); });
        }
        if (!step_in_direction)
            launch_cg_scalars_step(s->d_s, config->tolerance, s->d_hist, &s->h_poll->sequence, s->poll_sequence, s->compute, s->d_alpha_ring, L.slots);
    } else {
        timed(&stats->time_reductions_ms, &stats->time_dot_rs_new_ms, [&] {
            launch_reduce_partials_and_step(s->partials_blas, cg_partial_count(nl), &s->d_s->rr_new, skip, s->compute, s->scratch(), s->d_s,
                                            config->tolerance, s->d_hist, &s->h_poll->sequence, s->poll_sequence, s->d_alpha_ring, L.slots, L.mailbox,
                                            &s->h_poll->progress, 4 * s->poll_sequence + 2);
        });
    }
    mark(enqueued, 5);
    ++enqueued;  // from here on `enqueued` is the number of the iteration just stepped (1-based), as the kernels count
}

// Is iteration `iteration` (1-based) too far from the tolerance to be the converging one? Judged by the residual of iteration
// `basis`, which the host has SEEN (basis <= records_read; the history lies in host-coherent memory and is written before the record): CG on
// these systems halves the residual per iteration; a drop of 16 x per iteration between the known residual and `iteration` is
// the margin. A wrong guess costs empty dispatches, never a result: every kernel tests the flag itself.
bool SolveRun::far_from_convergence(int iteration, int basis) const {
#ifdef SPMV_AMD_LAB
    if (s->guess_far_always) return true;  // test hook: the wrong guess in every solve
    if (s->stop_at > 0) return iteration + 3 < s->stop_at;  // a stand-in's last iteration is declared: behave like a solve that converges there
#endif
    if (basis < 0 || basis > records_read || basis >= s->hist_cap || iteration <= basis) return false;
    const double ratio = basis == 0 ? 1.0 : s->d_hist[basis] / s->d_hist[0];  // ||r0|| itself may not have been written yet
    double bar = config->tolerance;
    for (int k = basis; k < iteration; ++k) bar *= 16.0;
    return ratio > bar;  // (a NaN compares false: "may converge", the careful path)
}

// The verdict on iteration k (1-based, k <= records_read) from the host-coherent history: the GPU's own test (reduce_device.hpp,
// cg_converging: sqrt(rr_new) / b_norm < tol, strict) on the same two doubles -- history[k] is that sqrt, history[0] is b_norm.
bool SolveRun::converged_at(int k) const {
    if (k < 1 || k >= s->hist_cap) return s->h_poll->converged != 0;
#ifdef SPMV_AMD_LAB
    if (s->stop_at > 0 && k == s->stop_at) return true;
#endif
    return s->d_hist[k] / s->d_hist[0] < config->tolerance;
}

void SolveRun::await_record(int iteration) {
    wait_for_record(s, sequence0 + iteration);
    if (iteration > records_read) records_read = iteration;
}

// Stage 5: p <- r + beta p and its halo exchange; nothing else before the host looks at the status: the GPU works on these
// while the host waits for the record. Late bulk: the piece the sweep walks first, then the status record, then -- unless the
// iteration converged -- the rest; the lead piece keeps the GPU busy while the host reads the record and launches.
void SolveRun::stage_direction_and_halo() {
    trace.push("BLAS_AXPBY");
    const size_t lo = L.bulk_lo, hi = L.bulk_hi;
    const bool two_pieces = L.late && hi - lo >= 4 * s->lead_rows && !(s->late_predict && far_from_convergence(enqueued, records_read));  // a local choice: any basis the host has will do
    const size_t cut = !two_pieces ? (backward ? lo : hi) : backward ? (hi - s->lead_rows) / 512 * 512 : lo + s->lead_rows;
    // first piece: [cut, hi) walking backward, [lo, cut) walking forward; the rest is the other side of the cut
    const size_t first_lo = backward ? cut : lo, first_rows = backward ? hi - cut : cut - lo;
    const size_t rest_lo = backward ? lo : cut, rest_rows = backward ? cut - lo : hi - cut;
    auto pieces = [&](auto&& first, auto&& rest) {
        timed(&stats->time_blas1_ms, &stats->time_axpby_update_p_ms, [&] {
            first(first_lo, first_rows);
            if (!two_pieces) return;
            s->enqueued_stage = "direction update (lead piece)";
            await_record(enqueued);
            if (!s->h_poll->converged) rest(rest_lo, rest_rows);  // else the loop ends here: nothing reads the rest of this direction
        });
    };
    if (L.slots == 1) {  // in-place form: x += alpha p rides with the direction update
        const double* x_in = enqueued == 1 ? s->x0 : s->x;
        auto px = [&](size_t off, size_t count) {
            if (count > 0) launch_cg_update_px(count, s->d_s, s->r + off, s->p + off, x_in + off, s->x + off, enqueued, s->compute, backward, s->device_form);
        };
        pieces(px, px);
    } else {
        // the slot the new direction goes to still holds p of iteration enqueued - slots: fold the whole window into x first
        // (alpha of this iteration is already on the stream: the step above wrote it)
        if (enqueued - window_start == L.slots) {
            timed(&stats->time_blas1_ms, nullptr, [&] {
                launch_cg_flush_x(nl, s->d_alpha_ring, ring_view, L.slots, window_start % L.slots, L.slots, window_start == 0 ? s->x0 : s->x, s->x, s->compute,
                                  s->d_s, window_start);
            });
            window_start = enqueued;
        }
        double* p_next = s->ring[(size_t)(enqueued % L.slots)];
        const double* p_in = s->p;
        auto plain = [&](size_t off, size_t count) {
            if (count > 0) launch_cg_update_p_ring(count, s->d_s, s->r + off, p_in + off, p_next + off, enqueued, s->compute, backward, s->device_form);
        };
        auto fused = [&](size_t off, size_t count) {
            // the edge rows first (they raise the flag the exchange waits for), then this piece; the step too where it is due
            launch_cg_direction(DirectionLaunch{s->d_s, config->tolerance, enqueued, s->d_hist, step_in_direction ? &s->h_poll->sequence : nullptr,
                                                s->poll_sequence, s->d_alpha_ring, L.slots, s->r, p_in, p_next, off, count, backward, s->device_form},
                                &L.edges, s->scratch(), s->compute);
            s->p = p_next;  // the exchange sends from / receives into the new direction buffer
            trace.pop();
            {
                TraceScope r(trace, "Halo_Exchange");
                start_halo(s->p, /*released_by_flag=*/true);
            }
            trace.push("BLAS_AXPBY");
        };
        if (L.pipeline) pieces(fused, plain);
        else pieces(plain, plain);
        s->p = p_next;
    }
    trace.pop();
    mark(enqueued - 1, 6);
    s->enqueued_stage = "direction update and halo exchange";
    if (!L.pipeline) {
        TraceScope r(trace, "Halo_Exchange");
        start_halo(s->p, /*released_by_flag=*/false);
    }
}

// The host reads the iteration's status record (the GPU is already busy with the direction update, the exchange and -- in the
// next turn of the loop -- the SpMV). Returns true when the loop is over.
bool SolveRun::read_status() {
    // Round 6: the host may run ONE iteration ahead of the records it has seen. While the residual it knows says the iteration
    // just enqueued cannot be the converging one, the host does not wait for that iteration's record: it goes on to enqueue the
    // next iteration and reads the record on the way. The GPU then always has an iteration's worth of work queued, and a host
    // that answers late (a CPU-throttled container: profiles/r06_throttle_probe.txt) no longer leaves it idle between a
    // direction update and the next SpMV. Near convergence every record is awaited before the next launch, as before.
    // The decision must be the SAME ON EVERY RANK (an iteration carries collectives): it is taken from the residual history
    // alone -- identical bits on all ranks -- never from whether a record happens to have arrived; and the verdict on an
    // iteration whose record was read late comes from the history too (converged_at), not from the record's flag, which a
    // later record may already have overwritten.
    // (that includes a rank that already KNOWS the outcome of this iteration because its slab is large enough for the late
    // bulk's lead / status / rest protocol: it follows the common rule too, and at worst enqueues an iteration of no-ops)
    const int j = enqueued;
    if (records_read < j - 1) {
        await_record(j - 1);
        if (converged_at(j - 1)) {  // the guess was wrong: iteration j was enqueued for nothing (its kernels saw the flag); the loop ends here
            await_record(j);
            return true;
        }
    }
    // basis j - 1, the one residual every rank is sure to have at this point (verbose >= 2 prints every iteration's scalars: no run-ahead)
    const bool deferred = s->run_ahead && config->verbose < 2 && j + 1 < s->hist_cap && far_from_convergence(j, j - 1);
    if (!deferred) await_record(j);
    mailbox_check(s->comm);
    if (const int gave_up = __atomic_load_n(&s->h_poll->halo_late, __ATOMIC_ACQUIRE)) {
        if (s->selfcheck) {  // creation check: a hand-over that never came is a verdict, not the end of the process
            s->selfcheck_late = true;
            __atomic_store_n(&s->h_poll->halo_late, 0, __ATOMIC_RELEASE);
            return true;
        }
        // worded like the host watchdog's report: bench.py's supervisors read that sentence and restart the ranks once without the overlap
        fprintf(stderr, "\n[spmv_amd watchdog] rank %d: no progress for %.1f s in stage '%s' (CG iteration %d)\n", s->comm->rank, halo_wait_limit_s(s),
                gave_up == 3 ? "edge rows' ready flag (side-stream wait in front of the halo exchange)"
                             : "halo arrival flag (in-kernel wait of the boundary rows)",
                enqueued - 1);
        report_slab_state(s, stderr);
        exit(EXIT_FAILURE);
    }
    if (config->verbose >= 2 && s->comm->rank == 0) {
        CgScalars now;
        HIP_CHECK(hipMemcpy(&now, s->d_s, sizeof now, hipMemcpyDeviceToHost));
        if (s->label)  // cg_solve_device's line (reference cg_solver.cu:604-607)
            printf("[%s] Iter %3d: residual = %e (rel = %e)\n", s->label, now.iterations, now.residual, now.residual / now.b_norm);
        else
            printf("[Iter %3d] Residual: %.6e (rel: %.6e, alpha: %.4e)\n", now.iterations, now.residual, now.residual / now.b_norm, now.alpha);
    }
    // not deferred: record `enqueued` is the latest there can be (nothing of the next iteration is on the stream): its flag is exact
    return deferred ? false : s->h_poll->converged != 0;
}

// Averages over the counted iterations of a timeline solve; order = kTimelineNames.
void SolveRun::resolve_timeline(const CgScalars& fin, float total_ms) {
    auto us = [&](hipEvent_t a, hipEvent_t b) {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        return (double)ms * 1e3;
    };
    const std::vector<hipEvent_t>& E = s->tl_compute;
    const size_t tl_flush = mark_index(enqueued, 0);  // [+0] before, [+1] after the final flush
    const int its = fin.iterations < enqueued ? fin.iterations : enqueued;
    // The direction update of the converging iteration does no work (its launch reads the flag and returns; with late
    // bulk only a lead piece is launched at all): it is left out of that stage's average, as the reference divides each
    // timer by the iterations that ran it (cg_solver_mgpu_partitioned.cu:770-800) and tests convergence before its p
    // update (:652-676). iteration_us stays the average over all counted iterations, the shorter last one included.
    const int direction_updates = fin.converged && its > 0 ? its - 1 : its;
    double stage[kTimelineMarks] = {0, 0, 0, 0, 0, 0, 0};  // [k] = mark k -> mark k+1; [6] = mark 6 -> next iteration's mark 0
    double iteration_us = 0.0;
    for (int it = 0; it < its; ++it) {
        const size_t base = mark_index(it, 0);
        for (int k = 0; k < kTimelineMarks - 1; ++k)
            if (k != 5 || it < direction_updates) stage[k] += us(E[base + k], E[base + k + 1]);
        const hipEvent_t next = it + 1 < enqueued ? E[base + kTimelineMarks] : E[tl_flush];  // the last counted iteration ends at the flush mark
        stage[6] += us(E[base + 6], next);
        iteration_us += us(E[base], next);
    }
    double side_us = 0.0;
    const int exchanges = tl_exchanges < its ? tl_exchanges : its;  // exchange j feeds the SpMV of iteration j
    for (int j = 0; j < exchanges; ++j) side_us += us(s->tl_side[2 * (size_t)j], s->tl_side[2 * (size_t)j + 1]);
    const double per = its > 0 ? 1.0 / its : 0.0;
    s->timeline_us = {(double)its, (double)total_ms, us(E[0], E[1]), stage[0] * per, stage[1] * per, stage[2] * per, stage[3] * per,
                      stage[4] * per, direction_updates > 0 ? stage[5] / direction_updates : 0.0, stage[6] * per, iteration_us * per,
                      exchanges > 0 ? side_us / exchanges : 0.0, us(E[tl_flush], E[tl_flush + 1]), (double)direction_updates};
    stats->time_spmv_ms = (s->timeline_us[3] + s->timeline_us[4]) * s->timeline_us[0] / 1e3;
}

// The last flush of x, the end of the timed region, and the outcome from what the scalar kernels left in host-coherent memory.
void SolveRun::finish() {
    s->shape.reverse = false;
    s->reduce_mailbox = nullptr;
    const size_t tl_flush = mark_index(enqueued, 0);
    if (timeline) HIP_CHECK(hipEventRecord(tl_event(s->tl_compute, tl_flush), s->compute));
    if (L.slots > 1 && enqueued > window_start)  // x <- x + the directions of the last window
        timed(&stats->time_blas1_ms, nullptr, [&] {
            launch_cg_flush_x(nl, s->d_alpha_ring, ring_view, L.slots, window_start % L.slots, enqueued - window_start, window_start == 0 ? s->x0 : s->x,
                              s->x, s->compute, s->d_s, window_start);
        });
    s->p = s->ring[0];
    if (enqueued == 0)  // no iteration ran (max_iters == 0): the solution is the initial guess
        HIP_CHECK(hipMemcpyAsync(s->x, s->x0, nl * sizeof(double), hipMemcpyDeviceToDevice, s->compute));
    if (timeline) HIP_CHECK(hipEventRecord(tl_event(s->tl_compute, tl_flush + 1), s->compute));
    total.end(s->compute);
    float total_ms = 0.f;
    {
        WatchdogScope guard("draining the streams after the loop", s->comm->rank, enqueued, report_slab_state, s);
        total_ms = total.elapsed_ms();
        if (s->side) HIP_CHECK(hipStreamSynchronize(s->side));
    }
    HIP_CHECK(hipGetLastError());
    // Iteration count and flag from the last status record, ||r|| of the last counted iteration from the history -- which is
    // fin.residual when converged and sqrt(fin.rr_old) when not (the step moves rr_new into rr_old exactly then). Only a solve
    // longer than the history (2^20 iterations) has to fetch the device scalars.
    CgScalars fin;
    memset(&fin, 0, sizeof fin);
    if (enqueued > 0 && !s->op_failed) {
        fin.iterations = s->h_poll->iterations;
        fin.converged = s->h_poll->converged;
    }
    if (fin.iterations < s->hist_cap && !s->op_failed) {
        fin.residual = s->d_hist[fin.iterations];
        fin.rr_old = fin.residual * fin.residual;
    } else {
        HIP_CHECK(hipMemcpy(&fin, s->d_s, sizeof fin, hipMemcpyDeviceToHost));
    }
    if (timeline) {
        resolve_timeline(fin, total_ms);
    } else if (!L.detail) {  // timed SpMV launches that did real work (one per counted iteration), scaled to all of them
        double ms_sum = 0.0;
        s->last_spmv_each.clear();
        for (int k = 0; k < sampled; ++k) {
            if (sampled_iteration[k] >= fin.iterations) continue;
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, s->spmv_ev[2 * k], s->spmv_ev[2 * k + 1]));
            ms_sum += ms;
            s->last_spmv_each.push_back(ms);
        }
        stats->time_spmv_ms = s->last_spmv_each.empty() ? 0.0 : ms_sum / (double)s->last_spmv_each.size() * fin.iterations;
    }
    ++s->spmv_event_phase;
    s->last_spmv_ms = stats->time_spmv_ms;
    s->last_spmv_launches = fin.iterations;
    stats->iterations = fin.iterations;
    stats->converged = fin.converged;
    // not converged: the reference reports sqrt(rs_old) of the last completed iteration (:720-725)
    stats->residual_norm = (fin.converged || fin.iterations < s->hist_cap) ? fin.residual : sqrt(fin.rr_old);
    stats->time_total_ms = total_ms;
    if (!fin.converged && s->comm->rank == 0 && s->label == nullptr) printf("\nMax iterations reached without convergence\n");
    if (L.detail && stats->iterations > 0) {
        stats->time_dot_rs_new_ms /= stats->iterations;
        stats->time_axpy_update_r_ms /= stats->iterations;
        stats->time_axpby_update_p_ms /= stats->iterations;
    }
    const int count = fin.iterations + 1 < s->hist_cap ? fin.iterations + 1 : s->hist_cap;
    s->history.assign(s->d_hist, s->d_hist + count);
    last_cg_history() = s->history;
}

}  // namespace

extern "C" int spmv_amd_cg_slab_solve(SpmvAmdCgSlab* s, const CGConfigMultiGPU* config, CGStatsMultiGPU* stats) {
    SolveRun run(s, config, stats);
    // the reference's NVTX ranges (:540-717) as roctx ranges, when detailed timers (or SPMV_AMD_ROCTX=1) ask for them
    TraceScope solver_range(run.trace, "CG_Solver");
    run.begin();
    run.initial_residual();
    bool done = false;
    while (!done && !s->op_failed && run.enqueued < config->max_iters) {
        TraceScope iteration_range(run.trace, "CG_Iteration");
        run.stage_spmv();
        if (s->op_failed) break;  // nothing of this iteration is awaited yet
        run.stage_allreduce_pAp();
        run.stage_update_r();
        run.stage_sum_rr_and_step();
        run.stage_direction_and_halo();
        done = run.read_status();
    }
    run.finish();
    return s->op_failed ? 1 : 0;
}

extern "C" int spmv_amd_cg_slab_gather(SpmvAmdCgSlab* s, double* x_full) {
    const int P = s->comm->world;
    std::vector<int> counts((size_t)P), displs((size_t)P);
    for (int r = 0; r < P; ++r) spmv_amd_partition_rows(s->n, P, r, &displs[r], &counts[r]);
    HIP_CHECK(hipStreamSynchronize(s->compute));
    if (s->comm->rank != 0)  // non-root ranks keep their own slab in place, like the reference (:831-833)
        download(x_full + s->row_offset, s->x, (size_t)s->n_local);
    WatchdogScope guard("gathering the solution on rank 0", s->comm->rank, -1, report_slab_state, s);
    s->comm->gather_to_root(s->x, s->n_local, x_full, counts.data(), displs.data());
    return 0;
}

extern "C" int spmv_amd_cg_slab_history(SpmvAmdCgSlab* s, double* out, int cap) {
    const int count = (int)s->history.size();
    for (int i = 0; i < count && i < cap; ++i) out[i] = s->history[i];
    return count;
}

extern "C" int spmv_amd_cg_slab_spmv(SpmvAmdCgSlab* s, const double* x_full, double* y_local) {
    upload(s->p, x_full + s->row_offset, (size_t)s->n_local);
    if (s->has_prev && s->row_offset >= s->halo)
        upload(s->p - s->halo, x_full + s->row_offset - s->halo, (size_t)s->halo);
    if (s->has_next && s->row_offset + s->n_local + s->halo <= s->n)
        upload(s->p + s->n_local, x_full + s->row_offset + s->n_local, (size_t)s->halo);
    slab_spmv(s, /*with_dot=*/false, /*overlap=*/false, nullptr);
    HIP_CHECK(hipStreamSynchronize(s->compute));
    HIP_CHECK(hipGetLastError());
    download(y_local, s->Ap, (size_t)s->n_local);
    return 0;
}

extern "C" void spmv_amd_cg_slab_info(const SpmvAmdCgSlab* s, int* row_offset, int* n_local,
                                      int* local_nnz) {
    if (row_offset) *row_offset = s->row_offset;
    if (n_local) *n_local = s->n_local;
    if (local_nnz) *local_nnz = (int)s->A.view.nnz_local;
}

extern "C" const char* spmv_amd_cg_slab_variant(const SpmvAmdCgSlab* s) { return s->variant_name; }

// What placement at creation did: {0, candidates timed, SpMV ms (mean of an early and a late direction buffer as x) where the
// coefficients were, ms where they are now}. 0 values = it did not run.
extern "C" int spmv_amd_cg_slab_placement(const SpmvAmdCgSlab* s, double* out, int cap) {
    const int count = (int)s->placement.size();
    for (int i = 0; i < count && i < cap; ++i) out[i] = s->placement[i];
    return count;
}
// Wall ms of creation's set-up phases (none of it inside any timed region): {matrix to HBM, streams + vectors, verification +
// launch plans, coefficient placement trial, tile-run trial}. Returns 5.
extern "C" int spmv_amd_cg_slab_setup_ms(const SpmvAmdCgSlab* s, double* out, int cap) {
    for (int i = 0; i < 5 && i < cap; ++i) out[i] = s->setup_ms[i];
    return 5;
}
// Row-lds tiles per XCD and run as tuned at creation: {rule, kept, SpMV ms with the rule, ms kept}. Returns 4, or 0 if the
// trial did not run (small slab, another kernel, SPMV_AMD_ROWLDS_GROUP set).
extern "C" int spmv_amd_cg_slab_tile_runs(const SpmvAmdCgSlab* s, double* out, int cap) {
    const int count = (int)s->tile_runs.size();
    for (int i = 0; i < count && i < cap; ++i) out[i] = s->tile_runs[i];
    return count;
}

// The in-loop SpMV launches of the last solve one by one (ms, iteration order; only the launches that were timed: every one on
// slabs of >= 1e8 rows, every fourth below). Returns how many there are.
extern "C" int spmv_amd_cg_slab_spmv_launch_ms(const SpmvAmdCgSlab* s, float* out, int cap) {
    const int count = (int)s->last_spmv_each.size();
    for (int i = 0; i < count && i < cap; ++i) out[i] = s->last_spmv_each[i];
    return count;
}

extern "C" void spmv_amd_cg_slab_set_timeline(SpmvAmdCgSlab* s, int on) { s->timeline_on = on != 0; }

#ifdef SPMV_AMD_LAB
// (LAB build only.) Options of an existing slab, for A/B measurements on the SAME allocations (two slabs of one process differ
// by up to +-1.3 % through their placement alone, profiles/r03_placement.txt): "no_overlap" (the PLAIN loop shape), "late_bulk",
// "lead_rows" -- none of which changes a bit of the results -- "spmv_event_stride", and the timing aid "stop_at".
// Returns 0, or -1 for an unknown name.
extern "C" int spmv_amd_cg_slab_set_option(SpmvAmdCgSlab* s, const char* name, long long value) {
    if (strcmp(name, "late_bulk") == 0) s->late_bulk = value != 0, s->late_predict = value == 2;  // 1: the protocol in every iteration, 2: where convergence is near (the default rule)
    else if (strcmp(name, "lead_rows") == 0) s->lead_rows = value < 512 ? 512 : (size_t)value / 512 * 512;
    else if (strcmp(name, "run_ahead") == 0) s->run_ahead = value != 0, s->guess_far_always = value == 2;
    else if (strcmp(name, "no_overlap") == 0) s->no_overlap = value != 0 || s->comm->pipeline_verdict < 0;  // a refused pipeline stays refused
    else if (strcmp(name, "stop_at") == 0) s->stop_at = value > 0 ? (int)value : 0;
    else if (strcmp(name, "spmv_event_stride") == 0) s->spmv_event_stride = (int)value;
    else return -1;
    return 0;
}
#endif  // SPMV_AMD_LAB
extern "C" const char* spmv_amd_cg_slab_timeline_names(void) { return kTimelineNames; }
extern "C" int spmv_amd_cg_slab_timeline(const SpmvAmdCgSlab* s, double* out, int cap) {
    const int count = (int)s->timeline_us.size();
    for (int i = 0; i < count && i < cap; ++i) out[i] = s->timeline_us[i];
    return count;
}

extern "C" int spmv_amd_cg_slab_time_spmv(SpmvAmdCgSlab* s, int reps, float* ms_each) {
    HIP_CHECK(hipMemsetAsync(&s->d_s->converged, 0, sizeof(int), s->compute));
    EventTimer t;
    for (int i = 0; i < reps; ++i) {
        t.begin(s->compute);
        (void)launch_stencil5_spmv(s->A.view, s->plan_whole, s->p, s->Ap, 1.0, s->fused_dot ? s->partials_spmv : nullptr,
                                   nullptr, false, s->compute);
        t.end(s->compute);
        ms_each[i] = t.elapsed_ms();
    }
    HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" void spmv_amd_cg_slab_destroy(SpmvAmdCgSlab* s) {
    if (!s) return;
    (void)hipStreamSynchronize(s->compute);
    (void)hipStreamSynchronize(s->side);
    s->A.release();
    device_release(s->x);
    device_release(s->x0_alloc);
    device_release(s->b);
    s->x0 = nullptr;
    device_release(s->vec_arena);  // r, Ap and the direction buffers
    s->r = s->Ap = s->p_alloc = s->p = nullptr;
    s->ring_alloc.clear();
    s->ring.clear();
    device_release(s->d_alpha_ring);
    device_release(s->partials_spmv);
    device_release(s->partials_blas);
    device_release(s->reduce_stage);
    device_release(s->d_halo_flag);
    device_release(s->d_s);
    if (s->d_hist) (void)hipHostFree(s->d_hist);
    if (s->h_poll) (void)hipHostFree(s->h_poll);
    for (hipEvent_t e : s->spmv_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->tl_compute) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->tl_side) (void)hipEventDestroy(e);
    (void)hipEventDestroy(s->ev_p_ready);
    (void)hipEventDestroy(s->ev_halo_done);
    if (s->owns_streams) {
        (void)hipStreamDestroy(s->compute);
        (void)hipStreamDestroy(s->side);
    }
    delete s;
}

// ---------------------------------------------------------------------------------------
// cg_solve_device on a borrowed operator (reference src/solvers/cg_solver.cu:436-706)
// ---------------------------------------------------------------------------------------
namespace {
SpmvAmdCgSlab* g_workspace = nullptr;  // vectors, direction ring, scalars of cg_solve_device, kept between calls
int g_workspace_device = -1;
// one solve at a time on the shared workspace (the reference's operators are not re-entrant either, SURVEY 8b "Threading";
// two host threads calling cg_solve_device are serialised here instead of racing on the buffers)
std::mutex g_workspace_lock;
// The thread inside cg_solve_on_operator (it holds the lock for the whole solve, callbacks into a foreign run_device included).
// A release asked for BY THAT THREAD -- a caller's run_device that calls spmv_amd_cg_release_workspace() or an operator's
// free() -- cannot take the lock again and must not pull the vectors from under the running loop: it is noted and carried
// out when the solve returns.
std::atomic<std::thread::id> g_workspace_owner{std::thread::id()};  // read without the lock by release_cg_workspace() on any thread
bool g_release_pending = false;  // only ever touched by the owner thread (which holds the lock)
void release_cg_workspace_locked() {
    if (g_workspace == nullptr) return;
    spmv_amd_cg_slab_destroy(g_workspace);
    g_workspace = nullptr;
}
}  // namespace

namespace spmv_amd {

void release_cg_workspace() {
    if (g_workspace_owner.load(std::memory_order_acquire) == std::this_thread::get_id()) {  // only ever equal on the thread that set it
        g_release_pending = true;
        return;
    }
    std::lock_guard<std::mutex> guard(g_workspace_lock);
    release_cg_workspace_locked();
}

// The loop of spmv_amd_cg_slab_solve driven through the caller's operator: the same fused kernels, direction ring,
// deferred x update, status record in host-coherent memory (no blocking read-back per iteration, no event pairs unless
// detailed timers ask for them), one rank, default stream. What stays the reference's: run_device(d_x, d_y) is the only
// thing asked of an operator this library does not own; b / x are host arrays, uploaded before and downloaded after the
// timed region (:458-474, :646-649); ||r0|| is the stopping test's denominator; the direction update is rounded as
// update_p_kernel rounds it. The reference allocates and frees its vectors in every call; here they are kept until an
// operator's free() (the harness solves the same system 13 times: 3 warm-ups + 10 runs, src/main/cg_solver.cu:154-178).
int cg_solve_on_operator(SpmvOperator* op, int n, const double* b, double* x, const CGConfig& config, CGStats* stats,
                         std::vector<double>* history) {
    std::lock_guard<std::mutex> guard(g_workspace_lock);
    struct Owner {  // marks this thread as the one inside the solve; a release it asked for meanwhile is carried out on the way out
        Owner() { g_release_pending = false, g_workspace_owner.store(std::this_thread::get_id(), std::memory_order_release); }
        ~Owner() {
            g_workspace_owner.store(std::thread::id(), std::memory_order_release);
            if (g_release_pending) release_cg_workspace_locked();
            g_release_pending = false;
        }
    } owner;
    int device = 0;
    HIP_CHECK(hipGetDevice(&device));
    if (g_workspace != nullptr && (g_workspace->n != n || g_workspace_device != device)) release_cg_workspace_locked();
    if (g_workspace == nullptr) {
        SpmvAmdCgSlab* s = new SpmvAmdCgSlab();
        s->comm = self_comm();
        s->op = op;
        s->n = s->n_local = n;
        s->grid = -1;
        s->label = "CG-DEVICE";
        s->device_form = true;
        make_common(s);
        g_workspace = s;
        g_workspace_device = device;
    } else {
        adopt_operator(g_workspace, op);
    }
    SpmvAmdCgSlab* s = g_workspace;
    upload(s->b, b, (size_t)n);
    upload(s->x0, x, (size_t)n);
    const CGConfigMultiGPU cfg = {config.max_iters, config.tolerance, config.verbose, config.enable_detailed_timers};
    CGStatsMultiGPU st;
    if (spmv_amd_cg_slab_solve(s, &cfg, &st) != 0) return 1;
    download(x, s->x, (size_t)n);
    *history = s->history;
    stats->iterations = st.iterations;
    // not converged: the reference's final_residual_norm is whatever it last copied back -- the residual of the last
    // iteration under verbose >= 2, else still ||r0|| (:535, :601-619)
    const double r0 = s->history.empty() ? 0.0 : s->history.front();
    stats->residual_norm = st.converged ? st.residual_norm : (config.verbose >= 2 && !s->history.empty() ? s->history.back() : r0);
    stats->time_total_ms = st.time_total_ms;
    stats->time_spmv_ms = st.time_spmv_ms + (config.enable_detailed_timers ? st.time_initial_r_ms : 0.0);
    stats->time_blas1_ms = st.time_blas1_ms;
    stats->time_reductions_ms = st.time_reductions_ms + st.time_dot_rs_initial_ms;
    stats->converged = (r0 > 0.0 && stats->residual_norm / r0 < config.tolerance) ? 1 : 0;
    return 0;
}

}  // namespace spmv_amd

// ---------------------------------------------------------------------------------------
// reference entry point
// ---------------------------------------------------------------------------------------
int cg_solve_mgpu_partitioned(SpmvOperator* spmv_op, MatrixData* mat, const double* b, double* x,
                              CGConfigMultiGPU config, CGStatsMultiGPU* stats) {
    (void)spmv_op;  // unused upstream as well; callers pass NULL
    SpmvAmdComm* comm = world_comm();
    const int rank = comm->rank, world = comm->world;
    if (rank == 0 && config.verbose >= 1) {
        printf("\n========================================\n");
        printf("Multi-GPU CG Solver (PARTITIONED CSR)\n");
        printf("========================================\n");
        printf("Ranks: %d (transport: %s)\n", world, comm->transport());
        printf("Problem size: %d unknowns\n", mat->rows);
        printf("Max iterations: %d\n", config.max_iters);
        printf("Tolerance: %.1e\n", config.tolerance);
        printf("========================================\n\n");
    }
    SpmvAmdCgSlab* s = create_from_matrix(mat, comm, /*setup_trials=*/false);  // one solve, then destroyed: no timed trials
    if (!s) return 1;
    if (config.verbose >= 1)
        printf("[Rank %d] Rows: [%d:%d) (%d rows), local nnz %lld\n", rank, s->row_offset,
               s->row_offset + s->n_local, s->n_local, s->A.view.nnz_local);
    spmv_amd_cg_slab_set_vectors(s, b, x);
    spmv_amd_cg_slab_solve(s, &config, stats);

    // slowest rank decides the wall time (reference :748-800 reduces six timers with MPI_MAX)
    if (world > 1) {
        // max over ranks through a sum of one-hot slots would need P slots; the timers only feed
        // rank 0's report, so gather them with the collective we have: all-reduce of per-rank slots.
        std::vector<double> slots((size_t)world * 6, 0.0);
        double mine[6] = {stats->time_total_ms,       stats->time_spmv_ms,      stats->time_blas1_ms,
                          stats->time_reductions_ms, stats->time_allreduce_ms, stats->time_allgather_ms};
        memcpy(&slots[(size_t)rank * 6], mine, sizeof mine);
        double* d_slots = device_alloc<double>(slots.size());
        upload(d_slots, slots.data(), slots.size());
        {
            WatchdogScope guard("all-reduce of the ranks' timers", rank, -1, report_slab_state, s);
            comm->allreduce_sum(d_slots, (int)slots.size(), s->compute);
            HIP_CHECK(hipStreamSynchronize(s->compute));
        }
        download(slots.data(), d_slots, slots.size());
        device_release(d_slots);
        if (rank == 0) {
            double mx[6], mn[6];
            for (int k = 0; k < 6; ++k) {
                mx[k] = mn[k] = slots[k];
                for (int r = 1; r < world; ++r) {
                    const double v = slots[(size_t)r * 6 + k];
                    mx[k] = v > mx[k] ? v : mx[k];
                    mn[k] = v < mn[k] ? v : mn[k];
                }
            }
            stats->time_total_ms = mx[0];
            stats->time_spmv_ms = mx[1];
            stats->time_blas1_ms = mx[2];
            stats->time_reductions_ms = mx[3];
            stats->time_allreduce_ms = mx[4];
            stats->time_allgather_ms = mx[5];
            printf("Total time: %.2f ms (max), %.2f ms (min) - Load imbalance: %.1f%%\n", mx[0], mn[0],
                   100.0 * (mx[0] - mn[0]) / mx[0]);
        }
    } else if (rank == 0 && config.verbose >= 1) {
        printf("Total time: %.2f ms\n", stats->time_total_ms);
    }

    spmv_amd_cg_slab_gather(s, x);
    if (rank == 0) {
        double sum = 0.0, sq = 0.0;
        for (int i = 0; i < mat->rows; i++) {
            sum += x[i];
            sq += x[i] * x[i];
        }
        stats->solution_sum = sum;
        stats->solution_norm = sqrt(sq);
    }
    spmv_amd_cg_slab_destroy(s);
    return 0;
}

// Frees what cg_solve_device keeps between calls (five vectors + the direction ring). A caller that drives its OWN operator
// table through cg_solve_device never passes through this library's operator free(), which is where the workspace is
// otherwise released.
extern "C" void spmv_amd_cg_release_workspace(void) { spmv_amd::release_cg_workspace(); }

extern "C" int spmv_amd_cg_solve_mgpu_partitioned(MatrixData* mat, const double* b, double* x,
                                                  const CGConfigMultiGPU* config,
                                                  CGStatsMultiGPU* stats) {
    return cg_solve_mgpu_partitioned(nullptr, mat, b, x, *config, stats);
}
