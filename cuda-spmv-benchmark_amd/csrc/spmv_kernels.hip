// spmv_kernels.hip -- hand-written gfx950 kernels for the three SpMV operators.
//
// Arithmetic contract (numerical-semantics checklist): every
// multiply-add below is an explicit fma() and the file is compiled with
// -ffp-contract=off, so the order and fusing of operations is exactly
//   interior stencil rows : t = vW*xW ; fma(vC,xC,t) ; fma(vE,xE,t) ; fma(vN,xN,t) ; fma(vS,xS,t)
//                           (reference src/spmv/spmv_stencil_csr_direct.cu:105-109 under nvcc -fmad)
//   every other row       : sum = 0 ; sum = fma(v[k], x[col[k]], sum) for ascending k
//                           (reference :116-119 ; cg_solver_mgpu_partitioned.cu:49-52)
// which is what oracle/spmv_oracle.c evaluates on the CPU.
//
// Memory contract: STENCIL5 is HBM-bound at 56 B per interior row (40 B values, 8 B x, 8 B y). Three kernels implement it:
//   row-lds     (grids of >= 512 columns, slabs made of whole grid rows: the benchmark's path) streams `values` with fully
//               coalesced nontemporal 8-byte loads through a wave-private LDS strip, keeps x / y at 8 bytes per lane and
//               deals tiles to the XCDs in runs;
//   row-direct  (smaller grids) one thread per row on a 2-D index space, the reference's own shape without its divisions;
//   row-generic (slabs that are not whole grid rows, unverified or non-stencil matrices) one thread per row, the
//               reference's kernel as it stands: analytic interior rows where the structure was verified, the CSR loop else.
// The shapes measured and closed in rounds 1-4 (wave-tile, column-march, row-lds march of 2 / 4 rows, five-plane
// coefficients) were removed in round 5; their tables are in profiles/ (DESIGN.md section 8 names the files) and their
// code in the history up to commit 258dcef.
// blockIdx -> tile mappings are performance choices only, never a correctness input.
#include "kernels.hpp"

#include <stdlib.h>

#include "reduce_device.hpp"
#include "stencil_geometry.hpp"

namespace spmv_amd {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
// Workgroups are dealt round-robin to the eight XCDs. logical_block() re-labels them so that each XCD works on
// `group` CONSECUTIVE logical blocks of every run of 8 * group (the row-lds finding, DESIGN.md section 3: with the
// same kernel body, which XCD touches which addresses is worth ~5 %). The launcher pads the grid to a multiple of
// 8 * group; blocks relabelled past `total` return. group <= 1: identity.
__device__ __forceinline__ long long logical_block(int group, long long total) {
    const long long b = blockIdx.x;
    if (group <= 1) return b;
    const long long span = 8LL * group;
    const long long t = (b / span) * span + (b & 7) * group + ((b >> 3) % group);
    return t < total ? t : -1;
}

// x[col - row_offset] if that index is readable, else 0 (halo kernel semantics,
// reference src/spmv/spmv_stencil_partitioned_halo_kernel.cu:77-94).
__device__ __forceinline__ double x_at(const double* __restrict__ x, long long local_col,
                                       int lo, int hi) {
    return (local_col >= lo && local_col < hi) ? x[local_col] : 0.0;
}

// One row, evaluated the way the reference kernels do: computed offset and computed
// columns for interior rows of a verified stencil, the CSR loop otherwise.
template <bool kAnalyticInterior>
__device__ __forceinline__ double row_reference(const SlabCsr& m, const double* __restrict__ x,
                                                int local_row, int i, int j) {
    const int n = m.grid_size;
    if (kAnalyticInterior && stencil_is_interior(i, j, n)) {
        const long long o = stencil_row_start(i, j, n) - m.nnz_base;
        const double* __restrict__ v = m.values + o;
        const double* __restrict__ xl = x + local_row;
        double sum = v[1] * xl[-1];
        sum = fma(v[2], xl[0], sum);
        sum = fma(v[3], xl[1], sum);
        sum = fma(v[0], xl[-n], sum);
        sum = fma(v[4], xl[n], sum);
        return sum;
    }
    const int lo = -m.halo_before, hi = m.n_local + m.halo_after;
    const int k0 = m.row_ptr[local_row], k1 = m.row_ptr[local_row + 1];
    double sum = 0.0;
    for (int k = k0; k < k1; ++k)
        sum = fma(m.values[k], x_at(x, (long long)m.col_idx[k] - m.row_offset, lo, hi), sum);
    return sum;
}


// ---------------------------------------------------------------------------------
// STENCIL5, row-direct variant (slabs made of whole grid rows, grids below 512 columns): one thread per row, a workgroup
// is 256 columns of ONE grid row, so the grid coordinates and the row's CSR offset cost no integer division. Everything
// else is the reference's own shape (src/spmv/spmv_stencil_csr_direct.cu:76-123): five strided 8-byte loads of the row's
// coefficients at the computed offset (plain loads on purpose: the five strided loads of a wave share cache lines through
// the vector L1; nontemporal loads bypass it and ran 5.10 ms instead of 3.95 ms at 20 000^2), x[row-1..row+1] from the same
// cache lines (taking W / E from the neighbouring lanes with __shfl measured slower) and x[row -+ n] from the lines the
// neighbouring grid rows pulled into L2 moments earlier. Blocks in dispatch order, column block fastest.
// ---------------------------------------------------------------------------------
template <bool kDot>
__global__ __launch_bounds__(kBlock) void stencil5_rowdirect_kernel(
    SlabCsr m, const double* __restrict__ x, double* __restrict__ y, double alpha, int gi_lo,
    int col_blocks, double* __restrict__ dot_partials, const int* __restrict__ skip_flag) {
    __shared__ double wave_part[kWavesPerBlock];
    if (skip_flag != nullptr && *skip_flag != 0) return;
    const int n = m.grid_size;
    const int row_group = (int)blockIdx.x / col_blocks;
    const int col_block = (int)blockIdx.x - row_group * col_blocks;
    const int li = gi_lo + row_group;                          // local grid row of this block
    const int gi = m.row_offset / n + li;                      // global grid row
    const int j = col_block * kBlock + (int)threadIdx.x;       // grid column
    double dot_acc = 0.0;
    if (j < n) {
        const long long lr = (long long)li * n + j;
        const double* __restrict__ xl = x + lr;
        double sum;
        if (j > 0 && j < n - 1 && gi > 0 && gi < n - 1) {
            const double* __restrict__ v = m.values + (stencil_gridrow_base(gi, n) + 5LL * j - 1 - m.nnz_base);
            const double v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3], v4 = v[4];
            sum = v1 * xl[-1];
            sum = fma(v2, xl[0], sum);
            sum = fma(v3, xl[1], sum);
            sum = fma(v0, xl[-n], sum);
            sum = fma(v4, xl[n], sum);
        } else {
            sum = row_reference<false>(m, x, (int)lr, gi, j);
        }
        if (kDot) dot_acc = fma(xl[0], sum, dot_acc);
        y[lr] = alpha * sum;
    }
    if (kDot) {
        // one partial per block: wave tree, then the four wave sums in wave order
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dot_acc += __shfl_down(dot_acc, off);
        if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = dot_acc;
        __syncthreads();
        if (threadIdx.x == 0)
            dot_partials[blockIdx.x] = ((wave_part[0] + wave_part[1]) + wave_part[2]) + wave_part[3];
    }
}

// ---------------------------------------------------------------------------------
// STENCIL5, row-lds variant (default on slabs made of whole grid rows, n >= 512). Same index space
// as row-direct -- a workgroup is a run of columns of ONE grid row, so no thread divides -- but the
// coefficients take the streaming path the probes found fastest on MI355X
// (tools/stream_probe3.hip, profiles/r01_stream_probe3.txt):
//   * a workgroup is ONE wavefront that owns 128 consecutive columns, two rows per lane (columns
//     j0 + lane and j0 + 64 + lane): every x / y access stays an 8-byte-per-lane coalesced access;
//   * the tile's 640 coefficients are one contiguous run of `values` (five entries per interior
//     row): ten fully coalesced 8-byte NONTEMPORAL loads per lane -- the run is read once and never
//     again, so it should not displace x in L2 / Infinity Cache -- parked in a 5 KiB wave-private
//     LDS strip and read back as [N,W,C,E,S] per row (lane stride 40 B = 10 banks, an even stride
//     over 64 banks: conflict-free for 8-byte reads);
//   * y leaves with a nontemporal store;
//   * x: centre and N/S (the lines the neighbouring grid rows pull through L2 / Infinity Cache) by plain
//     loads: these are the re-used bytes. W/E come from a 1 KiB LDS copy of the tile's own centre values
//     (only the tile's two outer neighbours are loaded): four fewer vector-memory instructions per wave,
//     0.8 % at 20 000^2 and 2.6 % at 10 000^2, bit-identical results;
//   * tile -> XCD: workgroups are dealt round-robin to the eight XCDs, so tile = blockIdx would
//     scatter every 1 KiB of x / y over eight L2s. Instead each XCD takes `group` consecutive tiles
//     of every run of 8 * group (group = 4: 512 columns = 4 KiB of x and y, 20 KiB of values per
//     XCD and run). Measured at 20 000^2 with the real coefficient layout: group 1 3.89 ms,
//     group 2 4.04 ms, group 4 / 8 3.63 ms, group 16 3.73 ms -- a plateau, not a spike.
// Columns 0 and n-1 of an interior grid row (4-entry rows) are evaluated from the same strip in the
// CSR loop's order (ascending column, sum started at 0: reference :116-119); only the first and
// last grid row of the whole grid, whose rows have no N or S entry, walk row_ptr / col_idx.
// No barrier: the strip is private to the wave, and LDS operations of one wave retire in order.
// Arithmetic per interior row is exactly row-direct's (and the reference's) W,C,E,N,S fma chain.
// ---------------------------------------------------------------------------------
constexpr int kLdsTileCols = 128;

// kMode 0: y = alpha A x. kMode 1: also one partial of x . (A x) per wave (the CG loop's p.Ap).
// kMode 2 (first SpMV of a solve, x = the initial guess): y is NOT stored; instead r = b - A x (one fma, the
// reference's axpy_kernel(-1, Ap, b), mgpu :475), p = r, and one partial of r.r per wave -- the initial residual
// without writing A x0 out and reading it back (16 B/row less, once per solve).
//
// One tile = local grid row li (global gi), columns [j0, j0 + 128), evaluated by one wave on its private LDS (`strip`:
// 640 doubles, `xrow`: 130). Returns false if the launch was enqueued past convergence (`skip`: the flag's value,
// REQUESTED by the caller before this call and tested here only after the tile's loads have been issued -- a wave does not
// sit on a scalar-load round trip before its first vector load; such a launch only reads). *dot: the tile's partial.
// kFreshHalo (boundary tiles evaluated behind a device-side arrival flag instead of a kernel boundary, see
// stencil5_rowlds_edges_reduce_kernel): the north / south values -- the halo rows another GPU's data was received into -- are
// read with SYSTEM-scope loads: the writer is another agent (a peer GPU's stores over xGMI, or the RCCL receive kernel acting
// for it), and no cache of this device may answer from a line it held before those rows arrived (2 x grid_size
// values per iteration, nothing on the clock).
template <int kMode, bool kFreshHalo = false>
__device__ __forceinline__ bool rowlds_tile(const SlabCsr& m, const double* __restrict__ x, double* __restrict__ y, double alpha,
                                            int li, int gi, int j0, int lane, int skip, double* __restrict__ strip,
                                            double* __restrict__ xrow, const ResidualOut& res, double* dot) {
    constexpr bool kDot = kMode == 1;
    constexpr bool kInit = kMode == 2;
    const int n = m.grid_size;
    double dot_acc = 0.0;
    if (gi > 0 && gi < n - 1) {
        // slab-local position of the tile's first coefficient: row (gi, j) starts at base + 5 j - 1
        // for j >= 1; the run of the first tile starts one entry early so that row j sits at strip
        // position 5 j there too (row 0 itself holds [N,C,E,S] at positions 1..4)
        const long long e = stencil_gridrow_base(gi, n) + 5LL * j0 - 1 - m.nnz_base + lane;
        const double* __restrict__ vals = m.values;
        double c[10];
        if (j0 == 0 || j0 + kLdsTileCols > n - 1) {
            // first / last tile of the grid row: part of the run lies outside the row (at the slab's
            // ends: outside the array): clamp the addresses, those strip slots feed no row
            const long long hi = m.nnz_local - 1;
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                long long idx = e + 64 * k;
                idx = idx < 0 ? 0 : (idx > hi ? hi : idx);
                c[k] = __builtin_nontemporal_load(vals + idx);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 10; ++k) c[k] = __builtin_nontemporal_load(vals + e + 64 * k);
        }
        double xc[2], xw[2], xe[2], xn[2], xs[2], bv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = j0 + lane + 64 * h;
            xc[h] = xw[h] = xe[h] = xn[h] = xs[h] = bv[h] = 0.0;
            if (j < n) {
                const double* __restrict__ xl = x + ((long long)li * n + j);
                xc[h] = xl[0];
                if (kFreshHalo) {
                    xn[h] = published_by_any_agent(reinterpret_cast<const unsigned long long*>(xl - n));
                    xs[h] = published_by_any_agent(reinterpret_cast<const unsigned long long*>(xl + n));
                } else {
                    xn[h] = xl[-n], xs[h] = xl[n];
                }
                if (kInit) bv[h] = __builtin_nontemporal_load(res.b + ((long long)li * n + j));
                // only the tile's two outer neighbours come from memory; the rest from the LDS copy below
                if (h == 0 && lane == 0 && j > 0) xw[0] = xl[-1];
                if (h == 1 && lane == 63 && j < n - 1) xe[1] = xl[1];
            }
        }
        if (skip != 0) return false;
#pragma unroll
        for (int k = 0; k < 10; ++k) strip[64 * k + lane] = c[k];
        xrow[1 + lane] = xc[0];
        xrow[65 + lane] = xc[1];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // columns beyond n hold 0 in xc, exactly what an absent neighbour contributes
        if (lane > 0) xw[0] = xrow[lane];
        xe[0] = xrow[2 + lane];
        xw[1] = xrow[64 + lane];
        if (lane < 63) xe[1] = xrow[66 + lane];
        if (j0 + lane == n - 1) xe[0] = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = j0 + lane + 64 * h;
            if (j < n) {
                const double* __restrict__ v = strip + 5 * (lane + 64 * h);
                double sum;
                if (j > 0 && j < n - 1) {            // [N,W,C,E,S], evaluated W,C,E,N,S
                    sum = v[1] * xw[h];
                    sum = fma(v[2], xc[h], sum);
                    sum = fma(v[3], xe[h], sum);
                    sum = fma(v[0], xn[h], sum);
                    sum = fma(v[4], xs[h], sum);
                } else if (j == 0) {                 // [N,C,E,S] at strip positions 1..4, CSR-loop order
                    sum = fma(v[1], xn[h], 0.0);
                    sum = fma(v[2], xc[h], sum);
                    sum = fma(v[3], xe[h], sum);
                    sum = fma(v[4], xs[h], sum);
                } else {                             // j == n-1: [N,W,C,S], CSR-loop order
                    sum = fma(v[0], xn[h], 0.0);
                    sum = fma(v[1], xw[h], sum);
                    sum = fma(v[2], xc[h], sum);
                    sum = fma(v[3], xs[h], sum);
                }
                if (kDot) dot_acc = fma(xc[h], sum, dot_acc);
                if (kInit) {
                    const long long lr = (long long)li * n + j;
                    const double rv = fma(-1.0, alpha * sum, bv[h]);
                    __builtin_nontemporal_store(rv, res.r + lr);
                    res.p[lr] = rv;  // plain: the next SpMV's neighbour loads re-use these lines
                    dot_acc = fma(rv, rv, dot_acc);
                } else {
                    __builtin_nontemporal_store(alpha * sum, y + ((long long)li * n + j));
                }
            }
        }
    } else {
        // first / last grid row of the whole grid: every row the reference's way
        if (skip != 0) return false;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = j0 + lane + 64 * h;
            if (j < n) {
                const long long lr = (long long)li * n + j;
                const double sum = row_reference<false>(m, x, (int)lr, gi, j);
                if (kDot) dot_acc = fma(x[lr], sum, dot_acc);
                if (kInit) {
                    const double rv = fma(-1.0, alpha * sum, res.b[lr]);
                    res.r[lr] = rv;
                    res.p[lr] = rv;
                    dot_acc = fma(rv, rv, dot_acc);
                } else {
                    y[lr] = alpha * sum;
                }
            }
        }
    }
    if (kDot || kInit) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dot_acc += __shfl_down(dot_acc, off);
    }
    *dot = dot_acc;  // lane 0 holds the tile's partial
    return true;
}

// Tile of workgroup b: each XCD takes `run` consecutive tiles of every run of 8 * run (workgroups are dealt round-robin to
// the eight XCDs); the launcher pads the grid to a multiple of 8 * run, tiles past `total` do not exist (-1).
__device__ __forceinline__ int rowlds_tile_of_block(int b, int run, int total, int reverse) {
    const int span = 8 * run;
    int tile = (b / span) * span + (b & 7) * run + ((b >> 3) % run);
    if (tile >= total) return -1;
    return reverse ? total - 1 - tile : tile;  // same tiles, same partial slots, walked from the end
}

template <int kMode>
__global__ __launch_bounds__(64) void stencil5_rowlds_kernel(
    SlabCsr m, const double* __restrict__ x, double* __restrict__ y, double alpha, int gi_lo, int row_step,
    int gfirst, int col_tiles, int total_tiles, int run, int reverse, double* __restrict__ dot_partials,
    const int* __restrict__ skip_flag, ResidualOut res) {
    __shared__ double strip[5 * kLdsTileCols];
    __shared__ double xrow[kLdsTileCols + 2];  // the tile's x values: W / E are read back from here
    const int skip = skip_flag != nullptr ? __builtin_nontemporal_load(skip_flag) : 0;
    const int tile = rowlds_tile_of_block((int)blockIdx.x, run, total_tiles, reverse);
    if (tile < 0) return;
    const int row_group = tile / col_tiles;
    const int col_tile = tile - row_group * col_tiles;
    const int li = gi_lo + row_group * row_step;  // local grid row (row_step > 1: a launch over separate grid rows)
    double dot = 0.0;
    if (!rowlds_tile<kMode>(m, x, y, alpha, li, gfirst + li, col_tile * kLdsTileCols, (int)threadIdx.x, skip, strip, xrow, res, &dot)) return;
    if (kMode != 0 && threadIdx.x == 0) dot_partials[tile] = dot;
}

// ---------------------------------------------------------------------------------
// The boundary rows of a solver slab AND the reduction of the whole SpMV's p.Ap partials in ONE launch (round 5).
// A rank with neighbours computes the rows that read halo values -- its first and / or last grid row -- once the halo has
// landed, behind the interior rows' launch; rounds 2-4 then reduced the partials with two more launches. On the P = 8 slab of
// the headline grid that tail (event wait + boundary launch + slices + final sum) cost 29 us of an 890 us iteration
// (profiles/r05_slab_attribution.txt). Here workgroups [0, edge_blocks) evaluate four boundary tiles each (one per wave, the
// row-lds tile as it is) and publish the tiles' partials as extra values of the reduction; workgroups behind them are the
// reduction's slice workgroups over the INTERIOR launch's partials (independent of the halo: they run beside the tiles);
// whichever workgroup finishes last sums slice sums + boundary partials and runs the tail (reduce_device.hpp).
// Each row is evaluated by the code every other launch uses: results do not depend on the launch a row falls in.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kReduceBlock) void stencil5_rowlds_edges_reduce_kernel(
    SlabCsr m, const double* __restrict__ x, double* __restrict__ y, double alpha, int li_first, int li_step, int gfirst,
    int col_tiles, int edge_tiles, int edge_blocks, const double* __restrict__ interior_partials, int interior_count, int slice,
    int slice_blocks, ReduceStage stage, ReduceTail tail, HaloArrival halo) {
    __shared__ double strip[kWavesPerBlock][5 * kLdsTileCols];
    __shared__ double xrow[kWavesPerBlock][kLdsTileCols + 2];
    __shared__ double s[kReduceBlock];
    __shared__ int s_last;
    const int skip = tail.skip_flag != nullptr ? __builtin_nontemporal_load(tail.skip_flag) : 0;
    const int tickets = edge_blocks + slice_blocks;
    if ((int)blockIdx.x < edge_blocks) {
        const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
        const int tile = (int)blockIdx.x * kWavesPerBlock + wave;
        if (tile < edge_tiles) {
            const int row_group = tile / col_tiles;
            const int li = li_first + row_group * li_step;
            double dot = 0.0;
            const ResidualOut none{nullptr, nullptr, nullptr};
            bool live;
            if (halo.flag != nullptr) {
                // No kernel boundary orders this launch behind the halo exchange: the side stream raises `flag` to the
                // exchange's sequence number once the received rows are in memory, and every boundary wave waits for it here
                // (bounded: a neighbour that never sends must not leave waves spinning; the host reads *late and ends the run).
                if (skip == 0 && lane == 0) {
                    const long long t0 = wall_clock64();
                    while ((int)(__hip_atomic_load(halo.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - halo.expected) < 0) {
                        if (wall_clock64() - t0 > halo.timeout_ticks) {
                            __hip_atomic_store(halo.late, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
                // Ordering point for the whole wave (lanes 1-63 never ran the loop): no load of the tile below may be moved above
                // the wait by the compiler. Wavefront scope: no cache operation is emitted, the halo values are read past the
                // caches by their own system-scope loads.
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                live = rowlds_tile<1, true>(m, x, y, alpha, li, gfirst + li, (tile - row_group * col_tiles) * kLdsTileCols, lane, skip, strip[wave],
                                            xrow[wave], none, &dot);
            } else {
                live = rowlds_tile<1>(m, x, y, alpha, li, gfirst + li, (tile - row_group * col_tiles) * kLdsTileCols, lane, skip, strip[wave],
                                      xrow[wave], none, &dot);
            }
            if (live && lane == 0) publish(stage.extra + tile, dot);
        }
        if (skip != 0) {  // the same value in every workgroup of the launch
            reduce_skipped(tail, blockIdx.x == 0);
            return;
        }
        draw_ticket_and_finish_if_last(stage, slice_blocks, stage.extra, edge_tiles, tickets, tail, s, &s_last);
    } else {
        if (skip != 0) return;
        reduce_slice_block(interior_partials, interior_count, slice, (int)blockIdx.x - edge_blocks, slice_blocks, stage.extra, edge_tiles,
                           tickets, stage, tail, s, &s_last);
    }
}

// ---------------------------------------------------------------------------------
// STENCIL5, row-generic variant: one thread per row, the reference's own shape. Used for
// small grids, for matrices that are not a complete 5-point stencil (kAnalytic = false:
// every row takes the CSR loop, as the reference does when grid_size = -1).
// ---------------------------------------------------------------------------------
template <bool kAnalytic, bool kDot>
__global__ __launch_bounds__(kBlock) void stencil5_row_kernel(SlabCsr m, const double* __restrict__ x,
                                                              double* __restrict__ y, double alpha,
                                                              int first_row, int last_row,
                                                              double* __restrict__ dot_partials,
                                                              const int* __restrict__ skip_flag) {
    if (skip_flag != nullptr && *skip_flag != 0) return;
    const long long row = (long long)first_row + (long long)blockIdx.x * kBlock + threadIdx.x;
    double dot_acc = 0.0;
    if (row < last_row) {
        int i = -1, j = 0;
        if (kAnalytic) {
            const int g = m.row_offset + (int)row;
            i = g / m.grid_size;
            j = g - i * m.grid_size;
        }
        const double sum = row_reference<kAnalytic>(m, x, (int)row, i, j);
        if (kDot) dot_acc = x[row] * sum;
        y[row] = alpha * sum;
    }
    if (kDot) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dot_acc += __shfl_down(dot_acc, off);
        if ((threadIdx.x & 63) == 0) dot_partials[blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)] = dot_acc;
    }
}

// ---------------------------------------------------------------------------------
// CSR baseline kernels (reference operator "cusparse-csr": arithmetic in closed-source
// cuSPARSE; semantic model = csr_spmv_kernel, cg_solver_mgpu_partitioned.cu:40-56).
// ---------------------------------------------------------------------------------
// One thread per row. The row is walked in chunks of eight entries: all column indices and values
// of a chunk are requested first, then the eight x gathers, then the fused multiply-adds in
// ascending order -- the same sequential sum as csr_spmv_kernel, with eight independent loads in
// flight per lane instead of a col -> x -> fma chain per entry.
__global__ __launch_bounds__(kBlock) void csr_row_scalar_kernel(SlabCsr m, const double* __restrict__ x,
                                                                double* __restrict__ y, double alpha) {
    const long long row = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (row >= m.n_local) return;
    const int lo = -m.halo_before, hi = m.n_local + m.halo_after;
    const int k0 = m.row_ptr[row], k1 = m.row_ptr[row + 1];
    double sum = 0.0;
    for (int base = k0; base < k1; base += 8) {
        int c[8];
        double v[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool live = base + u < k1;
            c[u] = live ? m.col_idx[base + u] : m.row_offset;
            v[u] = live ? m.values[base + u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = x_at(x, (long long)c[u] - m.row_offset, lo, hi);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (base + u < k1) sum = fma(v[u], xv[u], sum);
    }
    y[row] = alpha * sum;
}

// CSR-stream: a block owns `rows_per_block` consecutive rows whose entries form one contiguous span
// of col_idx / values. Phase 1 walks the span with all 256 threads, consecutive lanes on
// consecutive entries (fully coalesced 4- and 8-byte loads, no dependence on the per-row row_ptr),
// gathers x and parks (value, x) in LDS; phase 2 gives each row to one thread, which folds its
// entries from LDS with fma in ascending order -- the sequential sum of csr_spmv_kernel, bit for
// bit. A span that does not fit the LDS strip (very long rows) is walked in strip-sized chunks,
// each thread carrying its row's running sum from chunk to chunk. This is the shape of the "stream" half of CSR-adaptive
// (Greathouse & Daga, SC'14), without a preprocessing pass: the row count per block is fixed per
// matrix from its mean row length.
// kThreads threads fetch kPerThread entries each in phase 1 (LDS: 16 B per entry).
// kVectorLong ("csr/adaptive", the "vector" half of CSR-adaptive): inside a row that is longer than the strip, a chunk that
// lies entirely within that row is not staged at all -- every thread folds the entries it fetched into a private partial,
// and the partials are summed by a fixed tree when the row ends. 20 coalesced passes of 256 threads instead of 20 x 1024
// sequential fmas by one thread; the price is the summation order of such rows (a tree, like the sub-wavefront kernels).
template <int kThreads, int kPerThread, bool kVectorLong = false>
__global__ __launch_bounds__(kThreads) void csr_stream_kernel(SlabCsr m, const double* __restrict__ x,
                                                              double* __restrict__ y, double alpha,
                                                              int rows_per_block, int xcd_group, int total_blocks,
                                                              double* __restrict__ dot_partials) {
    constexpr int kCsrStreamCap = kThreads * kPerThread;
    constexpr int kCsrStreamPerThread = kPerThread;
    __shared__ double sv[kCsrStreamCap];
    __shared__ double sx[kCsrStreamCap];
    __shared__ int s_owner[2];  // by chunk parity: a slot is rewritten two chunks (two barriers) after it was read
    __shared__ double s_wave[kThreads / 64];
    const int lo = -m.halo_before, hi = m.n_local + m.halo_after;
    const long long blk = logical_block(xcd_group, total_blocks);
    if (blk < 0) return;
    const long long r0 = blk * rows_per_block;
    const long long r1 = min(r0 + rows_per_block, (long long)m.n_local);
    const int kb = m.row_ptr[r0], ke = m.row_ptr[r1];
    const long long row = r0 + threadIdx.x;
    const bool has_row = (int)threadIdx.x < rows_per_block && row < r1;
    int k0 = 0, k1 = 0;
    if (has_row) {
        k0 = m.row_ptr[row];
        k1 = m.row_ptr[row + 1];
    }
    double row_sum = 0.0;  // this thread's row, unscaled
    if (ke - kb <= kCsrStreamCap) {
        // phase 1, unrolled: all index and value loads first, then the gathers, then LDS
        int c[kCsrStreamPerThread];
        double v[kCsrStreamPerThread], xv[kCsrStreamPerThread];
#pragma unroll
        for (int u = 0; u < kCsrStreamPerThread; ++u) {
            const int e = kb + (int)threadIdx.x + u * kThreads;
            const bool live = e < ke;
            // plain loads: nontemporal ones measured 9 % slower here (block spans are not line-aligned and
            // neighbouring blocks share their edge lines; profiles/r01_csr_ell_nt_ab.txt)
            c[u] = live ? m.col_idx[e] : m.row_offset;
            v[u] = live ? m.values[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kCsrStreamPerThread; ++u) xv[u] = x_at(x, (long long)c[u] - m.row_offset, lo, hi);
#pragma unroll
        for (int u = 0; u < kCsrStreamPerThread; ++u) {
            sv[threadIdx.x + u * kThreads] = v[u];
            sx[threadIdx.x + u * kThreads] = xv[u];
        }
        __syncthreads();
        if (has_row) {
            for (int k = k0 - kb; k < k1 - kb; ++k) row_sum = fma(sv[k], sx[k], row_sum);
            y[row] = alpha * row_sum;
        }
    } else {
        // The block's span does not fit the strip (long rows): walk it in strip-sized chunks. Every chunk is fetched like
        // the single-chunk case -- coalesced index / value loads, gathers, LDS -- and every thread folds the part of ITS
        // row that lies in the chunk, continuing its running sum: still the sequential sum of csr_spmv_kernel, bit for bit,
        // but a 20 000-entry row now costs its thread 20 passes over LDS instead of 2 500 dependent trips to memory.
        // (Round 2 walked such rows with the chunked thread-per-row loop, one thread serialising the whole row.)
        double sum = 0.0;
        double vacc = 0.0;    // kVectorLong: this thread's share of the long row the block is walking through
        int long_owner = -1;  // the thread that owns that row (block-uniform), -1 outside such a row
        // Adds the block's private partials to the owner's running sum: wave trees, then the wave sums in wave order.
        auto flush_long_row = [&] {
            double w = vacc;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off);
            if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = w;
            __syncthreads();
            if ((int)threadIdx.x == long_owner) {
                double t = s_wave[0];
                for (int q = 1; q < kThreads / 64; ++q) t += s_wave[q];
                sum += t;
            }
            __syncthreads();
            vacc = 0.0;
            long_owner = -1;
        };
        int parity = 0;
        for (int base = kb; base < ke; base += kCsrStreamCap, parity ^= 1) {
            const int end = min(base + kCsrStreamCap, ke);
            int owner = -1;
            if (kVectorLong) {  // does ONE row cover this whole chunk? (rows are disjoint: at most one thread says yes)
                if (threadIdx.x == 0) s_owner[parity] = -1;
                __syncthreads();
                if (has_row && k0 <= base && k1 >= end) s_owner[parity] = (int)threadIdx.x;
                __syncthreads();
                owner = s_owner[parity];
                if (long_owner >= 0 && owner != long_owner) flush_long_row();
            }
            int c[kCsrStreamPerThread];
            double v[kCsrStreamPerThread], xv[kCsrStreamPerThread];
#pragma unroll
            for (int u = 0; u < kCsrStreamPerThread; ++u) {
                const int e = base + (int)threadIdx.x + u * kThreads;
                const bool live = e < end;
                c[u] = live ? m.col_idx[e] : m.row_offset;
                v[u] = live ? m.values[e] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kCsrStreamPerThread; ++u) {
                // a dead slot (past the chunk's end) gathers nothing and contributes +0.0: with x[0] in its place a
                // non-finite x[0] would turn 0 * x[0] into NaN in the no-staging fold below, for a row that has no column 0
                const bool live = base + (int)threadIdx.x + u * kThreads < end;
                xv[u] = live ? x_at(x, (long long)c[u] - m.row_offset, lo, hi) : 0.0;
            }
            if (kVectorLong && owner >= 0) {
                // the whole chunk belongs to one row: no staging, no barrier (dead slots carry v = 0 and x = 0)
#pragma unroll
                for (int u = 0; u < kCsrStreamPerThread; ++u) vacc = fma(v[u], xv[u], vacc);
                long_owner = owner;
                continue;
            }
#pragma unroll
            for (int u = 0; u < kCsrStreamPerThread; ++u) {
                sv[threadIdx.x + u * kThreads] = v[u];
                sx[threadIdx.x + u * kThreads] = xv[u];
            }
            __syncthreads();
            if (has_row) {
                const int from = max(k0, base), to = min(k1, end);
                for (int k = from; k < to; ++k) sum = fma(sv[k - base], sx[k - base], sum);
            }
            __syncthreads();  // the strip is overwritten by the next chunk
        }
        if (kVectorLong && long_owner >= 0) flush_long_row();
        if (has_row) y[row] = alpha * sum;
        row_sum = sum;
    }
    if (dot_partials != nullptr) {
        // x . (A x) of the block's rows (the CG loop's p.Ap, cg_single / cg_slab): one partial per logical block, wave
        // trees then the wave sums in wave order -- a fixed shape. Square operators only: x is indexed by the row.
        double d = has_row ? x[row] * row_sum : 0.0;
        __syncthreads();  // s_wave may still be read by a flush above
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off);
        if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = s_wave[0];
            for (int q = 1; q < kThreads / 64; ++q) t += s_wave[q];
            dot_partials[blk] = t;
        }
    }
}

// One row per wavefront (north_star's "plain coalesced row-per-wavefront kernel"): lanes stride the row's entries, so
// consecutive lanes read consecutive col_idx / values. The per-lane partial sums are combined by a fixed shuffle tree, so
// the summation order differs from the sequential loop (results agree to rounding; exact on the integer-valued benchmark
// inputs). The pick for matrices whose MEAN row length exceeds 192 (csr_auto_variant); 8.3 ms at 5 entries per row.
__global__ __launch_bounds__(kBlock) void csr_wavefront_kernel(SlabCsr m, const double* __restrict__ x,
                                                               double* __restrict__ y, double alpha) {
    const long long row = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63);
    const int lo = -m.halo_before, hi = m.n_local + m.halo_after;
    double sum = 0.0;
    if (row < m.n_local) {
        const int k1 = m.row_ptr[row + 1];
        for (int k = m.row_ptr[row] + lane; k < k1; k += 64)
            sum = fma(m.values[k], x_at(x, (long long)m.col_idx[k] - m.row_offset, lo, hi), sum);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (lane == 0 && row < m.n_local) y[row] = alpha * sum;
}

// ---------------------------------------------------------------------------------
// ELLPACK. Device layout is slot-major (element (r,k) at [k*rows + r]) so that one thread per
// row reads every array with unit stride across lanes; padding slots carry index -1.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void ell_transpose_kernel(int rows, int width,
                                                               const int* __restrict__ idx_rm,
                                                               const double* __restrict__ val_rm,
                                                               int* __restrict__ idx_sm,
                                                               double* __restrict__ val_sm) {
    const long long r = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (r >= rows) return;
    for (int k = 0; k < width; ++k) {
        idx_sm[(long long)k * rows + r] = idx_rm[r * (long long)width + k];
        val_sm[(long long)k * rows + r] = val_rm[r * (long long)width + k];
    }
}

template <bool kNt>
__device__ __forceinline__ double ell_row_walk(int rows, int width, const int* __restrict__ idx,
                                               const double* __restrict__ val,
                                               const double* __restrict__ x, long long r) {
    // slots in chunks of eight: indices and values first, then the gathers, then the sum in slot order
    double sum = 0.0;
    for (int base = 0; base < width; base += 8) {
        int c[8];
        double v[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool live = base + u < width;
            // slot-major planes are read once per SpMV: nontemporal, so they do not displace x
            const long long at = (long long)(base + u) * rows + r;
            c[u] = !live ? -1 : kNt ? __builtin_nontemporal_load(idx + at) : idx[at];
            v[u] = !live ? 0.0 : kNt ? __builtin_nontemporal_load(val + at) : val[at];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = c[u] >= 0 ? x[c[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c[u] >= 0) sum = fma(v[u], xv[u], sum);
    }
    return sum;
}

__device__ __forceinline__ double ell_finish(double alpha, double beta, double sum, double y_old) {
    return beta == 0.0 ? alpha * sum : fma(alpha, sum, beta * y_old);
}

// x . (A x) of one workgroup's rows into *out (the CG loop's p.Ap when cg_solve_device drives an ELLPACK operator): wave
// trees, then the wave sums in wave order. Every thread of the workgroup must call it (it has a barrier).
template <int kEllBlock>
__device__ __forceinline__ void ell_block_dot(double d, double* __restrict__ out) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off);
    if (kEllBlock == 64) {
        if (threadIdx.x == 0) *out = d;
        return;
    }
    __shared__ double s_wave[kEllBlock / 64];
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = s_wave[0];
        for (int q = 1; q < kEllBlock / 64; ++q) t += s_wave[q];
        *out = t;
    }
}

// Workgroup size and nontemporal plane loads / y store are template parameters chosen by the launcher
// from measurements (see launch_ell_spmv).
template <int kEllBlock, bool kNt>
__global__ __launch_bounds__(kEllBlock) void ell_spmv_kernel(int rows, int width, const int* __restrict__ idx,
                                                             const double* __restrict__ val,
                                                             const double* __restrict__ x,
                                                             double* __restrict__ y, double alpha,
                                                             double beta, int xcd_group, int total_blocks,
                                                             double* __restrict__ dot_partials) {
    const long long blk = logical_block(xcd_group, total_blocks);
    if (blk < 0) return;
    const long long r = blk * kEllBlock + threadIdx.x;
    if (r >= rows && dot_partials == nullptr) return;
    double sum = 0.0;
    if (r < rows) {
        sum = ell_row_walk<kNt>(rows, width, idx, val, x, r);
        const double out = ell_finish(alpha, beta, sum, beta == 0.0 ? 0.0 : y[r]);
        if (kNt) __builtin_nontemporal_store(out, y + r);
        else y[r] = out;
    }
    if (dot_partials != nullptr) ell_block_dot<kEllBlock>(r < rows ? x[r] * sum : 0.0, dot_partials + blk);
}

// Interior rows of a stencil stored as ELL: slots [N,W,C,E,S], columns computed, indices
// never read (the contract of reference include/spmv_stencil.h:25-42).
template <int kEllBlock, bool kNt>
__global__ __launch_bounds__(kEllBlock) void ell_stencil5_kernel(int rows, int width, int n,
                                                                 const int* __restrict__ idx,
                                                                 const double* __restrict__ val,
                                                                 const double* __restrict__ x,
                                                                 double* __restrict__ y, double alpha,
                                                                 double beta, int xcd_group, int total_blocks,
                                                                 double* __restrict__ dot_partials) {
    const long long blk = logical_block(xcd_group, total_blocks);
    if (blk < 0) return;
    const long long r = blk * kEllBlock + threadIdx.x;
    if (r >= rows && dot_partials == nullptr) return;
    double dot = 0.0;
    if (r < rows) {
        const int i = (int)(r / n), j = (int)(r - (long long)i * n);
        double sum;
        if (width >= 5 && stencil_is_interior(i, j, n)) {
            const double* __restrict__ v = val + r;
            const long long R = rows;
            const double v0 = kNt ? __builtin_nontemporal_load(v) : v[0], v1 = kNt ? __builtin_nontemporal_load(v + R) : v[R],
                         v2 = kNt ? __builtin_nontemporal_load(v + 2 * R) : v[2 * R],
                         v3 = kNt ? __builtin_nontemporal_load(v + 3 * R) : v[3 * R],
                         v4 = kNt ? __builtin_nontemporal_load(v + 4 * R) : v[4 * R];
            const double xc = x[r];
            sum = v1 * x[r - 1];
            sum = fma(v2, xc, sum);
            sum = fma(v3, x[r + 1], sum);
            sum = fma(v0, x[r - n], sum);
            sum = fma(v4, x[r + n], sum);
            dot = xc * sum;
        } else {
            sum = ell_row_walk<kNt>(rows, width, idx, val, x, r);
            if (dot_partials != nullptr) dot = x[r] * sum;
        }
        const double out = ell_finish(alpha, beta, sum, beta == 0.0 ? 0.0 : y[r]);
        if (kNt) __builtin_nontemporal_store(out, y + r);
        else y[r] = out;
    }
    if (dot_partials != nullptr) ell_block_dot<kEllBlock>(dot, dot_partials + blk);  // one call site: every thread of the workgroup arrives
}

// ---------------------------------------------------------------------------------
// Structure: generator and verifier of the complete 5-point pattern
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void generate_stencil5_kernel(int n, int row_offset, int n_local,
                                                                   long long nnz_base, double center,
                                                                   double off, int* __restrict__ row_ptr,
                                                                   int* __restrict__ col_idx,
                                                                   double* __restrict__ values) {
    const long long lr = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (lr > n_local) return;
    const long long g = (long long)row_offset + lr;
    if (lr == n_local) {
        row_ptr[lr] = (int)(stencil_row_start_flat(g, n) - nnz_base);
        return;
    }
    const int i = (int)(g / n), j = (int)(g - (long long)i * n);
    long long k = stencil_row_start(i, j, n) - nnz_base;
    row_ptr[lr] = (int)k;
    if (i > 0) col_idx[k] = (int)(g - n), values[k] = off, ++k;
    if (j > 0) col_idx[k] = (int)(g - 1), values[k] = off, ++k;
    col_idx[k] = (int)g, values[k] = center, ++k;
    if (j < n - 1) col_idx[k] = (int)(g + 1), values[k] = off, ++k;
    if (i < n - 1) col_idx[k] = (int)(g + n), values[k] = off, ++k;
}

__global__ __launch_bounds__(kBlock) void verify_stencil5_kernel(SlabCsr m, int* __restrict__ mismatch) {
    const long long lr = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (lr >= m.n_local) return;
    const int n = m.grid_size;
    const long long g = (long long)m.row_offset + lr;
    const int i = (int)(g / n), j = (int)(g - (long long)i * n);
    const long long start = stencil_row_start(i, j, n) - m.nnz_base;
    bool ok = m.row_ptr[lr] == start && m.row_ptr[lr + 1] - m.row_ptr[lr] == stencil_row_nnz(i, j, n);
    if (ok) {
        long long k = start;
        if (i > 0) ok = ok && m.col_idx[k++] == g - n;
        if (j > 0) ok = ok && m.col_idx[k++] == g - 1;
        ok = ok && m.col_idx[k++] == g;
        if (j < n - 1) ok = ok && m.col_idx[k++] == g + 1;
        if (i < n - 1) ok = ok && m.col_idx[k++] == g + n;
    }
    if (!ok) *mismatch = 1;  // benign race: every writer stores the same value
}

inline unsigned blocks_for(long long items) { return (unsigned)((items + kBlock - 1) / kBlock); }

}  // namespace

// ===================================================================================
// launchers
// ===================================================================================

void launch_generate_stencil5_csr(int n, int row_offset, int n_local, long long nnz_base,
                                  double center, double off, int* row_ptr, int* col_idx,
                                  double* values, hipStream_t stream) {
    hipLaunchKernelGGL(generate_stencil5_kernel, dim3(blocks_for((long long)n_local + 1)),
                       dim3(kBlock), 0, stream, n, row_offset, n_local, nnz_base, center, off,
                       row_ptr, col_idx, values);
}

void launch_verify_stencil5_csr(const SlabCsr& m, int* d_mismatch, hipStream_t stream) {
    if (m.n_local == 0) return;
    hipLaunchKernelGGL(verify_stencil5_kernel, dim3(blocks_for(m.n_local)), dim3(kBlock), 0, stream,
                       m, d_mismatch);
}

// Consecutive row-lds tiles one XCD takes of every run of 8 * run tiles. A run a little longer than one grid row lets an XCD
// find the x lines of the rows above / below in its own L2 (it fetched them one grid row earlier); one that exceeds the grid
// row by a whole XCD share hands every column to the NEXT XCD row after row and re-uses nothing. Measured per grid with
// everything else held still (profiles/r05_size_sweep.txt): optimum at 0.92-1.07 grid rows, and spiky -- 12 500^2: 13 tiles
// 1.386 ms, 14 tiles (round 2's "one grid row + ~1100 columns") 1.476 ms. The rule below is the starting point; operators
// and solver slabs of >= 16 Mi rows time its neighbours at set-up and keep the fastest (tune_rowlds_xcd_run).
int rowlds_xcd_run_rule(int n) {
    if (n < 8000) return 4;  // small grids keep short runs (4096^2: 0.150 vs 0.165 ms)
    const int tiles = (n + kLdsTileCols - 1) / kLdsTileCols;
    const int run = (int)(1.05 * tiles / 8.0 + 0.5);
    return run < 1 ? 1 : (run > 64 ? 64 : run);
}

Stencil5Plan plan_stencil5(const SlabCsr& m, int first_row, int last_row, Stencil5Variant variant,
                           const LaunchShape& shape) {
    Stencil5Plan p;
    const Tunables& knobs = shape.knobs;
    const int n = m.grid_size;
    // one workgroup = columns of ONE grid row: the slab and the launch range must be made of whole grid rows
    const bool direct_ok = m.verified_stencil && n >= 2 && m.row_offset % n == 0 &&
                           m.n_local % n == 0 && first_row % n == 0 && last_row % n == 0;
    if (variant == Stencil5Variant::Auto)
        // row-lds needs grid rows long enough that the two clamped edge tiles are a small share
        variant = !direct_ok ? Stencil5Variant::RowGeneric : n >= knobs.rowlds_min_grid ? Stencil5Variant::RowLds : Stencil5Variant::RowDirect;
    if (variant != Stencil5Variant::RowGeneric && !direct_ok) variant = Stencil5Variant::RowGeneric;
    p.variant = variant;
    if (variant == Stencil5Variant::RowGeneric) {
        p.row_blocks = (int)blocks_for((long long)last_row - first_row);
        p.partials = p.row_blocks * kWavesPerBlock;
        p.name = m.verified_stencil ? "stencil5/row-generic" : "stencil5/row-generic(csr-loop)";
    } else {
        p.gi_lo = first_row / n;
        p.gi_hi = last_row / n;
        if (variant == Stencil5Variant::RowDirect) {
            p.row_blocks = (int)blocks_for(n);  // column blocks per grid row
            p.name = "stencil5/row-direct";
        } else {
            p.row_blocks = (n + kLdsTileCols - 1) / kLdsTileCols;  // column tiles (= workgroups) per grid row
            p.xcd_run = knobs.rowlds_group > 0 ? knobs.rowlds_group : rowlds_xcd_run_rule(n);
            if (p.xcd_run < 1 || p.xcd_run > 64) p.xcd_run = 4;
            p.name = "stencil5/row-lds";
        }
        p.partials = p.row_blocks * (p.gi_hi - p.gi_lo);  // < 2^31 for any int32 CSR
    }
    p.first_row = first_row;
    p.last_row = last_row;
    return p;
}

// One-off form (plans, then launches): for callers outside a loop.
int launch_stencil5_spmv(const SlabCsr& m, const double* x, double* y, double alpha,
                         int first_row, int last_row, double* d_dot_partials,
                         const int* d_skip_flag, Stencil5Variant variant,
                         const LaunchShape& shape, hipStream_t stream) {
    if (last_row <= first_row) return 0;
    return launch_stencil5_spmv(m, plan_stencil5(m, first_row, last_row, variant, shape), x, y, alpha, d_dot_partials,
                                d_skip_flag, shape.reverse, stream);
}

namespace {
// the row-lds kernel over `tiles` tiles: `row_step` grid rows between consecutive row groups (1: a contiguous range)
void launch_rowlds(const SlabCsr& m, const Stencil5Plan& p, const double* x, double* y, double alpha, int gi_lo, int row_step,
                   int tiles, double* d_dot_partials, const int* d_skip_flag, bool reverse, hipStream_t stream, const ResidualOut* init) {
    const int span = 8 * p.xcd_run;
    const dim3 grid((unsigned)(((long long)tiles + span - 1) / span * span));
    const int gfirst = m.row_offset / m.grid_size;
    const ResidualOut res = init ? *init : ResidualOut{nullptr, nullptr, nullptr};
#define SPMV_AMD_LAUNCH_ROWLDS(MODE)                                                                                   \
    hipLaunchKernelGGL((stencil5_rowlds_kernel<MODE>), grid, dim3(64), 0, stream, m, x, y, alpha, gi_lo, row_step, gfirst, \
                       p.row_blocks, tiles, p.xcd_run, reverse ? 1 : 0, d_dot_partials, d_skip_flag, res)
    if (init) SPMV_AMD_LAUNCH_ROWLDS(2);
    else if (d_dot_partials) SPMV_AMD_LAUNCH_ROWLDS(1);
    else SPMV_AMD_LAUNCH_ROWLDS(0);
#undef SPMV_AMD_LAUNCH_ROWLDS
}
}  // namespace

int launch_stencil5_spmv(const SlabCsr& m, const Stencil5Plan& p, const double* x, double* y, double alpha,
                         double* d_dot_partials, const int* d_skip_flag, bool reverse, hipStream_t stream,
                         const ResidualOut* init) {
    if (p.last_row <= p.first_row) return 0;
    if (init != nullptr && (p.variant != Stencil5Variant::RowLds || d_dot_partials == nullptr)) {
        fprintf(stderr, "[spmv] the fused initial residual exists for the row-lds kernel only\n");
        exit(EXIT_FAILURE);
    }
    const bool dot = d_dot_partials != nullptr;
    if (p.variant == Stencil5Variant::RowLds) {
        launch_rowlds(m, p, x, y, alpha, p.gi_lo, 1, p.partials, d_dot_partials, d_skip_flag, reverse, stream, init);
    } else if (p.variant == Stencil5Variant::RowDirect) {
        const dim3 grid((unsigned)p.partials);
        if (dot)
            hipLaunchKernelGGL((stencil5_rowdirect_kernel<true>), grid, dim3(kBlock), 0, stream, m, x, y, alpha, p.gi_lo, p.row_blocks,
                               d_dot_partials, d_skip_flag);
        else
            hipLaunchKernelGGL((stencil5_rowdirect_kernel<false>), grid, dim3(kBlock), 0, stream, m, x, y, alpha, p.gi_lo, p.row_blocks,
                               d_dot_partials, d_skip_flag);
    } else {
        const dim3 grid((unsigned)p.row_blocks);
        const bool analytic = m.verified_stencil && m.grid_size >= 2;
#define SPMV_AMD_LAUNCH_ROWS(AN, DOT)                                                                           \
    hipLaunchKernelGGL((stencil5_row_kernel<AN, DOT>), grid, dim3(kBlock), 0, stream, m, x, y, alpha, p.first_row, \
                       p.last_row, d_dot_partials, d_skip_flag)
        if (analytic) {
            if (dot) SPMV_AMD_LAUNCH_ROWS(true, true);
            else SPMV_AMD_LAUNCH_ROWS(true, false);
        } else {
            if (dot) SPMV_AMD_LAUNCH_ROWS(false, true);
            else SPMV_AMD_LAUNCH_ROWS(false, false);
        }
#undef SPMV_AMD_LAUNCH_ROWS
    }
    return dot ? p.partials : 0;
}

int launch_stencil5_spmv_first_and_last_gridrow(const SlabCsr& m, const Stencil5Plan& head, const Stencil5Plan& tail, const double* x,
                                                double* y, double alpha, double* d_dot_partials, const int* d_skip_flag,
                                                hipStream_t stream, const ResidualOut* init) {
    const int n = m.grid_size;
    const int local_gridrows = n > 0 ? m.n_local / n : 0;
    if (head.variant != Stencil5Variant::RowLds || tail.variant != Stencil5Variant::RowLds || local_gridrows < 2) {
        // two launches over the two row ranges (any variant)
        int used = launch_stencil5_spmv(m, head, x, y, alpha, d_dot_partials, d_skip_flag, false, stream, init);
        used += launch_stencil5_spmv(m, tail, x, y, alpha, d_dot_partials ? d_dot_partials + used : nullptr, d_skip_flag, false, stream, init);
        return used;
    }
    // one launch: row group 0 is local grid row 0, row group 1 is the last local grid row; the partial slots
    // are those of the two separate launches back to back
    const int tiles = 2 * head.row_blocks;
    launch_rowlds(m, head, x, y, alpha, 0, local_gridrows - 1, tiles, d_dot_partials, d_skip_flag, false, stream, init);
    return d_dot_partials ? tiles : 0;
}

namespace {
__global__ void halo_arrived_kernel(unsigned* flag, unsigned sequence) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

void launch_halo_arrived(unsigned* d_flag, unsigned sequence, hipStream_t stream) {
    hipLaunchKernelGGL(halo_arrived_kernel, dim3(1), dim3(64), 0, stream, d_flag, sequence);
}

bool launch_stencil5_edges_and_reduce(const SlabCsr& m, const Stencil5Plan& interior, bool first_gridrow, bool last_gridrow, const double* x,
                                      double* y, double alpha, const double* d_interior_partials, double* d_out, const int* d_skip_flag,
                                      const ReduceScratch& scratch, int* host_progress, int progress_value, const PeerMailbox* mailbox,
                                      hipStream_t stream, const HaloArrival& halo) {
    const int n = m.grid_size;
    if (interior.variant != Stencil5Variant::RowLds || scratch.base == nullptr || (!first_gridrow && !last_gridrow) ||
        interior.partials <= 0 || n <= 0 || m.n_local % n != 0)
        return false;
    const int local_gridrows = m.n_local / n;
    const int col_tiles = interior.row_blocks;
    const int edge_tiles = (first_gridrow && last_gridrow ? 2 : 1) * col_tiles;
    if (local_gridrows < 3 || edge_tiles > kReduceExtraMax) return false;
    const int edge_blocks = (edge_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    int slice = 0, slice_blocks = 0;
    reduce_geometry(interior.partials, &slice, &slice_blocks);
    const ReduceTail tail{d_out, d_skip_flag, host_progress, progress_value, mailbox, StepArgs{nullptr, 0.0, nullptr, nullptr, 0, nullptr, 0}};
    hipLaunchKernelGGL(stencil5_rowlds_edges_reduce_kernel, dim3((unsigned)(edge_blocks + slice_blocks)), dim3(kReduceBlock), 0, stream, m, x, y,
                       alpha, first_gridrow ? 0 : local_gridrows - 1, local_gridrows - 1, m.row_offset / n, col_tiles, edge_tiles, edge_blocks,
                       d_interior_partials, interior.partials, slice, slice_blocks, reduce_stage_of(scratch.base), tail, halo);
    return true;
}

// Measured on MI355X, 10 000^2 stencil as CSR (5 nnz/row): stream 1.37 ms (256 threads x 4 entries, 176 rows per block;
// one-wave and 128-thread shapes, 3 / 5 / 6 entries per thread and one big LDS strip per 256 rows all slower), row-scalar
// 1.75 ms (chunked), 4 / 8 / 16 lanes per row 2.52 / 3.44 / 6.42 ms, one row per wavefront 8.34 ms. For scale:
// rocsparse_spmv (csr_adaptive) takes 1.33 ms (tools/rocsparse_compare.hip). The texture addresser is ~80 % busy in the
// thread-per-row kernels (rocprofv3 TA_BUSY_avr): the 20 / 40-byte lane strides of col_idx / values are what the coalesced
// phase 1 of the stream kernel removes.
CsrVariant csr_auto_variant(const SlabCsr& m) {
    // Round 3 (profiles/r03_generic_matrix_perf.txt), 10^8 entries each: uniform rows of 12 / 20 / 40 / 80 / 160 random
    // columns: the stream kernel is within 1-3 % of the best variant at EVERY length, with the sequential, bit-reproducible
    // row sum the sub-wavefront shapes give up; skewed (rows of 1-8 entries, one in a thousand with 2 000-20 000): stream
    // 4.2 ms, adaptive 2.8 ms. Round 4 (profiles/r04_generic_long_rows.txt), uniform rows of 320 / 640 / 1000: stream 0.85 /
    // 1.51 / 2.30 ms (the span leaves the strip and 16 of 256 threads fold sequentially), one row per wavefront 0.67 / 0.65 /
    // 0.64 ms; at 160 per row the two tie.
    // So: a MEAN row length above 192 -> one row per wavefront (its tree sum replaces the sequential one: 2e-15 relative,
    // and cg_solve_device then runs the unfused loop: SpMV, then a dot pass); else rows longer than the stream kernel's
    // strip -> adaptive; everything else -> stream.
    const double mean = m.n_local > 0 ? (double)m.nnz_local / (double)m.n_local : 0.0;
    if (mean > 192.0) return CsrVariant::Wavefront;
    return m.max_row_nnz > 1024 ? CsrVariant::Adaptive : CsrVariant::Stream;
}

namespace {
constexpr int kCsrStreamThreads = 256, kCsrStreamCap = 1024;  // 256 threads x 4 entries: the LDS strip of one block
// Launch geometry of the stream / adaptive kernels: rows per block and logical blocks (= dot partials of a fused launch).
void csr_stream_geometry(const SlabCsr& m, int* per_block_out, long long* blocks_out) {
    const long long rows = m.n_local;
    // rows per block: the mean span should fill about 90 % of the LDS strip, at most one row per thread
    const double avg = rows > 0 ? (double)m.nnz_local / rows : 1.0;
    int per_block = (int)(0.9 * kCsrStreamCap / (avg > 1.0 ? avg : 1.0));
    per_block = per_block > kCsrStreamThreads ? kCsrStreamThreads : (per_block < 16 ? 16 : per_block & ~15);
    *per_block_out = per_block;
    *blocks_out = (rows + per_block - 1) / per_block;
}
}  // namespace

int csr_fused_dot_partials(const SlabCsr& m, CsrVariant variant) {
    if (variant == CsrVariant::Auto) variant = csr_auto_variant(m);
    if ((variant != CsrVariant::Stream && variant != CsrVariant::Adaptive) || m.n_local == 0) return 0;
    int per_block = 0;
    long long blocks = 0;
    csr_stream_geometry(m, &per_block, &blocks);
    return blocks <= 0x7fffffffLL ? (int)blocks : 0;
}

void launch_csr_spmv(const SlabCsr& m, const double* x, double* y, double alpha,
                     CsrVariant variant, hipStream_t stream, double* d_dot_partials) {
    if (m.n_local == 0) return;
    if (variant == CsrVariant::Auto) variant = csr_auto_variant(m);
    const long long rows = m.n_local;
    switch (variant) {
        case CsrVariant::Adaptive:
        case CsrVariant::Stream: {
            int per_block = 0;
            long long blocks = 0;
            csr_stream_geometry(m, &per_block, &blocks);
            // dispatch order: relabelling blocks so that an XCD takes runs of consecutive blocks measured slower or neutral
            // here at 10 000^2 / 15 000^2 / 20 000^2 (profiles/r02_xcd_group.txt, r04_csr_runs.txt, r04_csr_runs_20k.txt)
            const dim3 grid((unsigned)blocks);
            if (variant == CsrVariant::Adaptive)
                hipLaunchKernelGGL((csr_stream_kernel<256, 4, true>), grid, dim3(256), 0, stream, m, x, y, alpha, per_block, 1, (int)blocks, d_dot_partials);
            else
                hipLaunchKernelGGL((csr_stream_kernel<256, 4, false>), grid, dim3(256), 0, stream, m, x, y, alpha, per_block, 1, (int)blocks, d_dot_partials);
            break;
        }
        case CsrVariant::RowScalar:
            hipLaunchKernelGGL(csr_row_scalar_kernel, dim3(blocks_for(rows)), dim3(kBlock), 0, stream, m, x, y, alpha);
            break;
        default:
            hipLaunchKernelGGL(csr_wavefront_kernel, dim3((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, stream, m, x, y, alpha);
            break;
    }
}

void launch_ell_transpose(int rows, int width, const int* idx_rowmajor, const double* val_rowmajor,
                          int* idx_slotmajor, double* val_slotmajor, hipStream_t stream) {
    if (rows == 0) return;
    hipLaunchKernelGGL(ell_transpose_kernel, dim3(blocks_for(rows)), dim3(kBlock), 0, stream, rows,
                       width, idx_rowmajor, val_rowmajor, idx_slotmajor, val_slotmajor);
}

// ELLPACK launch shape: 256-thread workgroups, nontemporal planes and y. Measured on MI355X at 15 000^2 (generic /
// stencil-aware): 256 threads plain 3.22 / 2.32 ms, one-wave workgroups 3.00 / 2.51, 256 threads nontemporal 2.96 / 2.26,
// one-wave nontemporal 3.00 / 2.47 (round 1, profiles/r01_csr_ell_nt_ab.txt).
constexpr int kEllBlock = 256;

int ell_fused_dot_partials(int rows) { return (int)(((long long)rows + kEllBlock - 1) / kEllBlock); }

void launch_ell_spmv(int rows, int width, const int* idx, const double* val, const double* x,
                     double* y, double alpha, double beta, hipStream_t stream, int grid_hint,
                     double* d_dot_partials) {
    if (rows == 0) return;
    // each XCD takes `run` consecutive 256-row blocks of every run of 8 * run (15 000^2: 2.95-3.03 ms in dispatch
    // order, 2.86 ms with runs of 9; 10 000^2: 1.35 -> 1.30 ms; profiles/r02_xcd_group.txt)
    const int run = grid_hint > 0 ? xcd_run_group(grid_hint, kEllBlock, 7) + 1 : 8;
    const long long span = 8LL * run, blocks = ell_fused_dot_partials(rows);
    hipLaunchKernelGGL((ell_spmv_kernel<kEllBlock, true>), dim3((unsigned)((blocks + span - 1) / span * span)), dim3(kEllBlock), 0, stream, rows,
                       width, idx, val, x, y, alpha, beta, run, (int)blocks, d_dot_partials);
}

void launch_ell_stencil5_spmv(int rows, int width, int grid_size, const int* idx,
                              const double* val, const double* x, double* y, double alpha,
                              double beta, hipStream_t stream, double* d_dot_partials) {
    if (rows == 0) return;
    if (grid_size < 3 || (long long)grid_size * grid_size != rows) {
        launch_ell_spmv(rows, width, idx, val, x, y, alpha, beta, stream, 0, d_dot_partials);
        return;
    }
    // 15 000^2: 2.27-2.29 ms in dispatch order, 2.01 ms with runs of 8 blocks per XCD; 20 000^2: 4.01 -> 3.6-3.7 ms
    const int run = xcd_run_group(grid_size, kEllBlock, 8);
    const long long span = 8LL * run, blocks = ell_fused_dot_partials(rows);
    hipLaunchKernelGGL((ell_stencil5_kernel<kEllBlock, true>), dim3((unsigned)((blocks + span - 1) / span * span)), dim3(kEllBlock), 0, stream,
                       rows, width, grid_size, idx, val, x, y, alpha, beta, run, (int)blocks, d_dot_partials);
}

}  // namespace spmv_amd
