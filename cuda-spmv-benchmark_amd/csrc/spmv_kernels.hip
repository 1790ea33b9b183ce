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
// Memory contract: STENCIL5 is HBM-bound at 56 B per interior row (40 B values, 8 B x,
// 8 B y). Five kernels implement it; the default (row-lds, further down) streams `values`
// with fully coalesced nontemporal 8-byte loads through a wave-private LDS strip, keeps x / y
// at 8 bytes per lane and deals tiles to the XCDs in groups; the others (row-direct,
// column-march, wave-tile, row-generic) are the earlier shapes, kept selectable and tested.
// blockIdx -> tile mappings are performance choices only, never a correctness input.
#include "kernels.hpp"

#include <stdlib.h>

#include "stencil_geometry.hpp"

namespace spmv_amd {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
constexpr int kTileRows = 128;           // rows per wave-tile: two per lane
constexpr int kLdsDoublesPerWave = 656;  // 640 values + 2 alignment slack, 5248 B per wave

typedef double d2 __attribute__((ext_vector_type(2)));

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
// STENCIL5, wave-tile variant. One wave owns 128 consecutive rows (two per lane): `values` are
// fetched as five fully coalesced 16-byte loads per lane and transposed through a wave-private
// LDS strip, x/y move as 16-byte pairs. Either one tile per wave in dispatch order (default) or
// persistent waves walking the tiles of their XCD's band. Works on any row range, so it serves
// slabs that are not made of whole grid rows.
// ---------------------------------------------------------------------------------
template <bool kVecXY, bool kVecNS, bool kDot>
__global__ __launch_bounds__(kBlock) void stencil5_wavetile_kernel(
    SlabCsr m, const double* __restrict__ x, double* __restrict__ y, double alpha, int first_row,
    int last_row, double* __restrict__ dot_partials, const int* __restrict__ skip_flag, int oneshot) {
    __shared__ __attribute__((aligned(16))) double lds[kWavesPerBlock * kLdsDoublesPerWave];
    if (skip_flag != nullptr && *skip_flag != 0) return;

    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    double* __restrict__ wlds = lds + wave_in_block * kLdsDoublesPerWave;
    const int n = m.grid_size;

    const int tile_first = first_row / kTileRows;
    const int tile_end = (last_row + kTileRows - 1) / kTileRows;
    const int per_band = (tile_end - tile_first + 7) >> 3;
    int band_lo = tile_first + (int)(blockIdx.x & 7) * per_band;
    int band_hi = min(band_lo + per_band, tile_end);
    int waves_per_band = (int)(gridDim.x >> 3) * kWavesPerBlock;
    int q = (int)(blockIdx.x >> 3) * kWavesPerBlock + wave_in_block;
    if (oneshot) {  // one tile per wave, tiles in dispatch order
        band_lo = tile_first;
        band_hi = tile_end;
        q = (int)blockIdx.x * kWavesPerBlock + wave_in_block;
        waves_per_band = 0x7fffffff - tile_end;
    }

    double dot_acc = 0.0;
    for (int t = band_lo + q; t < band_hi; t += waves_per_band) {
        const int l0 = t * kTileRows;
        const int g0 = m.row_offset + l0;
        const int i0 = g0 / n;
        const int j0 = g0 - i0 * n;
        const bool pure = l0 >= first_row && l0 + kTileRows <= last_row && i0 >= 1 &&
                          i0 <= n - 2 && j0 >= 1 && j0 + kTileRows - 1 <= n - 2;
        if (pure) {
            // -- values: 640 consecutive doubles starting at s0, fetched as aligned 16-byte pairs
            const long long s0 = stencil_gridrow_base(i0, n) + 5LL * j0 - 1 - m.nnz_base;
            const int sh = (int)(s0 & 1);
            const d2* __restrict__ vsrc = reinterpret_cast<const d2*>(m.values + (s0 - sh));
            const d2 c0 = vsrc[lane];
            const d2 c1 = vsrc[64 + lane];
            const d2 c2 = vsrc[128 + lane];
            const d2 c3 = vsrc[192 + lane];
            const d2 c4 = vsrc[256 + lane];
            d2 c5 = {0.0, 0.0};
            if (sh != 0 && lane == 0) c5 = vsrc[320];

            // -- x: centre pair, north pair, south pair, plus the two row-edge scalars
            const double* __restrict__ xl = x + l0 + 2 * lane;
            d2 xc, xn, xs;
            if (kVecXY) {
                xc = *reinterpret_cast<const d2*>(xl);
            } else {
                xc.x = xl[0];
                xc.y = xl[1];
            }
            if (kVecNS) {
                xn = *reinterpret_cast<const d2*>(xl - n);
                xs = *reinterpret_cast<const d2*>(xl + n);
            } else {
                xn.x = xl[-n];
                xn.y = xl[1 - n];
                xs.x = xl[n];
                xs.y = xl[n + 1];
            }
            double edge = 0.0;
            if (lane == 0) edge = xl[-1];
            if (lane == 63) edge = xl[2];

            // -- transpose values through the wave's LDS strip
            d2* __restrict__ w2 = reinterpret_cast<d2*>(wlds);
            w2[lane] = c0;
            w2[64 + lane] = c1;
            w2[128 + lane] = c2;
            w2[192 + lane] = c3;
            w2[256 + lane] = c4;
            if (sh != 0 && lane == 0) w2[320] = c5;
            __builtin_amdgcn_wave_barrier();
            const double* __restrict__ v = wlds + sh + 10 * lane;
            const double a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4];
            const double b0 = v[5], b1 = v[6], b2 = v[7], b3 = v[8], b4 = v[9];
            __builtin_amdgcn_wave_barrier();

            double w = __shfl_up(xc.y, 1);
            double e = __shfl_down(xc.x, 1);
            if (lane == 0) w = edge;
            if (lane == 63) e = edge;

            // row 2*lane: [N,W,C,E,S] = a0..a4 ; row 2*lane+1: b0..b4 ; order W,C,E,N,S
            double r0 = a1 * w;
            r0 = fma(a2, xc.x, r0);
            r0 = fma(a3, xc.y, r0);
            r0 = fma(a0, xn.x, r0);
            r0 = fma(a4, xs.x, r0);
            double r1 = b1 * xc.x;
            r1 = fma(b2, xc.y, r1);
            r1 = fma(b3, e, r1);
            r1 = fma(b0, xn.y, r1);
            r1 = fma(b4, xs.y, r1);
            if (kDot) {
                dot_acc = fma(xc.x, r0, dot_acc);
                dot_acc = fma(xc.y, r1, dot_acc);
            }
            double* __restrict__ yl = y + l0 + 2 * lane;
            if (kVecXY) {
                d2 out = {alpha * r0, alpha * r1};
                *reinterpret_cast<d2*>(yl) = out;
            } else {
                yl[0] = alpha * r0;
                yl[1] = alpha * r1;
            }
        } else {
            // tile touching a grid-row end, the first/last grid row, or the launch range's edge
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int lr = l0 + 2 * lane + e;
                if (lr >= first_row && lr < last_row) {
                    int j = j0 + 2 * lane + e;
                    int i = i0;
                    if (j >= n) {
                        j -= n;
                        ++i;
                    }
                    const double sum = row_reference<true>(m, x, lr, i, j);
                    if (kDot) dot_acc = fma(x[lr], sum, dot_acc);
                    y[lr] = alpha * sum;
                }
            }
        }
    }
    if (kDot) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dot_acc += __shfl_down(dot_acc, off);
        if (lane == 0) dot_partials[blockIdx.x * kWavesPerBlock + wave_in_block] = dot_acc;
    }
}

// ---------------------------------------------------------------------------------
// STENCIL5, column-march variant (the default on slabs made of whole grid rows).
// A wave owns a strip of 128 grid columns and marches down `rows_per_task` grid rows. The three
// x rows a stencil row needs (north, centre, south) stay in registers and rotate: every x element
// is fetched from memory exactly once per task (plus two start-up rows), whatever the caches do.
// `values` of the next grid row are prefetched into registers while the current row is computed.
// Four waves of a block take four adjacent strips, blocks are numbered strip-group fastest, so the
// chip sweeps bands of grid rows left to right and streams each array sequentially.
// Strips containing column 0 or n-1 (and any grid row that is a global boundary, which the
// launcher sends to the row kernel instead) evaluate rows the reference's way, row by row.
// ---------------------------------------------------------------------------------
struct ValueChunks {
    d2 c0, c1, c2, c3, c4, c5;
};

__device__ __forceinline__ void load_value_chunks(ValueChunks& v, const double* __restrict__ values,
                                                  long long s, int lane) {
    const int sh = (int)(s & 1);
    const d2* __restrict__ src = reinterpret_cast<const d2*>(values + (s - sh));
    v.c0 = src[lane];
    v.c1 = src[64 + lane];
    v.c2 = src[128 + lane];
    v.c3 = src[192 + lane];
    v.c4 = src[256 + lane];
    if (sh != 0 && lane == 0) v.c5 = src[320];
}

template <bool kVec, bool kDot>
__global__ __launch_bounds__(kBlock) void stencil5_colmarch_kernel(
    SlabCsr m, const double* __restrict__ x, double* __restrict__ y, double alpha, int gi_lo,
    int gi_hi, int rows_per_task, int strips, double* __restrict__ dot_partials,
    const int* __restrict__ skip_flag) {
    __shared__ __attribute__((aligned(16))) double lds[kWavesPerBlock * kLdsDoublesPerWave];
    if (skip_flag != nullptr && *skip_flag != 0) return;

    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    double* __restrict__ wlds = lds + wave_in_block * kLdsDoublesPerWave;
    const int n = m.grid_size;
    const int strip_groups = (strips + kWavesPerBlock - 1) / kWavesPerBlock;
    const int chunk = (int)blockIdx.x / strip_groups;
    const int strip = ((int)blockIdx.x - chunk * strip_groups) * kWavesPerBlock + wave_in_block;
    const int li0 = gi_lo + chunk * rows_per_task;  // local grid rows [li0, li1)
    const int li1 = min(li0 + rows_per_task, gi_hi);
    const int gfirst = m.row_offset / n;  // global grid row of the slab's first row
    const int j0 = strip * kTileRows;

    double dot_acc = 0.0;
    if (strip < strips && li0 < li1) {
        const bool pure = j0 >= 1 && j0 + kTileRows - 1 <= n - 2;
        if (pure) {
            long long s = stencil_gridrow_base(gfirst + li0, n) + 5LL * j0 - 1 - m.nnz_base;
            const long long s_step = 5LL * n - 2;
            const double* __restrict__ xl = x + ((long long)li0 * n + j0 + 2 * lane);
            double* __restrict__ yl = y + ((long long)li0 * n + j0 + 2 * lane);
            auto load_pair = [](const double* __restrict__ p) {
                d2 v;
                if (kVec) {
                    v = *reinterpret_cast<const d2*>(p);
                } else {
                    v.x = p[0];
                    v.y = p[1];
                }
                return v;
            };
            d2 xn = load_pair(xl - n);
            d2 xc = load_pair(xl);
            ValueChunks A, B;
            A.c5 = d2{0.0, 0.0};
            B.c5 = d2{0.0, 0.0};
            load_value_chunks(A, m.values, s, lane);

            auto step = [&](ValueChunks& cur, ValueChunks& nxt, bool more) {
                const d2 xs = load_pair(xl + n);
                double edge = 0.0;
                if (lane == 0) edge = xl[-1];
                if (lane == 63) edge = xl[2];
                if (more) load_value_chunks(nxt, m.values, s + s_step, lane);

                const int sh = (int)(s & 1);
                d2* __restrict__ w2 = reinterpret_cast<d2*>(wlds);
                w2[lane] = cur.c0;
                w2[64 + lane] = cur.c1;
                w2[128 + lane] = cur.c2;
                w2[192 + lane] = cur.c3;
                w2[256 + lane] = cur.c4;
                if (sh != 0 && lane == 0) w2[320] = cur.c5;
                __builtin_amdgcn_wave_barrier();
                const double* __restrict__ v = wlds + sh + 10 * lane;
                const double a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4];
                const double b0 = v[5], b1 = v[6], b2 = v[7], b3 = v[8], b4 = v[9];
                __builtin_amdgcn_wave_barrier();

                double w = __shfl_up(xc.y, 1);
                double e = __shfl_down(xc.x, 1);
                if (lane == 0) w = edge;
                if (lane == 63) e = edge;
                double r0 = a1 * w;
                r0 = fma(a2, xc.x, r0);
                r0 = fma(a3, xc.y, r0);
                r0 = fma(a0, xn.x, r0);
                r0 = fma(a4, xs.x, r0);
                double r1 = b1 * xc.x;
                r1 = fma(b2, xc.y, r1);
                r1 = fma(b3, e, r1);
                r1 = fma(b0, xn.y, r1);
                r1 = fma(b4, xs.y, r1);
                if (kDot) {
                    dot_acc = fma(xc.x, r0, dot_acc);
                    dot_acc = fma(xc.y, r1, dot_acc);
                }
                if (kVec) {
                    d2 out = {alpha * r0, alpha * r1};
                    *reinterpret_cast<d2*>(yl) = out;
                } else {
                    yl[0] = alpha * r0;
                    yl[1] = alpha * r1;
                }
                xn = xc;
                xc = xs;
                xl += n;
                yl += n;
                s += s_step;
            };
            int li = li0;
            for (; li + 1 < li1; li += 2) {
                step(A, B, true);
                step(B, A, li + 2 < li1);
            }
            if (li < li1) step(A, B, false);
        } else {
            for (int li = li0; li < li1; ++li) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = j0 + 2 * lane + e;
                    if (j < n) {
                        const long long lr = (long long)li * n + j;
                        const double sum = row_reference<true>(m, x, (int)lr, gfirst + li, j);
                        if (kDot) dot_acc = fma(x[lr], sum, dot_acc);
                        y[lr] = alpha * sum;
                    }
                }
            }
        }
    }
    if (kDot) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dot_acc += __shfl_down(dot_acc, off);
        if (lane == 0) dot_partials[blockIdx.x * kWavesPerBlock + wave_in_block] = dot_acc;
    }
}

// ---------------------------------------------------------------------------------
// STENCIL5, row-direct variant (default on slabs made of whole grid rows): one thread per row on
// a 2-D launch, blockIdx.y = grid row, blockIdx.x*256 + threadIdx.x = grid column, so the grid
// coordinates and the row's CSR offset cost no integer division. Everything else is the
// reference's own shape (src/spmv/spmv_stencil_csr_direct.cu:76-123): five strided 8-byte loads of
// the row's coefficients at the computed offset, x[row-1..row+1] from the same cache lines and
// x[row -+ n] from the lines the neighbouring grid rows pulled into L2 moments earlier.
// Measured on MI355X at 20 000^2 this plain shape beats both LDS-staged variants below
// (4.13 ms vs 4.25 column-march vs 4.65 one-shot wave-tile): the 40-byte lane stride is absorbed
// by the vector L1 (every line of `values` is still fetched from HBM exactly once), dispatch-order
// blocks keep the +-n rows L2-resident, and 8 waves/SIMD with no staging hide the latency.
// ---------------------------------------------------------------------------------
//
// kRows consecutive grid rows per thread (same column): the centre row of one step is the north
// row of the next, so x costs (kRows + 2) / kRows cache-served row loads per row instead of 3.
//
// Block order: a 1-D launch walked in dispatch order, column block fastest. Two XCD-affine
// mappings (column blocks pinned to an XCD, interleaved or as contiguous bands) were measured and
// rejected: they cut the fabric reads from 25.8 GB to 19.3 GB per launch at 20 000^2 (the +-n rows
// become L2 hits instead of Infinity-Cache hits) but ran 4.49-4.58 ms against 4.08 ms in the CG
// loop; the plain order keeps all eight XCDs on one moving front of the arrays.
template <int kRows, bool kDot>
__global__ __launch_bounds__(kBlock) void stencil5_rowdirect_kernel(
    SlabCsr m, const double* __restrict__ x, double* __restrict__ y, double alpha, int gi_lo,
    int gi_hi, int col_blocks, double* __restrict__ dot_partials, const int* __restrict__ skip_flag) {
    __shared__ double wave_part[kWavesPerBlock];
    if (skip_flag != nullptr && *skip_flag != 0) return;
    const int n = m.grid_size;
    const int row_group = (int)blockIdx.x / col_blocks;
    const int col_block = (int)blockIdx.x - row_group * col_blocks;
    const int li0 = gi_lo + row_group * kRows;                 // first local grid row of this block
    const int gfirst = m.row_offset / n;
    const int j = col_block * kBlock + (int)threadIdx.x;       // grid column
    double dot_acc = 0.0;
    if (j < n) {
        const bool col_interior = j > 0 && j < n - 1;
        double xn = 0.0, xc = 0.0;
        if (kRows > 1) {
            const double* __restrict__ x0 = x + ((long long)li0 * n + j);
            xc = x0[0];
            if (gfirst + li0 > 0) xn = x0[-n];
        }
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
            const int li = li0 + k;
            if (li < gi_hi) {
                const int gi = gfirst + li;
                const long long lr = (long long)li * n + j;
                const double* __restrict__ xl = x + lr;
                double sum, centre;
                if (col_interior && gi > 0 && gi < n - 1) {
                    const double* __restrict__ v =
                        m.values + (stencil_gridrow_base(gi, n) + 5LL * j - 1 - m.nnz_base);
                    const double xs = xl[n];
                    centre = kRows > 1 ? xc : xl[0];
                    const double north = kRows > 1 ? xn : xl[-n];
                    // plain loads on purpose: the five strided loads of a wave share cache lines through
                    // the vector L1; nontemporal loads bypass it and ran 5.10 ms instead of 3.95 ms
                    const double v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3], v4 = v[4];
                    // west / east straight from memory (same cache lines as the centre); taking them from
                    // the neighbouring lanes with __shfl instead measured 4.07 ms against 3.96 ms
                    const double west = xl[-1], east = xl[1];
                    sum = v1 * west;
                    sum = fma(v2, centre, sum);
                    sum = fma(v3, east, sum);
                    sum = fma(v0, north, sum);
                    sum = fma(v4, xs, sum);
                    if (kRows > 1) {
                        xn = xc;
                        xc = xs;
                    }
                } else {
                    sum = row_reference<false>(m, x, (int)lr, gi, j);
                    centre = xl[0];
                    if (kRows > 1) {
                        xn = centre;
                        if (k + 1 < kRows && li + 1 < gi_hi) xc = xl[n];
                    }
                }
                if (kDot) dot_acc = fma(centre, sum, dot_acc);
                y[lr] = alpha * sum;
            }
        }
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
//     0.8 % at 20 000^2 and 2.6 % at 10 000^2, bit-identical results (kWeLds; SPMV_AMD_ROWLDS_WE_LDS=0
//     restores the loads);
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
template <int kMode, bool kWeLds = false>
__global__ __launch_bounds__(64) void stencil5_rowlds_kernel(
    SlabCsr m, const double* __restrict__ x, double* __restrict__ y, double alpha, int gi_lo, int row_step,
    int gfirst, int col_tiles, int total_tiles, int group, int reverse, double* __restrict__ dot_partials,
    const int* __restrict__ skip_flag, ResidualOut res) {
    constexpr bool kDot = kMode == 1;
    constexpr bool kInit = kMode == 2;
    __shared__ double strip[5 * kLdsTileCols];
    __shared__ double xrow[kWeLds ? kLdsTileCols + 2 : 1];  // kWeLds: the tile's x values, W / E read back from here
    // The convergence flag is REQUESTED here and tested after the tile's loads have been issued: a wave does not sit
    // on a scalar-load round trip before its first vector load (a launch enqueued past convergence only reads).
    const int skip = skip_flag != nullptr ? __builtin_nontemporal_load(skip_flag) : 0;
    const int lane = (int)threadIdx.x;
    const int b = (int)blockIdx.x;
    const int span = 8 * group;
    int tile = (b / span) * span + (b & 7) * group + ((b >> 3) % group);
    if (tile >= total_tiles) return;
    if (reverse) tile = total_tiles - 1 - tile;  // same tiles, same partial slots, walked from the end
    const int n = m.grid_size;
    const int row_group = tile / col_tiles;
    const int col_tile = tile - row_group * col_tiles;
    const int li = gi_lo + row_group * row_step;  // local grid row (row_step > 1: a launch over separate grid rows)
    const int gi = gfirst + li;        // global grid row
    const int j0 = col_tile * kLdsTileCols;
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
                xc[h] = xl[0], xn[h] = xl[-n], xs[h] = xl[n];
                if (kInit) bv[h] = __builtin_nontemporal_load(res.b + ((long long)li * n + j));
                if (!kWeLds) {
                    if (j > 0) xw[h] = xl[-1];
                    if (j < n - 1) xe[h] = xl[1];
                } else {
                    // only the tile's two outer neighbours come from memory; the rest from the LDS copy below
                    if (h == 0 && lane == 0 && j > 0) xw[0] = xl[-1];
                    if (h == 1 && lane == 63 && j < n - 1) xe[1] = xl[1];
                }
            }
        }
        if (skip != 0) return;
#pragma unroll
        for (int k = 0; k < 10; ++k) strip[64 * k + lane] = c[k];
        if (kWeLds) {
            xrow[1 + lane] = xc[0];
            xrow[65 + lane] = xc[1];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (kWeLds) {
            // columns beyond n hold 0 in xc, exactly what an absent neighbour contributes
            if (lane > 0) xw[0] = xrow[lane];
            xe[0] = xrow[2 + lane];
            xw[1] = xrow[64 + lane];
            if (lane < 63) xe[1] = xrow[66 + lane];
            if (j0 + lane == n - 1) xe[0] = 0.0;
        }
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
        if (skip != 0) return;
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
        if (lane == 0) dot_partials[tile] = dot_acc;
    }
}

// ---------------------------------------------------------------------------------
// STENCIL5, row-lds MARCH (round 3, one bounded attempt at the x[row +- n] re-fetch; SPMV_AMD_ROWLDS_ROWS = 2 / 4).
// The row-lds tile, but a wave owns the same 128 columns of kRows CONSECUTIVE grid rows and walks down them: the
// north / centre / south values rotate in registers (xn <- xc <- xs), so a row's x line enters the wave once per
// kRows + 2 rows instead of three times per row, whatever the XCD-private L2s make of it; the coefficient strip is
// re-used row after row (LDS operations of one wave retire in order) and the NEXT row's ten coefficient loads and
// south-row loads are issued before the current row is evaluated. Tile groups -> XCDs by the same run rule.
// Per row the arithmetic, the clamped edge tiles, the LDS copy for W / E and the partial slot (one per row and
// column tile, the slot the one-row kernel writes) are row-lds's: results and dot products are bit-identical.
// Groups that are not kRows interior rows inside the launch's range (the grid's first / last row, the remainder at
// the end of a range) fall back to row-lds's own per-row evaluation with fresh loads.
// ---------------------------------------------------------------------------------
template <int kMode, int kRows>
__global__ __launch_bounds__(64) void stencil5_rowlds_march_kernel(
    SlabCsr m, const double* __restrict__ x, double* __restrict__ y, double alpha, int gi_lo, int gi_hi, int gfirst,
    int col_tiles, int total_groups, int group, int reverse, double* __restrict__ dot_partials,
    const int* __restrict__ skip_flag, ResidualOut res) {
    constexpr bool kDot = kMode == 1;
    constexpr bool kInit = kMode == 2;
    __shared__ double strip[5 * kLdsTileCols];
    __shared__ double xrow[kLdsTileCols + 2];
    const int skip = skip_flag != nullptr ? __builtin_nontemporal_load(skip_flag) : 0;
    const int lane = (int)threadIdx.x;
    const int b = (int)blockIdx.x;
    const int span = 8 * group;
    int tg = (b / span) * span + (b & 7) * group + ((b >> 3) % group);
    if (tg >= total_groups) return;
    if (reverse) tg = total_groups - 1 - tg;
    const int n = m.grid_size;
    const int row_group = tg / col_tiles;
    const int col_tile = tg - row_group * col_tiles;
    const int li0 = gi_lo + row_group * kRows;
    const int j0 = col_tile * kLdsTileCols;
    const int rows_here = min(kRows, gi_hi - li0);
    const bool pure = rows_here == kRows && gfirst + li0 > 0 && gfirst + li0 + kRows - 1 < n - 1;
    const bool edge_tile = j0 == 0 || j0 + kLdsTileCols > n - 1;
    const double* __restrict__ vals = m.values;
    const long long hi = m.nnz_local - 1;
    const int ja = j0 + lane, jb = j0 + lane + 64;

    // coefficient run of grid row gi for this tile: ten coalesced nontemporal loads (clamped on the row's edge tiles)
    auto load_coefficients = [&](int gi, double (&c)[10]) {
        const long long e = stencil_gridrow_base(gi, n) + 5LL * j0 - 1 - m.nnz_base + lane;
        if (edge_tile) {
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
    };
    auto load_row = [&](int li, double (&v)[2]) {  // x of local grid row li at this lane's two columns (0 beyond n)
        const double* __restrict__ xl = x + (long long)li * n;
        v[0] = ja < n ? xl[ja] : 0.0;
        v[1] = jb < n ? xl[jb] : 0.0;
    };

    if (pure) {
        double c[10], cn[10], xn[2], xc[2], xs[2], xsn[2] = {0.0, 0.0};
        load_coefficients(gfirst + li0, c);
        load_row(li0 - 1, xn);
        load_row(li0, xc);
        load_row(li0 + 1, xs);
        if (skip != 0) return;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const int li = li0 + r;
            const double* __restrict__ xl = x + (long long)li * n;
            double bv[2] = {0.0, 0.0}, xw0 = 0.0, xe1 = 0.0;
            // the tile's two outer neighbours and (mode 2) b: this row's only other loads
            if (lane == 0 && ja > 0) xw0 = xl[ja - 1];
            if (lane == 63 && jb < n - 1) xe1 = xl[jb + 1];
            if (kInit) {
                if (ja < n) bv[0] = __builtin_nontemporal_load(res.b + ((long long)li * n + ja));
                if (jb < n) bv[1] = __builtin_nontemporal_load(res.b + ((long long)li * n + jb));
            }
            if (r + 1 < kRows) {  // next row's streams go out before this row is evaluated
                load_coefficients(gfirst + li + 1, cn);
                load_row(li + 2, xsn);
            }
#pragma unroll
            for (int k = 0; k < 10; ++k) strip[64 * k + lane] = c[k];
            xrow[1 + lane] = xc[0];
            xrow[65 + lane] = xc[1];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double xw[2], xe[2];
            xw[0] = lane > 0 ? xrow[lane] : xw0;
            xe[0] = xrow[2 + lane];
            xw[1] = xrow[64 + lane];
            xe[1] = lane < 63 ? xrow[66 + lane] : xe1;
            if (ja == n - 1) xe[0] = 0.0;
            double dot_acc = 0.0;
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
                    const long long lr = (long long)li * n + j;
                    if (kInit) {
                        const double rv = fma(-1.0, alpha * sum, bv[h]);
                        __builtin_nontemporal_store(rv, res.r + lr);
                        res.p[lr] = rv;
                        dot_acc = fma(rv, rv, dot_acc);
                    } else {
                        __builtin_nontemporal_store(alpha * sum, y + lr);
                    }
                }
            }
            if (kDot || kInit) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) dot_acc += __shfl_down(dot_acc, off);
                if (lane == 0) dot_partials[(long long)(li - gi_lo) * col_tiles + col_tile] = dot_acc;
            }
            if (r + 1 < kRows) {
#pragma unroll
                for (int k = 0; k < 10; ++k) c[k] = cn[k];
                xn[0] = xc[0], xn[1] = xc[1];
                xc[0] = xs[0], xc[1] = xs[1];
                xs[0] = xsn[0], xs[1] = xsn[1];
                // the strip and the x copy are rewritten next: this wave's reads above have retired (in-order LDS)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }

    // not a full group of interior rows: every row on its own, as the one-row kernel evaluates it
    if (skip != 0) return;
    for (int r = 0; r < rows_here; ++r) {
        const int li = li0 + r;
        const int gi = gfirst + li;
        double dot_acc = 0.0;
        if (gi > 0 && gi < n - 1) {
            double c[10], xn[2], xc[2], xs[2];
            load_coefficients(gi, c);
            load_row(li - 1, xn);
            load_row(li, xc);
            load_row(li + 1, xs);
            const double* __restrict__ xl = x + (long long)li * n;
            double xw0 = 0.0, xe1 = 0.0;
            if (lane == 0 && ja > 0) xw0 = xl[ja - 1];
            if (lane == 63 && jb < n - 1) xe1 = xl[jb + 1];
#pragma unroll
            for (int k = 0; k < 10; ++k) strip[64 * k + lane] = c[k];
            xrow[1 + lane] = xc[0];
            xrow[65 + lane] = xc[1];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double xw[2], xe[2];
            xw[0] = lane > 0 ? xrow[lane] : xw0;
            xe[0] = xrow[2 + lane];
            xw[1] = xrow[64 + lane];
            xe[1] = lane < 63 ? xrow[66 + lane] : xe1;
            if (ja == n - 1) xe[0] = 0.0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = j0 + lane + 64 * h;
                if (j < n) {
                    const double* __restrict__ v = strip + 5 * (lane + 64 * h);
                    double sum;
                    if (j > 0 && j < n - 1) {
                        sum = v[1] * xw[h];
                        sum = fma(v[2], xc[h], sum);
                        sum = fma(v[3], xe[h], sum);
                        sum = fma(v[0], xn[h], sum);
                        sum = fma(v[4], xs[h], sum);
                    } else if (j == 0) {
                        sum = fma(v[1], xn[h], 0.0);
                        sum = fma(v[2], xc[h], sum);
                        sum = fma(v[3], xe[h], sum);
                        sum = fma(v[4], xs[h], sum);
                    } else {
                        sum = fma(v[0], xn[h], 0.0);
                        sum = fma(v[1], xw[h], sum);
                        sum = fma(v[2], xc[h], sum);
                        sum = fma(v[3], xs[h], sum);
                    }
                    if (kDot) dot_acc = fma(xc[h], sum, dot_acc);
                    const long long lr = (long long)li * n + j;
                    if (kInit) {
                        const double rv = fma(-1.0, alpha * sum, res.b[lr]);
                        __builtin_nontemporal_store(rv, res.r + lr);
                        res.p[lr] = rv;
                        dot_acc = fma(rv, rv, dot_acc);
                    } else {
                        __builtin_nontemporal_store(alpha * sum, y + lr);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
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
            if (lane == 0) dot_partials[(long long)(li - gi_lo) * col_tiles + col_tile] = dot_acc;
        }
    }
}

// ---------------------------------------------------------------------------------
// STENCIL5, row-planes variant (solver slabs; SlabCsr::planes). Index space, tile -> XCD runs, x handling, the three
// modes and the arithmetic are row-lds's; the coefficients come from five planes [N | W | C | E | S] instead of the
// CSR values array, so every coefficient load is a coalesced 8-byte-per-lane stream starting on a 1 KiB boundary and
// nothing is transposed through LDS (a CSR tile of 640 coefficients starts wherever row (gi, j0) starts: 8-byte
// aligned only -- the 3 % by which row-lds on tiles artificially aligned to 5 KiB beat the real layout).
// Same 56 B per row. Every row of the grid is handled here: interior rows in the reference's W,C,E,N,S order
// (spmv_stencil_csr_direct.cu:105-109), rows on the grid's border as the CSR loop would -- ascending column, i.e.
// N, W, C, E, S without the absent ones, sum started at 0 (:116-119) -- so row_ptr / col_idx are never read.
// MEASURED AND NOT ADOPTED (SPMV_AMD_SLAB_PLANES=1 selects it): bit-identical results, and as fast as row-lds in
// back-to-back launches (3.66 ms at 20 000^2), but inside the CG loop its launches average 3.79-3.81 ms against
// 3.62-3.69 ms for row-lds on the same box (solve 107.8-108.5 vs 104.6-106.4 ms; 13.85 vs 13.82 ms at 50 M rows):
// five coefficient streams 3.2 GB apart per wave instead of one contiguous 5 KiB run do worse between the other
// kernels of the loop than the alignment of the planes gains. The CSR array stays the solver's format.
// ---------------------------------------------------------------------------------
template <int kMode, bool kWeLds>
__global__ __launch_bounds__(64) void stencil5_planes_kernel(
    SlabCsr m, const double* __restrict__ x, double* __restrict__ y, double alpha, int gi_lo, int row_step,
    int gfirst, int col_tiles, int total_tiles, int group, int reverse, double* __restrict__ dot_partials,
    const int* __restrict__ skip_flag, ResidualOut res) {
    constexpr bool kDot = kMode == 1;
    constexpr bool kInit = kMode == 2;
    __shared__ double xrow[kWeLds ? kLdsTileCols + 2 : 1];
    const int skip = skip_flag != nullptr ? __builtin_nontemporal_load(skip_flag) : 0;  // tested after the loads are out
    const int lane = (int)threadIdx.x;
    const int b = (int)blockIdx.x;
    const int span = 8 * group;
    int tile = (b / span) * span + (b & 7) * group + ((b >> 3) % group);
    if (tile >= total_tiles) return;
    if (reverse) tile = total_tiles - 1 - tile;
    const int n = m.grid_size;
    const int row_group = tile / col_tiles;
    const int col_tile = tile - row_group * col_tiles;
    const int li = gi_lo + row_group * row_step;
    const int gi = gfirst + li;
    const int j0 = col_tile * kLdsTileCols;
    const bool has_n = gi > 0, has_s = gi < n - 1;
    const long long R = m.n_local;
    const double* __restrict__ planes = m.planes;
    double c[2][5], xc[2], xw[2], xe[2], xn[2], xs[2], bv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = j0 + lane + 64 * h;
        xc[h] = xw[h] = xe[h] = xn[h] = xs[h] = bv[h] = 0.0;
#pragma unroll
        for (int k = 0; k < 5; ++k) c[h][k] = 0.0;
        if (j < n) {
            const long long lr = (long long)li * n + j;
#pragma unroll
            for (int k = 0; k < 5; ++k) c[h][k] = __builtin_nontemporal_load(planes + k * R + lr);
            const double* __restrict__ xl = x + lr;
            xc[h] = xl[0];
            if (has_n) xn[h] = xl[-n];
            if (has_s) xs[h] = xl[n];
            if (!kWeLds) {
                if (j > 0) xw[h] = xl[-1];
                if (j < n - 1) xe[h] = xl[1];
            } else {
                if (h == 0 && lane == 0 && j > 0) xw[0] = xl[-1];
                if (h == 1 && lane == 63 && j < n - 1) xe[1] = xl[1];
            }
            if (kInit) bv[h] = __builtin_nontemporal_load(res.b + lr);
        }
    }
    if (skip != 0) return;
    if (kWeLds) {
        xrow[1 + lane] = xc[0];
        xrow[65 + lane] = xc[1];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane > 0) xw[0] = xrow[lane];
        xe[0] = xrow[2 + lane];
        xw[1] = xrow[64 + lane];
        if (lane < 63) xe[1] = xrow[66 + lane];
    }
    double dot_acc = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = j0 + lane + 64 * h;
        if (j < n) {
            const long long lr = (long long)li * n + j;
            const double vN = c[h][0], vW = c[h][1], vC = c[h][2], vE = c[h][3], vS = c[h][4];
            double sum;
            if (has_n && has_s && j > 0 && j < n - 1) {
                sum = vW * xw[h];
                sum = fma(vC, xc[h], sum);
                sum = fma(vE, xe[h], sum);
                sum = fma(vN, xn[h], sum);
                sum = fma(vS, xs[h], sum);
            } else {  // a row on the grid's border: the CSR loop's order over the entries that exist
                sum = 0.0;
                if (has_n) sum = fma(vN, xn[h], sum);
                if (j > 0) sum = fma(vW, xw[h], sum);
                sum = fma(vC, xc[h], sum);
                if (j < n - 1) sum = fma(vE, xe[h], sum);
                if (has_s) sum = fma(vS, xs[h], sum);
            }
            if (kDot) dot_acc = fma(xc[h], sum, dot_acc);
            if (kInit) {
                const double rv = fma(-1.0, alpha * sum, bv[h]);
                __builtin_nontemporal_store(rv, res.r + lr);
                res.p[lr] = rv;
                dot_acc = fma(rv, rv, dot_acc);
            } else {
                __builtin_nontemporal_store(alpha * sum, y + lr);
            }
        }
    }
    if (kDot || kInit) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dot_acc += __shfl_down(dot_acc, off);
        if (lane == 0) dot_partials[tile] = dot_acc;
    }
}

// One thread per local row of a verified stencil slab: the row's CSR entries [N,W,C,E,S minus the absent ones] spread
// over the five planes, absent entries 0.
__global__ __launch_bounds__(kBlock) void build_stencil5_planes_kernel(SlabCsr m, double* __restrict__ planes) {
    const long long lr = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (lr >= m.n_local) return;
    const int n = m.grid_size;
    const long long g = (long long)m.row_offset + lr;
    const int i = (int)(g / n), j = (int)(g - (long long)i * n);
    long long k = stencil_row_start(i, j, n) - m.nnz_base;
    const long long R = m.n_local;
    planes[lr] = i > 0 ? m.values[k++] : 0.0;
    planes[R + lr] = j > 0 ? m.values[k++] : 0.0;
    planes[2 * R + lr] = m.values[k++];
    planes[3 * R + lr] = j < n - 1 ? m.values[k++] : 0.0;
    planes[4 * R + lr] = i < n - 1 ? m.values[k++] : 0.0;
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

// kLanes lanes cooperate on one row (kLanes = 64: one row per wavefront). Lanes stride the
// row's entries, so consecutive lanes read consecutive col_idx/values: coalesced. The per-lane
// partial sums are combined by a fixed shuffle tree, so the summation order differs from the
// sequential loop (results agree to rounding; exact on the integer-valued benchmark inputs).
template <int kLanes>
__global__ __launch_bounds__(kBlock) void csr_subwave_kernel(SlabCsr m, const double* __restrict__ x,
                                                             double* __restrict__ y, double alpha) {
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long row = gid / kLanes;
    const int sub = (int)(gid % kLanes);
    const int lo = -m.halo_before, hi = m.n_local + m.halo_after;
    double sum = 0.0;
    if (row < m.n_local) {
        const int k1 = m.row_ptr[row + 1];
        for (int k = m.row_ptr[row] + sub; k < k1; k += kLanes)
            sum = fma(m.values[k], x_at(x, (long long)m.col_idx[k] - m.row_offset, lo, hi), sum);
    }
#pragma unroll
    for (int off = kLanes / 2; off > 0; off >>= 1) sum += __shfl_down(sum, off, kLanes);
    if (sub == 0 && row < m.n_local) y[row] = alpha * sum;
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

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

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

void launch_build_stencil5_planes(const SlabCsr& m, double* planes, hipStream_t stream) {
    if (m.n_local == 0) return;
    hipLaunchKernelGGL(build_stencil5_planes_kernel, dim3(blocks_for(m.n_local)), dim3(kBlock), 0, stream, m, planes);
}

static int wavetile_blocks(const LaunchShape& shape) {
    int blocks = shape.compute_units * shape.blocks_per_cu;
    blocks = (blocks + 7) & ~7;
    return blocks < 8 ? 8 : blocks;
}

static int plan_partials(const Stencil5Plan& p) {
    if (p.variant == Stencil5Variant::RowGeneric) return p.row_blocks * kWavesPerBlock;
    if (p.variant == Stencil5Variant::RowDirect)
        return p.row_blocks * ((p.gi_hi - p.gi_lo + p.rows_per_task - 1) / p.rows_per_task);
    if (p.variant == Stencil5Variant::RowLds || p.variant == Stencil5Variant::RowPlanes) return p.row_blocks * (p.gi_hi - p.gi_lo);
    if (p.variant == Stencil5Variant::WaveTile) return p.tile_blocks * kWavesPerBlock;
    return (p.march_blocks + (p.head_rows ? p.row_blocks : 0) + (p.tail_rows ? p.row_blocks : 0)) * kWavesPerBlock;
}

static const char* plan_name(const Stencil5Plan& p, const SlabCsr& m) {
    switch (p.variant) {
        case Stencil5Variant::RowDirect: return "stencil5/row-direct";
        case Stencil5Variant::RowLds: return "stencil5/row-lds";
        case Stencil5Variant::RowPlanes: return "stencil5/row-planes";
        case Stencil5Variant::ColumnMarch: return "stencil5/column-march";
        case Stencil5Variant::WaveTile: return "stencil5/wave-tile";
        default: return m.verified_stencil ? "stencil5/row-generic" : "stencil5/row-generic(csr-loop)";
    }
}

Stencil5Plan plan_stencil5(const SlabCsr& m, int first_row, int last_row, Stencil5Variant variant,
                           const LaunchShape& shape) {
    Stencil5Plan p;
    const Tunables& knobs = shape.knobs;
    const int n = m.grid_size;
    const bool tile_ok = m.verified_stencil && n >= kTileRows;
    const bool march_ok = tile_ok && m.row_offset % n == 0 && m.n_local % n == 0 &&
                          first_row % n == 0 && last_row % n == 0;
    const bool direct_ok = m.verified_stencil && n >= 2 && m.row_offset % n == 0 &&
                           m.n_local % n == 0 && first_row % n == 0 && last_row % n == 0;
    // row-lds needs grid rows long enough that the two clamped edge tiles are a small share
    const int lds_min_n = knobs.rowlds_min_grid;
    if (variant == Stencil5Variant::Auto)
        variant = direct_ok ? (n >= lds_min_n ? (m.planes != nullptr ? Stencil5Variant::RowPlanes : Stencil5Variant::RowLds)
                                              : Stencil5Variant::RowDirect)
                  : tile_ok ? Stencil5Variant::WaveTile
                            : Stencil5Variant::RowGeneric;
    if (variant == Stencil5Variant::RowPlanes && (m.planes == nullptr || !direct_ok)) variant = Stencil5Variant::RowLds;
    if ((variant == Stencil5Variant::RowDirect || variant == Stencil5Variant::RowLds) && !direct_ok)
        variant = Stencil5Variant::RowGeneric;
    if (variant == Stencil5Variant::ColumnMarch && !march_ok)
        variant = tile_ok ? Stencil5Variant::WaveTile : Stencil5Variant::RowGeneric;
    if (variant == Stencil5Variant::WaveTile && !tile_ok) variant = Stencil5Variant::RowGeneric;
    p.variant = variant;
    if (variant == Stencil5Variant::RowGeneric) {
        p.row_blocks = (int)blocks_for((long long)last_row - first_row);
    } else if (variant == Stencil5Variant::RowDirect) {
        p.gi_lo = first_row / n;
        p.gi_hi = last_row / n;
        p.row_blocks = (int)blocks_for(n);  // column blocks per grid row
        p.rows_per_task = knobs.direct_rows;
        if (p.rows_per_task != 2 && p.rows_per_task != 4) p.rows_per_task = 1;
    } else if (variant == Stencil5Variant::RowLds || variant == Stencil5Variant::RowPlanes) {
        p.gi_lo = first_row / n;
        p.gi_hi = last_row / n;
        p.row_blocks = (n + kLdsTileCols - 1) / kLdsTileCols;  // column tiles (= workgroups) per grid row
        // consecutive tiles per XCD: one grid row + ~1100 columns per run of 8 * group tiles (xcd_run_group)
        p.rows_per_task = knobs.rowlds_group > 0 ? knobs.rowlds_group : xcd_run_group(n, kLdsTileCols, 4);
        p.we_from_lds = knobs.rowlds_we_lds != 0;
        if (p.rows_per_task < 1 || p.rows_per_task > 64) p.rows_per_task = 4;
        // row-lds march (SPMV_AMD_ROWLDS_ROWS = 2 / 4): only where a wave has several grid rows to walk
        p.lds_march_rows = (variant == Stencil5Variant::RowLds && (knobs.rowlds_rows == 2 || knobs.rowlds_rows == 4) &&
                            p.gi_hi - p.gi_lo >= knobs.rowlds_rows) ? knobs.rowlds_rows : 1;
    } else if (variant == Stencil5Variant::WaveTile) {
        // one tile per wave in dispatch order by default (4.65 ms at 20 000^2); SPMV_AMD_WAVETILE_ONESHOT=0
        // selects the persistent, XCD-banded walk (5.87 ms), kept for the record
        p.oneshot = knobs.wavetile_oneshot != 0;
        const int tiles = (last_row + kTileRows - 1) / kTileRows - first_row / kTileRows;
        p.tile_blocks = p.oneshot ? (tiles + kWavesPerBlock - 1) / kWavesPerBlock : wavetile_blocks(shape);
    } else {
        const int gfirst = m.row_offset / n;
        p.gi_lo = first_row / n;
        p.gi_hi = last_row / n;
        if (gfirst + p.gi_lo == 0) p.head_rows = true, ++p.gi_lo;
        if (gfirst + p.gi_hi == n && p.gi_hi > p.gi_lo) p.tail_rows = true, --p.gi_hi;
        if (p.gi_hi < p.gi_lo) p.gi_hi = p.gi_lo;
        p.row_blocks = (int)blocks_for(n);
        p.strips = (n + kTileRows - 1) / kTileRows;
        const int strip_groups = (p.strips + kWavesPerBlock - 1) / kWavesPerBlock;
        const int G = p.gi_hi - p.gi_lo;
        if (G > 0) {
            // enough blocks for several rounds over the chip, few enough start-up rows per task
            const long long target = (long long)shape.compute_units * knobs.march_blocks_per_cu;
            long long R = ((long long)G * strip_groups) / (target > 0 ? target : 1);
            // measured on MI355X at n = 20000: 16 rows per task beats 8 / 32 / 64 (4.40 vs 4.53 / 4.66 / 4.58 ms)
            const int r_max = knobs.march_max_rows, r_min = 4;
            R = R > r_max ? r_max : (R < r_min ? r_min : R);
            if (R > G) R = G;
            p.rows_per_task = knobs.march_rows_per_task > 0 ? knobs.march_rows_per_task : (int)R;
            if (p.rows_per_task < 1) p.rows_per_task = 1;
            p.march_blocks = ((G + p.rows_per_task - 1) / p.rows_per_task) * strip_groups;
        }
    }
    p.first_row = first_row;
    p.last_row = last_row;
    p.partials = plan_partials(p);
    p.name = plan_name(p, m);
    return p;
}

int stencil5_partials_needed(const SlabCsr& m, int first_row, int last_row, Stencil5Variant variant,
                             const LaunchShape& shape) {
    return plan_stencil5(m, first_row, last_row, variant, shape).partials;
}

const char* stencil5_variant_name(const SlabCsr& m, int first_row, int last_row, Stencil5Variant variant,
                                  const LaunchShape& shape) {
    return plan_stencil5(m, first_row, last_row, variant, shape).name;
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

int launch_stencil5_spmv(const SlabCsr& m, const Stencil5Plan& p, const double* x, double* y, double alpha,
                         double* d_dot_partials, const int* d_skip_flag, bool reverse, hipStream_t stream,
                         const ResidualOut* init) {
    const int first_row = p.first_row, last_row = p.last_row;
    if (last_row <= first_row) return 0;
    if (init != nullptr && ((p.variant != Stencil5Variant::RowLds && p.variant != Stencil5Variant::RowPlanes) || d_dot_partials == nullptr)) {
        fprintf(stderr, "[spmv] the fused initial residual exists for the row-lds kernel only\n");
        exit(EXIT_FAILURE);
    }
    const int n = m.grid_size;
    const bool dot = d_dot_partials != nullptr;
    const bool analytic = m.verified_stencil && n >= 2;

    auto launch_rows = [&](int lo, int hi, double* partials) {
        const dim3 grid(blocks_for((long long)hi - lo));
#define SPMV_AMD_LAUNCH_ROWS(AN, DOT)                                                                   \
    hipLaunchKernelGGL((stencil5_row_kernel<AN, DOT>), grid, dim3(kBlock), 0, stream, m, x, y, alpha, lo, \
                       hi, partials, d_skip_flag)
        if (analytic) {
            if (dot) SPMV_AMD_LAUNCH_ROWS(true, true);
            else SPMV_AMD_LAUNCH_ROWS(true, false);
        } else {
            if (dot) SPMV_AMD_LAUNCH_ROWS(false, true);
            else SPMV_AMD_LAUNCH_ROWS(false, false);
        }
#undef SPMV_AMD_LAUNCH_ROWS
        return (int)grid.x * kWavesPerBlock;
    };

    if (p.variant == Stencil5Variant::RowGeneric) return launch_rows(first_row, last_row, d_dot_partials);

    if (p.variant == Stencil5Variant::RowDirect) {
        const int groups = (p.gi_hi - p.gi_lo + p.rows_per_task - 1) / p.rows_per_task;
        const long long blocks = (long long)p.row_blocks * groups;  // < 2^31 for any int32 CSR
        const dim3 grid((unsigned)blocks);
#define SPMV_AMD_LAUNCH_DIRECT(R, DOT)                                                                  \
    hipLaunchKernelGGL((stencil5_rowdirect_kernel<R, DOT>), grid, dim3(kBlock), 0, stream, m, x, y, alpha, \
                       p.gi_lo, p.gi_hi, p.row_blocks, d_dot_partials, d_skip_flag)
        if (p.rows_per_task == 4) {
            if (dot) SPMV_AMD_LAUNCH_DIRECT(4, true);
            else SPMV_AMD_LAUNCH_DIRECT(4, false);
        } else if (p.rows_per_task == 2) {
            if (dot) SPMV_AMD_LAUNCH_DIRECT(2, true);
            else SPMV_AMD_LAUNCH_DIRECT(2, false);
        } else {
            if (dot) SPMV_AMD_LAUNCH_DIRECT(1, true);
            else SPMV_AMD_LAUNCH_DIRECT(1, false);
        }
#undef SPMV_AMD_LAUNCH_DIRECT
        return (int)blocks;
    }

    if (p.variant == Stencil5Variant::RowPlanes) {
        const long long tiles = (long long)p.row_blocks * (p.gi_hi - p.gi_lo);
        const int span = 8 * p.rows_per_task;
        const dim3 grid((unsigned)((tiles + span - 1) / span * span));
        const int gfirst = m.row_offset / n;
        const ResidualOut res = init ? *init : ResidualOut{nullptr, nullptr, nullptr};
#define SPMV_AMD_LAUNCH_PLANES(MODE, WE)                                                                            \
    hipLaunchKernelGGL((stencil5_planes_kernel<MODE, WE>), grid, dim3(64), 0, stream, m, x, y, alpha, p.gi_lo, 1, gfirst, \
                       p.row_blocks, (int)tiles, p.rows_per_task, reverse ? 1 : 0, d_dot_partials, d_skip_flag, res)
        if (p.we_from_lds) {
            if (init) SPMV_AMD_LAUNCH_PLANES(2, true);
            else if (dot) SPMV_AMD_LAUNCH_PLANES(1, true);
            else SPMV_AMD_LAUNCH_PLANES(0, true);
        } else {
            if (init) SPMV_AMD_LAUNCH_PLANES(2, false);
            else if (dot) SPMV_AMD_LAUNCH_PLANES(1, false);
            else SPMV_AMD_LAUNCH_PLANES(0, false);
        }
#undef SPMV_AMD_LAUNCH_PLANES
        return (int)tiles;
    }

    if (p.variant == Stencil5Variant::RowLds && p.lds_march_rows > 1) {
        // row-lds march: a wave walks lds_march_rows consecutive grid rows of its 128 columns (same partial slots)
        const int R = p.lds_march_rows;
        const long long tiles = (long long)p.row_blocks * (p.gi_hi - p.gi_lo);
        const long long groups = (long long)p.row_blocks * ((p.gi_hi - p.gi_lo + R - 1) / R);
        const int span = 8 * p.rows_per_task;
        const dim3 grid((unsigned)((groups + span - 1) / span * span));
        const int gfirst = m.row_offset / n;
        const ResidualOut res = init ? *init : ResidualOut{nullptr, nullptr, nullptr};
#define SPMV_AMD_LAUNCH_MARCHLDS(MODE, ROWS)                                                                              \
    hipLaunchKernelGGL((stencil5_rowlds_march_kernel<MODE, ROWS>), grid, dim3(64), 0, stream, m, x, y, alpha, p.gi_lo, p.gi_hi, \
                       gfirst, p.row_blocks, (int)groups, p.rows_per_task, reverse ? 1 : 0, d_dot_partials, d_skip_flag, res)
        if (R == 2) {
            if (init) SPMV_AMD_LAUNCH_MARCHLDS(2, 2);
            else if (dot) SPMV_AMD_LAUNCH_MARCHLDS(1, 2);
            else SPMV_AMD_LAUNCH_MARCHLDS(0, 2);
        } else {
            if (init) SPMV_AMD_LAUNCH_MARCHLDS(2, 4);
            else if (dot) SPMV_AMD_LAUNCH_MARCHLDS(1, 4);
            else SPMV_AMD_LAUNCH_MARCHLDS(0, 4);
        }
#undef SPMV_AMD_LAUNCH_MARCHLDS
        return (int)tiles;
    }

    if (p.variant == Stencil5Variant::RowLds) {
        const long long tiles = (long long)p.row_blocks * (p.gi_hi - p.gi_lo);  // < 2^31 for any int32 CSR
        const int span = 8 * p.rows_per_task;  // rows_per_task carries the tiles-per-XCD group here
        const dim3 grid((unsigned)((tiles + span - 1) / span * span));
        const int gfirst = m.row_offset / n;
        const ResidualOut res = init ? *init : ResidualOut{nullptr, nullptr, nullptr};
#define SPMV_AMD_LAUNCH_ROWLDS(MODE, WE)                                                                            \
    hipLaunchKernelGGL((stencil5_rowlds_kernel<MODE, WE>), grid, dim3(64), 0, stream, m, x, y, alpha, p.gi_lo, 1, gfirst, \
                       p.row_blocks, (int)tiles, p.rows_per_task, reverse ? 1 : 0, d_dot_partials, d_skip_flag, res)
        if (p.we_from_lds) {
            if (init) SPMV_AMD_LAUNCH_ROWLDS(2, true);
            else if (dot) SPMV_AMD_LAUNCH_ROWLDS(1, true);
            else SPMV_AMD_LAUNCH_ROWLDS(0, true);
        } else {
            if (init) SPMV_AMD_LAUNCH_ROWLDS(2, false);
            else if (dot) SPMV_AMD_LAUNCH_ROWLDS(1, false);
            else SPMV_AMD_LAUNCH_ROWLDS(0, false);
        }
#undef SPMV_AMD_LAUNCH_ROWLDS
        return (int)tiles;
    }

    const bool vec_xy = aligned16(x) && aligned16(y);
    if (p.variant == Stencil5Variant::WaveTile) {
        // Fixed grid (independent of the row range) so that the dot partials keep their shape.
        const int oneshot = p.oneshot ? 1 : 0;
        const dim3 grid(p.tile_blocks);
        const bool vec_ns = vec_xy && (n % 2 == 0);
#define SPMV_AMD_LAUNCH_TILE(VXY, VNS, DOT)                                                       \
    hipLaunchKernelGGL((stencil5_wavetile_kernel<VXY, VNS, DOT>), grid, dim3(kBlock), 0, stream, m, \
                       x, y, alpha, first_row, last_row, d_dot_partials, d_skip_flag, oneshot)
        if (vec_ns) {
            if (dot) SPMV_AMD_LAUNCH_TILE(true, true, true);
            else SPMV_AMD_LAUNCH_TILE(true, true, false);
        } else if (vec_xy) {
            if (dot) SPMV_AMD_LAUNCH_TILE(true, false, true);
            else SPMV_AMD_LAUNCH_TILE(true, false, false);
        } else {
            if (dot) SPMV_AMD_LAUNCH_TILE(false, false, true);
            else SPMV_AMD_LAUNCH_TILE(false, false, false);
        }
#undef SPMV_AMD_LAUNCH_TILE
        return (int)grid.x * kWavesPerBlock;
    }

    // column-march: [global first grid row] + marched grid rows + [global last grid row]
    int used = 0;
    if (p.march_blocks > 0) {
        const dim3 grid(p.march_blocks);
        const bool vec = vec_xy && (n % 2 == 0);
#define SPMV_AMD_LAUNCH_MARCH(VEC, DOT)                                                               \
    hipLaunchKernelGGL((stencil5_colmarch_kernel<VEC, DOT>), grid, dim3(kBlock), 0, stream, m, x, y,    \
                       alpha, p.gi_lo, p.gi_hi, p.rows_per_task, p.strips, d_dot_partials, d_skip_flag)
        if (vec) {
            if (dot) SPMV_AMD_LAUNCH_MARCH(true, true);
            else SPMV_AMD_LAUNCH_MARCH(true, false);
        } else {
            if (dot) SPMV_AMD_LAUNCH_MARCH(false, true);
            else SPMV_AMD_LAUNCH_MARCH(false, false);
        }
#undef SPMV_AMD_LAUNCH_MARCH
        used += p.march_blocks * kWavesPerBlock;
    }
    if (p.head_rows) used += launch_rows(first_row, first_row + n, dot ? d_dot_partials + used : nullptr);
    if (p.tail_rows) used += launch_rows(last_row - n, last_row, dot ? d_dot_partials + used : nullptr);
    return used;
}

int launch_stencil5_spmv_first_and_last_gridrow(const SlabCsr& m, const Stencil5Plan& head, const double* x, double* y,
                                                double alpha, double* d_dot_partials, const int* d_skip_flag,
                                                const LaunchShape& shape, hipStream_t stream, const ResidualOut* init) {
    const int n = m.grid_size;
    const int local_gridrows = n > 0 ? m.n_local / n : 0;
    const bool tiled = head.variant == Stencil5Variant::RowLds || head.variant == Stencil5Variant::RowPlanes;
    if (init != nullptr && (!tiled || d_dot_partials == nullptr || local_gridrows < 2)) {
        fprintf(stderr, "[spmv] the fused initial residual exists for the row-lds kernel only\n");
        exit(EXIT_FAILURE);
    }
    if (n <= 0 || local_gridrows < 2 || !tiled) {
        // two launches over the two row ranges (any variant)
        int used = launch_stencil5_spmv(m, x, y, alpha, 0, n, d_dot_partials, d_skip_flag, Stencil5Variant::Auto, shape, stream);
        used += launch_stencil5_spmv(m, x, y, alpha, m.n_local - n, m.n_local, d_dot_partials ? d_dot_partials + used : nullptr,
                                     d_skip_flag, Stencil5Variant::Auto, shape, stream);
        return used;
    }
    // one launch: row group 0 is local grid row 0, row group 1 is the last local grid row; the partial slots
    // are those of the two separate launches back to back
    const int tiles = 2 * head.row_blocks;
    const int span = 8 * head.rows_per_task;
    const dim3 grid((unsigned)((tiles + span - 1) / span * span));
    const int gfirst = m.row_offset / n;
    const ResidualOut res = init ? *init : ResidualOut{nullptr, nullptr, nullptr};
    const int mode = init ? 2 : (d_dot_partials ? 1 : 0);
#define SPMV_AMD_LAUNCH_EDGES(KERNEL, MODE)                                                                             \
    hipLaunchKernelGGL((KERNEL<MODE, false>), grid, dim3(64), 0, stream, m, x, y, alpha, 0, local_gridrows - 1, gfirst, \
                       head.row_blocks, tiles, head.rows_per_task, 0, d_dot_partials, d_skip_flag, res)
    if (head.variant == Stencil5Variant::RowPlanes) {
        if (mode == 2) SPMV_AMD_LAUNCH_EDGES(stencil5_planes_kernel, 2);
        else if (mode == 1) SPMV_AMD_LAUNCH_EDGES(stencil5_planes_kernel, 1);
        else SPMV_AMD_LAUNCH_EDGES(stencil5_planes_kernel, 0);
    } else {
        if (mode == 2) SPMV_AMD_LAUNCH_EDGES(stencil5_rowlds_kernel, 2);
        else if (mode == 1) SPMV_AMD_LAUNCH_EDGES(stencil5_rowlds_kernel, 1);
        else SPMV_AMD_LAUNCH_EDGES(stencil5_rowlds_kernel, 0);
    }
#undef SPMV_AMD_LAUNCH_EDGES
    return tiles;
}

// Measured on MI355X, 10 000^2 stencil as CSR (5 nnz/row): stream 1.37 ms (4 entries per thread,
// 176 rows per block; 1.50 ms at 3 per thread, 1.40-1.45 at 5-6, 2.02-2.53 ms with one big LDS strip
// per 256 rows), row-scalar 1.75 ms (chunked), subwave4 2.52, subwave8 3.44, subwave16 6.42, one row
// per wavefront 8.34. For scale: rocsparse_spmv (csr_adaptive) takes 1.33 ms (tools/rocsparse_compare.hip).
// The texture addresser is ~80 % busy in the thread-per-row kernels (rocprofv3 TA_BUSY_avr): the
// 20/40-byte lane strides of col_idx/values are what the coalesced phase 1 of the stream kernel removes.
// Short rows: stream; longer rows: about four entries per lane.
CsrVariant csr_auto_variant(const SlabCsr& m) {
    // Round 3 (tools/generic_matrix_perf.py, profiles/r03_generic_matrix_perf.txt), 10^8 entries each:
    //  * uniform rows of 12 / 20 / 40 / 80 / 160 random columns: the stream kernel is within 1-3 % of the best variant at EVERY
    //    length (1.86 / 1.75 / 1.50 / 1.19 / 0.74 ms against 1.81 / 1.71 / 1.49 / 1.19 / 0.74 ms) and 5-12 % ahead of what the
    //    mean-row-length rule of rounds 1-2 picked (subwave4 / 8 / 16 / 32, wavefront) -- with the sequential, bit-reproducible
    //    row sum the others give up; banded (9 per row): stream 0.223 ms, subwave4 0.339 ms;
    //  * skewed (rows of 1-8 entries, one in a thousand with 2 000-20 000; mean 15.5): the mean sent it to subwave4, 7.5 ms;
    //    stream 4.2 ms, 32 lanes per row 3.1 ms, adaptive 2.8 ms.
    // Round 4 (profiles/r04_generic_long_rows.txt), the regime the round-3 table stopped short of -- uniform rows of 320 / 640 /
    // 1000 random columns, 10^8 entries: stream 0.85 / 1.51 / 2.30 ms (at >= 64 entries per row the block holds 16 rows, the
    // span leaves the strip and 16 of 256 threads fold sequentially), one row per wavefront 0.67 / 0.65 / 0.64 ms, 32 lanes
    // per row 0.68 / 0.65 / 0.63 ms. At 160 per row the two still tie (0.74 ms, round 3).
    // So: a MEAN row length above 192 -> one row per wavefront (its tree sum replaces the sequential one: 2e-15 relative);
    // else rows longer than the stream kernel's strip -> adaptive (stream for the short rows, the whole workgroup for the long
    // ones); everything else -> stream. The sub-wavefront kernels stay selectable (spmv_amd_operator_select_variant).
    const double mean = m.n_local > 0 ? (double)m.nnz_local / (double)m.n_local : 0.0;
    if (mean > 192.0) return CsrVariant::Wavefront;
    return m.max_row_nnz > 1024 ? CsrVariant::Adaptive : CsrVariant::Stream;
}

// Launch geometry of the stream / adaptive kernels: rows per block and logical blocks (= dot partials of a fused launch).
static void csr_stream_geometry(const SlabCsr& m, CsrVariant variant, const Tunables& knobs, int* threads_out, int* per_block_out,
                                long long* blocks_out) {
    const long long rows = m.n_local;
    // rows per block: the mean span should fill about 90 % of the LDS strip, at most one row per thread
    const double avg = rows > 0 ? (double)m.nnz_local / rows : 1.0;
    const int shape = variant == CsrVariant::Adaptive ? 0 : knobs.csr_stream_shape;  // 0: 256 x 4, 1: 64 x 6, 2: 64 x 8, 3: 128 x 5
    const int threads = shape == 0 ? 256 : (shape == 3 ? 128 : 64);
    const int cap = shape == 0 ? 1024 : (shape == 1 ? 384 : (shape == 2 ? 512 : 640));
    int per_block = (int)(0.9 * cap / (avg > 1.0 ? avg : 1.0));
    per_block = per_block > threads ? threads : (per_block < 16 ? 16 : per_block & ~15);
    if (knobs.csr_stream_rows > 0) per_block = knobs.csr_stream_rows;
    if (per_block > threads) per_block = threads;
    *threads_out = threads;
    *per_block_out = per_block;
    *blocks_out = (rows + per_block - 1) / per_block;
}

int csr_fused_dot_partials(const SlabCsr& m, CsrVariant variant, const Tunables& knobs) {
    if (variant == CsrVariant::Auto) variant = csr_auto_variant(m);
    if ((variant != CsrVariant::Stream && variant != CsrVariant::Adaptive) || m.n_local == 0) return 0;
    int threads = 0, per_block = 0;
    long long blocks = 0;
    csr_stream_geometry(m, variant, knobs, &threads, &per_block, &blocks);
    return blocks <= 0x7fffffffLL ? (int)blocks : 0;
}

void launch_csr_spmv(const SlabCsr& m, const double* x, double* y, double alpha,
                     CsrVariant variant, const Tunables& knobs, hipStream_t stream, double* d_dot_partials) {
    if (m.n_local == 0) return;
    if (variant == CsrVariant::Auto) variant = csr_auto_variant(m);
    const long long rows = m.n_local;
    switch (variant) {
        case CsrVariant::Adaptive:
        case CsrVariant::Stream: {
            // threads x entries per thread; measured on MI355X at 10 000^2 / 15 000^2: 256 x 4 1.43 / 3.13 ms,
            // 64 x 6 1.42 / 3.23-3.40, 64 x 8 1.55 / 3.30, 128 x 5 1.44 / 3.20 -> unlike the dense streams, the
            // one-wave shapes do not pay here (fewer rows per block = more shared edge lines)
            const int shape = variant == CsrVariant::Adaptive ? 0 : knobs.csr_stream_shape;
            int threads = 0, per_block = 0;
            long long blocks = 0;
            csr_stream_geometry(m, variant, knobs, &threads, &per_block, &blocks);
            // dispatch order: relabelling blocks so that an XCD takes runs of consecutive blocks measured SLOWER here
            // (10 000^2: 1.35 ms plain, 1.40-1.45 ms for runs of 2-16; profiles/r02_xcd_group.txt)
            const int group = knobs.xcd_group > 0 ? knobs.xcd_group : 1;
            const long long span = group > 1 ? 8LL * group : 1;
            const dim3 grid((unsigned)((blocks + span - 1) / span * span));
#define SPMV_AMD_CSR_STREAM(T, P) \
    hipLaunchKernelGGL((csr_stream_kernel<T, P>), grid, dim3(T), 0, stream, m, x, y, alpha, per_block, group, (int)blocks, d_dot_partials)
            if (variant == CsrVariant::Adaptive) {  // 256 x 4 only
                hipLaunchKernelGGL((csr_stream_kernel<256, 4, true>), grid, dim3(256), 0, stream, m, x, y, alpha, per_block, group, (int)blocks, d_dot_partials);
                break;
            }
            if (shape == 1) SPMV_AMD_CSR_STREAM(64, 6);
            else if (shape == 2) SPMV_AMD_CSR_STREAM(64, 8);
            else if (shape == 3) SPMV_AMD_CSR_STREAM(128, 5);
            else SPMV_AMD_CSR_STREAM(256, 4);
#undef SPMV_AMD_CSR_STREAM
            break;
        }
        case CsrVariant::RowScalar:
            hipLaunchKernelGGL(csr_row_scalar_kernel, dim3(blocks_for(rows)), dim3(kBlock), 0, stream,
                               m, x, y, alpha);
            break;
#define SPMV_AMD_SUBWAVE(L)                                                                        \
    hipLaunchKernelGGL(csr_subwave_kernel<L>, dim3(blocks_for(rows * L)), dim3(kBlock), 0, stream, m, \
                       x, y, alpha)
        case CsrVariant::SubWave4: SPMV_AMD_SUBWAVE(4); break;
        case CsrVariant::SubWave8: SPMV_AMD_SUBWAVE(8); break;
        case CsrVariant::SubWave16: SPMV_AMD_SUBWAVE(16); break;
        case CsrVariant::SubWave32: SPMV_AMD_SUBWAVE(32); break;
        default: SPMV_AMD_SUBWAVE(64); break;
#undef SPMV_AMD_SUBWAVE
    }
}

void launch_ell_transpose(int rows, int width, const int* idx_rowmajor, const double* val_rowmajor,
                          int* idx_slotmajor, double* val_slotmajor, hipStream_t stream) {
    if (rows == 0) return;
    hipLaunchKernelGGL(ell_transpose_kernel, dim3(blocks_for(rows)), dim3(kBlock), 0, stream, rows,
                       width, idx_rowmajor, val_rowmajor, idx_slotmajor, val_slotmajor);
}

int ell_fused_dot_partials(int rows, const Tunables& knobs) {
    const int B = (knobs.ell_shape & 1) ? 64 : 256;
    return (int)(((long long)rows + B - 1) / B);
}

void launch_ell_spmv(int rows, int width, const int* idx, const double* val, const double* x,
                     double* y, double alpha, double beta, const Tunables& knobs, hipStream_t stream, int grid_hint,
                     double* d_dot_partials) {
    if (rows == 0) return;
    // bit 0: one-wave workgroups, bit 1: nontemporal planes / y. Measured on MI355X at 15 000^2 (generic /
    // stencil-aware): 0: 3.22 / 2.32 ms, 1: 3.00 / 2.51, 2: 2.96 / 2.26, 3: 3.00 / 2.47 -> 2.
    const int shape = knobs.ell_shape;
    // each XCD takes `group` consecutive 256-row blocks of every run of 8 * group (15 000^2: 2.95-3.03 ms in dispatch
    // order, 2.86 ms with runs of 9; 10 000^2: 1.35 -> 1.30 ms; profiles/r02_xcd_group.txt)
    const int group = knobs.xcd_group > 0 ? knobs.xcd_group : (grid_hint > 0 ? xcd_run_group(grid_hint, 256, 7) + 1 : 8);
    const long long span = group > 1 ? 8LL * group : 1;
#define SPMV_AMD_ELL(B, NT)                                                                                              \
    hipLaunchKernelGGL((ell_spmv_kernel<B, NT>), dim3((unsigned)(((((long long)rows + B - 1) / B) + span - 1) / span * span)), \
                       dim3(B), 0, stream, rows, width, idx, val, x, y, alpha, beta, group, (int)(((long long)rows + B - 1) / B), d_dot_partials)
    if ((shape & 3) == 3) SPMV_AMD_ELL(64, true);
    else if (shape & 1) SPMV_AMD_ELL(64, false);
    else if (shape & 2) SPMV_AMD_ELL(256, true);
    else SPMV_AMD_ELL(256, false);
#undef SPMV_AMD_ELL
}

void launch_ell_stencil5_spmv(int rows, int width, int grid_size, const int* idx,
                              const double* val, const double* x, double* y, double alpha,
                              double beta, const Tunables& knobs, hipStream_t stream, double* d_dot_partials) {
    if (rows == 0) return;
    if (grid_size < 3 || (long long)grid_size * grid_size != rows) {
        launch_ell_spmv(rows, width, idx, val, x, y, alpha, beta, knobs, stream, 0, d_dot_partials);
        return;
    }
    const int shape = knobs.ell_shape;
    // 15 000^2: 2.27-2.29 ms in dispatch order, 2.01 ms with runs of 8 blocks per XCD; 20 000^2: 4.01 -> 3.6-3.7 ms
    const int group = knobs.xcd_group > 0 ? knobs.xcd_group : xcd_run_group(grid_size, 256, 8);
    const long long span = group > 1 ? 8LL * group : 1;
#define SPMV_AMD_ELL5(B, NT)                                                                                                 \
    hipLaunchKernelGGL((ell_stencil5_kernel<B, NT>), dim3((unsigned)(((((long long)rows + B - 1) / B) + span - 1) / span * span)), \
                       dim3(B), 0, stream, rows, width, grid_size, idx, val, x, y, alpha, beta, group,                        \
                       (int)(((long long)rows + B - 1) / B), d_dot_partials)
    if ((shape & 3) == 3) SPMV_AMD_ELL5(64, true);
    else if (shape & 1) SPMV_AMD_ELL5(64, false);
    else if (shape & 2) SPMV_AMD_ELL5(256, true);
    else SPMV_AMD_ELL5(256, false);
#undef SPMV_AMD_ELL5
}

}  // namespace spmv_amd
