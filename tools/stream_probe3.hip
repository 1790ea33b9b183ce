// tools/stream_probe3.hip -- measurement aid, not part of the product. Candidate shapes for the STENCIL5
// interior row (five coefficients at v[5 r + off .. +4], x centre + W/E/N/S, y store) without the
// boundary handling, to pick the one worth building into the library:
//   aos     : the shipped shape (five strided 8-byte loads per lane), nt on chosen loads / on the store
//   lds8    : coefficients fetched as five fully coalesced 8-byte loads per lane (optionally nontemporal),
//             transposed through a wave-private LDS strip
//   ldsdma  : coefficients of 128 rows per wave by five 16-byte-per-lane LDS-DMA instructions
// Every kernel of the runs logged in profiles/r01_stream_probe3.txt is in this file; main() holds the first block
// (argv[1] = passes, default 0) and the LAST experiment only -- the earlier ones were successive edits of main(),
// their launch macros (LG, LGP, LGR, LF, FG, LP ...) are quoted in the log headers.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/stream_probe3.hip -o tools/bin/stream_probe3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ double ld(const double* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(double* p, double v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

__device__ __forceinline__ double row5(double v0, double v1, double v2, double v3, double v4, double w, double c,
                                       double e, double nn, double s) {
    double t = v1 * w;
    t = fma(v2, c, t);
    t = fma(v3, e, t);
    t = fma(v0, nn, t);
    return fma(v4, s, t);
}

template <int BLOCK, int NTMASK, bool NTST>
__global__ __launch_bounds__(BLOCK) void aos(const double* __restrict__ v, const double* __restrict__ x,
                                             double* __restrict__ y, size_t rows, int n) {
    const size_t r = (size_t)n + (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r + n >= rows) return;
    const double* q = v + 5 * r;
    const double v0 = ld<(NTMASK & 1) != 0>(q), v1 = ld<(NTMASK & 2) != 0>(q + 1), v2 = ld<(NTMASK & 4) != 0>(q + 2),
                 v3 = ld<(NTMASK & 8) != 0>(q + 3), v4 = ld<(NTMASK & 16) != 0>(q + 4);
    const double* xl = x + r;
    st<NTST>(y + r, row5(v0, v1, v2, v3, v4, xl[-1], xl[0], xl[1], xl[-n], xl[n]));
}

template <int BLOCK, bool NTLD, bool NTST, bool SHFL>
__global__ __launch_bounds__(BLOCK) void lds8(const double* __restrict__ v, const double* __restrict__ x,
                                              double* __restrict__ y, size_t rows, int n) {
    __shared__ double lds[BLOCK * 5];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t r0 = (size_t)n + (size_t)blockIdx.x * BLOCK + (size_t)w * 64;
    if (r0 + 64 + n >= rows) return;  // whole waves only (rows is a multiple of 64)
    const double* src = v + 5 * r0 + lane;
    const double c0 = ld<NTLD>(src), c1 = ld<NTLD>(src + 64), c2 = ld<NTLD>(src + 128), c3 = ld<NTLD>(src + 192),
                 c4 = ld<NTLD>(src + 256);
    const size_t r = r0 + lane;
    const double* xl = x + r;
    const double xc = xl[0], xn = xl[-n], xs = xl[n];
    double xw, xe;
    if (SHFL) {
        xw = __shfl_up(xc, 1);
        xe = __shfl_down(xc, 1);
        if (lane == 0) xw = xl[-1];
        if (lane == 63) xe = xl[1];
    } else {
        xw = xl[-1];
        xe = xl[1];
    }
    double* wl = lds + w * 320;
    wl[lane] = c0; wl[64 + lane] = c1; wl[128 + lane] = c2; wl[192 + lane] = c3; wl[256 + lane] = c4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double* q = wl + 5 * lane;
    st<NTST>(y + r, row5(q[0], q[1], q[2], q[3], q[4], xw, xc, xe, xn, xs));
}

// 128 rows per wave: 5120 B of coefficients = five LDS-DMA instructions of 16 B per lane; the wave's two
// 64-row halves are handled as two 8-byte-per-lane accesses of x and y.
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
template <int WAVES, int AUX, bool NTST>
__global__ __launch_bounds__(WAVES * 64) void ldsdma(const double* __restrict__ v, const double* __restrict__ x,
                                                     double* __restrict__ y, size_t rows, int n, int misalign) {
    __shared__ __attribute__((aligned(16))) double lds[WAVES * 640];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t r0 = (size_t)n + ((size_t)blockIdx.x * WAVES + w) * 128;
    if (r0 + 128 + n >= rows) return;
    double* wl = lds + w * 640;
    const char* src = reinterpret_cast<const char*>(v + 5 * r0 + misalign) + lane * 16;
#pragma unroll
    for (int k = 0; k < 5; ++k)
        __builtin_amdgcn_global_load_lds((gbl_void*)(src + k * 1024), (lds_void*)(wl + k * 128), 16, 0, AUX);
    const size_t ra = r0 + lane, rb = ra + 64;
    const double *xa = x + ra, *xb = x + rb;
    const double ac = xa[0], aw = xa[-1], ae = xa[1], an = xa[-n], as = xa[n];
    const double bc = xb[0], bw = xb[-1], be = xb[1], bn = xb[-n], bs = xb[n];
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the DMA writes have landed
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double* qa = wl + 5 * lane;
    const double* qb = qa + 320;
    st<NTST>(y + ra, row5(qa[0], qa[1], qa[2], qa[3], qa[4], aw, ac, ae, an, as));
    st<NTST>(y + rb, row5(qb[0], qb[1], qb[2], qb[3], qb[4], bw, bc, be, bn, bs));
}

// two rows per lane (rows lane and lane + 64 of a 128-row wave tile), coalesced 8-byte nt loads via LDS
template <int WAVES, bool NTLD, bool NTST>
__global__ __launch_bounds__(WAVES * 64) void lds8x2(const double* __restrict__ v, const double* __restrict__ x,
                                                     double* __restrict__ y, size_t rows, int n) {
    __shared__ double lds[WAVES * 640];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t r0 = (size_t)n + ((size_t)blockIdx.x * WAVES + w) * 128;
    if (r0 + 128 + n >= rows) return;
    double* wl = lds + w * 640;
    const double* src = v + 5 * r0 + lane;
    double c[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) c[k] = ld<NTLD>(src + 64 * k);
    const size_t ra = r0 + lane, rb = ra + 64;
    const double *xa = x + ra, *xb = x + rb;
    const double ac = xa[0], aw = xa[-1], ae = xa[1], an = xa[-n], as = xa[n];
    const double bc = xb[0], bw = xb[-1], be = xb[1], bn = xb[-n], bs = xb[n];
#pragma unroll
    for (int k = 0; k < 10; ++k) wl[64 * k + lane] = c[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double* qa = wl + 5 * lane;
    const double* qb = qa + 320;
    st<NTST>(y + ra, row5(qa[0], qa[1], qa[2], qa[3], qa[4], aw, ac, ae, an, as));
    st<NTST>(y + rb, row5(qb[0], qb[1], qb[2], qb[3], qb[4], bw, bc, be, bn, bs));
}

// the library's index space: a wave = 128 columns of one grid row (157 tiles per 20 000-column row, the last
// one partial), coefficient run at the CSR position base(gi) + 5 j0 - 1 (8-byte aligned only)
template <bool NTLD, bool NTST, bool GRID2D>
__global__ __launch_bounds__(64) void lds8x2_grid(const double* __restrict__ v, const double* __restrict__ x,
                                                  double* __restrict__ y, int n, int tiles) {
    __shared__ double lds[640];
    const int lane = threadIdx.x;
    int gi, tile;
    if (GRID2D) { gi = 1 + blockIdx.y; tile = blockIdx.x; }
    else { gi = 1 + blockIdx.x / tiles; tile = blockIdx.x - (gi - 1) * tiles; }
    const int j0 = tile * 128;
    const long long base = (4LL * n - 2) + (long long)(gi - 1) * (5LL * n - 2);
    const long long e = base + 5LL * j0 - 1 + lane;
    const long long hi = 5LL * n * n - 4LL * n - 1;
    double c[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        long long idx = e + 64 * k;
        idx = idx > hi ? hi : idx;
        c[k] = ld<NTLD>(v + idx);
    }
    const int ja = j0 + lane, jb = ja + 64;
    const long long ra = (long long)gi * n + ja, rb = ra + 64;
    const bool fa = ja > 0 && ja < n - 1, fb = jb > 0 && jb < n - 1;
    double ac = 0, aw = 0, ae = 0, an = 0, as = 0, bc = 0, bw = 0, be = 0, bn = 0, bs = 0;
    if (fa) { const double* xa = x + ra; ac = xa[0], aw = xa[-1], ae = xa[1], an = xa[-n], as = xa[n]; }
    if (fb) { const double* xb = x + rb; bc = xb[0], bw = xb[-1], be = xb[1], bn = xb[-n], bs = xb[n]; }
#pragma unroll
    for (int k = 0; k < 10; ++k) lds[64 * k + lane] = c[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double* qa = lds + 5 * lane;
    const double* qb = qa + 320;
    if (fa) st<NTST>(y + ra, row5(qa[0], qa[1], qa[2], qa[3], qa[4], aw, ac, ae, an, as));
    if (fb) st<NTST>(y + rb, row5(qb[0], qb[1], qb[2], qb[3], qb[4], bw, bc, be, bn, bs));
}
// grid-row tiles of COLS columns (COLS / 64 rows per lane). ALIGNED: the coefficient run is fetched as an
// aligned window (whole 128-byte lines: no line is shared by two load instructions of the wave).
// PAIR: blocks b and b + 8 (same XCD under round-robin dispatch) take adjacent tiles.
template <int COLS, bool ALIGNED, int PAIR, bool PADROW = false>
__global__ __launch_bounds__(64) void lds_grid(const double* __restrict__ v, const double* __restrict__ x,
                                               double* __restrict__ y, int n, int tiles) {
    constexpr int R = COLS / 64;           // rows per lane
    constexpr int NL = 5 * R + (ALIGNED ? 1 : 0);  // load instructions
    __shared__ double lds[64 * NL];
    const int lane = threadIdx.x;
    unsigned b = blockIdx.x;
    int gi, tile;
    if (PADROW) {
        // every grid row is padded to a multiple of 8 * PAIR tiles: XCD k always owns the same columns
        const unsigned padded = (tiles + 8 * PAIR - 1) / (8 * PAIR) * (8 * PAIR);
        gi = 1 + b / padded;
        const unsigned w = b - (gi - 1) * padded;
        tile = (w / (8 * PAIR)) * (8 * PAIR) + (w & 7) * PAIR + ((w >> 3) % PAIR);
        if (tile >= tiles) return;
    } else {
        if (PAIR > 1) b = (b / (8 * PAIR)) * (8 * PAIR) + (b & 7) * PAIR + ((b >> 3) % PAIR);
        gi = 1 + b / tiles;
        tile = b - (gi - 1) * tiles;
    }
    if (gi > n - 2) return;
    const int j0 = tile * COLS;
    const long long base = (4LL * n - 2) + (long long)(gi - 1) * (5LL * n - 2);
    const long long s0 = base + 5LL * j0 - 1;
    const int sh = ALIGNED ? (int)(s0 & 15) : 0;
    const long long e = s0 - sh + lane;
    const long long hi = 5LL * n * n - 4LL * n - 1;
    double c[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        long long idx = e + 64 * k;
        idx = idx > hi ? hi : idx;
        if (ALIGNED && k == NL - 1) { if (lane < 16) c[k] = __builtin_nontemporal_load(v + idx); }
        else c[k] = __builtin_nontemporal_load(v + idx);
    }
    double xc[R], xw[R], xe[R], xn[R], xs[R];
    bool f[R];
#pragma unroll
    for (int h = 0; h < R; ++h) {
        const int j = j0 + lane + 64 * h;
        f[h] = j > 0 && j < n - 1;
        if (f[h]) { const double* xl = x + (long long)gi * n + j; xc[h] = xl[0], xw[h] = xl[-1], xe[h] = xl[1], xn[h] = xl[-n], xs[h] = xl[n]; }
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) if (!(ALIGNED && k == NL - 1) || lane < 16) lds[64 * k + lane] = c[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int h = 0; h < R; ++h) {
        const double* q = lds + sh + 5 * (lane + 64 * h);
        if (f[h]) __builtin_nontemporal_store(row5(q[0], q[1], q[2], q[3], q[4], xw[h], xc[h], xe[h], xn[h], xs[h]), y + (long long)gi * n + j0 + lane + 64 * h);
    }
}
// flat tiles: COLS consecutive rows of the flat index space (tiles cross grid-row ends; x / y tile-aligned),
// coefficient run at the real CSR position 5 r - 2 gi - n - 1; G = consecutive tiles per XCD.
template <int COLS, int G>
__global__ __launch_bounds__(64) void lds_flat(const double* __restrict__ v, const double* __restrict__ x,
                                               double* __restrict__ y, int n, long long ntiles) {
    constexpr int R = COLS / 64, NL = 5 * R;
    __shared__ double lds[64 * NL];
    const int lane = threadIdx.x;
    long long b = blockIdx.x;
    if (G > 1) b = (b / (8 * G)) * (8 * G) + (b & 7) * G + ((b >> 3) % G);
    if (b >= ntiles) return;
    const long long r0 = (long long)n + b * COLS;       // first flat row (grid row 0 skipped)
    const int gi0 = (int)(r0 / n);
    const long long s0 = 5 * r0 - 2LL * gi0 - n - 1;
    const long long hi = 5LL * n * n - 4LL * n - 1;
    double c[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) { long long idx = s0 + lane + 64 * k; idx = idx > hi ? hi : idx; c[k] = __builtin_nontemporal_load(v + idx); }
    double xc[R], xw[R], xe[R], xn[R], xs[R];
    bool f[R]; int off[R];
    const long long rend = (long long)(gi0 + 1) * n;    // first flat row of the next grid row
#pragma unroll
    for (int h = 0; h < R; ++h) {
        const long long r = r0 + lane + 64 * h;
        const bool crossed = r >= rend;
        const int j = (int)(r - (crossed ? rend : rend - n));
        off[h] = 5 * (lane + 64 * h) - (crossed ? 2 : 0);
        f[h] = j > 0 && j < n - 1 && r + n < (long long)n * n;
        if (f[h]) { const double* xl = x + r; xc[h] = xl[0], xw[h] = xl[-1], xe[h] = xl[1], xn[h] = xl[-n], xs[h] = xl[n]; }
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) lds[64 * k + lane] = c[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int h = 0; h < R; ++h) {
        const double* q = lds + off[h];
        if (f[h]) __builtin_nontemporal_store(row5(q[0], q[1], q[2], q[3], q[4], xw[h], xc[h], xe[h], xn[h], xs[h]), y + r0 + lane + 64 * h);
    }
}
// 128 columns x ROWS grid rows per wave: the centre row of one step is the north row of the next, so a
// tile fetches ROWS + 2 x rows instead of 3 * ROWS (fewer N/S re-reads through L2 / Infinity Cache)
template <int ROWS, int G>
__global__ __launch_bounds__(64) void lds_grid_rows(const double* __restrict__ v, const double* __restrict__ x,
                                                    double* __restrict__ y, int n, int tiles, int total) {
    __shared__ double lds[640 * ROWS];
    const int lane = threadIdx.x;
    unsigned b = blockIdx.x;
    if (G > 1) b = (b / (8 * G)) * (8 * G) + (b & 7) * G + ((b >> 3) % G);
    if ((int)b >= total) return;
    const int rg = b / tiles, tile = b - rg * tiles;
    const int gi0 = 1 + rg * ROWS;
    const int j0 = tile * 128;
    const long long hi = 5LL * n * n - 4LL * n - 1;
    double c[10 * ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const long long base = (4LL * n - 2) + (long long)(gi0 + r - 1) * (5LL * n - 2);
        const long long e = base + 5LL * j0 - 1 + lane;
#pragma unroll
        for (int k = 0; k < 10; ++k) { long long idx = e + 64 * k; idx = idx > hi ? hi : idx; c[10 * r + k] = __builtin_nontemporal_load(v + idx); }
    }
    double xr[ROWS + 2][2], xw[ROWS][2], xe[ROWS][2];
    bool f[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = j0 + lane + 64 * h;
        f[h] = j > 0 && j < n - 1;
        if (f[h]) {
            const double* xl = x + (long long)(gi0 - 1) * n + j;
#pragma unroll
            for (int r = 0; r < ROWS + 2; ++r) xr[r][h] = (gi0 - 1 + r < n) ? xl[(long long)r * n] : 0.0;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { xw[r][h] = xl[(long long)(r + 1) * n - 1]; xe[r][h] = xl[(long long)(r + 1) * n + 1]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 10 * ROWS; ++k) lds[64 * k + lane] = c[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double* q = lds + 640 * r + 5 * (lane + 64 * h);
            if (f[h] && gi0 + r < n - 1)
                __builtin_nontemporal_store(row5(q[0], q[1], q[2], q[3], q[4], xw[r][h], xr[r + 1][h], xe[r][h], xr[r][h], xr[r + 2][h]),
                                            y + (long long)(gi0 + r) * n + j0 + lane + 64 * h);
        }
}
// What would a CG iteration cost with the direction update folded into the SpMV? One pass that reads r, the old
// p (both at the five stencil points) and x, and writes x += a p_old, p_new = r + b p_old (other buffer) and
// Ap = A p_new: 88 B/row instead of 56 (SpMV) + 40 (x / p update) = 96 in two passes.
template <int G, bool WITHX = true>
__global__ __launch_bounds__(64) void fused_grid(const double* __restrict__ v, const double* __restrict__ r,
                                                 const double* __restrict__ pold, double* __restrict__ pnew,
                                                 double* __restrict__ xv, double* __restrict__ y, int n, int tiles,
                                                 double a, double bta) {
    __shared__ double lds[640];
    const int lane = threadIdx.x;
    unsigned b = blockIdx.x;
    if (G > 1) b = (b / (8 * G)) * (8 * G) + (b & 7) * G + ((b >> 3) % G);
    const int gi = 1 + b / tiles;
    if (gi > n - 2) return;
    const int tile = b - (gi - 1) * tiles;
    const int j0 = tile * 128;
    const long long base = (4LL * n - 2) + (long long)(gi - 1) * (5LL * n - 2);
    const long long e = base + 5LL * j0 - 1 + lane;
    const long long hi = 5LL * n * n - 4LL * n - 1;
    double c[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { long long idx = e + 64 * k; idx = idx > hi ? hi : idx; c[k] = __builtin_nontemporal_load(v + idx); }
    double pc[2], pw[2], pe[2], pn[2], ps[2], xo[2];
    bool f[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = j0 + lane + 64 * h;
        f[h] = j > 0 && j < n - 1;
        if (f[h]) {
            const long long at = (long long)gi * n + j;
            const double *rl = r + at, *pl = pold + at;
            const double p0 = pl[0];
            pc[h] = fma(1.0, rl[0], bta * p0);
            pw[h] = fma(1.0, rl[-1], bta * pl[-1]);
            pe[h] = fma(1.0, rl[1], bta * pl[1]);
            pn[h] = fma(1.0, rl[-n], bta * pl[-n]);
            ps[h] = fma(1.0, rl[n], bta * pl[n]);
            if (WITHX) xo[h] = fma(a, p0, __builtin_nontemporal_load(xv + at));
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) lds[64 * k + lane] = c[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double* q = lds + 5 * (lane + 64 * h);
        if (f[h]) {
            const long long at = (long long)gi * n + j0 + lane + 64 * h;
            if (WITHX) __builtin_nontemporal_store(xo[h], xv + at);
            pnew[at] = pc[h];
            __builtin_nontemporal_store(row5(q[0], q[1], q[2], q[3], q[4], pw[h], pc[h], pe[h], pn[h], ps[h]), y + at);
        }
    }
}
// cache-policy bits on the coefficient loads of the 128-column grid tile (G = 4): POL 0 plain, 1 nt, 2 sc1,
// 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc1 nt, 6 sc0, 7 sc0 nt
template <int POL> __device__ __forceinline__ double ld_pol(const double* p) {
    double v;
    if (POL == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (POL == 1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (POL == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (POL == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (POL == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if (POL == 5) asm volatile("global_load_dwordx2 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if (POL == 6) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (POL == 7) asm volatile("global_load_dwordx2 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POL>
__global__ __launch_bounds__(64) void lds_grid_pol(const double* __restrict__ v, const double* __restrict__ x,
                                                   double* __restrict__ y, int n, int tiles) {
    __shared__ double lds[640];
    const int lane = threadIdx.x;
    unsigned b = blockIdx.x;
    b = (b / 32) * 32 + (b & 7) * 4 + ((b >> 3) % 4);
    const int gi = 1 + b / tiles;
    if (gi > n - 2) return;
    const int tile = b - (gi - 1) * tiles;
    const int j0 = tile * 128;
    const long long base = (4LL * n - 2) + (long long)(gi - 1) * (5LL * n - 2);
    const long long e = base + 5LL * j0 - 1 + lane;
    const long long hi = 5LL * n * n - 4LL * n - 1;
    double c[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { long long idx = e + 64 * k; idx = idx > hi ? hi : idx; c[k] = ld_pol<POL>(v + idx); }
    double xc[2], xw[2], xe[2], xn[2], xs[2];
    bool f[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = j0 + lane + 64 * h;
        f[h] = j > 0 && j < n - 1;
        if (f[h]) { const double* xl = x + (long long)gi * n + j; xc[h] = xl[0], xw[h] = xl[-1], xe[h] = xl[1], xn[h] = xl[-n], xs[h] = xl[n]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 10; ++k) lds[64 * k + lane] = c[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double* q = lds + 5 * (lane + 64 * h);
        if (f[h]) __builtin_nontemporal_store(row5(q[0], q[1], q[2], q[3], q[4], xw[h], xc[h], xe[h], xn[h], xs[h]), y + (long long)gi * n + j0 + lane + 64 * h);
    }
}
// grid tiles of 128 columns, coefficient run fetched as five 16-byte nontemporal loads per lane from the 16-byte
// aligned address at or just below the run (sh = 0 / 1 doubles of lead-in), one extra pair by lane 0 when sh = 1
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int G>
__global__ __launch_bounds__(64) void lds_grid16(const double* __restrict__ v, const double* __restrict__ x,
                                                 double* __restrict__ y, int n, int tiles) {
    __shared__ __attribute__((aligned(16))) double lds[656];
    const int lane = threadIdx.x;
    unsigned b = blockIdx.x;
    if (G > 1) b = (b / (8 * G)) * (8 * G) + (b & 7) * G + ((b >> 3) % G);
    const int gi = 1 + b / tiles;
    if (gi > n - 2) return;
    const int tile = b - (gi - 1) * tiles;
    const int j0 = tile * 128;
    const long long base = (4LL * n - 2) + (long long)(gi - 1) * (5LL * n - 2);
    const long long s0 = base + 5LL * j0 - 1;
    const int sh = (int)(s0 & 1);
    const long long hi2 = (5LL * n * n - 4LL * n) / 2 - 1;  // last whole pair of the array
    const dbl2* src = reinterpret_cast<const dbl2*>(v) + (s0 - sh) / 2;
    dbl2 c[5], extra = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 5; ++k) { long long idx = (s0 - sh) / 2 + lane + 64 * k; idx = idx > hi2 ? hi2 : idx; c[k] = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(v) + idx); }
    if (sh && lane == 0) { long long idx = (s0 - sh) / 2 + 320; idx = idx > hi2 ? hi2 : idx; extra = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(v) + idx); }
    (void)src;
    double xc[2], xw[2], xe[2], xn[2], xs[2];
    bool f[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = j0 + lane + 64 * h;
        f[h] = j > 0 && j < n - 1;
        if (f[h]) { const double* xl = x + (long long)gi * n + j; xc[h] = xl[0], xw[h] = xl[-1], xe[h] = xl[1], xn[h] = xl[-n], xs[h] = xl[n]; }
    }
    dbl2* l2 = reinterpret_cast<dbl2*>(lds);
#pragma unroll
    for (int k = 0; k < 5; ++k) l2[64 * k + lane] = c[k];
    if (sh && lane == 0) l2[320] = extra;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double* q = lds + sh + 5 * (lane + 64 * h);
        if (f[h]) __builtin_nontemporal_store(row5(q[0], q[1], q[2], q[3], q[4], xw[h], xc[h], xe[h], xn[h], xs[h]), y + (long long)gi * n + j0 + lane + 64 * h);
    }
}
__global__ void fill_pattern(double* p, size_t count, int mode) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    // mode 1: stencil coefficients (5 at the centre slot, -1 elsewhere); mode 2: a varying x
    p[i] = mode == 1 ? ((i % 5) == 2 ? 5.0 : -1.0) : 1.0 + 1e-3 * (double)(i % 1000);
}

template <class F> double time_ms(F&& f, int reps = 9) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m); }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}
static unsigned blocks_for(size_t items, int block) { return (unsigned)((items + block - 1) / block); }

int main(int argc, char** argv) {
    const int n = 20000;
    const size_t rows = (size_t)n * n;
    double *v, *x, *y;
    CK(hipMalloc(&v, rows * 40 + 64)); CK(hipMalloc(&x, rows * 8)); CK(hipMalloc(&y, rows * 8));
    CK(hipMemset(v, 0, rows * 40 + 64)); CK(hipMemset(x, 0, rows * 8)); CK(hipMemset(y, 0, rows * 8));
    const double mixb = rows * 56.0;
#define RUN(label, ...) do { double ms = time_ms([&] { __VA_ARGS__; }); printf("%-64s : %7.3f ms  %8.1f GB/s\n", label, ms, mixb / ms / 1e6); fflush(stdout); } while (0)
#define AOS(B, M, S, label) RUN(label, hipLaunchKernelGGL((aos<B, M, S>), dim3(blocks_for(rows, B)), dim3(B), 0, 0, v, x, y, rows, n))
#define LDS8(B, L, S, H, label) RUN(label, hipLaunchKernelGGL((lds8<B, L, S, H>), dim3(blocks_for(rows, B)), dim3(B), 0, 0, v, x, y, rows, n))
#define DMA(W, A, S, MIS, label) RUN(label, hipLaunchKernelGGL((ldsdma<W, A, S>), dim3(blocks_for(rows, W * 128)), dim3(W * 64), 0, 0, v, x, y, rows, n, MIS))
#define L8X2(W, L, S, label) RUN(label, hipLaunchKernelGGL((lds8x2<W, L, S>), dim3(blocks_for(rows, W * 128)), dim3(W * 64), 0, 0, v, x, y, rows, n))
    for (int pass = 0; pass < (argc > 1 ? atoi(argv[1]) : 0); ++pass) {
        printf("-- pass %d\n", pass);
        AOS(256, 0, false, "aos block 256 plain (the shipped shape)");
        AOS(256, 0, true, "aos block 256 nt store");
        AOS(128, 0, true, "aos block 128 nt store");
        AOS(64, 0, true, "aos block  64 nt store");
        AOS(256, 1, true, "aos block 256 nt store, first coefficient load nt");
        AOS(256, 16, true, "aos block 256 nt store, last coefficient load nt");
        AOS(256, 31, true, "aos block 256 nt store, all coefficient loads nt");
        LDS8(256, false, true, false, "lds8 block 256 plain loads, nt store");
        LDS8(256, true, true, false, "lds8 block 256 nt loads, nt store");
        LDS8(128, true, true, false, "lds8 block 128 nt loads, nt store");
        LDS8(64, true, true, false, "lds8 block  64 nt loads, nt store");
        LDS8(64, true, false, false, "lds8 block  64 nt loads, plain store");
        LDS8(64, true, true, true, "lds8 block  64 nt loads, nt store, W/E by shuffle");
        LDS8(128, true, true, true, "lds8 block 128 nt loads, nt store, W/E by shuffle");
        L8X2(1, true, true, "lds8x2 1 wave/block (128 rows) nt loads, nt store");
        L8X2(2, true, true, "lds8x2 2 waves/block nt loads, nt store");
        DMA(1, 0, true, 0, "ldsdma 1 wave/block, default policy, nt store");
        DMA(1, 2, true, 0, "ldsdma 1 wave/block, nt, nt store");
        DMA(2, 2, true, 0, "ldsdma 2 waves/block, nt, nt store");
        DMA(4, 2, true, 0, "ldsdma 4 waves/block, nt, nt store");
        DMA(2, 2, true, 1, "ldsdma 2 waves/block, nt, nt store, source 8-byte aligned only");
    }
    const int tiles = (n + 127) / 128;
    for (int data = 0; data < 2; ++data) {
        if (data == 1) {
            hipLaunchKernelGGL(fill_pattern, dim3(blocks_for(rows * 5, 256)), dim3(256), 0, 0, v, rows * 5, 1);
            hipLaunchKernelGGL(fill_pattern, dim3(blocks_for(rows, 256)), dim3(256), 0, 0, x, rows, 2);
            CK(hipDeviceSynchronize());
        }
        printf("-- data: %s\n", data ? "stencil coefficients, varying x" : "all zero");
        L8X2(1, true, true, "lds8x2 1 wave/block, aligned 1-D tiles");
        RUN("lds8x2 grid-row tiles (157 per row), 1-D launch", hipLaunchKernelGGL((lds8x2_grid<true, true, false>), dim3((unsigned)tiles * (n - 2)), dim3(64), 0, 0, v, x, y, n, tiles));
        RUN("lds8x2 grid-row tiles (157 per row), 2-D launch", hipLaunchKernelGGL((lds8x2_grid<true, true, true>), dim3(tiles, n - 2), dim3(64), 0, 0, v, x, y, n, tiles));
        RUN("lds8x2 grid-row tiles, 2-D launch, plain loads", hipLaunchKernelGGL((lds8x2_grid<false, true, true>), dim3(tiles, n - 2), dim3(64), 0, 0, v, x, y, n, tiles));
        AOS(256, 0, false, "aos block 256 plain (the shipped shape)");
#define LG(COLS, AL, PR, label) do { const int tl = (n + COLS - 1) / COLS; const unsigned nb = ((unsigned)tl * (n - 2) + 31) & ~31u; RUN(label, hipLaunchKernelGGL((lds_grid<COLS, AL, PR>), dim3(nb), dim3(64), 0, 0, v, x, y, n, tl)); } while (0)
        if (data == 1) {
#define LF(COLS, G, label) do { const long long nt_ = ((long long)n * (n - 2)) / COLS; const unsigned nb = (unsigned)((nt_ + 8 * G - 1) / (8 * G) * (8 * G)); RUN(label, hipLaunchKernelGGL((lds_flat<COLS, G>), dim3(nb), dim3(64), 0, 0, v, x, y, n, nt_)); } while (0)
            {
            // coefficients in an UNCACHED / fine-grained allocation instead of the default one
            for (unsigned flag : {hipDeviceMallocUncached, hipDeviceMallocFinegrained}) {
                double* vu = nullptr;
                if (hipExtMallocWithFlags((void**)&vu, rows * 40 + 64, flag) != hipSuccess) { printf("allocation with flag %u failed\n", flag); continue; }
                CK(hipMemcpy(vu, v, rows * 40 + 64, hipMemcpyDeviceToDevice));
                const int tl = (n + 127) / 128; const unsigned nb = ((unsigned)tl * (n - 2) + 31) & ~31u;
                for (int rep = 0; rep < 2; ++rep) {
                    double ms = time_ms([&] { hipLaunchKernelGGL((lds_grid<128, false, 4>), dim3(nb), dim3(64), 0, 0, v, x, y, n, tl); });
                    printf("coefficients in the default allocation      : %7.3f ms\n", ms);
                    ms = time_ms([&] { hipLaunchKernelGGL((lds_grid<128, false, 4>), dim3(nb), dim3(64), 0, 0, vu, x, y, n, tl); });
                    printf("coefficients in allocation with flag 0x%x : %7.3f ms\n", flag, ms); fflush(stdout);
                }
                CK(hipFree(vu));
            }
            }
        }
    }
    return 0;
}
