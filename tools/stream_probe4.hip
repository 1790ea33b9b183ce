// tools/stream_probe4.hip -- measurement aid, not part of the product. The two BLAS1 passes of the CG
// iteration on non-zero data: r -= a Ap with the r.r partial (2 reads, 1 write) and the fused
// x += a p ; p = r + b p (3 reads, 2 writes), as functions of access width, workgroup size, load / store
// policy and the number of consecutive workgroups an XCD takes.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/stream_probe4.hip -o tools/bin/stream_probe4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int VEC> struct Vec;
template <> struct Vec<1> { typedef double T; };
template <> struct Vec<2> { typedef d2 T; };
template <bool NT, class T> __device__ __forceinline__ T ld(const T* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT, class T> __device__ __forceinline__ void st(T* p, T v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ double sq(double v) { return v * v; }
__device__ __forceinline__ double sq(d2 v) { return v.x * v.x + v.y * v.y; }

template <int G> __device__ __forceinline__ size_t regroup(size_t b) {
    return G > 1 ? (b / (8 * G)) * (8 * G) + (b & 7) * G + ((b >> 3) % G) : b;
}

template <int VEC, int BLOCK, bool NTLD, bool NTST, int G>
__global__ __launch_bounds__(BLOCK) void upd_r(const double* __restrict__ ap, double* __restrict__ r, size_t items,
                                               double a, double* __restrict__ partials) {
    typedef typename Vec<VEC>::T T;
    const size_t i = regroup<G>(blockIdx.x) * BLOCK + threadIdx.x;
    double acc = 0.0;
    if (i < items) {
        const T av = ld<NTLD>(reinterpret_cast<const T*>(ap) + i);
        T rv = ld<NTLD>(reinterpret_cast<const T*>(r) + i);
        rv = rv - a * av;
        st<NTST>(reinterpret_cast<T*>(r) + i, rv);
        acc = sq(rv);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) partials[(blockIdx.x * BLOCK + threadIdx.x) >> 6] = acc;
}

template <int VEC, int BLOCK, bool NTLD, bool NTST, int G>
__global__ __launch_bounds__(BLOCK) void upd_px(const double* __restrict__ r, double* __restrict__ p,
                                                double* __restrict__ x, size_t items, double a, double b) {
    typedef typename Vec<VEC>::T T;
    const size_t i = regroup<G>(blockIdx.x) * BLOCK + threadIdx.x;
    if (i < items) {
        T pv = ld<NTLD>(reinterpret_cast<const T*>(p) + i);
        T xv = ld<NTLD>(reinterpret_cast<const T*>(x) + i);
        const T rv = ld<NTLD>(reinterpret_cast<const T*>(r) + i);
        xv = xv + a * pv;
        pv = rv + b * pv;
        st<NTST>(reinterpret_cast<T*>(x) + i, xv);
        st<false>(reinterpret_cast<T*>(p) + i, pv);  // p is re-read by the SpMV's neighbour loads: plain store
    }
}
// one wave per workgroup, K elements per lane at stride 64 (8-byte accesses), all loads issued first
template <int K, bool NTLD, bool NTST>
__global__ __launch_bounds__(64) void upd_r_k(const double* __restrict__ ap, double* __restrict__ r, size_t n, double a,
                                              double* __restrict__ partials) {
    const size_t i0 = (size_t)blockIdx.x * (64 * K) + threadIdx.x;
    double av[K], rv[K], acc = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) if (i0 + 64 * k < n) { av[k] = ld<NTLD>(ap + i0 + 64 * k); rv[k] = ld<NTLD>(r + i0 + 64 * k); }
#pragma unroll
    for (int k = 0; k < K; ++k) if (i0 + 64 * k < n) { rv[k] = fma(-a, av[k], rv[k]); st<NTST>(r + i0 + 64 * k, rv[k]); acc = fma(rv[k], rv[k], acc); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
template <int K, bool NTLD, bool NTST>
__global__ __launch_bounds__(64) void upd_px_k(const double* __restrict__ r, double* __restrict__ p, double* __restrict__ x,
                                               size_t n, double a, double b) {
    const size_t i0 = (size_t)blockIdx.x * (64 * K) + threadIdx.x;
    double pv[K], xv[K], rv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) if (i0 + 64 * k < n) { pv[k] = ld<NTLD>(p + i0 + 64 * k); xv[k] = ld<NTLD>(x + i0 + 64 * k); rv[k] = ld<NTLD>(r + i0 + 64 * k); }
#pragma unroll
    for (int k = 0; k < K; ++k) if (i0 + 64 * k < n) { st<NTST>(x + i0 + 64 * k, fma(a, pv[k], xv[k])); p[i0 + 64 * k] = fma(1.0, rv[k], b * pv[k]); }
}
__global__ void fill_pattern(double* p, size_t count) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) p[i] = 1.0 + 1e-3 * (double)(i % 1000);
}

template <class F> double time_ms(F&& f, int reps = 9) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m); }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main() {
    const size_t rows = 400000000ULL;
    double *r, *p, *x, *ap, *part;
    CK(hipMalloc(&r, rows * 8)); CK(hipMalloc(&p, rows * 8)); CK(hipMalloc(&x, rows * 8)); CK(hipMalloc(&ap, rows * 8));
    CK(hipMalloc(&part, rows / 64 * 8 + 4096));
    for (double* q : {r, p, x, ap}) hipLaunchKernelGGL(fill_pattern, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, q, rows);
    CK(hipDeviceSynchronize());
#define GRID(VEC, BLOCK, G) dim3((unsigned)(((rows / VEC + BLOCK - 1) / BLOCK + 8 * G - 1) / (8 * G) * (8 * G)))
#define R(VEC, BLOCK, NL, NS, G) do { double ms = time_ms([&] { hipLaunchKernelGGL((upd_r<VEC, BLOCK, NL, NS, G>), GRID(VEC, BLOCK, G), dim3(BLOCK), 0, 0, ap, r, rows / VEC, 1e-9, part); }); \
    printf("upd_r  %2d B/lane block %4d ntld %d ntst %d group %2d : %7.3f ms  %8.1f GB/s\n", 8 * VEC, BLOCK, NL, NS, G, ms, rows * 24.0 / ms / 1e6); fflush(stdout); } while (0)
#define PX(VEC, BLOCK, NL, NS, G) do { double ms = time_ms([&] { hipLaunchKernelGGL((upd_px<VEC, BLOCK, NL, NS, G>), GRID(VEC, BLOCK, G), dim3(BLOCK), 0, 0, r, p, x, rows / VEC, 1e-9, 0.999); }); \
    printf("upd_px %2d B/lane block %4d ntld %d ntst %d group %2d : %7.3f ms  %8.1f GB/s\n", 8 * VEC, BLOCK, NL, NS, G, ms, rows * 40.0 / ms / 1e6); fflush(stdout); } while (0)
#define BOTH(...) R(__VA_ARGS__); PX(__VA_ARGS__)
#define RK(K, NL, NS) do { double ms = time_ms([&] { hipLaunchKernelGGL((upd_r_k<K, NL, NS>), dim3((unsigned)((rows + 64 * K - 1) / (64 * K))), dim3(64), 0, 0, ap, r, rows, 1e-9, part); }); \
    printf("upd_r_k  %d per lane ntld %d ntst %d : %7.3f ms  %8.1f GB/s\n", K, NL, NS, ms, rows * 24.0 / ms / 1e6); fflush(stdout); } while (0)
#define PK(K, NL, NS) do { double ms = time_ms([&] { hipLaunchKernelGGL((upd_px_k<K, NL, NS>), dim3((unsigned)((rows + 64 * K - 1) / (64 * K))), dim3(64), 0, 0, r, p, x, rows, 1e-9, 0.999); }); \
    printf("upd_px_k %d per lane ntld %d ntst %d : %7.3f ms  %8.1f GB/s\n", K, NL, NS, ms, rows * 40.0 / ms / 1e6); fflush(stdout); } while (0)
    for (int rep = 0; rep < 2; ++rep) {
        RK(1, true, true); PK(1, true, true); RK(2, true, true); PK(2, true, true); RK(4, true, true); PK(4, true, true); RK(8, true, true); PK(8, true, true);
        RK(2, true, false); PK(2, true, false); RK(2, false, false); PK(2, false, false);
    }
    BOTH(2, 256, false, false, 1);  // the shipped shape
    BOTH(1, 64, true, true, 1);
    BOTH(2, 64, true, true, 1);
    return 0;
}
