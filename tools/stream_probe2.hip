// tools/stream_probe2.hip -- measurement aid, not part of the product. Second set of questions about
// what bounds the STENCIL5 traffic mix (48 B read : 8 B written per row) on MI355X:
//   * is it the number of read streams, or the store stream? (reads only / stores only / mix)
//   * does the workgroup size, the store policy (plain / nontemporal) or the load policy matter?
//   * does the real layout (five coefficients at lane stride 40 B) cost anything against five planes?
//   * does batching a block's stores behind all of its loads help?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/stream_probe2.hip -o tools/bin/stream_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ double ld(const double* p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT> __device__ __forceinline__ void st(double* p, double v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// five planes + x read, y written (STORE) or folded into a never-true test (no store at all)
template <int BLOCK, bool STORE, bool NTLD, bool NTST>
__global__ __launch_bounds__(BLOCK) void mix_soa(const double* __restrict__ v, const double* __restrict__ x,
                                                 double* __restrict__ y, size_t rows) {
    const size_t r = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= rows) return;
    const double s = ld<NTLD>(v + r) * x[r] + ld<NTLD>(v + rows + r) + ld<NTLD>(v + 2 * rows + r) +
                     ld<NTLD>(v + 3 * rows + r) + ld<NTLD>(v + 4 * rows + r);
    if (STORE) st<NTST>(y + r, s);
    else if (s == 123.456) y[r] = s;
}
// store policy through the ISA's cache-control bits (POL: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc1 nt, 6 sc0)
template <int POL> __device__ __forceinline__ void st_pol(double* p, double v) {
    if (POL == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 5) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 6) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
template <int POL>
__global__ __launch_bounds__(256) void mix_soa_pol(const double* __restrict__ v, const double* __restrict__ x,
                                                   double* __restrict__ y, size_t rows) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const double s = v[r] * x[r] + v[rows + r] + v[2 * rows + r] + v[3 * rows + r] + v[4 * rows + r];
    st_pol<POL>(y + r, s);
}
// the real layout: v[5 r .. 5 r + 4], lane stride 40 B
template <int BLOCK, bool NTST, int XN>
__global__ __launch_bounds__(BLOCK) void mix_aos(const double* __restrict__ v, const double* __restrict__ x,
                                                 double* __restrict__ y, size_t rows, int n) {
    const size_t r = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= rows) return;
    const double* q = v + 5 * r;
    double s = q[1] * x[r];
    s = fma(q[2], q[3], s);
    s = fma(q[0], q[4], s);
    if (XN) {  // with the four neighbour loads of the real kernel
        if (r >= (size_t)n && r + n < rows) s += x[r - 1] + x[r + 1] + x[r - n] + x[r + n];
    }
    st<NTST>(y + r, s);
}
// stores only
template <bool NTST> __global__ __launch_bounds__(256) void write_only(double* __restrict__ y, size_t rows) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r < rows) st<NTST>(y + r, (double)r);
}
template <bool NTST> __global__ __launch_bounds__(256) void write_only16(d2* __restrict__ y, size_t pairs) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    d2 v = {(double)r, 1.0};
    if (r < pairs) { if (NTST) __builtin_nontemporal_store(v, y + r); else y[r] = v; }
}
// copy, one-shot, 16 B per lane
template <bool NTLD, bool NTST> __global__ __launch_bounds__(256) void copy_oneshot(const d2* __restrict__ a, d2* __restrict__ b, size_t n2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) { d2 v = NTLD ? __builtin_nontemporal_load(a + i) : a[i]; if (NTST) __builtin_nontemporal_store(v, b + i); else b[i] = v; }
}
// K rows per thread at block stride: every load of the block is issued before its first store
template <int K, bool NTST>
__global__ __launch_bounds__(256) void mix_soa_batched(const double* __restrict__ v, const double* __restrict__ x,
                                                       double* __restrict__ y, size_t rows) {
    const size_t r0 = (size_t)blockIdx.x * (256 * K) + threadIdx.x;
    double s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t r = r0 + (size_t)k * 256;
        s[k] = r < rows ? v[r] * x[r] + v[rows + r] + v[2 * rows + r] + v[3 * rows + r] + v[4 * rows + r] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t r = r0 + (size_t)k * 256;
        if (r < rows) st<NTST>(y + r, s[k]);
    }
}

template <class F> double time_ms(F&& f, int reps = 7) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m); }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}
static unsigned blocks_for(size_t items, int block) { return (unsigned)((items + block - 1) / block); }

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? (size_t)atoll(argv[1]) : 400000000ULL;
    const int n = 20000;
    double *v, *x, *y;
    CK(hipMalloc(&v, rows * 40)); CK(hipMalloc(&x, rows * 8)); CK(hipMalloc(&y, rows * 8));
    CK(hipMemset(v, 0, rows * 40)); CK(hipMemset(x, 0, rows * 8)); CK(hipMemset(y, 0, rows * 8));
    const double mixb = rows * 56.0, rdb = rows * 48.0;
#define RUN(label, bytes, ...) do { double ms = time_ms([&] { __VA_ARGS__; }); printf("%-58s : %7.3f ms  %8.1f GB/s\n", label, ms, (bytes) / ms / 1e6); fflush(stdout); } while (0)
    RUN("read 6 streams (5 planes + x), no store, block 256", rdb, hipLaunchKernelGGL((mix_soa<256, false, false, false>), dim3(blocks_for(rows, 256)), dim3(256), 0, 0, v, x, y, rows));
    RUN("read 6 streams, nontemporal loads", rdb, hipLaunchKernelGGL((mix_soa<256, false, true, false>), dim3(blocks_for(rows, 256)), dim3(256), 0, 0, v, x, y, rows));
    RUN("write only 8 B/lane", rows * 8.0, hipLaunchKernelGGL(write_only<false>, dim3(blocks_for(rows, 256)), dim3(256), 0, 0, y, rows));
    RUN("write only 8 B/lane nontemporal", rows * 8.0, hipLaunchKernelGGL(write_only<true>, dim3(blocks_for(rows, 256)), dim3(256), 0, 0, y, rows));
    RUN("write only 16 B/lane", rows * 8.0, hipLaunchKernelGGL(write_only16<false>, dim3(blocks_for(rows / 2, 256)), dim3(256), 0, 0, (d2*)y, rows / 2));
    RUN("write only 16 B/lane nontemporal", rows * 8.0, hipLaunchKernelGGL(write_only16<true>, dim3(blocks_for(rows / 2, 256)), dim3(256), 0, 0, (d2*)y, rows / 2));
    RUN("copy one-shot 16 B/lane (3.2 GB -> 3.2 GB)", rows * 16.0, hipLaunchKernelGGL((copy_oneshot<false, false>), dim3(blocks_for(rows / 2, 256)), dim3(256), 0, 0, (const d2*)x, (d2*)y, rows / 2));
    RUN("copy one-shot 16 B/lane, nt store", rows * 16.0, hipLaunchKernelGGL((copy_oneshot<false, true>), dim3(blocks_for(rows / 2, 256)), dim3(256), 0, 0, (const d2*)x, (d2*)y, rows / 2));
    RUN("copy one-shot 16 B/lane, nt load + nt store", rows * 16.0, hipLaunchKernelGGL((copy_oneshot<true, true>), dim3(blocks_for(rows / 2, 256)), dim3(256), 0, 0, (const d2*)x, (d2*)y, rows / 2));
#define MIXB(B) \
    RUN("mix SoA block " #B " plain", mixb, hipLaunchKernelGGL((mix_soa<B, true, false, false>), dim3(blocks_for(rows, B)), dim3(B), 0, 0, v, x, y, rows)); \
    RUN("mix SoA block " #B " nt store", mixb, hipLaunchKernelGGL((mix_soa<B, true, false, true>), dim3(blocks_for(rows, B)), dim3(B), 0, 0, v, x, y, rows)); \
    RUN("mix SoA block " #B " nt load + nt store", mixb, hipLaunchKernelGGL((mix_soa<B, true, true, true>), dim3(blocks_for(rows, B)), dim3(B), 0, 0, v, x, y, rows));
    MIXB(64) MIXB(128) MIXB(256) MIXB(512) MIXB(1024)
#define POLR(P, name) RUN("mix SoA block 256 store policy " name, mixb, hipLaunchKernelGGL(mix_soa_pol<P>, dim3(blocks_for(rows, 256)), dim3(256), 0, 0, v, x, y, rows));
    POLR(0, "plain(asm)") POLR(1, "nt") POLR(2, "sc1") POLR(3, "sc0 sc1") POLR(4, "sc0 sc1 nt") POLR(5, "sc1 nt") POLR(6, "sc0")
    RUN("mix AoS (lane stride 40 B) block 256 plain", mixb, hipLaunchKernelGGL((mix_aos<256, false, 0>), dim3(blocks_for(rows, 256)), dim3(256), 0, 0, v, x, y, rows, n));
    RUN("mix AoS block 256 nt store", mixb, hipLaunchKernelGGL((mix_aos<256, true, 0>), dim3(blocks_for(rows, 256)), dim3(256), 0, 0, v, x, y, rows, n));
    RUN("mix AoS + 4 neighbour loads, plain", mixb, hipLaunchKernelGGL((mix_aos<256, false, 1>), dim3(blocks_for(rows, 256)), dim3(256), 0, 0, v, x, y, rows, n));
    RUN("mix AoS + 4 neighbour loads, nt store", mixb, hipLaunchKernelGGL((mix_aos<256, true, 1>), dim3(blocks_for(rows, 256)), dim3(256), 0, 0, v, x, y, rows, n));
    RUN("mix AoS block 512 nt store", mixb, hipLaunchKernelGGL((mix_aos<512, true, 0>), dim3(blocks_for(rows, 512)), dim3(512), 0, 0, v, x, y, rows, n));
    RUN("mix AoS block 1024 nt store", mixb, hipLaunchKernelGGL((mix_aos<1024, true, 0>), dim3(blocks_for(rows, 1024)), dim3(1024), 0, 0, v, x, y, rows, n));
    RUN("mix SoA batched 4 rows/thread plain", mixb, hipLaunchKernelGGL((mix_soa_batched<4, false>), dim3(blocks_for(rows, 1024)), dim3(256), 0, 0, v, x, y, rows));
    RUN("mix SoA batched 4 rows/thread nt store", mixb, hipLaunchKernelGGL((mix_soa_batched<4, true>), dim3(blocks_for(rows, 1024)), dim3(256), 0, 0, v, x, y, rows));
    RUN("mix SoA batched 8 rows/thread nt store", mixb, hipLaunchKernelGGL((mix_soa_batched<8, true>), dim3(blocks_for(rows, 2048)), dim3(256), 0, 0, v, x, y, rows));
    return 0;
}
