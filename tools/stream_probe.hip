// tools/stream_probe.hip -- measurement aid, not part of the product: what does this MI355X
// sustain for the access shapes the STENCIL5 kernel is made of? Prints GB/s per shape.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o gpurun_out/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// shape A: pure read, 16 B/lane, grid-stride, 4 loads in flight per lane
__global__ __launch_bounds__(256) void read_only(const d2* __restrict__ a, size_t n2, double* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    d2 acc = {0, 0};
    for (; i + 3 * stride < n2; i += 4 * stride) {
        d2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        acc += v0 + v1 + v2 + v3;
    }
    if (acc.x + acc.y == 123.456) out[0] = acc.x;
}
// shape A': the same bytes, one 16-byte load per thread x 4 arrays-worth per block, one-shot grid
__global__ __launch_bounds__(256) void read_only_oneshot(const d2* __restrict__ a, size_t n2, double* out) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    d2 acc = {0, 0};
    if (i + 3 < n2) acc = a[i] + a[i + 1] + a[i + 2] + a[i + 3];
    if (acc.x + acc.y == 123.456) out[0] = acc.x;
}
// shape D: the stencil's traffic mix with every stream fully coalesced at 8 B/lane, one row per thread,
// one-shot: 5 value planes (SoA) + x + y = 48 B read : 8 B written, no neighbours, no indices
__global__ __launch_bounds__(256) void mix_soa_oneshot(const double* __restrict__ v, const double* __restrict__ x,
                                                       double* __restrict__ y, size_t rows) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r < rows) y[r] = v[r] * x[r] + v[rows + r] + v[2 * rows + r] + v[3 * rows + r] + v[4 * rows + r];
}
// shape D-nt: shape D with a nontemporal store of y
__global__ __launch_bounds__(256) void mix_soa_oneshot_nt(const double* __restrict__ v, const double* __restrict__ x,
                                                          double* __restrict__ y, size_t rows) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r < rows) __builtin_nontemporal_store(v[r] * x[r] + v[rows + r] + v[2 * rows + r] + v[3 * rows + r] + v[4 * rows + r], &y[r]);
}
// shape D2: the same mix with two adjacent rows per thread, every access 16 B/lane
__global__ __launch_bounds__(256) void mix_soa_oneshot2(const d2* __restrict__ v, const d2* __restrict__ x,
                                                        d2* __restrict__ y, size_t pairs) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r < pairs) y[r] = v[r] * x[r] + v[pairs + r] + v[2 * pairs + r] + v[3 * pairs + r] + v[4 * pairs + r];
}
// shape D4: four adjacent rows per thread (two 16-byte accesses per stream)
__global__ __launch_bounds__(256) void mix_soa_oneshot4(const d2* __restrict__ v, const d2* __restrict__ x,
                                                        d2* __restrict__ y, size_t pairs) {
    const size_t r = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (r + 1 < pairs) {
        d2 a = v[r] * x[r] + v[pairs + r] + v[2 * pairs + r] + v[3 * pairs + r] + v[4 * pairs + r];
        d2 b = v[r + 1] * x[r + 1] + v[pairs + r + 1] + v[2 * pairs + r + 1] + v[3 * pairs + r + 1] + v[4 * pairs + r + 1];
        y[r] = a; y[r + 1] = b;
    }
}
// shape B: copy 16 B/lane
__global__ __launch_bounds__(256) void copy16(const d2* __restrict__ a, d2* __restrict__ b, size_t n2) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i < n2; i += stride) b[i] = a[i];
}
// shape C: per "tile" of 128 rows a wave reads 5 KiB (5 x dwordx4 per lane) of values and 1 KiB of x,
// writes 1 KiB of y: the traffic mix of STENCIL5 (48 read : 8 write) without LDS or N/S neighbours.
template <int MODE>
__global__ __launch_bounds__(256) void mix_kernel(const double* __restrict__ values, const double* __restrict__ x,
                                                  double* __restrict__ y, long long ntiles, int banded) {
    __shared__ __attribute__((aligned(16))) double lds[4 * 656];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long t0, t1, step;
    if (banded) {
        const long long per = (ntiles + 7) >> 3;
        t0 = (blockIdx.x & 7) * per + (long long)(blockIdx.x >> 3) * 4 + w;
        t1 = min((long long)((blockIdx.x & 7) + 1) * per, ntiles);
        step = (long long)(gridDim.x >> 3) * 4;
    } else {
        t0 = (long long)blockIdx.x * 4 + w; t1 = ntiles; step = (long long)gridDim.x * 4;
    }
    double* wl = lds + w * 656;
    for (long long t = t0; t < t1; t += step) {
        const d2* v = reinterpret_cast<const d2*>(values + t * 640);
        d2 c0 = v[lane], c1 = v[64 + lane], c2 = v[128 + lane], c3 = v[192 + lane], c4 = v[256 + lane];
        d2 xc = *reinterpret_cast<const d2*>(x + t * 128 + 2 * lane);
        d2 r;
        if (MODE == 0) {
            r = c0 + c1 + c2 + c3 + c4 + xc;
        } else {
            d2* w2 = reinterpret_cast<d2*>(wl);
            w2[lane] = c0; w2[64 + lane] = c1; w2[128 + lane] = c2; w2[192 + lane] = c3; w2[256 + lane] = c4;
            __builtin_amdgcn_wave_barrier();
            const double* q = wl + 10 * lane;
            double a = q[0] + q[1] + q[2] + q[3] + q[4], b = q[5] + q[6] + q[7] + q[8] + q[9];
            __builtin_amdgcn_wave_barrier();
            r.x = a * xc.x; r.y = b * xc.y;
        }
        *reinterpret_cast<d2*>(y + t * 128 + 2 * lane) = r;
    }
}

// BLAS1 shapes of the CG loop: p = r + b*p (2 reads, 1 write) and the x/r double update (4 reads, 2 writes)
template <int UNROLL>
__global__ __launch_bounds__(256) void upd_p(const d2* __restrict__ r, d2* __restrict__ p, size_t n2, double b, int oneshot) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * (oneshot ? UNROLL : 1);
    const size_t stride = oneshot ? n2 : (size_t)gridDim.x * 256;
    if (oneshot) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) if (i + u < n2) { d2 rv = r[i + u], pv = p[i + u]; pv.x = fma(b, pv.x, rv.x); pv.y = fma(b, pv.y, rv.y); p[i + u] = pv; }
    } else {
        for (; i < n2; i += stride) { d2 rv = r[i], pv = p[i]; pv.x = fma(b, pv.x, rv.x); pv.y = fma(b, pv.y, rv.y); p[i] = pv; }
    }
}
__global__ __launch_bounds__(256) void upd_xr(const d2* __restrict__ p, const d2* __restrict__ ap, d2* __restrict__ x, d2* __restrict__ r, size_t n2, double a, int oneshot, double* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = oneshot ? n2 : (size_t)gridDim.x * 256;
    double acc = 0;
    for (; i < n2; i += stride) {
        d2 pv = p[i], av = ap[i], xv = x[i], rv = r[i];
        xv.x = fma(a, pv.x, xv.x); xv.y = fma(a, pv.y, xv.y); rv.x = fma(-a, av.x, rv.x); rv.y = fma(-a, av.y, rv.y);
        x[i] = xv; r[i] = rv; acc = fma(rv.x, rv.x, acc); acc = fma(rv.y, rv.y, acc);
    }
    if (acc == 123.456) out[0] = acc;
}

template <class F> double time_ms(F&& f, int reps = 7) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m); }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char** argv) {
    const long long rows = argc > 1 ? atoll(argv[1]) : 400000000LL;
    const long long ntiles = rows / 128;
    double *values, *x, *y;
    CK(hipMalloc(&values, (size_t)ntiles * 640 * 8)); CK(hipMalloc(&x, (size_t)rows * 8)); CK(hipMalloc(&y, (size_t)rows * 8));
    CK(hipMemset(values, 0, (size_t)ntiles * 640 * 8)); CK(hipMemset(x, 0, (size_t)rows * 8));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, rows %lld\n", p.name, p.multiProcessorCount, rows);
    const size_t vb = (size_t)ntiles * 640 * 8;
    for (int bpc : {4, 8, 16}) {
        int grid = p.multiProcessorCount * bpc;
        double ms = time_ms([&] { hipLaunchKernelGGL(read_only, dim3(grid), dim3(256), 0, 0, (const d2*)values, vb / 16, y); });
        printf("read_only   blocks/CU %2d : %7.3f ms  %8.1f GB/s\n", bpc, ms, vb / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const d2*)values, (d2*)x, (size_t)rows * 8 / 16); });
        printf("copy16      blocks/CU %2d : %7.3f ms  %8.1f GB/s (read+write)\n", bpc, ms, 2.0 * rows * 8 / ms / 1e6);
    }
    {
        double ms = time_ms([&] { hipLaunchKernelGGL(read_only_oneshot, dim3((unsigned)((vb / 16 / 4 + 255) / 256)), dim3(256), 0, 0, (const d2*)values, vb / 16, y); });
        printf("read_only   one-shot 64 B/thread   : %7.3f ms  %8.1f GB/s\n", ms, vb / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(mix_soa_oneshot, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, values, x, y, (size_t)rows); });
        printf("mix SoA     one-shot 1 row/thread  : %7.3f ms  %8.1f GB/s (48 B read : 8 B write per row)\n", ms, rows * 56.0 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(mix_soa_oneshot_nt, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, values, x, y, (size_t)rows); });
        printf("mix SoA     one-shot 1 row/thread, nontemporal store : %7.3f ms  %8.1f GB/s\n", ms, rows * 56.0 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(mix_soa_oneshot2, dim3((unsigned)((rows / 2 + 255) / 256)), dim3(256), 0, 0, (const d2*)values, (const d2*)x, (d2*)y, (size_t)rows / 2); });
        printf("mix SoA     one-shot 2 rows/thread : %7.3f ms  %8.1f GB/s (16 B/lane)\n", ms, rows * 56.0 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(mix_soa_oneshot4, dim3((unsigned)((rows / 4 + 255) / 256)), dim3(256), 0, 0, (const d2*)values, (const d2*)x, (d2*)y, (size_t)rows / 2); });
        printf("mix SoA     one-shot 4 rows/thread : %7.3f ms  %8.1f GB/s (2 x 16 B/lane)\n", ms, rows * 56.0 / ms / 1e6);
    }
    const double mixbytes = (double)rows * 56;
    for (int banded : {0, 1}) for (int bpc : {4, 7, 8, 12, 16}) {
        int grid = (p.multiProcessorCount * bpc + 7) & ~7;
        double ms = time_ms([&] { hipLaunchKernelGGL(mix_kernel<0>, dim3(grid), dim3(256), 0, 0, values, x, y, ntiles, banded); });
        printf("mix nolds   banded %d blocks/CU %2d : %7.3f ms  %8.1f GB/s\n", banded, bpc, ms, mixbytes / ms / 1e6);
        if (bpc <= 7) {
            ms = time_ms([&] { hipLaunchKernelGGL(mix_kernel<1>, dim3(grid), dim3(256), 0, 0, values, x, y, ntiles, banded); });
            printf("mix lds     banded %d blocks/CU %2d : %7.3f ms  %8.1f GB/s\n", banded, bpc, ms, mixbytes / ms / 1e6);
        }
    }
    {
        double *a1, *a2;
        CK(hipMalloc(&a1, (size_t)rows * 8)); CK(hipMalloc(&a2, (size_t)rows * 8));
        CK(hipMemset(a1, 0, (size_t)rows * 8)); CK(hipMemset(a2, 0, (size_t)rows * 8));
        const size_t n2 = (size_t)rows / 2;
        for (int blocks : {512, 1024, 2048, 4096, 8192}) {
            double ms = time_ms([&] { hipLaunchKernelGGL(upd_p<1>, dim3(blocks), dim3(256), 0, 0, (const d2*)x, (d2*)y, n2, 0.5, 0); });
            printf("upd_p  grid-stride blocks %5d : %7.3f ms  %8.1f GB/s\n", blocks, ms, rows * 24.0 / ms / 1e6);
            ms = time_ms([&] { hipLaunchKernelGGL(upd_xr, dim3(blocks), dim3(256), 0, 0, (const d2*)x, (const d2*)y, (d2*)a1, (d2*)a2, n2, 0.5, 0, values); });
            printf("upd_xr grid-stride blocks %5d : %7.3f ms  %8.1f GB/s\n", blocks, ms, rows * 48.0 / ms / 1e6);
        }
        double ms = time_ms([&] { hipLaunchKernelGGL(upd_p<1>, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, (const d2*)x, (d2*)y, n2, 0.5, 1); });
        printf("upd_p  one-shot 16 B/thread      : %7.3f ms  %8.1f GB/s\n", ms, rows * 24.0 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(upd_p<2>, dim3((unsigned)((n2 / 2 + 255) / 256)), dim3(256), 0, 0, (const d2*)x, (d2*)y, n2, 0.5, 1); });
        printf("upd_p  one-shot 32 B/thread      : %7.3f ms  %8.1f GB/s\n", ms, rows * 24.0 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(upd_p<4>, dim3((unsigned)((n2 / 4 + 255) / 256)), dim3(256), 0, 0, (const d2*)x, (d2*)y, n2, 0.5, 1); });
        printf("upd_p  one-shot 64 B/thread      : %7.3f ms  %8.1f GB/s\n", ms, rows * 24.0 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(upd_xr, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, (const d2*)x, (const d2*)y, (d2*)a1, (d2*)a2, n2, 0.5, 1, values); });
        printf("upd_xr one-shot 16 B/thread      : %7.3f ms  %8.1f GB/s\n", ms, rows * 48.0 / ms / 1e6);
        CK(hipFree(a1)); CK(hipFree(a2));
    }
    // one tile per wave, non-persistent
    {
        double ms = time_ms([&] { hipLaunchKernelGGL(mix_kernel<0>, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, 0, values, x, y, ntiles, 0); });
        printf("mix nolds   one tile per wave (grid %lld) : %7.3f ms  %8.1f GB/s\n", (ntiles + 3) / 4, ms, mixbytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(mix_kernel<1>, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, 0, values, x, y, ntiles, 0); });
        printf("mix lds     one tile per wave : %7.3f ms  %8.1f GB/s\n", ms, mixbytes / ms / 1e6);
    }
    return 0;
}
