// tools/placement_probe.hip -- measurement aid, not part of the product. Does the RATE of a streaming kernel depend on
// WHICH allocation it runs on? Round 3 saw the same SpMV kernel at 3.66 and 3.78 ms on two `values` arrays of one process
// (profiles/r03_cpu_baseline_full_size_and_variance.txt); virtual offsets inside an allocation were ruled out
// (profiles/r03_placement.txt). This probe allocates COUNT buffers of BYTES bytes, fills them with non-zero data and times,
// per buffer, a read-only pass (16-byte nontemporal loads, one-wave workgroups, the solver's access shape) and a
// read-modify-write pass, twice in opposite buffer orders: a buffer that is slow both times is slow because of where it lies.
//   tools/bin/placement_probe [count=16] [GiB per buffer=3] [reps=7]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e = (x);                                                               \
        if (e != hipSuccess) {                                                            \
            printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__);             \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64) void fill_kernel(d2* p, size_t pairs) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i < pairs) p[i] = d2{1.0 + (double)(i & 1023) * 1e-3, -0.5};
}
__global__ __launch_bounds__(64) void read_kernel(const d2* __restrict__ p, size_t pairs, double* __restrict__ sink) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    double acc = 0.0;
    if (i < pairs) {
        const d2 v = __builtin_nontemporal_load(p + i);
        acc = v.x + v.y;
    }
    if (acc == 123456.789) sink[0] = acc;  // never true: keeps the load
}
__global__ __launch_bounds__(64) void rmw_kernel(d2* p, size_t pairs) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i < pairs) {
        d2 v = __builtin_nontemporal_load(p + i);
        v.x = v.x * 1.0000001;
        v.y = v.y * 0.9999999;
        __builtin_nontemporal_store(v, p + i);
    }
}

static float median_ms(std::vector<float> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char** argv) {
    const int count = argc > 1 ? atoi(argv[1]) : 16;
    const double gib = argc > 2 ? atof(argv[2]) : 3.0;
    const int reps = argc > 3 ? atoi(argv[3]) : 7;
    const size_t bytes = (size_t)(gib * (double)(1ull << 30)) / 4096 * 4096;
    const size_t pairs = bytes / 16;
    const unsigned grid = (unsigned)((pairs + 63) / 64);
    std::vector<d2*> buf((size_t)count, nullptr);
    double* sink = nullptr;
    CK(hipMalloc(&sink, 8));
    for (int b = 0; b < count; ++b) {
        CK(hipMalloc(&buf[b], bytes));
        hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(64), 0, 0, buf[b], pairs);
    }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_one = [&](int b, int kind) {
        std::vector<float> ms;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(e0, 0));
            if (kind == 0)
                hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(64), 0, 0, buf[b], pairs, sink);
            else
                hipLaunchKernelGGL(rmw_kernel, dim3(grid), dim3(64), 0, 0, buf[b], pairs);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t = 0.f;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) ms.push_back(t);
        }
        return median_ms(ms);
    };
    printf("%d buffers of %.2f GiB, median of %d launches each; pass A in allocation order, pass B in reverse order\n", count, gib, reps);
    std::vector<float> rdA((size_t)count), rdB((size_t)count), wrA((size_t)count), wrB((size_t)count);
    for (int b = 0; b < count; ++b) rdA[b] = time_one(b, 0), wrA[b] = time_one(b, 1);
    for (int b = count - 1; b >= 0; --b) rdB[b] = time_one(b, 0), wrB[b] = time_one(b, 1);
    printf("%4s %18s  %10s %10s %9s  %10s %10s %9s\n", "buf", "address", "read A ms", "read B ms", "read GB/s", "rmw A ms", "rmw B ms", "rmw GB/s");
    for (int b = 0; b < count; ++b)
        printf("%4d %18p  %10.4f %10.4f %9.1f  %10.4f %10.4f %9.1f\n", b, (void*)buf[b], rdA[b], rdB[b],
               (double)bytes / (0.5 * (rdA[b] + rdB[b]) * 1e-3) / 1e9, wrA[b], wrB[b], 2.0 * (double)bytes / (0.5 * (wrA[b] + wrB[b]) * 1e-3) / 1e9);
    auto spread = [&](const std::vector<float>& a, const std::vector<float>& b2, const char* what) {
        std::vector<float> m((size_t)count);
        float noise = 0.f;
        for (int b = 0; b < count; ++b) m[b] = 0.5f * (a[b] + b2[b]), noise = std::max(noise, std::abs(a[b] - b2[b]) / m[b]);
        const float lo = *std::min_element(m.begin(), m.end()), hi = *std::max_element(m.begin(), m.end());
        printf("%s: fastest %.4f ms, slowest %.4f ms (+%.2f %%); largest A-vs-B difference of one buffer %.2f %%\n", what, lo, hi,
               100.0 * (hi / lo - 1.0), 100.0 * noise);
    };
    spread(rdA, rdB, "read");
    spread(wrA, wrB, "read-modify-write");
    for (int b = 0; b < count; ++b) CK(hipFree(buf[b]));
    CK(hipFree(sink));
    return 0;
}
