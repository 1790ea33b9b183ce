// tools/vmm_probe.hip -- measurement aid, not part of the product. Can the solver BUILD its vectors out of physical memory of a
// chosen class? HIP's virtual-memory API hands out physical memory in chunks (hipMemCreate) that are mapped
// into a reserved address range: allocate `count` chunks of 1 GiB, sort them into classes with the r-update pair kernel
// (chunk k against chunk 0, and the others against one another), then map vectors of 3 chunks (3 GiB) from chunks of ONE class
// and from mixed classes and time the same kernel on them -- next to plain hipMalloc vectors of the same size.
//   tools/bin/vmm_probe [chunks=96]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                          \
        }                                                                     \
    } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64) void fill_kernel(d2* p, size_t pairs, double v) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i < pairs) p[i] = d2{v, -v};
}
__global__ __launch_bounds__(64) void upd_r(const d2* __restrict__ ap, d2* __restrict__ r, size_t pairs, double a,
                                            double* __restrict__ partials) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    double acc = 0.0;
    if (i < pairs) {
        const d2 av = __builtin_nontemporal_load(ap + i);
        d2 rv = __builtin_nontemporal_load(r + i);
        rv.x = fma(-a, av.x, rv.x);
        rv.y = fma(-a, av.y, rv.y);
        __builtin_nontemporal_store(rv, r + i);
        acc = rv.x * rv.x + rv.y * rv.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int count = argc > 1 ? atoi(argv[1]) : 96;
    const size_t GiB = (size_t)1 << 30;
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("allocation granularity (recommended): %zu bytes; %d chunks of 1 GiB\n", gran, count);
    std::vector<hipMemGenericAllocationHandle_t> h((size_t)count);
    for (int k = 0; k < count; ++k) CK(hipMemCreate(&h[k], GiB, &prop, 0));
    // view 1: every chunk on its own, in creation order
    char* flat = nullptr;
    CK(hipMemAddressReserve((void**)&flat, (size_t)count * GiB, GiB, nullptr, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int k = 0; k < count; ++k) CK(hipMemMap(flat + (size_t)k * GiB, GiB, 0, h[k], 0));
    CK(hipMemSetAccess(flat, (size_t)count * GiB, &acc, 1));
    const size_t chunk_pairs = GiB / 16;
    const unsigned chunk_grid = (unsigned)(chunk_pairs / 64);
    double* partials = nullptr;
    CK(hipMalloc(&partials, (size_t)3 * chunk_grid * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const d2* a, d2* b, size_t pairs) {
        std::vector<float> ms;
        const unsigned grid = (unsigned)((pairs + 63) / 64);
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(upd_r, dim3(grid), dim3(64), 0, 0, a, b, pairs, 1e-9, partials);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t = 0.f;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return ms[1];
    };
    for (int k = 0; k < count; ++k)
        hipLaunchKernelGGL(fill_kernel, dim3(chunk_grid), dim3(64), 0, 0, reinterpret_cast<d2*>(flat + (size_t)k * GiB), chunk_pairs, 1.0 + k);
    CK(hipDeviceSynchronize());
    auto chunk = [&](int k) { return reinterpret_cast<d2*>(flat + (size_t)k * GiB); };
    // classes: a chunk joins the first class whose representative it pairs fast with
    std::vector<float> t0((size_t)count, 0.f);
    for (int k = 1; k < count; ++k) t0[k] = timed(chunk(k), chunk(0), chunk_pairs);
    const float fast = *std::min_element(t0.begin() + 1, t0.end());
    std::vector<int> reps = {0};
    std::vector<int> cls((size_t)count, -1);
    cls[0] = 0;
    for (int k = 1; k < count; ++k) {
        for (size_t c = 0; c < reps.size() && cls[k] < 0; ++c) {
            const float t = reps[c] == 0 ? t0[k] : timed(chunk(k), chunk(reps[c]), chunk_pairs);
            if (t < 1.03f * fast) cls[k] = (int)c;
        }
        if (cls[k] < 0 && reps.size() < 4) {
            reps.push_back(k);
            cls[k] = (int)reps.size() - 1;
        }
    }
    printf("pair kernel on 1 GiB chunks: fast mode %.4f ms; class of every chunk in creation order:\n   ", fast);
    for (int k = 0; k < count; ++k) printf("%c", cls[k] < 0 ? '?' : (char)('A' + cls[k]));
    printf("\n");
    // view 2: vectors of 3 chunks mapped from chosen chunks (a second mapping of the same physical memory)
    std::vector<std::vector<int>> by((size_t)4);
    for (int k = 0; k < count; ++k)
        if (cls[k] >= 0) by[(size_t)cls[k]].push_back(k);
    char* vecs = nullptr;
    const int nvec = 6;
    CK(hipMemAddressReserve((void**)&vecs, (size_t)nvec * 3 * GiB, GiB, nullptr, 0));
    auto build = [&](int v, int c0, int c1, int c2) {
        const int src[3] = {c0, c1, c2};
        for (int j = 0; j < 3; ++j) CK(hipMemMap(vecs + ((size_t)v * 3 + j) * GiB, GiB, 0, h[src[j]], 0));
    };
    if (by[0].size() >= 9 && by[1].size() >= 6) {
        const std::vector<int>&A = by[0], &Bc = by[1];
        build(0, A[0], A[1], A[2]);      // class A
        build(1, A[3], A[4], A[5]);      // class A
        build(2, Bc[0], Bc[1], Bc[2]);   // class B
        build(3, A[6], Bc[3], A[7]);     // mixed A B A
        build(4, Bc[4], A[8], Bc[5]);    // mixed B A B
        build(5, A[A.size() - 1], A[A.size() - 2], A[A.size() - 3]);  // class A, chunks from the far end, reverse order
        CK(hipMemSetAccess(vecs, (size_t)nvec * 3 * GiB, &acc, 1));
        auto vec = [&](int v) { return reinterpret_cast<d2*>(vecs + (size_t)v * 3 * GiB); };
        const size_t vpairs = 3 * chunk_pairs;
        CK(hipFree(partials));
        CK(hipMalloc(&partials, (size_t)(vpairs / 64 + 1) * 8));
        printf("3 GiB vectors mapped from chosen chunks, r-update pair kernel (ms):\n");
        printf("   A A A  with  A A A            %.4f\n", timed(vec(0), vec(1), vpairs));
        printf("   A A A  with  A A A (far end)  %.4f\n", timed(vec(0), vec(5), vpairs));
        printf("   A A A  with  B B B            %.4f\n", timed(vec(0), vec(2), vpairs));
        printf("   A A A  with  A B A            %.4f\n", timed(vec(0), vec(3), vpairs));
        printf("   A B A  with  B A B            %.4f\n", timed(vec(3), vec(4), vpairs));
        printf("   A B A  with  A B A (itself shifted: vec 3 with vec 0 swapped roles) %.4f\n", timed(vec(3), vec(0), vpairs));
        d2 *m0 = nullptr, *m1 = nullptr;
        CK(hipMalloc(&m0, 3 * GiB));
        CK(hipMalloc(&m1, 3 * GiB));
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(vpairs / 64)), dim3(64), 0, 0, m0, vpairs, 1.0);
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(vpairs / 64)), dim3(64), 0, 0, m1, vpairs, 2.0);
        printf("   two plain hipMalloc vectors   %.4f\n", timed(m0, m1, vpairs));
        CK(hipFree(m0));
        CK(hipFree(m1));
        CK(hipMemUnmap(vecs, (size_t)nvec * 3 * GiB));
    } else {
        printf("not enough chunks in two classes (%zu, %zu) to build the vectors\n", by[0].size(), by[1].size());
    }
    CK(hipMemAddressFree(vecs, (size_t)nvec * 3 * GiB));
    CK(hipMemUnmap(flat, (size_t)count * GiB));
    CK(hipMemAddressFree(flat, (size_t)count * GiB));
    for (int k = 0; k < count; ++k) CK(hipMemRelease(h[k]));
    printf("done\n");
    return 0;
}
