// tools/offset_probe.hip -- measurement aid, not part of the product. The CG loop's BLAS1 passes read / write several vectors in
// LOCK STEP (same index, same stride): r -= a Ap touches r[i] and Ap[i] together, p' = r + b p touches r[i], p[i], p'[i].
// profiles/r04_placement_culprit.txt: moving ONE of those vectors into a fresh allocation changes such a kernel by up to 6 %
// (r update 1.454 -> 1.546 ms at 4e8 rows), while a single stream does not care where it lies (r04_placement_probe.txt).
// Hypothesis: what matters is the distance between the streams modulo the memory system's interleave. This probe puts the
// vectors into ONE allocation and sweeps the distance between them.
//   tools/bin/offset_probe [rows=400000000] [reps=7]
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
// r -= a * Ap + wave partial of r.r (the shape of cg_update_r_kernel)
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
// p_out = r + b * p_in (the shape of cg_update_p_ring_kernel)
__global__ __launch_bounds__(64) void upd_p(const d2* __restrict__ r, const d2* __restrict__ p_in, d2* __restrict__ p_out, size_t pairs,
                                            double b) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i < pairs) {
        const d2 rv = __builtin_nontemporal_load(r + i);
        d2 pv = __builtin_nontemporal_load(p_in + i);
        pv.x = fma(b, pv.x, rv.x);
        pv.y = fma(b, pv.y, rv.y);
        p_out[i] = pv;
    }
}

static float median_ms(std::vector<float> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? (size_t)atoll(argv[1]) : 400000000;
    const int reps = argc > 2 ? atoi(argv[2]) : 7;
    const size_t pairs = rows / 2, vbytes = rows * 8;
    const unsigned grid = (unsigned)((pairs + 63) / 64);
    const size_t slack = (size_t)512 << 20;  // room for the largest distance swept
    char* arena = nullptr;
    CK(hipMalloc(&arena, 3 * (vbytes + slack) + slack));
    double* partials = nullptr;
    CK(hipMalloc(&partials, (size_t)grid * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](auto&& launch) {
        std::vector<float> ms;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t = 0.f;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (r > 0) ms.push_back(t);
        }
        return median_ms(ms);
    };
    // vector k starts at arena + k * (vbytes rounded up to 1 GiB) + delta_k: distance between vector 0 and vector 1 = a whole
    // number of GiB + delta
    const size_t gib = (size_t)1 << 30;
    const size_t pitch = (vbytes + gib - 1) / gib * gib;
    const size_t deltas[] = {0,       64,      128,     256,      512,      1024,     2048,      4096,      8192,      16384,    32768,
                             65536,   131072,  262144,  524288,   1048576,  2097152,  4194304,   8388608,   16777216,  33554432, 67108864,
                             134217728, 268435456, 4096 + 256, 65536 + 4096, 1048576 + 65536, 3 * 4096, 5 * 65536, 3 * 1048576, 7 * 2097152, 100663296};
    printf("rows %zu (%.2f GB per vector), one allocation; vector k at k * %zu GiB + delta; median of %d launches\n", rows, vbytes / 1e9,
           pitch / gib, reps);
    printf("%12s  %12s %9s   %14s %9s\n", "delta bytes", "r update ms", "GB/s", "p update ms", "GB/s");
    for (size_t delta : deltas) {
        d2* r = reinterpret_cast<d2*>(arena);
        d2* ap = reinterpret_cast<d2*>(arena + pitch + delta);
        d2* p2 = reinterpret_cast<d2*>(arena + 2 * pitch + 2 * delta);
        hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(64), 0, 0, r, pairs, 1.0);
        hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(64), 0, 0, ap, pairs, 0.5);
        hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(64), 0, 0, p2, pairs, 0.25);
        const float t_r = timed([&] { hipLaunchKernelGGL(upd_r, dim3(grid), dim3(64), 0, 0, ap, r, pairs, 1e-9, partials); });
        // in place in p2's role: read r, read ap (as p_in), write p2
        const float t_p = timed([&] { hipLaunchKernelGGL(upd_p, dim3(grid), dim3(64), 0, 0, r, ap, p2, pairs, 0.999); });
        printf("%12zu  %12.4f %9.1f   %14.4f %9.1f\n", delta, t_r, 3.0 * vbytes / (t_r * 1e-3) / 1e9, t_p, 3.0 * vbytes / (t_p * 1e-3) / 1e9);
        fflush(stdout);
    }
    // Separately allocated vectors, as the solver has them (every hipMalloc of this size returns a 2 MiB-aligned address, so
    // the distance between any two vectors is a multiple of 2 MiB): eight sets, each measured with the vectors at the start
    // of their allocations and with small PHASE offsets added (vector k shifted by phase[k] bytes inside its allocation).
    printf("separate hipMalloc per vector (+64 KiB slack each), eight sets; phases in bytes for (r, Ap, p'):\n");
    const size_t phases[][3] = {{0, 0, 0}, {0, 4096, 8192}, {0, 8192, 4096}, {0, 1024, 2048}, {0, 2048, 6144}, {0, 8192, 8192}, {0, 0, 8192}, {0, 4096, 12288}};
    const int n_phase = (int)(sizeof(phases) / sizeof(phases[0]));
    std::vector<std::vector<float>> tr((size_t)n_phase), tp((size_t)n_phase);
    for (int s = 0; s < 8; ++s) {
        char *r0 = nullptr, *ap0 = nullptr, *p0 = nullptr;
        CK(hipMalloc(&r0, vbytes + 65536));
        CK(hipMalloc(&ap0, vbytes + 65536));
        CK(hipMalloc(&p0, vbytes + 65536));
        printf("   set %d  r %p ap %p p %p :", s, (void*)r0, (void*)ap0, (void*)p0);
        for (int f = 0; f < n_phase; ++f) {
            d2* r = reinterpret_cast<d2*>(r0 + phases[f][0]);
            d2* ap = reinterpret_cast<d2*>(ap0 + phases[f][1]);
            d2* p2 = reinterpret_cast<d2*>(p0 + phases[f][2]);
            hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(64), 0, 0, r, pairs, 1.0);
            hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(64), 0, 0, ap, pairs, 0.5);
            hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(64), 0, 0, p2, pairs, 0.25);
            const float t_r = timed([&] { hipLaunchKernelGGL(upd_r, dim3(grid), dim3(64), 0, 0, ap, r, pairs, 1e-9, partials); });
            const float t_p = timed([&] { hipLaunchKernelGGL(upd_p, dim3(grid), dim3(64), 0, 0, r, ap, p2, pairs, 0.999); });
            tr[f].push_back(t_r), tp[f].push_back(t_p);
            printf("  %.3f/%.3f", t_r, t_p);
        }
        printf("\n");
        fflush(stdout);
        // sets 0-3 are freed at once, sets 4-7 keep one vector alive so that the next set lands elsewhere
        CK(hipFree(r0));
        CK(hipFree(ap0));
        if (s < 4) CK(hipFree(p0));
    }
    printf("per phase over the eight sets (r update | p update: min median max ms):\n");
    for (int f = 0; f < n_phase; ++f) {
        std::sort(tr[f].begin(), tr[f].end());
        std::sort(tp[f].begin(), tp[f].end());
        printf("   phases (%5zu, %5zu, %5zu)   r update %.4f %.4f %.4f   p update %.4f %.4f %.4f\n", phases[f][0], phases[f][1], phases[f][2], tr[f].front(),
               tr[f][4], tr[f].back(), tp[f].front(), tp[f][4], tp[f].back());
    }
    CK(hipFree(arena));
    CK(hipFree(partials));
    return 0;
}
