// tools/arena_probe.hip -- measurement aid, not part of the product. Follow-up of offset_probe: inside ONE allocation the
// rate of the lock-step BLAS1 kernels is a reproducible function of the distance between the vectors (same table on two
// boxes), while separately allocated vectors land in a fast or a slow mode at random (6-7 % apart). If the solver's vectors
// (r, Ap, 16 direction buffers) are carved out of one arena, which PITCH between consecutive vectors puts every pair the
// loop's kernels combine -- (r, Ap), (r, p_k, p_k+1), (p_k, Ap) -- into the fast mode?
//   tools/bin/arena_probe [rows=400000000] [vectors=18] [reps=5]
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

// the 48 B read : 8 B written mix of a STENCIL5 row (csrc/stream_ceiling.hip): five coefficients + x read, y written, per row
__global__ __launch_bounds__(64) void spmv_mix(const double* __restrict__ v, const double* __restrict__ x, double* __restrict__ y, size_t tiles) {
    const size_t tile = blockIdx.x;
    if (tile >= tiles) return;
    const int lane = threadIdx.x;
    const double* src = v + tile * 640 + lane;
    double c[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) c[k] = __builtin_nontemporal_load(src + 64 * k);
    const size_t ra = tile * 128 + lane, rb = ra + 64;
    const double xa = __builtin_nontemporal_load(x + ra), xb = __builtin_nontemporal_load(x + rb);
    double ya = c[0] * xa, yb = c[5] * xb;
#pragma unroll
    for (int k = 1; k < 5; ++k) {
        ya = fma(c[k], xa, ya);
        yb = fma(c[5 + k], xb, yb);
    }
    __builtin_nontemporal_store(ya, y + ra);
    __builtin_nontemporal_store(yb, y + rb);
}

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? (size_t)atoll(argv[1]) : 400000000;
    const int vectors = argc > 2 ? atoi(argv[2]) : 18;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const size_t pairs = rows / 2, vbytes = rows * 8;
    const unsigned grid = (unsigned)((pairs + 63) / 64);
    const size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30, KiB = 1024;
    const size_t natural = (vbytes + 2 * MiB - 1) / (2 * MiB) * (2 * MiB);
    const size_t gib = (vbytes + GiB - 1) / GiB * GiB;
    struct Option {
        const char* name;
        size_t pitch;
    } options[] = {{"vector rounded up to 2 MiB", natural},
                   {"rounded up to 1 GiB", gib},
                   {"1 GiB multiple + 2 MiB", gib + 2 * MiB},
                   {"1 GiB multiple + 14 MiB", gib + 14 * MiB},
                   {"1 GiB multiple + 4 KiB", gib + 4 * KiB},
                   {"1 GiB multiple + 2 MiB + 4 KiB", gib + 2 * MiB + 4 * KiB},
                   {"2 MiB multiple + 4 KiB", natural + 4 * KiB},
                   {"2 MiB multiple + 2 MiB", natural + 2 * MiB},
                   {"2 MiB multiple + 2 MiB + 4 KiB", natural + 2 * MiB + 4 * KiB},
                   {"1 GiB multiple + 256 MiB", gib + 256 * MiB}};
    size_t widest = 0;
    for (const Option& o : options) widest = std::max(widest, o.pitch);
    char* arena = nullptr;
    CK(hipMalloc(&arena, widest * (size_t)vectors + vbytes));
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
        std::sort(ms.begin(), ms.end());
        return ms[ms.size() / 2];
    };
    printf("rows %zu (%.2f GB per vector), %d vectors in one allocation of %.1f GB at %p; median of %d launches\n", rows, vbytes / 1e9, vectors,
           (widest * (size_t)vectors + vbytes) / 1e9, (void*)arena, reps);
    const bool quick = getenv("ARENA_QUICK") != nullptr;
    for (const Option& o : options) {
        if (quick) break;
        auto vec = [&](int i) { return reinterpret_cast<d2*>(arena + (size_t)i * o.pitch); };
        for (int i = 0; i < vectors; ++i) hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(64), 0, 0, vec(i), pairs, 1.0 / (1 + i));
        printf("pitch = %s (%zu bytes)\n   r update, r = vector 0, Ap = vector m, m = 1..%d:", o.name, o.pitch, vectors - 1);
        std::vector<float> all_r, all_p;
        for (int m = 1; m < vectors; ++m) {
            const float t = timed([&] { hipLaunchKernelGGL(upd_r, dim3(grid), dim3(64), 0, 0, vec(m), vec(0), pairs, 1e-9, partials); });
            all_r.push_back(t);
            printf(" %.3f", t);
        }
        printf("\n   p update, r = vector 0, p_in = vector m, p_out = vector m + 1, m = 1..%d:", vectors - 2);
        for (int m = 1; m + 1 < vectors; ++m) {
            const float t = timed([&] { hipLaunchKernelGGL(upd_p, dim3(grid), dim3(64), 0, 0, vec(0), vec(m), vec(m + 1), pairs, 0.999); });
            all_p.push_back(t);
            printf(" %.3f", t);
        }
        std::sort(all_r.begin(), all_r.end());
        std::sort(all_p.begin(), all_p.end());
        printf("\n   r update min %.4f median %.4f max %.4f | p update min %.4f median %.4f max %.4f\n", all_r.front(), all_r[all_r.size() / 2], all_r.back(),
               all_p.front(), all_p[all_p.size() / 2], all_p.back());
        fflush(stdout);
    }
    if (!quick) {
        // every pair (i, j), natural pitch: is "fast" a property of the DISTANCE, or of the two vectors lying in the same large
        // region of the address space?
        const size_t pitch = natural;
        auto vec = [&](int i) { return reinterpret_cast<d2*>(arena + (size_t)i * pitch); };
        printf("r update for every pair, pitch %zu: row = r (vector i), column = Ap (vector j); ms\n      ", pitch);
        for (int j = 0; j < vectors; ++j) printf(" %5d", j);
        printf("\n");
        for (int i = 0; i < vectors; ++i) {
            printf("   %2d ", i);
            for (int j = 0; j < vectors; ++j) {
                if (i == j) {
                    printf("     -");
                    continue;
                }
                const float t = timed([&] { hipLaunchKernelGGL(upd_r, dim3(grid), dim3(64), 0, 0, vec(j), vec(i), pairs, 1e-9, partials); });
                printf(" %.3f", t);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    if (vectors >= 36) {
        // the SpMV's byte mix: the coefficient stream (5 vectors long) at vector slot a, x at slot b, y at slot c. Regions of the
        // all-pairs table above: slots 0-9 | 11-20 | 22-31 | 33-... (10, 21, 32 straddle)
        const size_t pitch = natural;
        auto vecd = [&](int i) { return reinterpret_cast<double*>(arena + (size_t)i * pitch); };
        const size_t tiles = rows / 128;
        const int cfg[][3] = {{0, 5, 6},   {0, 5, 7},   {0, 11, 12}, {0, 5, 12},  {0, 11, 6},  {0, 22, 23}, {0, 33, 34}, {11, 16, 17}, {11, 5, 6},
                              {11, 33, 34}, {22, 27, 28}, {22, 5, 6},  {22, 33, 34}, {33, 38, 39}, {33, 5, 6},  {33, 11, 12}, {5, 0, 1},   {6, 16, 17}};
        printf("SpMV byte mix (40 B coefficients + 8 B x read, 8 B y written per row), %zu rows: coefficients at slot a (5 slots long), x at b, y at c\n", rows);
        for (const auto& c : cfg) {
            if (c[0] + 5 > vectors || c[1] >= vectors || c[2] >= vectors) continue;
            const float t = timed([&] { hipLaunchKernelGGL(spmv_mix, dim3((unsigned)tiles), dim3(64), 0, 0, vecd(c[0]), vecd(c[1]), vecd(c[2]), tiles); });
            printf("   a = %2d  b = %2d  c = %2d   %.4f ms   %.1f GB/s\n", c[0], c[1], c[2], t, 56.0 * rows / (t * 1e-3) / 1e9);
        }
    }
    CK(hipFree(arena));
    CK(hipFree(partials));
    return 0;
}
