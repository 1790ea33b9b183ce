// stream_ceiling.hip -- what this GPU sustains for the byte mix of a STENCIL5 row with ideal accesses.
//
// A measurement kernel, not an operator: per row it reads five coefficients and one x value and writes one
// y value (48 B read : 8 B written, the algorithmic 56 B of SURVEY.md section 8d), every access an 8-byte,
// fully coalesced, nontemporal load or store of a one-wave workgroup (the access shape section 3 of DESIGN.md
// found fastest), no neighbour reads, no LDS transpose, no row structure. Its rate on non-zero data is the
// ceiling bench.py reports next to the spec-sheet peak as `roofline.ceiling_measured`: the fraction of THAT is
// what the SpMV kernel's own organisation costs; the rest is between the part and its data sheet.
#include <stdio.h>

#include <vector>

#include "device_runtime.hpp"

namespace spmv_amd {
namespace {

constexpr int kRowsPerWave = 128;

__global__ __launch_bounds__(64) void ceiling_fill_kernel(double* __restrict__ v, double* __restrict__ x, size_t rows) {
    const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= rows) return;
    // the benchmark's own data: rows [N W C E S] = [-1 -1 5 -1 -1], x = 1
    double* q = v + 5 * r;
    q[0] = -1.0;
    q[1] = -1.0;
    q[2] = 5.0;
    q[3] = -1.0;
    q[4] = -1.0;
    x[r] = 1.0;
}

__global__ __launch_bounds__(64) void stream_ceiling_kernel(const double* __restrict__ v, const double* __restrict__ x,
                                                            double* __restrict__ y, size_t tiles) {
    const size_t tile = blockIdx.x;
    if (tile >= tiles) return;
    const int lane = threadIdx.x;
    const double* src = v + tile * (5 * kRowsPerWave) + lane;
    double c[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) c[k] = __builtin_nontemporal_load(src + 64 * k);
    const size_t ra = tile * kRowsPerWave + lane, rb = ra + 64;
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

}  // namespace
}  // namespace spmv_amd

// Runs the probe on `rows` rows (rounded down to whole 128-row tiles): `warmup` untimed launches, then `reps`
// launches each timed with HIP events on the launch stream. Returns the bytes one launch moves (56 per row) or 0.
extern "C" double spmv_amd_stream_ceiling(size_t rows, int warmup, int reps, float* ms_each) {
    using namespace spmv_amd;
    const size_t tiles = rows / kRowsPerWave;
    if (tiles == 0 || tiles > 0x7fffffffULL || reps < 1) return 0.0;
    rows = tiles * kRowsPerWave;
    double* v = device_alloc<double>(5 * rows);
    double* x = device_alloc<double>(rows);
    double* y = device_alloc<double>(rows);
    hipStream_t stream = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    hipLaunchKernelGGL(ceiling_fill_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, stream, v, x, rows);
    for (int i = 0; i < warmup; ++i)
        hipLaunchKernelGGL(stream_ceiling_kernel, dim3((unsigned)tiles), dim3(64), 0, stream, v, x, y, tiles);
    EventTimer t;
    for (int i = 0; i < reps; ++i) {
        t.begin(stream);
        hipLaunchKernelGGL(stream_ceiling_kernel, dim3((unsigned)tiles), dim3(64), 0, stream, v, x, y, tiles);
        t.end(stream);
        ms_each[i] = t.elapsed_ms();
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamDestroy(stream));
    device_release(v);
    device_release(x);
    device_release(y);
    return 56.0 * (double)rows;
}
