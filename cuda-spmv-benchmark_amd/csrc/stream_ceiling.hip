// stream_ceiling.hip -- what this GPU sustains for the byte mix of a STENCIL5 row with ideal accesses.
//
// A measurement kernel, not an operator: per row it reads five coefficients and one x value and writes one
// y value (48 B read : 8 B written, the algorithmic 56 B of SURVEY.md section 8d), every access an 8-byte,
// fully coalesced, nontemporal load or store of a one-wave workgroup (the access shape section 3 of DESIGN.md
// found fastest), no neighbour reads, no LDS transpose, no row structure. Its rate on non-zero data is what bench.py
// reports as `roofline.mix_probe_gbs` -- one yardstick among three (with a read-only pass, mix 3 below, and the rate
// the loop's r update reaches in the same run): `roofline.best_stream_gbs_this_run` is their maximum. None of them is
// "the ceiling": in round 4's run the r update streamed 7 % faster than this probe.
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

// Read-only pass over the same arrays (mix 3): 48 B/row read, one 8-byte partial per wave written.
__global__ __launch_bounds__(64) void stream_read_only_kernel(const double* __restrict__ v, const double* __restrict__ x,
                                                              double* __restrict__ y, size_t tiles) {
    const size_t tile = blockIdx.x;
    if (tile >= tiles) return;
    const int lane = threadIdx.x;
    const double* src = v + tile * (5 * kRowsPerWave) + lane;
    double c[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) c[k] = __builtin_nontemporal_load(src + 64 * k);
    const size_t ra = tile * kRowsPerWave + lane;
    double acc = __builtin_nontemporal_load(x + ra) + __builtin_nontemporal_load(x + ra + 64);
#pragma unroll
    for (int k = 0; k < 10; ++k) acc += c[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) y[tile] = acc;
}

// The byte mix of the CSR and ELLPACK formats at five entries per row, with the same ideal accesses: per row five 8-byte
// coefficients, five 4-byte column indices (kIndex), optionally one 4-byte row pointer (kRowPtr), one x and one y.
// CSR: 60 + 4 + 8 read, 8 written = 80 B/row; ELLPACK width 5: 60 + 8 read, 8 written = 76 B/row. The indices
// are read and folded into the result (so the loads cannot be dropped) but nothing is gathered through them: the
// ceiling is what the streams cost, the gather is what the real kernels add.
template <bool kRowPtr>
__global__ __launch_bounds__(64) void stream_ceiling_indexed_kernel(const double* __restrict__ v, const int* __restrict__ idx,
                                                                    const int* __restrict__ row_ptr, const double* __restrict__ x,
                                                                    double* __restrict__ y, size_t tiles) {
    const size_t tile = blockIdx.x;
    if (tile >= tiles) return;
    const int lane = threadIdx.x;
    const double* src = v + tile * (5 * kRowsPerWave) + lane;
    const int* isrc = idx + tile * (5 * kRowsPerWave) + lane;
    double c[10];
    int q[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        c[k] = __builtin_nontemporal_load(src + 64 * k);
        q[k] = __builtin_nontemporal_load(isrc + 64 * k);
    }
    const size_t ra = tile * kRowsPerWave + lane, rb = ra + 64;
    const double xa = __builtin_nontemporal_load(x + ra), xb = __builtin_nontemporal_load(x + rb);
    int extra = 0;
    if (kRowPtr) extra = __builtin_nontemporal_load(row_ptr + ra) + __builtin_nontemporal_load(row_ptr + rb);
    double ya = c[0] * xa, yb = c[5] * xb;
#pragma unroll
    for (int k = 1; k < 5; ++k) {
        ya = fma(c[k], xa, ya);
        yb = fma(c[5 + k], xb, yb);
    }
    int fold = extra;
#pragma unroll
    for (int k = 0; k < 10; ++k) fold ^= q[k];
    if (fold == 0x7fffffff) ya = 0.0;  // never true for the indices below; keeps the index loads alive
    __builtin_nontemporal_store(ya, y + ra);
    __builtin_nontemporal_store(yb, y + rb);
}

__global__ __launch_bounds__(64) void ceiling_fill_index_kernel(int* __restrict__ idx, int* __restrict__ row_ptr, size_t rows) {
    const size_t r = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= rows) return;
    for (int k = 0; k < 5; ++k) idx[5 * r + k] = (int)((r + (size_t)k) & 0x3fffffff);
    row_ptr[r] = (int)((5 * r) & 0x3fffffff);
}

}  // namespace
}  // namespace spmv_amd

// Runs the probe on `rows` rows (rounded down to whole 128-row tiles): `warmup` untimed launches, then `reps`
// launches each timed with HIP events on the launch stream. Returns the bytes one launch moves (56 per row) or 0.
static double stream_probe(size_t rows, int warmup, int reps, float* ms_each, bool read_only) {
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
    auto launch = [&] {
        if (read_only)
            hipLaunchKernelGGL(stream_read_only_kernel, dim3((unsigned)tiles), dim3(64), 0, stream, v, x, y, tiles);
        else
            hipLaunchKernelGGL(stream_ceiling_kernel, dim3((unsigned)tiles), dim3(64), 0, stream, v, x, y, tiles);
    };
    for (int i = 0; i < warmup; ++i) launch();
    EventTimer t;
    for (int i = 0; i < reps; ++i) {
        t.begin(stream);
        launch();
        t.end(stream);
        ms_each[i] = t.elapsed_ms();
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamDestroy(stream));
    device_release(v);
    device_release(x);
    device_release(y);
    return (read_only ? 48.0 : 56.0) * (double)rows;
}

extern "C" double spmv_amd_stream_ceiling(size_t rows, int warmup, int reps, float* ms_each) {
    return stream_probe(rows, warmup, reps, ms_each, false);
}

// The same probe for another format's byte mix: mix 1 = CSR at five entries per row (80 B/row: values, column indices,
// row pointers, x, y), mix 2 = ELLPACK width 5 (76 B/row: values, column indices, x, y), mix 0 = the STENCIL5 mix above,
// mix 3 = the STENCIL5 arrays read only (48 B/row).
// Returns the bytes one launch moves, or 0.
extern "C" double spmv_amd_stream_ceiling_mix(int mix, size_t rows, int warmup, int reps, float* ms_each) {
    using namespace spmv_amd;
    if (mix == 0) return spmv_amd_stream_ceiling(rows, warmup, reps, ms_each);
    if (mix == 3) return stream_probe(rows, warmup, reps, ms_each, true);
    if (mix != 1 && mix != 2) return 0.0;
    const size_t tiles = rows / kRowsPerWave;
    if (tiles == 0 || tiles > 0x7fffffffULL || reps < 1) return 0.0;
    rows = tiles * kRowsPerWave;
    double* v = device_alloc<double>(5 * rows);
    int* idx = device_alloc<int>(5 * rows);
    int* rp = device_alloc<int>(rows);
    double* x = device_alloc<double>(rows);
    double* y = device_alloc<double>(rows);
    hipStream_t stream = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    hipLaunchKernelGGL(ceiling_fill_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, stream, v, x, rows);
    hipLaunchKernelGGL(ceiling_fill_index_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, stream, idx, rp, rows);
    auto launch = [&] {
        if (mix == 1)
            hipLaunchKernelGGL(stream_ceiling_indexed_kernel<true>, dim3((unsigned)tiles), dim3(64), 0, stream, v, idx, rp, x, y, tiles);
        else
            hipLaunchKernelGGL(stream_ceiling_indexed_kernel<false>, dim3((unsigned)tiles), dim3(64), 0, stream, v, idx, rp, x, y, tiles);
    };
    for (int i = 0; i < warmup; ++i) launch();
    EventTimer t;
    for (int i = 0; i < reps; ++i) {
        t.begin(stream);
        launch();
        t.end(stream);
        ms_each[i] = t.elapsed_ms();
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamDestroy(stream));
    device_release(v);
    device_release(idx);
    device_release(rp);
    device_release(x);
    device_release(y);
    return (mix == 1 ? 80.0 : 76.0) * (double)rows;
}
