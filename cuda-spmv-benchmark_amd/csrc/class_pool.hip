// class_pool.hip -- see class_pool.hpp.
#include "class_pool.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

namespace spmv_amd {
namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

// The shape of cg_update_r_kernel with alpha = 0: a is read, b is read and written back unchanged; one-wave workgroups, 16 bytes
// per lane, nontemporal. Two streams in lock step: the kernel whose rate tells whether a and b lie in the same class.
__global__ __launch_bounds__(64) void class_probe_kernel(const d2* __restrict__ a, d2* __restrict__ b, size_t pairs, double* __restrict__ partials) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    double acc = 0.0;
    if (i < pairs) {
        const d2 av = __builtin_nontemporal_load(a + i);
        d2 bv = __builtin_nontemporal_load(b + i);
        bv.x = fma(-0.0, av.x, bv.x);
        bv.y = fma(-0.0, av.y, bv.y);
        __builtin_nontemporal_store(bv, b + i);
        acc = bv.x * bv.x + bv.y * bv.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
__global__ __launch_bounds__(64) void class_fill_kernel(d2* p, size_t pairs, double v) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i < pairs) p[i] = d2{v, -v};
}

bool ok(hipError_t e) {
    if (e == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

}  // namespace

ClassPool::ClassPool(size_t vector_bytes, hipStream_t stream) : stream_(stream) {
    // opt-in: measured at 0.2-0.8 % per solve, depending on how fragmented the device's free memory is when the slab is created
    // (class_pool.hpp); the default layout is the vector arena
    const char* enabled = getenv("SPMV_AMD_CLASS_POOL");
    if (enabled == nullptr || enabled[0] != '1') return;
    int dev = 0;
    if (!ok(hipGetDevice(&dev))) return;
    prop_.type = hipMemAllocationTypePinned;
    prop_.location.type = hipMemLocationTypeDevice;
    prop_.location.id = dev;
    access_.location = prop_.location;
    access_.flags = hipMemAccessFlagsProtReadWrite;
    size_t gran = 0;
    if (!ok(hipMemGetAllocationGranularity(&gran, &prop_, hipMemAllocationGranularityRecommended)) || gran == 0) return;
    // a vector occupies whole chunks: 1 GiB chunks for vectors of 2 GiB and more, 256 MiB below; a chunk must be long enough for
    // the pair kernel's two modes (6.5 % apart) to be told apart by a median of five launches
    chunk_bytes_ = vector_bytes >= ((size_t)2 << 30) ? (size_t)1 << 30 : (size_t)256 << 20;
    if (const char* v = getenv("SPMV_AMD_CLASS_CHUNK_MIB"))
        if (atoi(v) >= 64) chunk_bytes_ = (size_t)atoi(v) << 20;
    if (chunk_bytes_ % gran != 0) return;
    size_t free_b = 0, total_b = 0;
    if (!ok(hipMemGetInfo(&free_b, &total_b))) return;
    max_chunks_ = free_b > ((size_t)24 << 30) ? (free_b - ((size_t)24 << 30)) / chunk_bytes_ : 0;
    if (max_chunks_ < 8) return;
    // every chunk gets an address of its own for as long as it is being judged (chunk i at flat_ + i * chunk): an address is
    // never mapped twice in a row, so no translation of an earlier chunk can be looked up for a later one
    if (!ok(hipMemAddressReserve((void**)&flat_, max_chunks_ * chunk_bytes_, chunk_bytes_, nullptr, 0))) {
        flat_ = nullptr;
        return;
    }
    if (!ok(hipMalloc((void**)&partials_, (chunk_bytes_ / 16 / 64 + 1) * sizeof(double))) || !ok(hipEventCreate(&e0_)) || !ok(hipEventCreate(&e1_))) return;
    usable_ = true;
}

ClassPool::~ClassPool() {
    (void)hipDeviceSynchronize();
    for (const Mapping& m : mappings_) {
        (void)hipMemUnmap(m.base, m.bytes);
        (void)hipMemAddressFree(m.base, m.bytes);
    }
    if (flat_) {
        for (size_t i = 0; i < chunks_.size(); ++i)
            if (chunks_[i].flat_mapped) (void)hipMemUnmap(flat_ + i * chunk_bytes_, chunk_bytes_);
        (void)hipMemAddressFree(flat_, max_chunks_ * chunk_bytes_);
    }
    for (const Chunk& c : chunks_)
        if (c.cls != -2) (void)hipMemRelease(c.handle);
    if (partials_) (void)hipFree(partials_);
    if (e0_) (void)hipEventDestroy(e0_);
    if (e1_) (void)hipEventDestroy(e1_);
}

double ClassPool::pair_ms(const char* a, char* b) {
    const size_t pairs = chunk_bytes_ / 16;
    const unsigned grid = (unsigned)((pairs + 63) / 64);
    float ms[5];
    for (int i = 0; i < 6; ++i) {
        (void)hipEventRecord(e0_, stream_);
        hipLaunchKernelGGL(class_probe_kernel, dim3(grid), dim3(64), 0, stream_, reinterpret_cast<const d2*>(a), reinterpret_cast<d2*>(b), pairs, partials_);
        (void)hipEventRecord(e1_, stream_);
        (void)hipEventSynchronize(e1_);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0_, e1_);
        if (i > 0) ms[i - 1] = t;
    }
    std::sort(ms, ms + 5);
    return (double)ms[2];
}

bool ClassPool::map_flat(size_t index) {
    Chunk& c = chunks_[index];
    if (c.flat_mapped) return true;
    char* at = flat_ + index * chunk_bytes_;
    if (!ok(hipMemMap(at, chunk_bytes_, 0, c.handle, 0)) || !ok(hipMemSetAccess(at, chunk_bytes_, &access_, 1))) return false;
    c.flat_mapped = true;
    const size_t pairs = chunk_bytes_ / 16;
    hipLaunchKernelGGL(class_fill_kernel, dim3((unsigned)((pairs + 63) / 64)), dim3(64), 0, stream_, reinterpret_cast<d2*>(at), pairs, 1.0 + (double)(index % 7));
    return true;
}

// What the fast mode costs on this part: every pair of the first five chunks is timed; with at most four classes two of the
// five share one, so the fastest of the ten pairings is the fast mode.
bool ClassPool::calibrate() {
    const size_t n = chunks_.size() < 5 ? chunks_.size() : 5;
    for (size_t k = 0; k < n; ++k)
        if (!map_flat(k)) return false;
    // a time that would mean more than 7.5 TB/s (the part's data sheet says 8) is not a mode of the memory system but a
    // fluke: it must not become the yardstick
    const double floor_ms = 3.0 * (double)chunk_bytes_ / 7.5e12 * 1e3;
    double best = 0.0;
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j) {
            const double t = pair_ms(flat_ + i * chunk_bytes_, flat_ + j * chunk_bytes_);
            if (t >= floor_ms && (best == 0.0 || t < best)) best = t;
        }
    fast_ms_ = best;
    return best > 0.0;
}

// The chunk is timed against every class's reference in turn: within 2 % of the fast mode = same class; more than 4.5 % above it
// with every reference = a new class (the chunk becomes its reference and is never handed out); anything else is set aside
// (a chunk that lies across a region boundary, or a noisy measurement) and used for nothing.
bool ClassPool::classify(size_t index) {
    Chunk& c = chunks_[index];
    if (!map_flat(index)) return false;
    char* at = flat_ + index * chunk_bytes_;
    const double floor_ms = 3.0 * (double)chunk_bytes_ / 7.5e12 * 1e3;
    int joined = -1;
    bool clearly_slow_with_all = true;
    for (size_t k = 0; k < refs_.size() && joined < 0; ++k) {
        const double t = pair_ms(at, flat_ + refs_[k] * chunk_bytes_);
        if (t < fast_ms_ && t >= floor_ms) fast_ms_ = t;
        if (t < 1.02 * fast_ms_) joined = (int)k;
        if (t < 1.045 * fast_ms_) clearly_slow_with_all = false;
    }
    if (joined >= 0) {
        c.cls = joined;
    } else if (clearly_slow_with_all && (int)refs_.size() < kMaxClasses) {
        c.cls = (int)refs_.size();
        c.used = true;
        refs_.push_back(index);
    } else {
        c.cls = -1;
        ++set_aside_;
    }
    return true;
}

int ClassPool::grow(int count) {
    if (!usable_) return 0;
    const size_t before = chunks_.size();
    for (int k = 0; k < count && chunks_.size() < max_chunks_; ++k) {
        Chunk c{};
        if (!ok(hipMemCreate(&c.handle, chunk_bytes_, &prop_, 0))) break;
        c.cls = -3;  // created, not yet classified
        c.used = false;
        c.flat_mapped = false;
        chunks_.push_back(c);
    }
    if (fast_ms_ == 0.0 && (chunks_.size() < 5 || !calibrate())) {
        usable_ = false;
        return 0;
    }
    for (size_t i = 0; i < chunks_.size(); ++i)
        if (chunks_[i].cls == -3 && !classify(i)) {
            usable_ = false;
            return 0;
        }
    (void)hipStreamSynchronize(stream_);
    return (int)(chunks_.size() - before);
}

int ClassPool::available(int cls) const {
    int n = 0;
    for (const Chunk& c : chunks_)
        if (!c.used && c.cls == cls) ++n;
    return n;
}

// Every chunk leaves the flat view; those that are in no vector are released.
void ClassPool::trim() {
    (void)hipStreamSynchronize(stream_);
    for (size_t i = 0; i < chunks_.size(); ++i) {
        Chunk& c = chunks_[i];
        if (c.flat_mapped) {
            (void)hipMemUnmap(flat_ + i * chunk_bytes_, chunk_bytes_);
            c.flat_mapped = false;
        }
        const bool reference = std::find(refs_.begin(), refs_.end(), i) != refs_.end();
        if ((!c.used || reference) && c.cls != -2) {
            (void)hipMemRelease(c.handle);
            c.cls = -2;
            c.used = true;
        }
    }
    refs_.clear();  // no more vectors after a trim
}

double* ClassPool::vector(size_t bytes, int cls) {
    if (!usable_ || bytes == 0 || cls < 0 || cls >= classes()) return nullptr;
    const size_t n = chunks_for(bytes);
    if ((size_t)available(cls) < n) return nullptr;
    std::vector<size_t> picked;
    for (size_t i = 0; i < chunks_.size() && picked.size() < n; ++i)
        if (!chunks_[i].used && chunks_[i].cls == cls) picked.push_back(i);
    char* base = nullptr;
    if (!ok(hipMemAddressReserve((void**)&base, n * chunk_bytes_, (size_t)2 << 20, nullptr, 0))) return nullptr;
    for (size_t k = 0; k < n; ++k) {
        if (!ok(hipMemMap(base + k * chunk_bytes_, chunk_bytes_, 0, chunks_[picked[k]].handle, 0))) {
            if (k > 0) (void)hipMemUnmap(base, k * chunk_bytes_);
            (void)hipMemAddressFree(base, n * chunk_bytes_);
            return nullptr;
        }
    }
    if (!ok(hipMemSetAccess(base, n * chunk_bytes_, &access_, 1)) || !ok(hipMemsetAsync(base, 0, n * chunk_bytes_, stream_))) {
        (void)hipMemUnmap(base, n * chunk_bytes_);
        (void)hipMemAddressFree(base, n * chunk_bytes_);
        return nullptr;
    }
    for (size_t i : picked) chunks_[i].used = true;
    mappings_.push_back({base, n * chunk_bytes_});
    in_vectors_ += (int)n;
    return reinterpret_cast<double*>(base);
}

}  // namespace spmv_amd
