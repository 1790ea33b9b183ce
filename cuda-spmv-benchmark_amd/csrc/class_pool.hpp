// class_pool.hpp -- device vectors built from physical memory of a chosen CLASS (round 4).
//
// MI355X's physical address space falls into classes of 32 GiB regions, and what a streaming kernel achieves depends on which
// classes its streams lie in (device_runtime.hpp): the CG loop is fastest -- 103.3 ms against 104.3 with
// all vectors in one allocation and up to 109.4 in an unlucky lottery of separate hipMallocs (profiles/r04_loop_regions.txt) --
// when Ap and r share one class and nothing else the loop streams (the direction buffers, the coefficients) lies in it.
// hipMalloc gives no control over the class; HIP's virtual-memory API does, indirectly: physical memory comes in chunks
// (hipMemCreate) that are mapped into a reserved address range afterwards, so a chunk can be timed against reference chunks first
// (the r-update pair kernel: same class = fast, other class = +6.5 %) and then mapped into a vector that wants its class
// (profiles/r04_vmm_probe.txt: vectors mapped from chunks of one class run in the fast mode, 1.45 ms, mixed ones in between,
// 1.51-1.56 ms; two plain hipMalloc vectors: whatever the lottery gives).
//
// ClassPool creates chunks, sorts them into classes (a chunk joins the first class whose reference chunk it pairs fast with,
// founds a new class if it is clearly slow with all of them, and is set aside if it is neither) and maps vectors from the unused
// chunks of one class. Every failure -- the API missing, memory short -- shows as usable() == false, grow() == false or
// vector() == nullptr: the caller then allocates the plain way. Set-up work, outside every timed region; addresses only.
//
// OPT-IN (SPMV_AMD_CLASS_POOL=1). Measured on the 4e8-row slab (profiles/r04_class_pool_*.txt): in a process whose first large
// allocation is the slab, 103.35-104.03 ms per solve against the arena's 103.87-104.81 (in-loop SpMV 3.56-3.58 ms = 0.78 of
// 8 TB/s, flat from launch to launch, against 3.64-3.67), through bench.py 104.10-104.59 against 104.79-105.62; behind other
// large allocations and frees, which leave the device's free memory in pieces, a quarter of the 1 GiB chunks fit no class
// cleanly and the gain shrinks to 0.2 %; with 256 MiB chunks the SpMV falls to 3.90 ms (vectors made of many small physical
// pieces). Set-up: 2.2 s for the first slab of a process, 17.8 s for a second one -- the reference's benchmark wrapper creates
// thirteen (cg_benchmark_with_stats_mgpu_partitioned): hence not the default. Two lessons kept in the code: an
// address is never mapped twice in a row (a chunk mapped where another had just been unmapped was timed as if it were the
// earlier one: every chunk "joined" the reference's class), and a pair time that implies more than 7.5 TB/s is no yardstick.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

#include <vector>

namespace spmv_amd {

class ClassPool {
   public:
    static constexpr int kMaxClasses = 4;
    ClassPool(size_t vector_bytes, hipStream_t stream);
    ~ClassPool();
    ClassPool(const ClassPool&) = delete;
    ClassPool& operator=(const ClassPool&) = delete;
    bool usable() const { return usable_; }
    size_t chunk_bytes() const { return chunk_bytes_; }
    size_t chunks_for(size_t bytes) const { return chunk_bytes_ ? (bytes + chunk_bytes_ - 1) / chunk_bytes_ : 0; }
    // creates and classifies up to `count` more chunks; returns how many were added
    int grow(int count);
    int classes() const { return (int)refs_.size(); }
    int available(int cls) const;  // unused chunks of that class
    // A zero-filled device vector of `bytes` bytes mapped from unused chunks of class `cls`; nullptr if there are not enough
    // (nothing is consumed then). The pool owns the memory.
    double* vector(size_t bytes, int cls);
    // releases the chunks that ended up in no vector (call when every vector has been made)
    void trim();
    int chunks_created() const { return (int)chunks_.size(); }
    int chunks_in_vectors() const { return in_vectors_; }
    int chunks_set_aside() const { return set_aside_; }
    double fast_ms() const { return fast_ms_; }

   private:
    struct Chunk {
        hipMemGenericAllocationHandle_t handle;
        int cls;  // >= 0 class, -1 set aside (fits no class cleanly), -2 released, -3 created but not yet classified
        bool used;
        bool flat_mapped;
    };
    struct Mapping {
        char* base;
        size_t bytes;
    };
    bool map_flat(size_t index);
    bool calibrate();
    bool classify(size_t index);
    double pair_ms(const char* a, char* b);
    hipStream_t stream_;
    size_t chunk_bytes_ = 0;
    bool usable_ = false;
    hipMemAllocationProp prop_{};
    hipMemAccessDesc access_{};
    std::vector<Chunk> chunks_;
    std::vector<Mapping> mappings_;
    std::vector<size_t> refs_;  // chunk index of every class's reference
    char* flat_ = nullptr;      // address range in which chunk i is mapped at i * chunk_bytes while the pool is sorting
    double* partials_ = nullptr;
    hipEvent_t e0_ = nullptr, e1_ = nullptr;
    double fast_ms_ = 0.0;
    int in_vectors_ = 0, set_aside_ = 0;
    size_t max_chunks_ = 0;
};

}  // namespace spmv_amd
