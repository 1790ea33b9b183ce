// device_runtime.hpp -- small host-side helpers around the HIP runtime used by the operators
// and solvers. Error convention: HIP_CHECK prints and exits (reference CUDA_CHECK, spmv.h:46-53);
// the product never falls back to a CPU path when the GPU or a kernel is unavailable.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "spmv_amd.h"

namespace spmv_amd {

// The launch switches of the environment (kernels.hpp, Tunables), read now.
LaunchShape current_launch_shape();

// hipMalloc that reports instead of exiting: nullptr when the device cannot provide `count` T (optional buffers only --
// placement candidates, spacers; everything a solve needs goes through device_alloc and fails loudly).
template <class T>
inline T* device_try_alloc(size_t count) {
    void* p = nullptr;
    if (hipMalloc(&p, (count ? count : 1) * sizeof(T)) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return static_cast<T*>(p);
}

template <class T>
inline T* device_alloc(size_t count) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, (count ? count : 1) * sizeof(T)));
    return static_cast<T*>(p);
}

template <class T>
inline void device_release(T*& p) {
    if (p) {
        HIP_CHECK(hipFree((void*)p));
        p = nullptr;
    }
}

template <class T>
inline void upload(T* d_dst, const T* h_src, size_t count) {
    if (count) HIP_CHECK(hipMemcpy(d_dst, h_src, count * sizeof(T), hipMemcpyHostToDevice));
}

template <class T>
inline void download(T* h_dst, const T* d_src, size_t count) {
    if (count) HIP_CHECK(hipMemcpy(h_dst, d_src, count * sizeof(T), hipMemcpyDeviceToHost));
}

// Output placement (round 4). On MI355X the physical address space falls into three classes of 32 GiB regions (repeating every
// 96 GiB; presumably the stack-level ranks of the HBM3E stacks) and the rate of a streaming kernel depends on which classes its
// streams lie in. Measured with this library's own kernels on operands taken from chosen classes (tools/spmv_regions.py,
// profiles/r04_spmv_regions.txt; 4e8 rows):
//   STENCIL5 SpMV (coefficients V and x read, y written)   y in a class of its own 3.54-3.57 ms | y with x 3.68-3.71 | y with V but not x 3.88-3.92
//   r -= a Ap (Ap read, r read and written)                same class 1.43-1.45 ms | different 1.53-1.54
//   p' = r + b p (r, p read, p' written)                   r and p in the same class 1.47-1.50 ms (p' anywhere) | different 1.55-1.56
// hipMalloc decides the class (the same virtual address is fast after one free / malloc cycle and slow after the next), large
// allocations span several regions, and nothing in the API tells: so where ONE buffer decides -- the vector a SpMV writes --
// the owner allocates a few candidates, times the kernel on each and keeps the fastest. Set-up work, outside every timed region
// (the reference times kernels after init, main.cu:136-187); only an address changes, never a result.
// Consecutive hipMallocs are neighbours in physical memory as a rule, so a handful of candidates allocated back to back all
// lie in one region (six of them, 19 GB, never found a better class: profiles/r04_output_placement.txt): the candidates are
// therefore spaced one region apart -- between two of them a spacer allocation fills the rest of 32 GiB and is freed once the
// choice is made -- so that three candidates see all three classes.
// SPMV_AMD_PLACEMENT_CANDIDATES=<k> (default 3; 1 = take the first allocation as it comes).
int placement_candidates();
// Placement trial for ONE buffer of `count` T. `first` is the buffer as it is now (may be null: it is then allocated here,
// fatally on failure, like any buffer a solve cannot do without). Up to placement_candidates() - 1 FURTHER buffers, one
// region apart, are tried with plain hipMalloc: a candidate (or spacer) the device cannot provide ends the trial -- an
// optional copy must never abort a solve that fits without it -- and so does free memory below
// bytes + spacer + 4 GiB. cost_ms(candidate) is evaluated on each; the cheapest is returned, the others are freed -- `first`
// too if it lost and release_first is set. *tried / *gain (optional): candidates timed, cost of the first over the one kept.
// SPMV_AMD_PLACEMENT_FAIL_AFTER=<k> (test hook, LAB build only): the k-th further candidate "does not fit".
int placement_fail_after();
template <class T, class Cost>
inline T* device_alloc_best_of(size_t count, size_t min_count, Cost&& cost_ms, int* tried = nullptr, double* gain = nullptr,
                               T* first = nullptr, bool release_first = true) {
    const int want = count >= min_count ? placement_candidates() : 1;
    if (tried) *tried = 1;
    if (gain) *gain = 1.0;
    if (first == nullptr) first = device_alloc<T>(count);
    if (want <= 1) return first;
    constexpr size_t kRegion = (size_t)32 << 30;
    const size_t bytes = count * sizeof(T);
    const size_t spacer_bytes = bytes < kRegion ? kRegion - bytes : 0;
    const int fail_after = placement_fail_after();
    T* cand[16];
    void* spacer[16];
    double cost[16];
    int n = 1, spacers = 0;
    cand[0] = first;
    cost[0] = cost_ms(first);
    for (; n < want && n < 16; ++n) {
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        if (free_b < spacer_bytes + bytes + ((size_t)4 << 30)) break;
        if (spacer_bytes > 0) {
            void* sp = device_try_alloc<char>(spacer_bytes);
            if (sp == nullptr) break;
            spacer[spacers++] = sp;
        }
        T* c = (fail_after > 0 && n >= fail_after) ? nullptr : device_try_alloc<T>(count);
        if (c == nullptr) break;  // another rank on this device took the memory in between, or the test hook says so
        cand[n] = c;
        cost[n] = cost_ms(c);
    }
    int best = 0;
    for (int k = 1; k < n; ++k)
        if (cost[k] < cost[best]) best = k;
    for (int k = 0; k < spacers; ++k) HIP_CHECK(hipFree(spacer[k]));
    for (int k = 0; k < n; ++k)
        if (k != best && (k > 0 || release_first)) device_release(cand[k]);
    if (tried) *tried = n;
    if (gain) *gain = cost[best] > 0.0 ? cost[0] / cost[best] : 1.0;
    return cand[best];
}

// Row-lds tile -> XCD run length by measurement (round 5; spmv_kernels.hip, rowlds_xcd_run_rule has the why): times the whole
// slab's SpMV on (x, y) for the rule's neighbours and for 4, and returns the fastest run length -- 0 where tuning does not
// apply (not a row-lds slab, fewer than 16 Mi rows, SPMV_AMD_ROWLDS_GROUP forces a value). Set-up work: the mapping of
// workgroups to tiles is a performance choice only, results and partial slots do not depend on it.
// record (optional, 4 doubles): {rule, kept, ms with the rule, ms kept}.
int tune_rowlds_xcd_run(const SlabCsr& m, const LaunchShape& shape, const double* x, double* y, double* d_partials, hipStream_t stream,
                        double* record = nullptr);

// Pair of events for on-stream timing of one region.
struct EventTimer {
    hipEvent_t start = nullptr, stop = nullptr;
    EventTimer() {
        HIP_CHECK(hipEventCreate(&start));
        HIP_CHECK(hipEventCreate(&stop));
    }
    ~EventTimer() {
        (void)hipEventDestroy(start);
        (void)hipEventDestroy(stop);
    }
    void begin(hipStream_t s) { HIP_CHECK(hipEventRecord(start, s)); }
    void end(hipStream_t s) { HIP_CHECK(hipEventRecord(stop, s)); }
    float elapsed_ms() {
        HIP_CHECK(hipEventSynchronize(stop));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, start, stop));
        return ms;
    }
};

// What cg_solve_device may ask of an operator BEYOND the vtable (reference include/spmv.h:125-134 has run_device(d_x, d_y)
// only). This library's own stencil5-csr operator can write the p.Ap partial sums -- and, for the first SpMV of a solve,
// r0 = b - A x0, p0 and the r0.r0 partials -- from inside its SpMV launch, which saves a 16 B/row dot-product pass per
// iteration; any other operator (the CSR / ELLPACK ones, a table supplied by the caller) reports partials == 0 and is
// driven through run_device followed by a dot kernel. Defined in operators.hip.
struct FusedSpmv {
    int partials = 0;       // dot-partial slots one launch writes; 0 = the operator has no fused form
    bool can_init = false;  // the launch can produce the initial residual (ResidualOut)
    int (*launch)(const double* d_x, double* d_y, double* d_partials, const int* d_skip, bool reverse, const ResidualOut* init,
                  hipStream_t stream) = nullptr;
};
FusedSpmv fused_spmv_of(const SpmvOperator* op);
// Frees the vectors cg_solve_device keeps between calls (cg_slab.hip); called by every operator's free().
void release_cg_workspace();

// Device-resident CSR of one operator or one slab (owning).
struct DeviceCsr {
    char* block = nullptr;         // ONE allocation: [values |] col_idx | row_ptr (allocate())
    bool separate_values = false;  // set before the arrays are made: `values` outside the block (replace_values can then free the loser)
    double* values_own = nullptr;  // `values` when it is an allocation of its own (separate_values, or after replace_values)
    int* row_ptr = nullptr;
    int* col_idx = nullptr;
    double* values = nullptr;
    void allocate(size_t n_local, size_t local_nnz);
    void replace_values(double* fresh);
    SlabCsr view;  // non-owning descriptor handed to the kernels

    // Uploads rows [row_offset, row_offset + n_local) of a host CSR, rebasing row_ptr to 0.
    void upload_slab(const CSRMatrix& host, int row_offset, int n_local, int grid_size);
    // Generates the same slab of the synthetic n x n stencil in HBM.
    void generate_stencil5(int n, int row_offset, int n_local, double center, double off,
                           hipStream_t stream);
    // Runs the structure check and records the verdict in view.verified_stencil.
    void verify_stencil(hipStream_t stream);
    void release();
};

}  // namespace spmv_amd
