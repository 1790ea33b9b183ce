// device_runtime.hpp -- small host-side helpers around the HIP runtime used by the operators
// and solvers. Error convention: HIP_CHECK prints and exits (reference CUDA_CHECK, spmv.h:46-53);
// the product never falls back to a CPU path when the GPU or a kernel is unavailable.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "spmv_amd.h"

namespace spmv_amd {

// Launch geometry for the persistent SpMV kernel on the current device.
LaunchShape current_launch_shape();

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
    int* row_ptr = nullptr;
    int* col_idx = nullptr;
    double* values = nullptr;
    double* planes = nullptr;  // optional plane copy of a verified stencil's coefficients (SlabCsr::planes)
    SlabCsr view;  // non-owning descriptor handed to the kernels

    // Uploads rows [row_offset, row_offset + n_local) of a host CSR, rebasing row_ptr to 0.
    void upload_slab(const CSRMatrix& host, int row_offset, int n_local, int grid_size);
    // Generates the same slab of the synthetic n x n stencil in HBM.
    void generate_stencil5(int n, int row_offset, int n_local, double center, double off,
                           hipStream_t stream);
    // Runs the structure check and records the verdict in view.verified_stencil.
    void verify_stencil(hipStream_t stream);
    // After a successful verify_stencil: builds the five coefficient planes and publishes them in view.planes.
    void build_planes(hipStream_t stream);
    void release();
};

}  // namespace spmv_amd
