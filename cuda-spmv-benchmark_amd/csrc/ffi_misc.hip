// ffi_misc.hip -- the small extern "C" helpers of include/spmv_amd/api.h part 2: integer
// arithmetic that has to match the reference bit for bit, and device plumbing for callers
// that have no HIP binding of their own (the Python tests and bench.py bind only this library).
#include <string.h>

#include "device_runtime.hpp"
#include "stencil_geometry.hpp"

using namespace spmv_amd;

extern "C" int spmv_amd_interior_csr_offset(int row, int grid_size) {
    return reference_interior_csr_offset(row, grid_size);
}

extern "C" void spmv_amd_partition_rows(int n, int world, int rank, int* row_offset, int* n_local) {
    int count = n / world;
    const int first = rank * count;
    if (rank == world - 1) count = n - first;
    *row_offset = first;
    *n_local = count;
}

extern "C" int spmv_amd_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

extern "C" int spmv_amd_set_device(int device) {
    HIP_CHECK(hipSetDevice(device));
    return 0;
}

extern "C" int spmv_amd_current_device(char* pci_bus_id, int cap) {
    int dev = -1;
    HIP_CHECK(hipGetDevice(&dev));
    if (pci_bus_id != nullptr && cap > 0) {
        pci_bus_id[0] = '\0';
        (void)hipDeviceGetPCIBusId(pci_bus_id, cap, dev);
    }
    return dev;
}

extern "C" void* spmv_amd_device_alloc(size_t bytes) { return device_alloc<char>(bytes); }

extern "C" void spmv_amd_device_free(void* d_ptr) {
    if (d_ptr) HIP_CHECK(hipFree(d_ptr));
}

extern "C" int spmv_amd_copy_to_device(void* d_dst, const void* h_src, size_t bytes) {
    if (bytes) HIP_CHECK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int spmv_amd_copy_to_host(void* h_dst, const void* d_src, size_t bytes) {
    if (bytes) HIP_CHECK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int spmv_amd_device_fill_f64(double* d_ptr, size_t count, double value) {
    launch_fill(d_ptr, count, value, nullptr);
    HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int spmv_amd_device_synchronize(void) {
    HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

// ---- the CG building blocks one by one, for callers (tests) that check each kernel against the reference's
// element-wise form instead of through a whole solve. Device pointers; each call synchronises. ----

extern "C" int spmv_amd_blas1_axpy(size_t n, double a, const double* d_x, double* d_y) {
    launch_axpy(n, a, d_x, d_y, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

extern "C" int spmv_amd_blas1_axpby(size_t n, double a, const double* d_x, double b, const double* d_y, double* d_z) {
    launch_axpby(n, a, d_x, b, d_y, d_z, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

extern "C" int spmv_amd_blas1_axpy_dev(size_t n, double a, const double* d_x, double* d_y, int subtract) {
    double* d_a = device_alloc<double>(1);
    upload(d_a, &a, 1);
    launch_axpy_dev(n, d_a, d_x, d_y, subtract != 0, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    device_release(d_a);
    return 0;
}

extern "C" int spmv_amd_blas1_update_p_dev(size_t n, const double* d_r, double b, double* d_p) {
    double* d_b = device_alloc<double>(1);
    upload(d_b, &b, 1);
    launch_update_p_dev(n, d_r, d_b, d_p, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    device_release(d_b);
    return 0;
}

extern "C" int spmv_amd_blas1_dot(size_t n, const double* d_x, const double* d_y, double* result) {
    double* scratch = device_alloc<double>(dot_scratch_doubles(n));
    HIP_CHECK(hipMemset(scratch, 0, dot_scratch_doubles(n) * sizeof(double)));
    double* d_out = device_alloc<double>(1);
    launch_dot(n, d_x, d_y, scratch, d_out, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    download(result, d_out, 1);
    device_release(d_out);
    device_release(scratch);
    return 0;
}

// The slab solver's fused steps on caller data. which: 0 = r -= (rr_old/pAp) Ap with r.r (cg_update_r),
// 1 = p_out = r + beta p_in (cg_update_p_ring), 2 = r = b - Ap, p = r with r.r (cg_init_residual; a = b, c = p out).
// scalars = {rr_old, pAp, beta}; *dot_out receives the reduced r.r where the step produces one.
extern "C" int spmv_amd_cg_fused_step(int which, size_t n, const double* scalars, const double* d_a, double* d_b,
                                      double* d_c, int reverse, double* dot_out) {
    CgScalars h;
    memset(&h, 0, sizeof h);
    h.rr_old = scalars[0];
    h.pAp = scalars[1];
    h.beta = scalars[2];
    h.iterations = 1;
    CgScalars* d_s = device_alloc<CgScalars>(1);
    upload(d_s, &h, 1);
    double* partials = device_alloc<double>(dot_scratch_doubles(n));
    HIP_CHECK(hipMemset(partials, 0, dot_scratch_doubles(n) * sizeof(double)));
    double* stage = reduce_scratch_alloc();
    double* d_out = device_alloc<double>(1);
    bool has_dot = false;
    if (which == 0) {  // d_a = Ap, d_b = r (in place)
        launch_cg_update_r(n, d_s, d_a, d_b, partials, nullptr, reverse != 0);
        has_dot = true;
    } else if (which == 1) {  // d_a = r, d_b = p_in, d_c = p_out
        launch_cg_update_p_ring(n, d_s, d_a, d_b, d_c, /*iteration=*/1, nullptr, reverse != 0);
    } else if (which == 2) {  // d_a = b, d_b = Ap ... outputs: d_c = r, and p written to d_c + n
        launch_cg_init_residual(n, d_a, d_b, d_c, d_c + n, partials, nullptr);
        has_dot = true;
    } else {
        return 1;
    }
    if (has_dot) {
        launch_reduce_partials(partials, cg_partial_count(n), d_out, nullptr, nullptr, ReduceScratch{stage});
        HIP_CHECK(hipDeviceSynchronize());
        if (dot_out) download(dot_out, d_out, 1);
    }
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipGetLastError());
    device_release(d_out);
    device_release(stage);
    device_release(partials);
    device_release(d_s);
    return 0;
}

extern "C" const char* spmv_amd_version(void) {
#ifdef SPMV_AMD_LAB
    return "libspmv_amd_lab 0.1 (gfx950, HIP " __DATE__ "; LAB build: test and measurement hooks compiled in)";
#else
    return "libspmv_amd 0.1 (gfx950, HIP " __DATE__ ")";
#endif
}
