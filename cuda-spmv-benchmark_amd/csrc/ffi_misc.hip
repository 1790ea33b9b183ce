// ffi_misc.hip -- the small extern "C" helpers of include/spmv_amd/api.h part 2: integer
// arithmetic that has to match the reference bit for bit, and device plumbing for callers
// that have no HIP binding of their own (the Python tests and bench.py bind only this library).
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

extern "C" const char* spmv_amd_version(void) {
    return "libspmv_amd 0.1 (gfx950, HIP " __DATE__ ")";
}
