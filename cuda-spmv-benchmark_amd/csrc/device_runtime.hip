// device_runtime.hip -- see device_runtime.hpp.
#include "device_runtime.hpp"

#include <stdlib.h>

#include <vector>

#include "stencil_geometry.hpp"

namespace spmv_amd {

namespace {
int env_int(const char* name, int fallback) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : fallback;
}

}  // namespace

int placement_candidates() {
    const int k = env_int("SPMV_AMD_PLACEMENT_CANDIDATES", 3);
    return k < 1 ? 1 : (k > 16 ? 16 : k);
}

namespace {
// The SPMV_AMD_* measurement switches of the launch paths (tools/README.md), out-of-range values replaced by the
// defaults. Called when an operator is initialised or a solver slab is created, never per launch.
Tunables read_tunables() {
    Tunables k;
    k.rowlds_min_grid = env_int("SPMV_AMD_ROWLDS_MIN_GRID", k.rowlds_min_grid);
    k.rowlds_group = env_int("SPMV_AMD_ROWLDS_GROUP", k.rowlds_group);
    if (k.rowlds_group < 0 || k.rowlds_group > 64) k.rowlds_group = 0;
    k.rowlds_we_lds = env_int("SPMV_AMD_ROWLDS_WE_LDS", k.rowlds_we_lds);
    k.rowlds_rows = env_int("SPMV_AMD_ROWLDS_ROWS", k.rowlds_rows);
    if (k.rowlds_rows != 2 && k.rowlds_rows != 4) k.rowlds_rows = 1;
    k.slab_planes = env_int("SPMV_AMD_SLAB_PLANES", k.slab_planes);
    k.direct_rows = env_int("SPMV_AMD_DIRECT_ROWS", k.direct_rows);
    if (k.direct_rows != 2 && k.direct_rows != 4) k.direct_rows = 1;
    k.wavetile_oneshot = env_int("SPMV_AMD_WAVETILE_ONESHOT", k.wavetile_oneshot);
    k.march_blocks_per_cu = env_int("SPMV_AMD_MARCH_BLOCKS_PER_CU", k.march_blocks_per_cu);
    k.march_max_rows = env_int("SPMV_AMD_MARCH_MAX_ROWS", k.march_max_rows);
    k.march_rows_per_task = env_int("SPMV_AMD_ROWS_PER_TASK", 0);
    k.csr_stream_shape = env_int("SPMV_AMD_CSR_STREAM_SHAPE", k.csr_stream_shape);
    k.csr_stream_rows = env_int("SPMV_AMD_CSR_STREAM_ROWS", 0);
    k.ell_shape = env_int("SPMV_AMD_ELL_SHAPE", k.ell_shape);
    k.xcd_group = env_int("SPMV_AMD_XCD_GROUP", k.xcd_group);
    if (k.xcd_group < 0 || k.xcd_group > 512) k.xcd_group = 0;
    return k;
}
}  // namespace

LaunchShape current_launch_shape() {
    static int cached_device = -1;
    static LaunchShape cached;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev != cached_device) {
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        cached.compute_units = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        // 5248 B of LDS per wave x 4 waves per block: seven blocks (28 waves) fit one CU's 160 KiB
        cached.blocks_per_cu = 7;
        cached_device = dev;
    }
    cached.knobs = read_tunables();
    cached.reverse = false;
    return cached;
}

void DeviceCsr::allocate(size_t n_local, size_t local_nnz) {
    const char* v = getenv("SPMV_AMD_CSR_ARENA");
    if (v != nullptr && v[0] == '0') {
        row_ptr = device_alloc<int>(n_local + 1);
        col_idx = device_alloc<int>(local_nnz);
        values = device_alloc<double>(local_nnz);
        return;
    }
    constexpr size_t k4KiB = 4096;
    auto up = [](size_t bytes) { return (bytes + k4KiB - 1) / k4KiB * k4KiB; };
    const size_t v_bytes = up((local_nnz ? local_nnz : 1) * sizeof(double)), c_bytes = up((local_nnz ? local_nnz : 1) * sizeof(int)),
                 r_bytes = up((n_local + 1) * sizeof(int));
    block = device_alloc<char>(v_bytes + c_bytes + r_bytes);
    values = reinterpret_cast<double*>(block);
    col_idx = reinterpret_cast<int*>(static_cast<char*>(block) + v_bytes);
    row_ptr = reinterpret_cast<int*>(static_cast<char*>(block) + v_bytes + c_bytes);
}

void DeviceCsr::replace_values(double* fresh) {
    if (values_moved != nullptr) device_release(values_moved);
    else if (block == nullptr && !values_borrowed) device_release(values);
    values_borrowed = false;
    values_moved = fresh;
    values = fresh;
    view.values = fresh;
}

void DeviceCsr::borrow_values(double* theirs) {
    if (values_moved != nullptr) device_release(values_moved);
    else if (block == nullptr && !values_borrowed) device_release(values);
    values_borrowed = true;
    values = theirs;
    view.values = theirs;
}

void DeviceCsr::upload_slab(const CSRMatrix& host, int row_offset, int n_local, int grid_size) {
    release();
    const long long base = host.row_ptr[row_offset];
    const long long local_nnz = (long long)host.row_ptr[row_offset + n_local] - base;
    allocate((size_t)n_local, (size_t)local_nnz);
    if (base == 0) {
        upload(row_ptr, host.row_ptr, (size_t)n_local + 1);
    } else {
        std::vector<int> rebased((size_t)n_local + 1);
        for (int i = 0; i <= n_local; ++i) rebased[i] = host.row_ptr[row_offset + i] - (int)base;
        upload(row_ptr, rebased.data(), rebased.size());
    }
    upload(col_idx, host.col_indices + base, (size_t)local_nnz);
    upload(values, host.values + base, (size_t)local_nnz);
    view = SlabCsr{};
    view.row_ptr = row_ptr;
    view.col_idx = col_idx;
    view.values = values;
    view.n_local = n_local;
    view.row_offset = row_offset;
    view.nnz_local = local_nnz;
    view.nnz_base = base;
    view.n_global = host.nb_rows;
    view.grid_size = grid_size;
    view.verified_stencil = false;
    for (int i = 0; i < n_local; ++i) {
        const int len = host.row_ptr[row_offset + i + 1] - host.row_ptr[row_offset + i];
        if (len > view.max_row_nnz) view.max_row_nnz = len;
    }
}

void DeviceCsr::generate_stencil5(int n, int row_offset, int n_local, double center, double off,
                                  hipStream_t stream) {
    release();
    const long long base = stencil_row_start_flat(row_offset, n);
    const long long end = stencil_row_start_flat((long long)row_offset + n_local, n);
    const long long local_nnz = end - base;
    allocate((size_t)n_local, (size_t)local_nnz);
    launch_generate_stencil5_csr(n, row_offset, n_local, base, center, off, row_ptr, col_idx, values,
                                 stream);
    view = SlabCsr{};
    view.row_ptr = row_ptr;
    view.col_idx = col_idx;
    view.values = values;
    view.n_local = n_local;
    view.row_offset = row_offset;
    view.nnz_local = local_nnz;
    view.nnz_base = base;
    view.n_global = n * n;
    view.grid_size = n;
    view.verified_stencil = false;
    view.max_row_nnz = n >= 3 ? 5 : (n == 2 ? 3 : 1);
}

void DeviceCsr::verify_stencil(hipStream_t stream) {
    view.verified_stencil = false;
    const int n = view.grid_size;
    if (n < 2 || (long long)n * n != view.n_global) return;
    // the pattern's total must also match, or the analytic offsets would run past the arrays
    if (stencil_row_start_flat((long long)view.row_offset + view.n_local, n) -
            stencil_row_start_flat(view.row_offset, n) != view.nnz_local)
        return;
    int* d_flag = device_alloc<int>(1);
    HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(int), stream));
    launch_verify_stencil5_csr(view, d_flag, stream);
    int h_flag = 1;
    HIP_CHECK(hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    device_release(d_flag);
    view.verified_stencil = (h_flag == 0);
}

void DeviceCsr::build_planes(hipStream_t stream) {
    if (!view.verified_stencil || view.grid_size < 2 || view.n_local <= 0) return;
    device_release(planes);
    planes = device_alloc<double>(5 * (size_t)view.n_local);
    launch_build_stencil5_planes(view, planes, stream);
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipGetLastError());
    view.planes = planes;
}

void DeviceCsr::release() {
    device_release(planes);
    if (block != nullptr) {
        char* b = static_cast<char*>(block);
        device_release(b);
        block = nullptr;
        row_ptr = nullptr;
        col_idx = nullptr;
    } else {
        device_release(row_ptr);
        device_release(col_idx);
        if (values_moved == nullptr && !values_borrowed) device_release(values);
    }
    if (values_moved != nullptr) device_release(values_moved);
    values = nullptr;
    values_borrowed = false;
    view = SlabCsr{};
}

}  // namespace spmv_amd
