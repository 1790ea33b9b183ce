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
int placement_fail_after() {
#ifdef SPMV_AMD_LAB  // test hook of the lab build: the k-th further candidate "does not fit"
    return env_int("SPMV_AMD_PLACEMENT_FAIL_AFTER", 0);
#else
    return 0;
#endif
}

LaunchShape current_launch_shape() {
    // the launch switches (kernels.hpp, Tunables), out-of-range values replaced by the defaults: read when an operator is
    // initialised or a solver slab is created, never per launch
    LaunchShape shape;
    shape.knobs.rowlds_min_grid = env_int("SPMV_AMD_ROWLDS_MIN_GRID", shape.knobs.rowlds_min_grid);
    shape.knobs.rowlds_group = env_int("SPMV_AMD_ROWLDS_GROUP", shape.knobs.rowlds_group);
    if (shape.knobs.rowlds_group < 0 || shape.knobs.rowlds_group > 64) shape.knobs.rowlds_group = 0;
    return shape;
}

int tune_rowlds_xcd_run(const SlabCsr& m, const LaunchShape& shape, const double* x, double* y, double* d_partials, hipStream_t stream,
                        double* record) {
    if (shape.knobs.rowlds_group > 0 || m.n_local < (16 << 20)) return 0;
    LaunchShape trial = shape;
    const Stencil5Plan base = plan_stencil5(m, 0, m.n_local, Stencil5Variant::Auto, trial);
    if (base.variant != Stencil5Variant::RowLds) return 0;
    const int rule = base.xcd_run;
    int cand[6] = {rule, rule - 1, rule + 1, rule - 2, rule + 2, 4};
    EventTimer timer;
    double best_ms = 0.0, rule_ms = 0.0;
    int best = rule;
    for (int k = 0; k < 6; ++k) {
        const int g = cand[k];
        bool seen = g < 1 || g > 64;
        for (int j = 0; j < k; ++j) seen = seen || cand[j] == g;
        if (seen) continue;
        trial.knobs.rowlds_group = g;
        const Stencil5Plan p = plan_stencil5(m, 0, m.n_local, Stencil5Variant::RowLds, trial);
        float ms[3];
        for (int i = 0; i < 4; ++i) {
            timer.begin(stream);
            (void)launch_stencil5_spmv(m, p, x, y, 1.0, d_partials, nullptr, false, stream);
            timer.end(stream);
            const float t = timer.elapsed_ms();
            if (i > 0) ms[i - 1] = t;
        }
        const double med = ms[0] < ms[1] ? (ms[1] < ms[2] ? ms[1] : (ms[0] < ms[2] ? ms[2] : ms[0])) : (ms[0] < ms[2] ? ms[0] : (ms[1] < ms[2] ? ms[2] : ms[1]));
        if (g == rule) rule_ms = med;
        if (best_ms == 0.0 || med < best_ms) best_ms = med, best = g;
    }
    HIP_CHECK(hipGetLastError());
    if (best_ms > 0.995 * rule_ms) best = rule, best_ms = rule_ms;  // within noise: keep the rule
    if (record) record[0] = rule, record[1] = best, record[2] = rule_ms, record[3] = best_ms;
    return best;
}

// values | col_idx | row_ptr in ONE allocation, values first: the CSR kernels read values[e] and col_idx[e] in lock step,
// and lock-step streams in different 32 GiB classes of the address space run ~6 % slower (device_runtime.hpp).
// separate_values (solver slabs whose coefficient stream may be re-placed, cg_slab.hip): `values` is an allocation of its
// own, so that the copy that loses a placement trial can be freed (16 GB at 4e8 rows).
void DeviceCsr::allocate(size_t n_local, size_t local_nnz) {
    constexpr size_t k4KiB = 4096;
    auto up = [](size_t bytes) { return (bytes + k4KiB - 1) / k4KiB * k4KiB; };
    const size_t v_bytes = separate_values ? 0 : up((local_nnz ? local_nnz : 1) * sizeof(double)),
                 c_bytes = up((local_nnz ? local_nnz : 1) * sizeof(int)), r_bytes = up((n_local + 1) * sizeof(int));
    if (separate_values) values_own = device_alloc<double>(local_nnz);
    block = device_alloc<char>(v_bytes + c_bytes + r_bytes);
    values = separate_values ? values_own : reinterpret_cast<double*>(block);
    col_idx = reinterpret_cast<int*>(block + v_bytes);
    row_ptr = reinterpret_cast<int*>(block + v_bytes + c_bytes);
}

// `values` := fresh (same contents, the caller copied them); the previous array is freed if it was an allocation of its own.
void DeviceCsr::replace_values(double* fresh) {
    device_release(values_own);
    values_own = fresh;
    values = fresh;
    view.values = fresh;
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

void DeviceCsr::release() {
    device_release(block);
    device_release(values_own);
    row_ptr = nullptr;
    col_idx = nullptr;
    values = nullptr;
    view = SlabCsr{};
}

}  // namespace spmv_amd
