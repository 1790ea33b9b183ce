// tools/rocsparse_compare.hip -- comparator slot (the role AmgX / cuSPARSE play for the
// reference): the vendor library's CSR SpMV on the same synthetic 5-point matrix, same timing rule.
// Not part of the product and not linked into libspmv_amd.so.
//   make -C tools            (-> tools/bin/rocsparse_compare)
//   tools/bin/rocsparse_compare 10000                  timing of every rocsparse_spmv CSR algorithm, x = 1
//   tools/bin/rocsparse_compare 2048 x.bin y.bin       x read from x.bin (rows doubles), y of the default algorithm
//                                                      written to y.bin: what tests/test_comparators_gpu.py compares
//                                                      with this repository's operators
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define RS(x) do { rocsparse_status s = (x); if (s != rocsparse_status_success) { printf("rocsparse error %d line %d\n", (int)s, __LINE__); exit(1);} } while (0)

__global__ void gen(int n, int* rp, int* ci, double* va) {
    long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long rows = (long long)n * n;
    if (g > rows) return;
    auto start = [n](long long row) -> long long {
        if (row >= (long long)n * n) return 5LL * n * n - 4LL * n;
        int i = (int)(row / n), j = (int)(row % n);
        long long base = i == 0 ? 0 : (4LL * n - 2) + (long long)(i - 1) * (5LL * n - 2);
        int vert = (i > 0) + (i < n - 1);
        return base + (j > 0 ? (2 + vert) + (long long)(j - 1) * (3 + vert) : 0);
    };
    long long k = start(g);
    rp[g] = (int)k;
    if (g == rows) return;
    int i = (int)(g / n), j = (int)(g % n);
    if (i > 0) ci[k] = (int)(g - n), va[k] = -1.0, ++k;
    if (j > 0) ci[k] = (int)(g - 1), va[k] = -1.0, ++k;
    ci[k] = (int)g, va[k] = 5.0, ++k;
    if (j < n - 1) ci[k] = (int)(g + 1), va[k] = -1.0, ++k;
    if (i < n - 1) ci[k] = (int)(g + n), va[k] = -1.0, ++k;
}
__global__ void fill(double* p, size_t n, double v) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 10000;
    const char* x_file = argc > 3 ? argv[2] : nullptr;
    const char* y_file = argc > 3 ? argv[3] : nullptr;
    const long long rows = (long long)n * n, nnz = 5LL * n * n - 4LL * n;
    int *rp, *ci; double *va, *x, *y;
    CK(hipMalloc(&rp, (rows + 1) * 4)); CK(hipMalloc(&ci, nnz * 4)); CK(hipMalloc(&va, nnz * 8));
    CK(hipMalloc(&x, rows * 8)); CK(hipMalloc(&y, rows * 8));
    hipLaunchKernelGGL(gen, dim3((unsigned)((rows + 256) / 256)), dim3(256), 0, 0, n, rp, ci, va);
    hipLaunchKernelGGL(fill, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, x, (size_t)rows, 1.0);
    CK(hipDeviceSynchronize());
    if (x_file) {
        std::vector<double> hx((size_t)rows);
        FILE* f = fopen(x_file, "rb");
        if (!f || fread(hx.data(), sizeof(double), (size_t)rows, f) != (size_t)rows) { printf("cannot read %s\n", x_file); return 1; }
        fclose(f);
        CK(hipMemcpy(x, hx.data(), (size_t)rows * 8, hipMemcpyHostToDevice));
    }
    rocsparse_handle h; RS(rocsparse_create_handle(&h));
    rocsparse_spmat_descr A; rocsparse_dnvec_descr vx, vy;
    RS(rocsparse_create_csr_descr(&A, rows, rows, nnz, rp, ci, va, rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    RS(rocsparse_create_dnvec_descr(&vx, rows, x, rocsparse_datatype_f64_r));
    RS(rocsparse_create_dnvec_descr(&vy, rows, y, rocsparse_datatype_f64_r));
    const double alpha = 1.0, beta = 0.0;
    const double bytes = 12.0 * nnz + 4.0 * (rows + 1) + 16.0 * rows;
    struct { rocsparse_spmv_alg alg; const char* name; } algs[] = {{rocsparse_spmv_alg_default, "default"}, {rocsparse_spmv_alg_csr_adaptive, "csr_adaptive"},
                                                                   {rocsparse_spmv_alg_csr_rowsplit, "csr_rowsplit"}, {rocsparse_spmv_alg_csr_lrb, "csr_lrb"}};
    for (auto& a : algs) {
        size_t bs = 0; void* buf = nullptr;
        RS(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, a.alg, rocsparse_spmv_stage_buffer_size, &bs, nullptr));
        CK(hipMalloc(&buf, bs ? bs : 8));
        RS(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, a.alg, rocsparse_spmv_stage_preprocess, &bs, buf));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<float> ms;
        for (int it = 0; it < 15; ++it) {
            CK(hipEventRecord(e0));
            RS(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, a.alg, rocsparse_spmv_stage_compute, &bs, buf));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (it >= 5) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const double med = 0.5 * (ms[4] + ms[5]);
        std::vector<double> hy(8);
        CK(hipMemcpy(hy.data(), y, 64, hipMemcpyDeviceToHost));
        printf("rocsparse_spmv CSR %-13s grid %d: median %.3f ms  %.1f GB/s (reference 'effective' bytes)  y[0..2]=%g %g %g\n", a.name, n, med, bytes / med / 1e6, hy[0], hy[1], hy[2]);
        if (y_file && a.alg == rocsparse_spmv_alg_default) {
            std::vector<double> out((size_t)rows);
            CK(hipMemcpy(out.data(), y, (size_t)rows * 8, hipMemcpyDeviceToHost));
            FILE* f = fopen(y_file, "wb");
            if (!f || fwrite(out.data(), sizeof(double), (size_t)rows, f) != (size_t)rows) { printf("cannot write %s\n", y_file); return 1; }
            fclose(f);
            printf("wrote y of rocsparse_spmv (default algorithm) to %s\n", y_file);
        }
        CK(hipFree(buf));
    }
    return 0;
}
