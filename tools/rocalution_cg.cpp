// tools/rocalution_cg.cpp -- comparator slot (the role AmgX plays in the reference's
// external/benchmarks/amgx): the vendor library's unpreconditioned CG (rocALUTION) on the same generator stencil,
// b = 1, x0 = 0, relative tolerance 1e-6. Not part of the product; nothing here is linked into libspmv_amd.so.
//   make -C tools  (-> tools/bin/rocalution_cg) ;  tools/bin/rocalution_cg 20000 [runs]
// The last line, "RESULT iterations=<k> residual=<%.17e> ...", is what tests/test_comparators_gpu.py parses.
#include <hip/hip_runtime.h>
#include <rocalution/rocalution.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <vector>

using namespace rocalution;

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 10000;
    const int runs = argc > 2 ? atoi(argv[2]) : 5;
    const int64_t rows = (int64_t)n * n, nnz = 5 * rows - 4 * (int64_t)n;
    if (nnz > 0x7fffffffLL) { printf("grid too large for 32-bit CSR\n"); return 1; }
    init_rocalution();
    {
        LocalMatrix<double> A;
        LocalVector<double> x, b;
        A.AllocateCSR("A", nnz, rows, rows);
        {
            std::vector<PtrType> rp((size_t)rows + 1);
            std::vector<int> ci((size_t)nnz);
            std::vector<double> va((size_t)nnz);
            int64_t k = 0;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    const int64_t r = (int64_t)i * n + j;
                    rp[(size_t)r] = (PtrType)k;
                    if (i > 0) { ci[(size_t)k] = (int)(r - n); va[(size_t)k++] = -1.0; }
                    if (j > 0) { ci[(size_t)k] = (int)(r - 1); va[(size_t)k++] = -1.0; }
                    ci[(size_t)k] = (int)r; va[(size_t)k++] = 5.0;
                    if (j < n - 1) { ci[(size_t)k] = (int)(r + 1); va[(size_t)k++] = -1.0; }
                    if (i < n - 1) { ci[(size_t)k] = (int)(r + n); va[(size_t)k++] = -1.0; }
                }
            rp[(size_t)rows] = (PtrType)k;
            A.CopyFromCSR(rp.data(), ci.data(), va.data());
        }
        x.Allocate("x", rows);
        b.Allocate("b", rows);
        A.MoveToAccelerator();
        x.MoveToAccelerator();
        b.MoveToAccelerator();
        b.Ones();

        CG<LocalMatrix<double>, LocalVector<double>, double> cg;
        cg.SetOperator(A);
        cg.Init(0.0, 1e-6, 1e8, 1000);
        cg.Verbose(0);
        cg.Build();

        std::vector<double> ms;
        int iterations = 0;
        double residual = 0.0;
        for (int run = 0; run < runs + 2; ++run) {
            x.Zeros();
            (void)hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            cg.Solve(b, &x);
            (void)hipDeviceSynchronize();
            const auto t1 = std::chrono::steady_clock::now();
            if (run >= 2) ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
            iterations = cg.GetIterationCount();
            residual = cg.GetCurrentResidual();
        }
        std::sort(ms.begin(), ms.end());
        printf("rocALUTION CG (no preconditioner), %d x %d stencil, %lld unknowns: %d iterations, final residual %.6e, "
               "median of %d solves %.2f ms (min %.2f, max %.2f) = %.1f iterations/s\n",
               n, n, (long long)rows, iterations, residual, runs, ms[ms.size() / 2], ms.front(), ms.back(),
               iterations / (ms[ms.size() / 2] / 1e3));
        printf("RESULT iterations=%d residual=%.17e grid=%d median_ms=%.4f\n", iterations, residual, n, ms[ms.size() / 2]);
        cg.Clear();
    }
    stop_rocalution();
    return 0;
}
