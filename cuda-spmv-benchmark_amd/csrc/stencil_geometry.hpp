// stencil_geometry.hpp -- integer structure of the CSR form of an n x n 5-point
// stencil whose rows are sorted by column ([N,W,C,E,S] minus the absent ones).
// Shared by host and device code. All of it must agree bit for bit with the
// reference's index arithmetic: for interior rows
// stencil_row_start() equals calculate_interior_csr_offset()
// (reference src/spmv/spmv_stencil_csr_direct.cu:50-67).
#pragma once

#if defined(__HIPCC__)
#define SPMV_HD __host__ __device__ inline
#else
#define SPMV_HD inline
#endif

namespace spmv_amd {

// nnz in grid rows [0, i): row 0 holds 4n-2 entries, rows 1..n-2 hold 5n-2 each.
SPMV_HD long long stencil_gridrow_base(int i, int n) {
    if (i <= 0) return 0;
    long long first = 4LL * n - 2;
    if (i <= n - 1) return first + (long long)(i - 1) * (5LL * n - 2);
    return first + (long long)(n - 2) * (5LL * n - 2) + (4LL * n - 2);  // i == n: total nnz
}

// CSR start of row (i, j), valid for every row of the grid when n >= 2.
SPMV_HD long long stencil_row_start(int i, int j, int n) {
    int vertical = (i > 0) + (i < n - 1);
    long long s = stencil_gridrow_base(i, n);
    if (j > 0) s += (2 + vertical) + (long long)(j - 1) * (3 + vertical);
    return s;
}

// Same for a flat row index, also accepting row == n*n (-> nnz).
SPMV_HD long long stencil_row_start_flat(long long row, int n) {
    int i = (int)(row / n);
    int j = (int)(row - (long long)i * n);
    if (i >= n) return stencil_gridrow_base(n, n);
    return stencil_row_start(i, j, n);
}

SPMV_HD int stencil_row_nnz(int i, int j, int n) {
    return 1 + (i > 0) + (i < n - 1) + (j > 0) + (j < n - 1);
}

SPMV_HD bool stencil_is_interior(int i, int j, int n) {
    return i > 0 && i < n - 1 && j > 0 && j < n - 1;
}

// The reference's closed form, kept in 32-bit int exactly as written there.
SPMV_HD int reference_interior_csr_offset(int row, int grid_size) {
    int i = row / grid_size;
    int j = row % grid_size;
    int row0_nnz = 3 + (grid_size - 2) * 4 + 3;
    int interior_row_nnz = 4 + (grid_size - 2) * 5 + 4;
    int offset = row0_nnz + (i - 1) * interior_row_nnz;
    offset += 4 + (j - 1) * 5;
    return offset;
}

}  // namespace spmv_amd
