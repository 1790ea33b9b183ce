// host_matrix.cpp -- host-side matrix formats behind the operators: the process-wide
// csr_mat / ellpack_matrix and their builders.
//   build_csr_struct              <- reference src/spmv/spmv_cusparse_csr.cu:62-170
//   build_ellpack_from_csr_struct <- reference include/spmv_ellpack.h:50-51 (declaration only)
//   build_ellpack_from_csr_local,
//   ensure_ellpack_structure_built<- reference include/spmv.h:37-38 (declarations only)
//   convert_csr_to_ellpack        <- reference include/io.h:124-125 (declaration only)
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "spmv_amd.h"

CSRMatrix csr_mat = {0, 0, 0, nullptr, nullptr, nullptr};
ELLPACKMatrix ellpack_matrix = {0, 0, 0, -1, nullptr, 0, nullptr};

int build_csr_struct(struct MatrixData* mat) {
    // Reuse across operators in one process, keyed on (rows, nnz) exactly as upstream.
    if (csr_mat.row_ptr != nullptr && csr_mat.nb_rows == mat->rows &&
        csr_mat.nb_nonzeros == mat->nnz) {
        printf("CSR structure already built, reusing (%dx%d, %d nnz)\n", mat->rows, mat->cols,
               mat->nnz);
        return EXIT_SUCCESS;
    }
    printf("Building CSR structure (%dx%d, %d nnz)...\n", mat->rows, mat->cols, mat->nnz);
    fflush(stdout);

    const size_t rows = (size_t)mat->rows, nnz = (size_t)mat->nnz;
    int* row_ptr = (int*)calloc(rows + 1, sizeof(int));
    if (!row_ptr) {
        fprintf(stderr, "[ERROR] calloc failed for row_ptr\n");
        return EXIT_FAILURE;
    }
    const Entry* e = mat->entries;
    for (size_t k = 0; k < nnz; ++k) row_ptr[e[k].row + 1]++;
    for (size_t r = 1; r <= rows; ++r) row_ptr[r] += row_ptr[r - 1];

    int* col = (int*)malloc((nnz ? nnz : 1) * sizeof(int));
    double* val = (double*)malloc((nnz ? nnz : 1) * sizeof(double));
    int* cursor = (int*)malloc((rows ? rows : 1) * sizeof(int));
    if (!col || !val || !cursor) {
        free(row_ptr);
        free(col);
        free(val);
        free(cursor);
        return EXIT_FAILURE;
    }
    memcpy(cursor, row_ptr, rows * sizeof(int));
    // scatter in file order: the k-th entry of a row lands in the row's k-th slot
    for (size_t k = 0; k < nnz; ++k) {
        const int dst = cursor[e[k].row]++;
        col[dst] = e[k].col;
        val[dst] = e[k].value;
    }
    free(cursor);

    // Stable sort of every row by column index. The reference sorts each row by insertion (spmv_cusparse_csr.cu:137-160),
    // which is quadratic in the row length: a 20 000-entry row costs 10^8 moves, a matrix with 10^4 of them minutes (90 s
    // measured, tools/generic_matrix_perf.py). A stable sort has ONE possible output -- ascending columns, equal columns
    // in input order -- so long rows go through std::stable_sort and produce the array insertion would; short rows (every
    // row of the stencil) keep the insertion loop.
    constexpr int kInsertionLimit = 64;
    std::vector<std::pair<int, double>> scratch;
    for (size_t r = 0; r < rows; ++r) {
        const int lo = row_ptr[r], hi = row_ptr[r + 1];
        if (hi - lo > kInsertionLimit) {
            scratch.resize((size_t)(hi - lo));
            for (int a = lo; a < hi; ++a) scratch[(size_t)(a - lo)] = {col[a], val[a]};
            std::stable_sort(scratch.begin(), scratch.end(),
                             [](const std::pair<int, double>& x, const std::pair<int, double>& y) { return x.first < y.first; });
            for (int a = lo; a < hi; ++a) {
                col[a] = scratch[(size_t)(a - lo)].first;
                val[a] = scratch[(size_t)(a - lo)].second;
            }
            continue;
        }
        for (int a = lo + 1; a < hi; ++a) {
            const int c = col[a];
            const double v = val[a];
            int b = a - 1;
            for (; b >= lo && col[b] > c; --b) {
                col[b + 1] = col[b];
                val[b + 1] = val[b];
            }
            col[b + 1] = c;
            val[b + 1] = v;
        }
    }

    // Host arrays of an earlier, different matrix are released here; upstream leaks them.
    free(csr_mat.row_ptr);
    free(csr_mat.col_indices);
    free(csr_mat.values);
    csr_mat.row_ptr = row_ptr;
    csr_mat.col_indices = col;
    csr_mat.values = val;
    csr_mat.nb_rows = mat->rows;
    csr_mat.nb_cols = mat->cols;
    csr_mat.nb_nonzeros = mat->nnz;
    printf("CSR structure built successfully\n");
    fflush(stdout);
    return EXIT_SUCCESS;
}

int build_ellpack_from_csr_struct(const struct CSRMatrix* csr, ELLPACKMatrix* ell, int* max_width) {
    if (!csr || !ell || !csr->row_ptr) return EXIT_FAILURE;
    int width = 0;
    for (int r = 0; r < csr->nb_rows; ++r) {
        const int len = csr->row_ptr[r + 1] - csr->row_ptr[r];
        if (len > width) width = len;
    }
    if (max_width) *max_width = width;
    if (width > MAX_WIDTH) {
        fprintf(stderr, "[ERROR] ELLPACK width %d exceeds MAX_WIDTH %d\n", width, MAX_WIDTH);
        return EXIT_FAILURE;
    }
    const size_t slots = (size_t)csr->nb_rows * (size_t)width;
    int* idx = (int*)malloc((slots ? slots : 1) * sizeof(int));
    double* val = (double*)malloc((slots ? slots : 1) * sizeof(double));
    if (!idx || !val) {
        free(idx);
        free(val);
        return EXIT_FAILURE;
    }
    for (int r = 0; r < csr->nb_rows; ++r) {
        const int lo = csr->row_ptr[r], len = csr->row_ptr[r + 1] - lo;
        int* irow = idx + (size_t)r * width;
        double* vrow = val + (size_t)r * width;
        for (int k = 0; k < width; ++k) {
            const bool live = k < len;
            irow[k] = live ? csr->col_indices[lo + k] : -1;
            vrow[k] = live ? csr->values[lo + k] : 0.0;
        }
    }
    free(ell->indices);
    free(ell->values);
    ell->nb_rows = csr->nb_rows;
    ell->nb_cols = csr->nb_cols;
    ell->ell_width = width;
    ell->nb_nonzeros = csr->nb_nonzeros;
    ell->indices = idx;
    ell->values = val;
    return EXIT_SUCCESS;
}

extern "C" int build_ellpack_from_csr_local(CSRMatrix* csr_matrix) {
    int width = 0;
    return build_ellpack_from_csr_struct(csr_matrix, &ellpack_matrix, &width);
}

extern "C" int ensure_ellpack_structure_built(MatrixData* mat) {
    if (ellpack_matrix.values != nullptr && ellpack_matrix.nb_rows == mat->rows &&
        ellpack_matrix.nb_nonzeros == mat->nnz) {
        ellpack_matrix.grid_size = mat->grid_size;
        return EXIT_SUCCESS;
    }
    if (build_csr_struct(mat) != EXIT_SUCCESS) return EXIT_FAILURE;
    if (build_ellpack_from_csr_local(&csr_mat) != EXIT_SUCCESS) return EXIT_FAILURE;
    ellpack_matrix.grid_size = mat->grid_size;
    return EXIT_SUCCESS;
}

extern "C" void convert_csr_to_ellpack(const struct CSRMatrix* csr_matrix,
                                       struct ELLPACKMatrix* ell, int* max_width) {
    ell->indices = nullptr;
    ell->values = nullptr;
    ell->grid_size = -1;
    build_ellpack_from_csr_struct(csr_matrix, ell, max_width);
}

extern "C" void spmv_amd_reset_host_matrices(void) {
    free(csr_mat.row_ptr);
    free(csr_mat.col_indices);
    free(csr_mat.values);
    csr_mat = CSRMatrix{0, 0, 0, nullptr, nullptr, nullptr};
    free(ellpack_matrix.indices);
    free(ellpack_matrix.values);
    ellpack_matrix = ELLPACKMatrix{0, 0, 0, -1, nullptr, 0, nullptr};
}

extern "C" int spmv_amd_build_csr_struct(MatrixData* mat) { return build_csr_struct(mat); }

extern "C" int spmv_amd_build_ellpack_from_csr_struct(const CSRMatrix* csr, ELLPACKMatrix* ell,
                                                      int* max_width) {
    return build_ellpack_from_csr_struct(csr, ell, max_width);
}
