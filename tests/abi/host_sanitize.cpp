// host_sanitize.cpp -- the host layer of the boundary (Matrix Market reader / writer, build_csr_struct, the ELLPACK builders)
// compiled with -fsanitize=address,undefined and driven through well-formed and malformed inputs. GPU AddressSanitizer is
// not available on the test pool, so the sanitizers cover what runs on the CPU: csrc/matrix_market.cpp and csrc/host_matrix.cpp,
// which are plain C++ and link without HIP. Built and run by tests/test_host_logic.py::test_host_layer_under_sanitizers.
// usage: host_sanitize <scratch directory>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "spmv_amd.h"

static int failures = 0;
#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) {                                                        \
            fprintf(stderr, "host_sanitize: line %d: %s\n", __LINE__, #cond); \
            ++failures;                                                       \
        }                                                                     \
    } while (0)

static void write_text(const std::string& path, const char* text) {
    FILE* f = fopen(path.c_str(), "w");
    if (!f) exit(2);
    fputs(text, f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string dir = argv[1];

    // writer -> reader -> CSR -> ELLPACK on a small stencil
    const std::string stencil = dir + "/s7.mtx";
    EXPECT(write_matrix_market_stencil5(7, stencil.c_str()) == 0);
    MatrixData m;
    memset(&m, 0, sizeof m);
    EXPECT(load_matrix_market(stencil.c_str(), &m) == 0);
    EXPECT(m.rows == 49 && m.cols == 49 && m.nnz == 5 * 49 - 4 * 7 && m.grid_size == 7);
    EXPECT(build_csr_struct(&m) == EXIT_SUCCESS);
    EXPECT(csr_mat.nb_rows == 49 && csr_mat.row_ptr[49] == m.nnz);
    for (int r = 0; r < 49; ++r)
        for (int k = csr_mat.row_ptr[r] + 1; k < csr_mat.row_ptr[r + 1]; ++k) EXPECT(csr_mat.col_indices[k - 1] < csr_mat.col_indices[k]);
    EXPECT(ensure_ellpack_structure_built(&m) == EXIT_SUCCESS);
    EXPECT(ellpack_matrix.ell_width == 5 && ellpack_matrix.nb_rows == 49);
    free(m.entries);
    spmv_amd_reset_host_matrices();

    // rows far beyond the insertion limit, with repeated columns, in scrambled order (the stable_sort path)
    {
        const int rows = 5, per_row = 700;
        std::vector<Entry> e;
        unsigned state = 12345u;
        for (int r = 0; r < rows; ++r)
            for (int k = 0; k < (r == 3 ? 2 : per_row); ++k) {
                state = state * 1664525u + 1013904223u;
                e.push_back(Entry{r, (int)((state >> 8) % 37u), (double)k});
            }
        for (size_t i = e.size() - 1; i > 0; --i) {
            state = state * 1664525u + 1013904223u;
            std::swap(e[i], e[(state >> 8) % (i + 1)]);
        }
        MatrixData big = {rows, 37, (int)e.size(), -1, e.data()};
        EXPECT(build_csr_struct(&big) == EXIT_SUCCESS);
        EXPECT(csr_mat.row_ptr[rows] == (int)e.size());
        for (int r = 0; r < rows; ++r)
            for (int k = csr_mat.row_ptr[r] + 1; k < csr_mat.row_ptr[r + 1]; ++k) EXPECT(csr_mat.col_indices[k - 1] <= csr_mat.col_indices[k]);
        int width = 0;
        ELLPACKMatrix ell;
        memset(&ell, 0, sizeof ell);
        EXPECT(build_ellpack_from_csr_struct(&csr_mat, &ell, &width) == EXIT_SUCCESS && width == per_row);
        free(ell.indices);
        free(ell.values);
        spmv_amd_reset_host_matrices();
    }

    // symmetric files: a good one, and the malformed ones the reader must refuse without touching memory it does not own
    const std::string sym = dir + "/sym.mtx";
    write_text(sym, "%%MatrixMarket matrix coordinate real symmetric\n% STENCIL_GRID_SIZE 2\n4 4 6\n1 1 2.0\n2 1 -1.0\n2 2 2.0\n3 3 2.0\n4 3 -1.0\n4 4 2.0\n");
    {
        int rows = 0, cols = 0, nnz = 0, full = 0, *rp = nullptr, *ci = nullptr;
        double* va = nullptr;
        MatrixData s;
        memset(&s, 0, sizeof s);
        read_matrix_symtogen(&s, sym.c_str(), &rows, &cols, &nnz, &rp, &ci, &va, &full);
        EXPECT(s.entries != nullptr && rows == 4 && nnz == 6 && full == 8 && rp != nullptr && rp[4] == 8);
        free(s.entries), free(rp), free(ci), free(va);
    }
    const char* bad[] = {
        "%%MatrixMarket matrix coordinate real symmetric\n3 3 2\n1 1 2.0\n4 1 -1.0\n",   // row beyond the matrix
        "%%MatrixMarket matrix coordinate real symmetric\n3 3 2\n1 1 2.0\n2 0 -1.0\n",   // index 0 in a 1-based file
        "%%MatrixMarket matrix coordinate real symmetric\n3 4 2\n1 1 2.0\n2 4 -1.0\n",   // not square
        "%%MatrixMarket matrix coordinate real general\n3 3 4\n1 1 2.0\n2 2\n",          // truncated entry, fewer entries than announced
        "%%MatrixMarket matrix coordinate real general\n",                                // no size line
        "",                                                                                // empty file
    };
    for (size_t i = 0; i < sizeof bad / sizeof bad[0]; ++i) {
        const std::string path = dir + "/bad" + std::to_string(i) + ".mtx";
        write_text(path, bad[i]);
        MatrixData b;
        memset(&b, 0, sizeof b);
        const int rc = load_matrix_market(path.c_str(), &b);
        EXPECT(rc != 0 && b.entries == nullptr);
    }
    MatrixData none;
    memset(&none, 0, sizeof none);
    EXPECT(load_matrix_market((dir + "/does_not_exist.mtx").c_str(), &none) != 0);

    printf("host_sanitize: %s (%d failure%s)\n", failures ? "FAILED" : "ok", failures, failures == 1 ? "" : "s");
    return failures ? 1 : 0;
}
