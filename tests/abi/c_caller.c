/* A plain-C caller of the drop-in boundary: what a cgo / JNI / N-API binding sees. Compiled with gcc -std=c11 against
 * include/spmv_amd/api.h and linked against libspmv_amd.so by tests/test_host_logic.py (no GPU needed: only the
 * integer helpers, the operator table lookup and the Matrix Market writer / reader are called). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spmv_amd/api.h"

int main(int argc, char** argv) {
    int off = -1, n_local = -1;
    /* reference src/spmv/spmv_stencil_csr_direct.cu:50-67: row 82 of the 81 x 81 grid starts at 326 */
    if (spmv_amd_interior_csr_offset(82, 81) != 326) return 1;
    /* reference cg_solver_mgpu_partitioned.cu:261-268: 400 M rows over 8 ranks, last rank takes the remainder */
    spmv_amd_partition_rows(400000000, 8, 7, &off, &n_local);
    if (off != 350000000 || n_local != 50000000) return 2;
    spmv_amd_partition_rows(10, 3, 2, &off, &n_local);
    if (off != 6 || n_local != 4) return 3;
    if (get_operator("stencil5-csr") == NULL || get_operator("cusparse-csr") == NULL || get_operator("ellpack") == NULL) return 4;
    if (get_operator("no-such-mode") != NULL) return 5;
    if (strcmp(get_operator("stencil5")->name, "stencil5-csr") != 0) return 6;
    if (argc > 1) { /* write a 3 x 3 stencil file and read it back through the reference-named I/O entry points */
        MatrixData m;
        memset(&m, 0, sizeof m);
        if (write_matrix_market_stencil5(3, argv[1]) != 0) return 7;
        if (load_matrix_market(argv[1], &m) != 0) return 8;
        if (m.rows != 9 || m.cols != 9 || m.nnz != 33 || m.grid_size != 3 || m.entries[0].value != 5.0) return 9;
        free(m.entries);
    }
    printf("c caller ok: %s\n", spmv_amd_version());
    return 0;
}
