/* spmv_csr.h -- drop-in name for the reference header of the same name.
 * Replaces reference include/spmv_csr.h (CSRMatrix, build_csr_struct).
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef SPMV_CSR_H
#define SPMV_CSR_H
#include "spmv_amd/types.h"
#include "spmv_amd/api.h"
#endif
