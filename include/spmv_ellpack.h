/* spmv_ellpack.h -- drop-in name for the reference header of the same name.
 * Replaces reference include/spmv_ellpack.h (ELLPACKMatrix, MAX_WIDTH, builder).
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef SPMV_ELLPACK_H
#define SPMV_ELLPACK_H
#include "spmv_amd/types.h"
#include "spmv_amd/api.h"
#endif
