/* io.h -- drop-in name for the reference header of the same name.
 * Replaces reference include/io.h (Entry, MatrixData, Matrix Market reader/writer).
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef SPMV_IO_H
#define SPMV_IO_H
#include "spmv_amd/types.h"
#include "spmv_amd/api.h"
#endif
