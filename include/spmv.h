/* spmv.h -- drop-in name for the reference header of the same name.
 * Replaces reference include/spmv.h (operator table, globals, metrics, error macro).
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef SPMV_H
#define SPMV_H
#include "spmv_amd/types.h"
#include "spmv_amd/api.h"
#include "spmv_amd/hip_check.h"
#endif
