/* spmv_amd.h -- drop-in name for the reference header of the same name.
 * Umbrella header of libspmv_amd.so.
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef SPMV_AMD_H
#define SPMV_AMD_H
#include "spmv_amd/types.h"
#include "spmv_amd/api.h"
#include "spmv_amd/hip_check.h"
#endif
