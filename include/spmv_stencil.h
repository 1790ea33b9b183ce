/* spmv_stencil.h -- drop-in name for the reference header of the same name.
 * Replaces reference include/spmv_stencil.h; the ELLPACK stencil kernel it prototypes is reached through get_operator("stencil5-ellpack").
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef SPMV_STENCIL_H
#define SPMV_STENCIL_H
#include "spmv_amd/types.h"
#include "spmv_amd/api.h"
#endif
