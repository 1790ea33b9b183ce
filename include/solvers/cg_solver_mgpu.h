/* solvers/cg_solver_mgpu.h -- drop-in name for the reference header of the same name.
 * Replaces reference include/solvers/cg_solver_mgpu.h (CGConfigMultiGPU, CGStatsMultiGPU); the full-replication cg_solve_mgpu it declares is undefined upstream and not provided.
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef CG_SOLVER_MGPU_H
#define CG_SOLVER_MGPU_H
#include "spmv.h"
#include "solvers/cg_solver.h"
#endif
