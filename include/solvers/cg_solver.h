/* solvers/cg_solver.h -- drop-in name for the reference header of the same name.
 * Replaces reference include/solvers/cg_solver.h (CGConfig, CGStats, cg_solve, cg_solve_device).
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef CG_SOLVER_H
#define CG_SOLVER_H
#include "spmv.h"
#endif
