/* solvers/cg_solver_mgpu_partitioned.h -- drop-in name for the reference header of the same name.
 * Replaces reference include/solvers/cg_solver_mgpu_partitioned.h (cg_solve_mgpu_partitioned).
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef CG_SOLVER_MGPU_PARTITIONED_H
#define CG_SOLVER_MGPU_PARTITIONED_H
#include "spmv.h"
#include "solvers/cg_solver.h"
#include "solvers/cg_solver_mgpu.h"
#endif
