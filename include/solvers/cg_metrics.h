/* solvers/cg_metrics.h -- drop-in name for the reference header of the same name.
 * Replaces reference include/solvers/cg_metrics.h (export_cg_json, export_cg_mgpu_json, export_cg_csv).
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef CG_METRICS_H
#define CG_METRICS_H
#include "solvers/cg_solver.h"
#include "solvers/cg_solver_mgpu.h"
#include "benchmark_stats.h"
#endif
