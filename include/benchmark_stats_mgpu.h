/* benchmark_stats_mgpu.h -- drop-in name for the reference header of the same name.
 * Replaces reference include/benchmark_stats_mgpu.h.
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef BENCHMARK_STATS_MGPU_H
#define BENCHMARK_STATS_MGPU_H
#include "benchmark_stats.h"
#include "solvers/cg_solver_mgpu.h"
#endif
