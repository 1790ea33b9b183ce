/* benchmark_stats.h -- drop-in name for the reference header of the same name.
 * Replaces reference include/benchmark_stats.h.
 * The declarations live in spmv_amd/types.h and spmv_amd/api.h. */
#ifndef BENCHMARK_STATS_H
#define BENCHMARK_STATS_H
#include "spmv.h"
#include "io.h"
#include "solvers/cg_solver.h"
#endif
