// watchdog.hpp -- wall-clock watchdog for every point where a solver rank can wait on another rank.
//
// The reference blocks in MPI_Waitall / MPI_Allreduce / cudaStreamSynchronize with no bound
// (cg_solver_mgpu_partitioned.cu:202-231,531,583,645): a wedged rank hangs the job until the scheduler
// kills it. Here each such wait runs inside a WatchdogScope. A helper thread checks the armed scope every
// 100 ms; when one outlives its limit the thread prints which rank, which stage, which iteration and what
// the GPU queues look like, then ends the process with a non-zero status -- a deadlock becomes a diagnosable
// exit instead of a timeout kill.
//
// Limit: SPMV_AMD_WATCHDOG_S seconds (default 60, 0 disables), read once per process.
#pragma once

#include <stdio.h>

namespace spmv_amd {

// Extra diagnostics printed by the watchdog thread before it ends the process (stream / event queries of
// the solver that armed the scope). Must not block.
typedef void (*WatchdogReportFn)(void* user, FILE* out);

class WatchdogScope {
public:
    // `stage` must outlive the scope (string literals). iteration < 0: not inside the CG loop.
    WatchdogScope(const char* stage, int rank, int iteration = -1, WatchdogReportFn report = nullptr,
                  void* user = nullptr);
    ~WatchdogScope();
    WatchdogScope(const WatchdogScope&) = delete;
    WatchdogScope& operator=(const WatchdogScope&) = delete;

private:
    bool armed_ = false;
    unsigned long long id_ = 0;  // the frame this scope armed; frames are removed by id, so scopes of several host threads may overlap
};

double watchdog_limit_seconds();

}  // namespace spmv_amd
