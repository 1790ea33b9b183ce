// trace_ranges.hpp -- named ranges for rocprofv3 / roctx traces, the counterpart of the reference's NVTX ranges
// (cg_solver_mgpu_partitioned.cu:540-717: CG_Solver, CG_Iteration, SpMV, Dot_Product, BLAS_AXPY, BLAS_AXPBY,
// Halo_Exchange_MPI). librocprofiler-sdk-roctx is opened at run time, the first time ranges are asked for
// (CGConfig.enable_detailed_timers, or SPMV_AMD_ROCTX=1), so the library has no link-time dependency on the
// profiler SDK and the default path makes no call at all. `rocprofv3 --marker-trace` shows the ranges.
#pragma once

namespace spmv_amd {

class TraceRanges {
public:
    explicit TraceRanges(bool enabled);
    bool enabled() const { return push_ != nullptr; }
    void push(const char* name) const {
        if (push_) push_(name);
    }
    void pop() const {
        if (pop_) pop_();
    }

private:
    int (*push_)(const char*) = nullptr;
    int (*pop_)() = nullptr;
};

// push in the constructor, pop in the destructor
struct TraceScope {
    const TraceRanges& t;
    TraceScope(const TraceRanges& ranges, const char* name) : t(ranges) { t.push(name); }
    ~TraceScope() { t.pop(); }
};

}  // namespace spmv_amd
