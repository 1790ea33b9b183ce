// watchdog.cpp -- see watchdog.hpp.
#include "watchdog.hpp"

#include <stdlib.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

namespace spmv_amd {
namespace {

using Clock = std::chrono::steady_clock;

struct Frame {
    unsigned long long id;  // handed to the scope that armed it: scopes of different host threads leave in any order
    const char* stage;
    int rank;
    int iteration;
    WatchdogReportFn report;
    void* user;
    Clock::time_point since;
};

constexpr int kMaxDepth = 8;

// Heap-allocated and never freed: the helper thread may outlive static destruction.
struct State {
    std::mutex lock;
    Frame frames[kMaxDepth];
    int depth = 0;
    unsigned long long next_id = 1;
    double limit_s = 60.0;
    bool thread_started = false;
};

State& state() {
    static State* s = [] {
        State* st = new State();
        if (const char* v = getenv("SPMV_AMD_WATCHDOG_S")) st->limit_s = atof(v);
        return st;
    }();
    return *s;
}

[[noreturn]] void fire(const Frame& f, double waited_s) {
    fprintf(stderr, "\n[spmv_amd watchdog] rank %d: no progress for %.1f s in stage '%s'", f.rank, waited_s, f.stage);
    if (f.iteration >= 0) fprintf(stderr, " (CG iteration %d)", f.iteration);
    fprintf(stderr, "; limit SPMV_AMD_WATCHDOG_S=%g\n", state().limit_s);
    if (f.report) {
        // the report queries HIP / RCCL state; should one of those calls block behind the wedged main thread,
        // the process still ends five seconds from now
        std::atomic<bool>* done = new std::atomic<bool>(false);
        std::thread([f, done] {
            f.report(f.user, stderr);
            done->store(true);
        }).detach();
        for (int i = 0; i < 50 && !done->load(); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!done->load()) fprintf(stderr, "[spmv_amd watchdog] rank %d: the state report itself did not return\n", f.rank);
    }
    fprintf(stderr, "[spmv_amd watchdog] rank %d: ending the process with status %d\n", f.rank, EXIT_FAILURE);
    fflush(stderr);
    _exit(EXIT_FAILURE);  // the main thread is wedged: no atexit handlers, no destructors
}

void patrol() {
    State& st = state();
    for (;;) {
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        Frame overdue;
        double waited = 0.0;
        bool found = false;
        {
            std::lock_guard<std::mutex> g(st.lock);
            // Within one thread the innermost scope names where the rank is actually stuck and outer scopes are at least
            // as old; with scopes of several threads armed, any overdue frame is a finding: report the most recently
            // armed one that is overdue.
            const Clock::time_point now = Clock::now();
            for (int k = st.depth - 1; k >= 0 && !found; --k) {
                const Frame& f = st.frames[k];
                waited = std::chrono::duration<double>(now - f.since).count();
                if (waited > st.limit_s) {
                    overdue = f;
                    found = true;
                }
            }
        }
        if (found) fire(overdue, waited);
    }
}

}  // namespace

double watchdog_limit_seconds() { return state().limit_s; }

WatchdogScope::WatchdogScope(const char* stage, int rank, int iteration, WatchdogReportFn report, void* user) {
    State& st = state();
    if (!(st.limit_s > 0.0)) return;
    std::lock_guard<std::mutex> g(st.lock);
    if (st.depth >= kMaxDepth) return;
    id_ = st.next_id++;
    st.frames[st.depth++] = Frame{id_, stage, rank, iteration, report, user, Clock::now()};
    armed_ = true;
    if (!st.thread_started) {
        st.thread_started = true;
        std::thread(patrol).detach();
    }
}

WatchdogScope::~WatchdogScope() {
    if (!armed_) return;
    State& st = state();
    std::lock_guard<std::mutex> g(st.lock);
    // remove THIS scope's frame, wherever it sits: another host thread may have armed a scope after it
    for (int k = st.depth - 1; k >= 0; --k) {
        if (st.frames[k].id != id_) continue;
        for (int j = k; j + 1 < st.depth; ++j) st.frames[j] = st.frames[j + 1];
        --st.depth;
        break;
    }
}

}  // namespace spmv_amd
