// trace_ranges.cpp -- see trace_ranges.hpp.
#include "trace_ranges.hpp"

#include <dlfcn.h>
#include <stdio.h>

namespace spmv_amd {
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};

const Roctx& roctx() {
    static const Roctx r = [] {
        Roctx x;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (h == nullptr) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (h != nullptr) {
            x.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        }
        if (x.push == nullptr || x.pop == nullptr) {
            fprintf(stderr, "[spmv_amd] roctx ranges requested but no roctx library could be opened\n");
            x = Roctx{};
        }
        return x;
    }();
    return r;
}
}  // namespace

TraceRanges::TraceRanges(bool enabled) {
    if (!enabled) return;
    push_ = roctx().push;
    pop_ = roctx().pop;
}

}  // namespace spmv_amd
