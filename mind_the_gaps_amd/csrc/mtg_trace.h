// mtg_trace.h -- roctx ranges around the host-side phases of the C-ABI:
// `rocprofv3 --marker-trace --kernel-trace -- <cmd>` then shows prepare / solve / gather / simulate
// spans next to the kernels.  The marker library is looked up at run time (librocprofiler-sdk-roctx,
// then the older libroctx64); without it, or outside a profiler, a range costs one predictable branch.
#pragma once
#include <dlfcn.h>

namespace mtg_trace {

struct Api {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Api()
    {
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(name, RTLD_LAZY | RTLD_LOCAL)) {
                push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};

inline const Api &api()
{
    static const Api a;
    return a;
}

struct Range {  // one nested range for the lifetime of the object
    bool on;
    explicit Range(const char *label) : on(api().push != nullptr) { if (on) api().push(label); }
    ~Range() { if (on) api().pop(); }
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
};

}  // namespace mtg_trace
