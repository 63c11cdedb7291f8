// roctx ranges around the phases of a substep and of the contact solve (the reference has no
// tracing hooks; a caller profiling a Drake run with `rocprofv3 --marker-trace` wants the engine's time attributed
// without a special build).  Off unless MPM_ROCTX=1 is set when the first engine is created; the marker library is
// bound at run time (the one rocprofv3 preloads, else the first that can be opened) so that nobody else needs it.
// Ranges mark the HOST span in which a phase's kernels are enqueued; rocprofv3 correlates the dispatches inside it.
#pragma once
#include <dlfcn.h>

#include <cstdlib>

namespace roctx_rt {
using Push = int (*)(const char*);
using Pop = int (*)();
struct Api {
    Push push = nullptr;
    Pop pop = nullptr;
};
static const Api* api() {
    static Api a;
    static int state = 0;   // 0 untried, 1 on, 2 off
    if (state == 0) {
        state = 2;
        const char* on = getenv("MPM_ROCTX");
        if (on && atoi(on) != 0) {
            const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
            void* lib = nullptr;
            for (const char* n : names)
                if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
            if (!lib)
                for (const char* n : names)
                    if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
            if (lib) {
                a.push = (Push)dlsym(lib, "roctxRangePushA");
                a.pop = (Pop)dlsym(lib, "roctxRangePop");
                if (a.push && a.pop) state = 1;
            }
        }
    }
    return state == 1 ? &a : nullptr;
}
}  // namespace roctx_rt

struct TraceRange {
    bool on = false;
    explicit TraceRange(const char* name) {
        if (const roctx_rt::Api* a = roctx_rt::api()) {
            a->push(name);
            on = true;
        }
    }
    ~TraceRange() {
        if (on) roctx_rt::api()->pop();
    }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};
