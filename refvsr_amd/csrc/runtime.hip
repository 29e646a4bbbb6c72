// Error reporting + one-time init for the C-ABI library.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void refvsr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* refvsr_last_error(void) { return g_err; }
extern "C" int refvsr_abi_version(void) { return REFVSR_ABI_VERSION; }

extern "C" int refvsr_init(void) {
    static int state = 0;   // 0 = not done, 1 = ok, 2 = failed
    if (state == 1) return 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        refvsr_set_error("refvsr_init: no HIP device visible (%s)", hipGetErrorString(e));
        state = 2;
        return 3;
    }
    state = 1;
    return 0;
}

// Debug knob shared by the fused-block kernels (resblock_lean.hip, resblock24.hip, resblock48.hip): when set, their PROBE
// instantiations record s_memtime stamps per workgroup (tools/probe_resblock24.py, tools/probe_resblock48.py).
unsigned long long* g_rb_probe = nullptr;
int g_rb_probe_iter = 0;
extern "C" int refvsr_set_probe(void* buf, int iter) {
    g_rb_probe = (unsigned long long*)buf;
    g_rb_probe_iter = iter;
    return 0;
}
