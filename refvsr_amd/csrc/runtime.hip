// Error reporting + one-time init for the C-ABI library.
#include <stdarg.h>

#include "common.h"

#define RV_MAX_MASK_WORDS 16

static thread_local char g_err[512] = "";

void refvsr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* refvsr_last_error(void) { return g_err; }
extern "C" int refvsr_abi_version(void) { return REFVSR_ABI_VERSION; }
extern "C" int refvsr_max_maps(void) { return REFVSR_MAX_MAPS; }

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

// ---- CU partitions (round 6): streams restricted to a subset of the compute units ----------------------------------------
// A stream created with a CU mask runs its kernels only on the CUs of the mask, so kernels of two such streams with disjoint
// masks run side by side whatever LDS / wave budget each needs (two full-chip kernels of different streams cannot co-reside
// on a CU when one of them takes the whole LDS: they serialise).  The persistent launchers size their grids from the budget
// registered for the stream they launch on (rv_stream_cus) so that a workgroup still owns a CU's worth of tiles.
// Mask bit i = CU (i / 8) of XCD (i % 8) on gfx950 (the driver deals the bits round-robin over the XCDs): the first n bits are
// n / 8 CUs of every XCD, which keeps the workgroup -> XCD round-robin (b % 8) of rv_tile_range intact.
#define RV_MAX_STREAM_BUDGETS 16
static struct { hipStream_t st; int cus; } g_budget[RV_MAX_STREAM_BUDGETS];
static int g_n_budget = 0;

int rv_stream_cus(hipStream_t st) {
    for (int i = 0; i < g_n_budget; ++i)
        if (g_budget[i].st == st) return g_budget[i].cus;
    return rv_num_cus();
}

extern "C" int refvsr_num_cus(void) { return rv_num_cus(); }

extern "C" int refvsr_stream_set_cu_budget(void* stream, int n_cus) {
    hipStream_t st = (hipStream_t)stream;
    RV_CHECK(n_cus >= 0 && n_cus <= rv_num_cus(), "stream_set_cu_budget: n_cus %d outside [0, %d]", n_cus, rv_num_cus());
    RV_CHECK(n_cus == 0 || n_cus % 8 == 0, "stream_set_cu_budget: n_cus must be a multiple of 8 (one share per XCD)");
    for (int i = 0; i < g_n_budget; ++i)
        if (g_budget[i].st == st) {
            if (n_cus == 0) g_budget[i] = g_budget[--g_n_budget];
            else g_budget[i].cus = n_cus;
            return 0;
        }
    if (n_cus == 0) return 0;
    RV_CHECK(g_n_budget < RV_MAX_STREAM_BUDGETS, "stream_set_cu_budget: table full (%d streams)", RV_MAX_STREAM_BUDGETS);
    g_budget[g_n_budget].st = st;
    g_budget[g_n_budget].cus = n_cus;
    ++g_n_budget;
    return 0;
}

extern "C" int refvsr_stream_create_cu_range(int first_cu, int n_cus, void** stream) {
    RV_CHECK(stream, "stream_create_cu_range: null output");
    const int total = rv_num_cus();
    RV_CHECK(first_cu >= 0 && n_cus > 0 && first_cu + n_cus <= total && first_cu % 8 == 0 && n_cus % 8 == 0,
             "stream_create_cu_range: [%d, %d) must be a multiple-of-8 range inside [0, %d)", first_cu, first_cu + n_cus, total);
    RV_CHECK(refvsr_init() == 0, "init failed");
    uint32_t mask[RV_MAX_MASK_WORDS];
    const int words = (total + 31) / 32;
    RV_CHECK(words <= RV_MAX_MASK_WORDS, "stream_create_cu_range: %d CUs exceed the mask size", total);
    memset(mask, 0, sizeof(mask));
    for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st = nullptr;
    RV_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
    *stream = (void*)st;
    return refvsr_stream_set_cu_budget((void*)st, n_cus);
}

extern "C" int refvsr_stream_destroy(void* stream) {
    RV_CHECK(stream, "stream_destroy: null stream");
    refvsr_stream_set_cu_budget(stream, 0);
    RV_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
