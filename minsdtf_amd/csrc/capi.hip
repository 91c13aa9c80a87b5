// Library-level entry points of the C ABI (include/minsdtf_hip.h).
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void msd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int msd_conv_gemm_init();
int msd_attention_init();

extern "C" int msd_abi_version(void) { return MSD_ABI_VERSION; }
extern "C" const char* msd_last_error(void) { return g_err; }
extern "C" int msd_init(void) {
    int rc = msd_conv_gemm_init();
    if (rc) return rc;
    return msd_attention_init();
}
