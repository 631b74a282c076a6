// Error reporting and ABI version of liblime_hip.so.
#include "common.h"

static thread_local char g_err[512] = "";

void lime_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_kernel[160] = "";

void lime_set_last_linear_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

extern "C" const char* lime_last_linear_kernel(void) { return g_kernel; }
extern "C" int lime_abi_version(void) { return LIME_ABI_VERSION; }
extern "C" const char* lime_last_error_string(void) { return g_err; }
