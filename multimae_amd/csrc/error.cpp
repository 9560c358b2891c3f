// Thread-local error message of the C ABI.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mmae.h"

static thread_local char g_err[512] = "";

void mmae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mmae_last_error(void) { return g_err; }
extern "C" int mmae_version(void) { return 1; }
