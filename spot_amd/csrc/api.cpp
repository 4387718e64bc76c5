// Library-wide C-ABI plumbing: version and thread-local error string.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/spot_hip.h"

static thread_local char g_err[512] = "";

void spot_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int spot_version(void) { return SPOT_HIP_VERSION; }

extern "C" const char* spot_last_error(void) { return g_err; }
