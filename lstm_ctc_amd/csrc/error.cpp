// error.cpp — thread-local error string + version for the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/lstm_ctc_hip.h"

static thread_local char g_err[512] = "";

void lc_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *lc_last_error(void) { return g_err; }
extern "C" int lc_version(void) { return 1; }
