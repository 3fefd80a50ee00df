// error.cpp — thread-local error string, version and the development switches of the C ABI.
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

// ---- development switches -------------------------------------------------------------------------------------------
// Every switch has ONE reader, lc_option(): a per-thread override set through lc_set_option() wins, then the LC_*
// environment variable (read per call: the tests flip schedules between calls), then the built-in default.  The
// override is what in-process recovery uses (re-running a step on the launch train must not mutate the process
// environment under threads that are reading it).
static const char *const g_opt_name[] = {"lstm_persistent", "lstm_spin_limit", "gemm_f32_big", "gemm_bf16_big", "ctc_lse2",
                                         "gemm_bf16_persist", "gemm_tail"};
static const char *const g_opt_env[] = {"LC_LSTM_PERSISTENT", "LC_LSTM_SPIN_LIMIT", "LC_GEMM_F32_BIG",
                                        "LC_GEMM_BF16_BIG", "LC_CTC_LSE2", "LC_GEMM_BF16_PERSIST",
                                        "LC_GEMM_TAIL"};
enum { N_OPTS = sizeof(g_opt_name) / sizeof(g_opt_name[0]) };
static thread_local long g_opt_val[N_OPTS];
static thread_local bool g_opt_set[N_OPTS];

long lc_option(int opt, long dflt)
{
    if (opt < 0 || opt >= N_OPTS) return dflt;
    if (g_opt_set[opt]) return g_opt_val[opt];
    const char *env = getenv(g_opt_env[opt]);
    return env ? atol(env) : dflt;
}
static int opt_index(const char *name)
{
    for (int i = 0; name && i < N_OPTS; ++i)
        if (strcmp(name, g_opt_name[i]) == 0) return i;
    return -1;
}
extern "C" int lc_set_option(const char *name, long value)
{
    const int i = opt_index(name);
    if (i < 0) { lc_set_error("lc_set_option: unknown option '%s'", name ? name : "(null)"); return LC_EINVAL; }
    g_opt_set[i] = value != LC_OPTION_UNSET;
    g_opt_val[i] = value;
    return LC_OK;
}
extern "C" int lc_get_option(const char *name, long *value)
{
    const int i = opt_index(name);
    if (i < 0) { lc_set_error("lc_get_option: unknown option '%s'", name ? name : "(null)"); return LC_EINVAL; }
    const long v = lc_option(i, LC_OPTION_UNSET);
    if (value) *value = v;
    return LC_OK;
}
