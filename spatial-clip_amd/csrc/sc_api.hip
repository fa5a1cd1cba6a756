// Error plumbing of the C ABI: thread-local last-error string, no exceptions cross the boundary.
#include "sc_common.h"
#include "sc_kernels.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void sc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* sc_last_error(void) { return g_err; }
extern "C" int sc_abi_version(void) { return 1; }
