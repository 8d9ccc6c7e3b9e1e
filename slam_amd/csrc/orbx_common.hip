// Error reporting shared by the orbx translation units.
#include <string.h>

#include "orbx_internal.h"

static thread_local char g_err[512] = "";

void orbx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* orbx_last_error(void) { return g_err; }
