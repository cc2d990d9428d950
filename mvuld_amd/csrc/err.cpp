// mvuld_last_error / mvuld_version for libmvuld_hip.so
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void mvuld_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mvuld_last_error(void) { return g_err; }
extern "C" int mvuld_version(void) { return 100; }
