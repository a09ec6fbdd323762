// asan_stubs.cpp -- test infrastructure for `make asan`: the two functions of nvk_paged_attn.hip that the host-only
// translation units call (the error text), so they link without the HIP half of the library.
#include <cstdarg>
#include <cstdio>

#include "nvk_internal.h"

static thread_local char g_err[512] = "";
int nvk::fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char *nvk_last_error(void) { return g_err; }
