// nvk_internal.h -- shared by the translation units of libnvk_paged_attn.so.
#pragma once

#include <cstdarg>
#include <cstdio>
#include <new>

#include "nvk_paged_attn.h"

namespace nvk {

// Records `msg` (printf-style) as the calling thread's last error and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

} // namespace nvk
