// Library-level entry points and the thread-local error string.
#include <stdarg.h>

#include "s2d_common.h"

namespace s2d {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace s2d

extern "C" int s2d_version(void) { return 100; }

extern "C" int s2d_last_error(char *buf, size_t buf_len) {
    size_t n = strlen(s2d::g_err);
    if (buf && buf_len) {
        size_t m = n < buf_len - 1 ? n : buf_len - 1;
        memcpy(buf, s2d::g_err, m);
        buf[m] = 0;
    }
    return (int)n;
}
