// Library-wide pieces of the C ABI (include/gfs3d.h): version + error text.
#include <stdarg.h>

#include "common.h"

namespace gfs3d {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace gfs3d

extern "C" int gfs3d_version(void) { return 100; }   // 0.1.0
extern "C" const char *gfs3d_last_error(void) { return gfs3d::g_err; }
