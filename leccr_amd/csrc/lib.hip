// lib.hip - library plumbing: version, thread-local error text, launch check.
#include <stdarg.h>

#include "common.h"

namespace leccr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return -1000 - (int)e;
    }
    return LECCR_OK;
}
}  // namespace leccr

extern "C" int leccr_version(void) { return 1; }
extern "C" const char* leccr_last_error(void) { return leccr::g_err; }
