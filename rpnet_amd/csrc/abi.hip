// Error plumbing and version of the C ABI (include/rpnet_abi.h).
#include <stdarg.h>

#include "common.h"

namespace rpnet {

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
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return RPNET_OK;
}

}  // namespace rpnet

extern "C" int rpnet_version(void) { return 100; }
extern "C" const char* rpnet_last_error_string(void) { return rpnet::g_err; }
