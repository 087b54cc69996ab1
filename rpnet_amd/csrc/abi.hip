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

// RPNET_ABI_VERSION (include/rpnet_abi.h): bumped whenever an entry point's arguments or struct rpnet_conv_desc change; the ctypes
// binding (rpnet_amd/hip.py) refuses a library of another version
extern "C" int rpnet_version(void) { return RPNET_ABI_VERSION; }
extern "C" const char* rpnet_last_error_string(void) { return rpnet::g_err; }
