// capi.hip -- error reporting and introspection entry points of libgfnet_hip.so.
#include "common.h"

#include <cstring>

namespace gfn {

char *last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace gfn

GFN_EXPORT int gfn_abi_version(void) { return GFN_ABI_VERSION; }

GFN_EXPORT const char *gfn_last_error(void) { return gfn::last_error_buf(); }

GFN_EXPORT int gfn_device_arch(char *buf, int buflen) {
    if (!buf || buflen <= 0) return gfn::fail(GFN_ERR_INVALID_ARG, "device_arch: no buffer");
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return gfn::fail(GFN_ERR_LAUNCH, "device_arch: %s", hipGetErrorString(e));
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
    return GFN_OK;
}
