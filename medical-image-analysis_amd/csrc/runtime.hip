// Library-level plumbing of libmia_hip.so: last-error record, version, device query.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void mia_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mia_last_error(void) { return g_err; }

extern "C" int mia_version(void) { return 100; }

// 0 when the current device is a gfx950 part the kernels were built for, negative otherwise.
extern "C" int mia_check_device(void) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    MIA_CHECK_ARG(e == hipSuccess, "mia_check_device: hipGetDevice failed: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    MIA_CHECK_ARG(e == hipSuccess, "mia_check_device: hipGetDeviceProperties failed: %s", hipGetErrorString(e));
    MIA_CHECK_ARG(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "mia_check_device: device is %s, library is built for gfx950",
                  prop.gcnArchName);
    return 0;
}
