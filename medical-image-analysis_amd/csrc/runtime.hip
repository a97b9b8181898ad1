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

// Launches are made from one host thread at a time (the forward on the caller's thread, the backward on autograd's
// worker while the caller waits), so a plain counter is exact there; concurrent callers may lose counts, nothing else.
unsigned long long g_mia_launches = 0;
extern "C" long long mia_launch_count(void) { return (long long)g_mia_launches; }

int mia_preprocess_init_device();   // preprocess.hip

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
    // per-device set-up that must not happen for the first time inside a HIP graph capture
    return mia_preprocess_init_device();
}
