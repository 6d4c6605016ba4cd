// Error reporting / version / device helpers of the weclip_hip C-ABI library.
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void wc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* wc_last_error(void) { return g_err; }

extern "C" int wc_version(void) { return 100; }   // 0.1.0

// Number of visible HIP devices, or -1 (with wc_last_error set) when the runtime is unusable.
extern "C" int wc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        wc_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return -1;
    }
    return n;
}

// gcnArchName of device `dev` copied into buf (e.g. "gfx950:sramecc+:xnack-").
extern "C" int wc_device_arch(int dev, char* buf, int buflen) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) {
        wc_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
        return WC_ERR_HIP;
    }
    strncpy(buf, p.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
    return WC_OK;
}
