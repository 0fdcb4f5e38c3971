// Error plumbing and device queries for libsegnb_hip.so.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void segnb_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int segnb_num_cus() {
    static int cus = 0;
    if (cus > 0) return cus;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return cus;
}

extern "C" const char* segnb_last_error(void) { return g_err; }
extern "C" int segnb_version(void) { return 1; }
extern "C" int segnb_device_cus(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        segnb_set_error("segnb_device_cus: no HIP device");
        return -1;
    }
    return segnb_num_cus();
}
