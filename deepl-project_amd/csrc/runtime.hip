// Library runtime: error string, ABI version, device zero page.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>

#include "common.h"

namespace {
thread_local char g_err[512] = "";
std::mutex g_init_mu;
void* g_zero[64] = {nullptr};  // one zero page per device ordinal
constexpr size_t ZERO_BYTES = 4096;
}  // namespace

void tv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const void* tv_zero_page() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    return g_zero[dev];
}

extern "C" const char* tv_last_error(void) { return g_err; }
extern "C" int tv_abi_version(void) { return 1; }

extern "C" int tv_init(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        tv_set_error("tv_init: no HIP device");
        return TV_ERR_INIT;
    }
    if (g_zero[dev]) return TV_OK;
    std::lock_guard<std::mutex> lk(g_init_mu);
    if (g_zero[dev]) return TV_OK;
    void* p = nullptr;
    if (hipMalloc(&p, ZERO_BYTES) != hipSuccess || hipMemset(p, 0, ZERO_BYTES) != hipSuccess) {
        tv_set_error("tv_init: cannot allocate the device zero page");
        return TV_ERR_INIT;
    }
    (void)hipDeviceSynchronize();
    g_zero[dev] = p;
    return TV_OK;
}
