// Error reporting shared by all translation units of libmonoflex_hip.so.
#pragma once
#include <hip/hip_runtime.h>

int mfx_fail(int code, const char* msg);          // records msg (thread-local) and returns code
int mfx_fail_hip(hipError_t e, const char* what); // records the HIP error string, returns MFX_ERR_LAUNCH

#define MFX_HIP_CHECK(expr)                                            \
    do {                                                               \
        hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) return mfx_fail_hip(_e, #expr);          \
    } while (0)
