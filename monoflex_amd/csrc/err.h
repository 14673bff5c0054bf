// Error reporting shared by all translation units of libmonoflex_hip.so.
#pragma once
#include <hip/hip_runtime.h>

int mfx_fail(int code, const char* msg);          // records msg (thread-local) and returns code
int mfx_fail_hip(hipError_t e, const char* what); // records the HIP error string, returns MFX_ERR_LAUNCH

// option "deterministic" (capi.hip): every floating-point reduction of the training path runs in a fixed order (single-writer
// partial sums, one pixel slab per weight-gradient tile, 64-bit fixed-point accumulation of the DCN input gradient) so that two
// runs -- eager or replayed from a hipGraph -- produce bit-identical results.  Slower; the default (0) keeps the atomics.
extern int g_opt_det;

#define MFX_HIP_CHECK(expr)                                            \
    do {                                                               \
        hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) return mfx_fail_hip(_e, #expr);          \
    } while (0)
