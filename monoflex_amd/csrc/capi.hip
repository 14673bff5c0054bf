// ABI version + error string plumbing of libmonoflex_hip.so.
#include "../../include/monoflex_hip.h"
#include "err.h"
#include <cstdio>
#include <cstring>

int g_opt_det = 0;

static thread_local char g_err[512] = "";

int mfx_fail(int code, const char* msg) {
    std::snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
int mfx_fail_hip(hipError_t e, const char* what) {
    std::snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return MFX_ERR_LAUNCH;
}
extern "C" const char* mfx_last_error(void) { return g_err; }
extern "C" int mfx_abi_version(void) { return MFX_ABI_VERSION; }

// split-precision range sentinel: the per-translation-unit flags of common.h's lds_operand<f32s_t>
int mfx_range_flag_conv_halo(int), mfx_range_flag_conv_kernels(int), mfx_range_flag_dcn_wave(int), mfx_range_flag_f1_fused(int), mfx_range_flag_heads(int),
    mfx_range_flag_stem(int), mfx_range_flag_dcn_lds(int), mfx_range_flag_conv_cws(int);
extern "C" int mfx_f16x2_range_check(int reset) {
    int (*const f[])(int) = {mfx_range_flag_conv_halo, mfx_range_flag_conv_kernels, mfx_range_flag_dcn_wave, mfx_range_flag_f1_fused, mfx_range_flag_heads,
                             mfx_range_flag_stem, mfx_range_flag_dcn_lds, mfx_range_flag_conv_cws};
    int any = 0;
    for (auto fn : f) {
        const int v = fn(reset);
        if (v < 0) return mfx_fail(MFX_ERR_LAUNCH, "f16x2_range_check: reading the device flag failed");
        any |= v;
    }
    return any ? 1 : 0;
}
