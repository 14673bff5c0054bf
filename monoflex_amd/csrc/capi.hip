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
