// The reference's native boundary for this path: `_ext.dcn_v2_forward` / `_ext.dcn_v2_backward`
// (/root/reference/model/backbone/DCNv2/src/vision.cpp:3-8, src/dcn_v2.h:9-92) with the reference's
// NCHW fp32 layouts, served by the NHWC gfx950 kernels: layout transforms in, fused DCN kernel, out.
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"

namespace mfx {

static inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int cdivi(long a, long b) { return (int)((a + b - 1) / b); }

struct DcnExtDims {
    int B, C, H, W, Cout, kh, kw, stride, pad_h, pad_w, dil;
    int Ho, Wo, Cp, Coutp, K, M;
    size_t off_x, off_om, off_w, off_shift, off_y, total_fwd;
};

static DcnExtDims make_dims(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int ph, int pw, int dh) {
    DcnExtDims d;
    d.B = B; d.C = C; d.H = H; d.W = W; d.Cout = Cout; d.kh = kh; d.kw = kw; d.stride = sh; d.pad_h = ph; d.pad_w = pw; d.dil = dh;
    d.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;           // src/cpu/dcn_v2_cpu.cpp:59-60
    d.Wo = (W + 2 * pw - (dh * (kw - 1) + 1)) / sh + 1;
    d.Cp = next_pow2(C < 16 ? 16 : C);
    d.Coutp = ((Cout + 63) / 64) * 64;
    d.K = kh * kw * d.Cp;
    d.M = B * d.Ho * d.Wo;
    size_t o = 0;
    d.off_x = o;     o += align256((size_t)B * H * W * d.Cp * 4);
    d.off_om = o;    o += align256((size_t)d.M * 32 * 4);
    d.off_w = o;     o += align256((size_t)d.Coutp * d.K * 4);
    d.off_shift = o; o += align256((size_t)d.Coutp * 4);
    d.off_y = o;     o += align256((size_t)d.M * d.Coutp * 4);
    d.total_fwd = o;
    return d;
}

// offset (B,2*kk,Ho,Wo) + mask (B,kk,Ho,Wo) -> om [M][32]: ch 2k = dh, 2k+1 = dw, 18+k = mask
__global__ void pack_offmask_kernel(const float* offset, const float* mask, float* om, int B, int HW, int kk) {
    const long total = (long)B * HW * 32;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i & 31);
        const long m = i >> 5;
        const int b = (int)(m / HW), p = (int)(m - (long)b * HW);
        float v = 0.f;
        if (ch < 18) { if (ch < 2 * kk) v = offset[((size_t)b * 2 * kk + ch) * HW + p]; }
        else if (ch < 27) { if (ch - 18 < kk) v = mask[((size_t)b * kk + (ch - 18)) * HW + p]; }
        om[i] = v;
    }
}

// weight (Cout,C,kh,kw) -> packed [Coutp][K], k = tap*Cp + c, zero padded
__global__ void pack_weight_kernel(const float* w, float* wp, int Cout, int C, int kk, int Coutp, int Cp) {
    const long total = (long)Coutp * kk * Cp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        const int tap = (int)((i / Cp) % kk);
        const int o = (int)(i / ((long)Cp * kk));
        wp[i] = (o < Cout && c < C) ? w[((size_t)o * C + c) * kk + tap] : 0.f;
    }
}

__global__ void pad_copy_kernel(const float* src, float* dst, int n, int npad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) dst[i] = i < n ? src[i] : 0.f;
}

}  // namespace mfx
using namespace mfx;

extern "C" size_t mfx_dcn_v2_backward_workspace_bytes_(int B, int C, int H, int W, int Cout, int kh, int kw, int s, int p, int d);  // dcn_bwd.hip

#define EXT_GRID(total) dim3((unsigned)(cdivi((total), 256) < 8192 ? cdivi((total), 256) : 8192))

extern "C" size_t mfx_dcn_v2_workspace_bytes(int B, int C, int H, int W, int Cout, int kh, int kw,
                                             int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                             int backward) {
    (void)stride_w; (void)dil_w;
    if (backward) return mfx_dcn_v2_backward_workspace_bytes_(B, C, H, W, Cout, kh, kw, stride_h, pad_h, dil_h);
    return make_dims(B, C, H, W, Cout, kh, kw, stride_h, pad_h, pad_w, dil_h).total_fwd;
}

static int check_ext_args(int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg) {
    if (dg != 1) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_v2: deformable_group must be 1 (MonoFlex uses 1, dla_dcn.py:391)");
    if (sh != sw || dh != dw || ph != pw) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_v2: stride/pad/dilation must be square");
    if (kh * kw > 9) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_v2: at most 9 taps");
    return MFX_OK;
}

extern "C" int mfx_dcn_v2_forward(const float* input, const float* weight, const float* bias,
                                  const float* offset, const float* mask, float* output,
                                  int B, int C, int H, int W, int Cout, int kh, int kw,
                                  int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                  int deformable_group, void* workspace, size_t workspace_bytes, void* stream) {
    if (!input || !weight || !bias || !offset || !mask || !output) return mfx_fail(MFX_ERR_ARG, "dcn_v2_forward: null pointer");
    int rc = check_ext_args(kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, deformable_group);
    if (rc) return rc;
    const DcnExtDims d = make_dims(B, C, H, W, Cout, kh, kw, stride_h, pad_h, pad_w, dil_h);
    if (d.Ho <= 0 || d.Wo <= 0) return mfx_fail(MFX_ERR_ARG, "dcn_v2_forward: empty output");
    if (!workspace || workspace_bytes < d.total_fwd) return mfx_fail(MFX_ERR_WORKSPACE, "dcn_v2_forward: workspace too small");
    if (d.M == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(workspace);
    float* x_nhwc = reinterpret_cast<float*>(ws + d.off_x);
    float* om = reinterpret_cast<float*>(ws + d.off_om);
    float* wp = reinterpret_cast<float*>(ws + d.off_w);
    float* shift = reinterpret_cast<float*>(ws + d.off_shift);
    float* y_nhwc = reinterpret_cast<float*>(ws + d.off_y);

    rc = mfx_nchw_to_nhwc(input, x_nhwc, B, C, H, W, d.Cp, MFX_F32, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(pack_offmask_kernel, EXT_GRID((long)d.M * 32), dim3(256), 0, st, offset, mask, om, B, d.Ho * d.Wo, kh * kw);
    hipLaunchKernelGGL(pack_weight_kernel, EXT_GRID((long)d.Coutp * d.K), dim3(256), 0, st, weight, wp, Cout, C, kh * kw, d.Coutp, d.Cp);
    hipLaunchKernelGGL(pad_copy_kernel, dim3(cdivi(d.Coutp, 256)), dim3(256), 0, st, bias, shift, Cout, d.Coutp);
    MFX_HIP_CHECK(hipGetLastError());

    mfx_dcn_desc dd = {};
    dd.x = x_nhwc; dd.offmask = om; dd.w = wp; dd.w_frag = nullptr; dd.scale = nullptr; dd.shift = shift; dd.y = y_nhwc;
    dd.B = B; dd.H = H; dd.W = W; dd.C = d.Cp; dd.kh = kh; dd.kw = kw; dd.stride = stride_h; dd.pad = pad_h; dd.dil = dil_h;
    dd.Ho = d.Ho; dd.Wo = d.Wo; dd.Cout = d.Coutp; dd.Cout_pad = d.Coutp; dd.K_pad = d.K; dd.ldy = d.Coutp;
    dd.act = MFX_ACT_NONE; dd.dtype = MFX_F32;
    rc = mfx_dcn_nhwc(&dd, stream);
    if (rc) return rc;
    return mfx_nhwc_to_nchw(y_nhwc, output, B, Cout, d.Ho, d.Wo, d.Coutp, MFX_F32, stream);
}
