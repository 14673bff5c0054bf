// The reference's native boundary for this path: `_ext.dcn_v2_forward` / `_ext.dcn_v2_backward`
// (/root/reference/model/backbone/DCNv2/src/vision.cpp:3-8, src/dcn_v2.h:9-92) with the reference's
// NCHW fp32 layouts, served by the NHWC gfx950 kernels: layout transforms in, fused DCN kernel, out.
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"
#include "fill.h"

namespace mfx {

static inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int cdivi(long a, long b) { return (int)((a + b - 1) / b); }

struct DcnExtDims {
    int B, C, H, W, Cout, kh, kw, stride, stride_w, pad_h, pad_w, dil, dil_w;
    int Ho, Wo, Cp, Coutp, K, M;
    size_t off_x, off_om, off_w, off_shift, off_y, total_fwd;
};

static DcnExtDims make_dims(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw) {
    DcnExtDims d;
    d.B = B; d.C = C; d.H = H; d.W = W; d.Cout = Cout; d.kh = kh; d.kw = kw; d.stride = sh; d.stride_w = sw; d.pad_h = ph; d.pad_w = pw; d.dil = dh; d.dil_w = dw;
    d.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;           // src/cpu/dcn_v2_cpu.cpp:59-60
    d.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
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

// dst[b][cd0 + c][p] (+)= src[b][cs0 + c][p], c < Cg: a channel slice of an NCHW tensor (deformable groups; weights as (Cout, C, kk))
__global__ void ext_slice_copy_kernel(const float* src, float* dst, int B, int Cs, int cs0, int Cd, int cd0, int Cg, int HW, int accumulate) {
    const long total = (long)B * Cg * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW), c = (int)((i / HW) % Cg), b = (int)(i / ((long)HW * Cg));
        const float v = src[((size_t)b * Cs + cs0 + c) * HW + p];
        float* o = dst + ((size_t)b * Cd + cd0 + c) * HW + p;
        *o = accumulate ? *o + v : v;
    }
}

}  // namespace mfx
using namespace mfx;

// library-internal (shared with dcn_bwd.hip)
int mfx_internal_ext_slice(const float* src, float* dst, int B, int Cs, int cs0, int Cd, int cd0, int Cg, int HW, int accumulate, void* stream) {
    const long total = (long)B * Cg * HW;
    if (total == 0) return MFX_OK;
    hipLaunchKernelGGL(ext_slice_copy_kernel, dim3((unsigned)(cdivi(total, 256) < 8192 ? cdivi(total, 256) : 8192)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       src, dst, B, Cs, cs0, Cd, cd0, Cg, HW, accumulate);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

#define EXT_GRID(total) dim3((unsigned)(cdivi((total), 256) < 8192 ? cdivi((total), 256) : 8192))

// temporaries of the deformable-group loop behind the one-group regions: contiguous copies of a group's input / weight / offset / mask slices and
// its output (forward), or its five gradients (backward).  Sized for one group of ALL C channels (an upper bound: the size query has no group count)
static size_t group_tmp_bytes(int B, int C, int H, int W, int Cout, int kk, int Ho, int Wo, int backward) {
    size_t o = 0;
    o += align256((size_t)B * C * H * W * 4) + align256((size_t)Cout * C * kk * 4) + align256((size_t)B * 2 * kk * Ho * Wo * 4) + align256((size_t)B * kk * Ho * Wo * 4);
    if (!backward) o += align256((size_t)B * Cout * Ho * Wo * 4) + align256((size_t)Cout * 4);
    else o += align256((size_t)B * C * H * W * 4) + align256((size_t)Cout * C * kk * 4) + align256((size_t)B * 2 * kk * Ho * Wo * 4) + align256((size_t)B * kk * Ho * Wo * 4) +
              align256((size_t)Cout * 4);
    return o;
}

extern "C" size_t mfx_dcn_v2_backward_workspace_bytes_(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw);  // dcn_bwd.hip

extern "C" size_t mfx_dcn_v2_workspace_bytes_g(int B, int C, int H, int W, int Cout, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                                               int dil_h, int dil_w, int deformable_group, int backward);
extern "C" size_t mfx_dcn_v2_workspace_bytes(int B, int C, int H, int W, int Cout, int kh, int kw,
                                             int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                             int backward) {
    return mfx_dcn_v2_workspace_bytes_g(B, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, /*deformable_group (any)*/ 2, backward);
}

// the same for a known group count: one deformable group (the only value the model uses) needs none of the group loop's temporaries -- about
// half of the bound above at full size; the entries check against THIS size
extern "C" size_t mfx_dcn_v2_workspace_bytes_g(int B, int C, int H, int W, int Cout, int kh, int kw,
                                               int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                               int deformable_group, int backward) {
    const DcnExtDims d = make_dims(B, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w);
    const size_t one = backward ? mfx_dcn_v2_backward_workspace_bytes_(B, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w) : d.total_fwd;
    return one + (deformable_group > 1 ? group_tmp_bytes(B, C, H, W, Cout, kh * kw, d.Ho > 0 ? d.Ho : 0, d.Wo > 0 ? d.Wo : 0, backward) : 0);
}

static int check_ext_args(int C, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg) {
    if (dg < 1 || C % dg != 0) return mfx_fail(MFX_ERR_ARG, "dcn_v2: deformable_group must divide the input channels");
    if (sh < 1 || sw < 1 || dh < 1 || dw < 1 || ph < 0 || pw < 0) return mfx_fail(MFX_ERR_ARG, "dcn_v2: bad stride / padding / dilation");
    if (kh * kw > 9) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_v2: at most 9 taps");
    return MFX_OK;
}

// one deformable group: contiguous NCHW fp32 operands
static int forward_one(const float* input, const float* weight, const float* bias, const float* offset, const float* mask, float* output,
                       const DcnExtDims& d, char* ws, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* x_nhwc = reinterpret_cast<float*>(ws + d.off_x);
    float* om = reinterpret_cast<float*>(ws + d.off_om);
    float* wp = reinterpret_cast<float*>(ws + d.off_w);
    float* shift = reinterpret_cast<float*>(ws + d.off_shift);
    float* y_nhwc = reinterpret_cast<float*>(ws + d.off_y);
    int rc = mfx_nchw_to_nhwc(input, x_nhwc, d.B, d.C, d.H, d.W, d.Cp, MFX_F32, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(pack_offmask_kernel, EXT_GRID((long)d.M * 32), dim3(256), 0, st, offset, mask, om, d.B, d.Ho * d.Wo, d.kh * d.kw);
    hipLaunchKernelGGL(pack_weight_kernel, EXT_GRID((long)d.Coutp * d.K), dim3(256), 0, st, weight, wp, d.Cout, d.C, d.kh * d.kw, d.Coutp, d.Cp);
    hipLaunchKernelGGL(pad_copy_kernel, dim3(cdivi(d.Coutp, 256)), dim3(256), 0, st, bias, shift, d.Cout, d.Coutp);
    MFX_HIP_CHECK(hipGetLastError());

    mfx_dcn_desc dd = {};
    dd.x = x_nhwc; dd.offmask = om; dd.w = wp; dd.w_frag = nullptr; dd.scale = nullptr; dd.shift = shift; dd.y = y_nhwc;
    dd.B = d.B; dd.H = d.H; dd.W = d.W; dd.C = d.Cp; dd.kh = d.kh; dd.kw = d.kw; dd.stride = d.stride; dd.pad = d.pad_h; dd.dil = d.dil;
    dd.Ho = d.Ho; dd.Wo = d.Wo; dd.Cout = d.Coutp; dd.Cout_pad = d.Coutp; dd.K_pad = d.K; dd.ldy = d.Coutp;
    dd.act = MFX_ACT_NONE; dd.dtype = MFX_F32;
    if (d.stride != d.stride_w || d.pad_h != d.pad_w || d.dil != d.dil_w) { dd.nonsquare = 1; dd.stride_w = d.stride_w; dd.pad_w = d.pad_w; dd.dil_w = d.dil_w; }
    rc = mfx_dcn_nhwc(&dd, stream);
    if (rc) return rc;
    return mfx_nhwc_to_nchw(y_nhwc, output, d.B, d.Cout, d.Ho, d.Wo, d.Coutp, MFX_F32, stream);
}

extern "C" int mfx_dcn_v2_forward(const float* input, const float* weight, const float* bias,
                                  const float* offset, const float* mask, float* output,
                                  int B, int C, int H, int W, int Cout, int kh, int kw,
                                  int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                  int deformable_group, void* workspace, size_t workspace_bytes, void* stream) {
    if (!input || !weight || !bias || !offset || !mask || !output) return mfx_fail(MFX_ERR_ARG, "dcn_v2_forward: null pointer");
    int rc = check_ext_args(C, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, deformable_group);
    if (rc) return rc;
    const int dg = deformable_group, Cg = C / dg, kk = kh * kw;
    const DcnExtDims d = make_dims(B, Cg, H, W, Cout, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w);
    if (d.Ho <= 0 || d.Wo <= 0) return mfx_fail(MFX_ERR_ARG, "dcn_v2_forward: empty output");
    if (!workspace || workspace_bytes < mfx_dcn_v2_workspace_bytes_g(B, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, dg, 0))
        return mfx_fail(MFX_ERR_WORKSPACE, "dcn_v2_forward: workspace too small (mfx_dcn_v2_workspace_bytes_g)");
    if (d.M == 0) return MFX_OK;
    char* ws = reinterpret_cast<char*>(workspace);
    if (dg == 1) return forward_one(input, weight, bias, offset, mask, output, d, ws, stream);
    // deformable groups (src/cuda/dcn_v2_im2col_cuda.cu:147-156): the layer is the sum over g of a one-group layer on channel slice g with
    // its own 2 kk offset and kk mask channels; the bias enters once
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int HW = H * W, HWo = d.Ho * d.Wo;
    char* t = ws + make_dims(B, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w).total_fwd;
    float* xg = reinterpret_cast<float*>(t); t += align256((size_t)B * C * HW * 4);
    float* wg = reinterpret_cast<float*>(t); t += align256((size_t)Cout * C * kk * 4);
    float* og = reinterpret_cast<float*>(t); t += align256((size_t)B * 2 * kk * HWo * 4);
    float* mg = reinterpret_cast<float*>(t); t += align256((size_t)B * kk * HWo * 4);
    float* yg = reinterpret_cast<float*>(t); t += align256((size_t)B * Cout * HWo * 4);
    float* zb = reinterpret_cast<float*>(t);
    MFX_HIP_CHECK(mfx::zero_async(zb, (size_t)Cout * 4, st));
    for (int g = 0; g < dg; ++g) {
        if ((rc = mfx_internal_ext_slice(input, xg, B, C, g * Cg, Cg, 0, Cg, HW, 0, stream))) return rc;
        if ((rc = mfx_internal_ext_slice(weight, wg, Cout, C, g * Cg, Cg, 0, Cg, kk, 0, stream))) return rc;
        if ((rc = mfx_internal_ext_slice(offset, og, B, dg * 2 * kk, g * 2 * kk, 2 * kk, 0, 2 * kk, HWo, 0, stream))) return rc;
        if ((rc = mfx_internal_ext_slice(mask, mg, B, dg * kk, g * kk, kk, 0, kk, HWo, 0, stream))) return rc;
        if ((rc = forward_one(xg, wg, g == 0 ? bias : zb, og, mg, g == 0 ? output : yg, d, ws, stream))) return rc;
        if (g > 0 && (rc = mfx_internal_ext_slice(yg, output, B, Cout, 0, Cout, 0, Cout, HWo, 1, stream))) return rc;
    }
    return MFX_OK;
}
