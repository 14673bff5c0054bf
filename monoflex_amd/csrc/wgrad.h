// Weight-gradient launch geometry shared by train_kernels.hip (first-generation kernels, reduce pass) and wgrad_tr.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace mfx {

struct WgradGeom { int B, H, W, Ho, Wo, x_pixstride, Ck, kh, kw, stride, pad_h, pad_w, dil_w, M, K, Cout, ldy, m_per_block;
                   int oihw, Cin_out, Cout_out;      // oihw: write dW as (Cout_out, Cin_out, kh, kw), dropping padded channels
                   float* ws; int ws_ld; long ws_slab;   // ws != null: partial tiles go to ws[slab][o][k] with plain stores
                   int direct; };                    // direct: the A operand is a dense [M][K] matrix (row stride x_pixstride), no im2col addressing

}  // namespace mfx

// wgrad_tr.hip (dtype = MFX_BF16 or MFX_F16): returns 1 if it launched (partial tiles in g.ws, *nslab slabs: run wgrad_reduce_kernel), 0 to fall through
int try_conv_wgrad_tr(const void* x, const void* dy, mfx::WgradGeom& g, int dtype, void* workspace, size_t workspace_bytes, int* nslab, hipStream_t st);
// wgrad_tr.hip, 3x3 / s1 / p1 form with the input patch in LDS: same contract
int try_conv_wgrad_patch(const void* x, const void* dy, mfx::WgradGeom& g, int dtype, void* workspace, size_t workspace_bytes, int* nslab, hipStream_t st);
