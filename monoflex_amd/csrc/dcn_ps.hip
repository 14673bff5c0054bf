// DCNv2 as "project, then sample" (r06), the sampling half.
//
// The modulated deformable convolution (reference: model/backbone/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:125-195 + dcn_v2_cuda.cu:139-163)
//     out[m][n] = sum_tap mask[m,tap] * sum_c W[n][tap,c] * bilinear(x[:,:,c] @ p(m,tap))
// is linear in x, and bilinear interpolation commutes with the contraction over channels:
//     out[m][n] = sum_tap mask[m,tap] * bilinear(P_tap[:,:,n] @ p(m,tap)),      P_tap = W_tap . x   (a dense 1x1 convolution C -> 9 N of the whole map).
// The projection is a plain GEMM with NO gather in it (mfx_conv2d_nhwc, M x C x 9N -- the same 9 C N multiply-adds per pixel the fused kernels spend,
// at dense-convolution efficiency instead of 0.07-0.13 of the MFMA peak), and what is left to gather is N channels per corner instead of C:
// the channel-reducing `proj` modules of DLAUp / IDAUp (512->256, 256->128, 256->64, 128->64: 8 of the 16 modules, dla_dcn.py:398-425) gather
// 2-4x fewer bytes than the im2col form.  The price is the projected map itself: 9 N values per pixel written and read back -- 18-71 MB per layer
// at B = 8, which stays inside the 256 MB Infinity Cache -- and one more rounding of the 16-bit modes' intermediates (P is stored in the
// activation type; the sums below are fp32).  Corners outside the image contribute zero exactly as in the reference (P is undefined there: weight 0).
//
// This file: out[m][n] = act(scale[n] * (sum_tap sum_corner w(m,tap,corner) * P[pos(m,tap,corner)][tap*N + n]) + shift[n]).
//   workgroup = 256 threads = PX pixels (a small 2-D block: neighbouring pixels sample neighbouring rows of P, which meet in L1) x N/8 lanes per pixel,
//   every lane owning 8 consecutive output channels: a corner is one 16-byte load per lane, N*2 contiguous bytes per pixel;
//   phase 1: the (pixel, tap) geometry -- four weights (mask folded in, zero for corners outside the image) and four element offsets -- is computed
//   once by one thread each and parked in LDS; phase 2 streams the nine taps in batches of three (12 loads in flight per lane), fp32 accumulate.
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "common.h"

namespace mfx {

template <typename T, int N, int TB = 3, int TWO = 0>
__global__ __launch_bounds__(256, 4) void dcn_sample_kernel(const T* __restrict__ P, const float* __restrict__ om, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, T* __restrict__ y, int B, int H, int W, int ldy, int act,
                                                        int tiles_x, int tiles_y) {
    constexpr int CH = N / 8;                 // lanes per pixel
    constexpr int PX = 256 / CH;              // pixels per workgroup
    constexpr int TW = TWO > 0 ? TWO : (PX >= 32 ? 8 : 4);      // block width; height = PX / TW  (32 -> 8 x 4, 16 -> 4 x 4, 8 -> 4 x 2)
    constexpr int TH = PX / TW;
    constexpr int LDP = 9 * N;                // elements per pixel row of P
    __shared__ __attribute__((aligned(16))) float gw[PX * 9][4];
    __shared__ __attribute__((aligned(16))) int go[PX * 9][4];
    const int tid = threadIdx.x;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y, b = tile / tiles_y;

    // ---- phase 1: geometry of (pixel, tap)
    for (int it = tid; it < PX * 9; it += 256) {
        const int pl = it / 9, tap = it - pl * 9;
        const int yo = ty * TH + pl / TW, xo = tx * TW + pl % TW;
        const bool ok = yo < H && xo < W;
        const float* r = om + ((size_t)(b * H + min(yo, H - 1)) * W + min(xo, W - 1)) * 32;
        const float dh = r[2 * tap], dw = r[2 * tap + 1], mk = ok ? r[18 + tap] : 0.f;
        const int th = tap / 3, tw = tap - th * 3;
        const float h = (float)(yo - 1 + th) + dh, w = (float)(xo - 1 + tw) + dw;
        const bool inside = h > -1.f && w > -1.f && h < (float)H && w < (float)W;
        const float hf = floorf(h), wf = floorf(w);
        const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
        const float m_ = inside ? mk : 0.f;
        // clamp before the int conversion: a wild offset must not overflow (the sample is outside the image then: weight 0)
        const int h0 = (int)fminf(fmaxf(hf, -2.f), 30000.f), w0 = (int)fminf(fmaxf(wf, -2.f), 30000.f);
        const bool t0 = h0 >= 0 && h0 < H, t1 = h0 + 1 >= 0 && h0 + 1 < H, l0 = w0 >= 0 && w0 < W, l1 = w0 + 1 >= 0 && w0 + 1 < W;
        const int ch0 = min(max(h0, 0), H - 1), ch1 = min(max(h0 + 1, 0), H - 1), cw0 = min(max(w0, 0), W - 1), cw1 = min(max(w0 + 1, 0), W - 1);
        const int rb = b * H;
        gw[it][0] = (t0 && l0) ? hh * hw * m_ : 0.f; go[it][0] = ((rb + ch0) * W + cw0) * LDP + tap * N;
        gw[it][1] = (t0 && l1) ? hh * lw * m_ : 0.f; go[it][1] = ((rb + ch0) * W + cw1) * LDP + tap * N;
        gw[it][2] = (t1 && l0) ? lh * hw * m_ : 0.f; go[it][2] = ((rb + ch1) * W + cw0) * LDP + tap * N;
        gw[it][3] = (t1 && l1) ? lh * lw * m_ : 0.f; go[it][3] = ((rb + ch1) * W + cw1) * LDP + tap * N;
    }
    __syncthreads();

    // ---- phase 2: lane (pixel pl, channel chunk c8)
    const int pl = tid / CH, c8 = (tid - pl * CH) * 8;
    const int yo = ty * TH + pl / TW, xo = tx * TW + pl % TW;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const T* Pc = P + c8;
    // (batches stay batches: fully unrolled, the compiler hoists all 36 loads -- 202 VGPRs, two waves per SIMD)
#pragma unroll 1
    for (int t0 = 0; t0 < 9; t0 += TB) {
        u32x4 v[TB][4]; f32x4 wq[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const int4 o = *reinterpret_cast<const int4*>(go[pl * 9 + t0 + t]);
            wq[t] = *reinterpret_cast<const f32x4*>(gw[pl * 9 + t0 + t]);
            v[t][0] = *reinterpret_cast<const u32x4*>(Pc + (uint32_t)o.x);
            v[t][1] = *reinterpret_cast<const u32x4*>(Pc + (uint32_t)o.y);
            v[t][2] = *reinterpret_cast<const u32x4*>(Pc + (uint32_t)o.z);
            v[t][3] = *reinterpret_cast<const u32x4*>(Pc + (uint32_t)o.w);
        }
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float f[8];
                ElemTraits<T>::unpack(v[t][q], f);
                const float wv = wq[t][q];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(wv, f[e], acc[e]);
            }
    }
    if (yo < H && xo < W) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = acc[e] * (scale ? scale[c8 + e] : 1.f) + (shift ? shift[c8 + e] : 0.f);
        apply_act_chunk<8>(acc, act, c8);
        *reinterpret_cast<u32x4*>(y + ((size_t)(b * H + yo) * W + xo) * ldy + c8) = ElemTraits<T>::pack(acc);
    }
}

template <typename T, int N, int TB = 3, int TWO = 0>
static int launch_dcn_sample_v(const void* P, const float* om, const float* scale, const float* shift, void* y, int B, int H, int W, int ldy, int act, hipStream_t st) {
    constexpr int CH = N / 8, PX = 256 / CH, TW = TWO > 0 ? TWO : (PX >= 32 ? 8 : 4), TH = PX / TW;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    hipLaunchKernelGGL((dcn_sample_kernel<T, N, TB, TWO>), dim3(B * tiles_y * tiles_x), dim3(256), 0, st, reinterpret_cast<const T*>(P), om, scale, shift,
                       reinterpret_cast<T*>(y), B, H, W, ldy, act, tiles_x, tiles_y);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}
// (r06 sweep, tools/probes/dcn_sample_sweep.py, 128 -> 64 @ 48 x 160: 22-24 us whatever the offsets' spread (0 .. 6 px), the block shape (8 x 4, 16 x 2, 4 x 8,
// 32 x 1) or the batch (1 or 3 taps): 283 MB of 16-byte-per-lane row gathers in 22.5 us = 12.6 TB/s -- the same rate the fused gather kernels reach
// (566 MB in 45 us).  That is this chip's practical texture-path rate for scattered 128-byte rows, and the bound of every gather-based DCN form.)
template <typename T, int N>
static int launch_dcn_sample(const void* P, const float* om, const float* scale, const float* shift, void* y, int B, int H, int W, int ldy, int act, hipStream_t st) {
    return launch_dcn_sample_v<T, N>(P, om, scale, shift, y, B, H, W, ldy, act, st);
}

template <typename T>
static int dispatch_dcn_sample(int N, const void* P, const float* om, const float* scale, const float* shift, void* y, int B, int H, int W, int ldy, int act, hipStream_t st) {
    switch (N) {
        case 64: return launch_dcn_sample<T, 64>(P, om, scale, shift, y, B, H, W, ldy, act, st);
        case 128: return launch_dcn_sample<T, 128>(P, om, scale, shift, y, B, H, W, ldy, act, st);
        case 256: return launch_dcn_sample<T, 256>(P, om, scale, shift, y, B, H, W, ldy, act, st);
        default: return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_sample: output channels must be 64, 128 or 256");
    }
}

}  // namespace mfx
using namespace mfx;

extern "C" int mfx_dcn_sample_nhwc(const void* P, const float* offmask, const float* scale, const float* shift, void* y,
                                   int B, int H, int W, int N, int ldy, int act, int dtype, void* stream) {
    if (!P || !offmask || !y) return mfx_fail(MFX_ERR_ARG, "dcn_sample: null pointer");
    if (B < 0 || H < 1 || W < 1 || ldy < N) return mfx_fail(MFX_ERR_ARG, "dcn_sample: bad shape");
    if ((size_t)B * H * W * 9 * N >= ((size_t)1 << 31)) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_sample: projected map of 2^31 elements or more (32-bit element offsets)");
    if (B == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MFX_BF16) return dispatch_dcn_sample<bf16_t>(N, P, offmask, scale, shift, y, B, H, W, ldy, act, st);
    if (dtype == MFX_F16) return dispatch_dcn_sample<half_t>(N, P, offmask, scale, shift, y, B, H, W, ldy, act, st);
    return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_sample: 16-bit maps only (MFX_BF16 / MFX_F16)");
}
