// DLA stem: 7x7 / stride 1 / pad 3 convolution of the fp32 NCHW image with 16 output channels + BN + ReLU
// (reference model/backbone/dla_dcn.py:268-272), bf16 perf mode.
//
// The generic implicit-GEMM kernel needed the image repacked to a zero-padded NHWC4 buffer and then spent 190 us on a layer
// whose traffic (47 MB of fp32 image in, 126 MB of bf16 features out) is worth ~45 us.  Here a workgroup (4 waves) owns an
// 8 x 64 block of output pixels:
//   * its (8+6) x 72 input patch is read straight from the three NCHW planes (coalesced rows), converted to bf16 and
//     kept in LDS as 4-channel pixels (8 bytes; channel 3 = 0): no padded copy of the image exists;
//   * K = 7 rows x 8 columns x 4 channels = 224 in the "super-tap" order of ops.pack_stem (two adjacent pixels per 16-byte
//     lane chunk), i.e. 7 k-steps of 32; the whole 16 x 224 weight matrix is 7 MFMA B fragments = 28 VGPRs per lane,
//     loaded once;
//   * each wave computes 2 rows x 64 pixels: 8 M-fragments x 7 MFMAs, A fragments are two ds_read_b64 per k-step
//     (a pixel pair is only 8-byte aligned);
//   * epilogue: scale/shift (folded BN) + ReLU, transposed through a wave-private LDS buffer -> 16-byte stores.
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"

namespace mfx {

constexpr int kStemRows = 8, kStemCols = 64, kStemPW = 72, kStemPH = kStemRows + 6;

template <typename T>                       // bf16_t or half_t (16-bit activations: 4 x T per patch pixel)
__global__ __launch_bounds__(256) void stem_conv7x7_kernel(const float* __restrict__ img, const T* __restrict__ w, int K_pad,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          T* __restrict__ y, int B, int H, int W, int act) {
    __shared__ __attribute__((aligned(16))) uint2 patch[kStemPH * kStemPW];          // 4 x bf16 per pixel
    __shared__ float stage_all[4][16 * 20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xl = lane & 15, kq = lane >> 4;
    const int tiles_x = (W + kStemCols - 1) / kStemCols, tiles_y = (H + kStemRows - 1) / kStemRows;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y, b = tile / tiles_y;
    const int x0 = tx * kStemCols, y0 = ty * kStemRows;

    // weights: fragment s = rows n (lane&15), k = s*32 + kq*8 .. +8
    u32x4 wf[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) wf[s] = *reinterpret_cast<const u32x4*>(w + (size_t)xl * K_pad + s * 32 + kq * 8);

    // patch: pixel (py, px) = image (y0 - 3 + py, x0 - 3 + px), zero outside
    const float* ib = img + (size_t)b * 3 * H * W;
    for (int i = tid; i < kStemPH * kStemPW; i += 256) {
        const int py = i / kStemPW, px = i - py * kStemPW;
        const int gy = y0 - 3 + py, gx = x0 - 3 + px;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t o = (size_t)gy * W + gx;
            v0 = ib[o]; v1 = ib[o + (size_t)H * W]; v2 = ib[o + 2 * (size_t)H * W];
        }
        const float q[8] = {v0, v1, v2, 0.f, 0.f, 0.f, 0.f, 0.f};
        const u32x4 pk = ElemTraits<T>::pack(q);
        patch[i] = uint2{pk.x, pk.y};
    }
    __syncthreads();

    float* stage = stage_all[wave];
    const float sc = scale ? scale[xl] : 1.f, sh = shift ? shift[xl] : 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ly = wave * 2 + r;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int lx = f * 16 + xl;                        // this lane's output pixel within the tile row
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 7; ++s) {                      // kernel row s: pixel pair (lx + 2 kq, lx + 2 kq + 1) of patch row ly + s
                const uint2* p = &patch[(ly + s) * kStemPW + lx + 2 * kq];
                const uint2 a = p[0], c = p[1];
                const u32x4 af = u32x4{a.x, a.y, c.x, c.y};
                mma_chunk<T>(af, wf[s], acc);
            }
            // D: col (lane&15) = channel, row (lane>>4)*4 + q = pixel within the fragment
#pragma unroll
            for (int q = 0; q < 4; ++q) stage[(kq * 4 + q) * 20 + xl] = apply_act(acc[q] * sc + sh, act, xl);
            __builtin_amdgcn_wave_barrier();
            if (lane < 32) {
                const int px = lane >> 1, half = lane & 1;
                const int gy = y0 + ly, gx = x0 + f * 16 + px;
                if (gy < H && gx < W) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; e += 4) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(stage + px * 20 + half * 8 + e);
                        v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
                    }
                    *reinterpret_cast<u32x4*>(y + (((size_t)b * H + gy) * W + gx) * 16 + half * 8) = ElemTraits<T>::pack(v);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Split-precision form (MFX_F16X2: fp32 image in, fp32 features out, fp16 (hi, lo) operand pairs; csrc/common.h f32s_t).  Same tiling;
// a patch pixel is 16 bytes [h0 h1 h2 0 | l0 l1 l2 0], the A operand of kernel row s is the hi (lo) halves of the lane's pixel pair -- one
// ds_read2_b64 each --, the weights are the fp16 super-tap matrix twice (hi, then lo: w = [2][16][K_pad]) = 56 VGPRs, and a k-step is the
// three products hi.hi + hi.lo + lo.hi.  The generic implicit-GEMM kernel this replaces took ~370 us of the B = 8 step for a layer whose
// traffic (47 MB in, 252 MB fp32 out) is worth ~60 us.
__global__ __launch_bounds__(256) void stem_conv7x7_split_kernel(const float* __restrict__ img, const uint16_t* __restrict__ w, int K_pad,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                float* __restrict__ y, int B, int H, int W, int act) {
    __shared__ __attribute__((aligned(16))) u32x4 patch[kStemPH * kStemPW];          // [4 x hi | 4 x lo] per pixel
    __shared__ float stage_all[4][16 * 20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xl = lane & 15, kq = lane >> 4;
    const int tiles_x = (W + kStemCols - 1) / kStemCols, tiles_y = (H + kStemRows - 1) / kStemRows;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y, b = tile / tiles_y;
    const int x0 = tx * kStemCols, y0 = ty * kStemRows;

    u32x4 wh[7], wl[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        wh[s] = *reinterpret_cast<const u32x4*>(w + (size_t)xl * K_pad + s * 32 + kq * 8);
        wl[s] = *reinterpret_cast<const u32x4*>(w + (size_t)(16 + xl) * K_pad + s * 32 + kq * 8);
    }
    const float* ib = img + (size_t)b * 3 * H * W;
    for (int i = tid; i < kStemPH * kStemPW; i += 256) {
        const int py = i / kStemPW, px = i - py * kStemPW;
        const int gy = y0 - 3 + py, gx = x0 - 3 + px;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t o = (size_t)gy * W + gx;
            v0 = ib[o]; v1 = ib[o + (size_t)H * W]; v2 = ib[o + 2 * (size_t)H * W];
        }
        const float q[4] = {v0, v1, v2, 0.f};
        patch[i] = lds_operand<f32s_t>(ElemTraits<float>::pack(q));
    }
    __syncthreads();

    float* stage = stage_all[wave];
    const float sc = scale ? scale[xl] : 1.f, sh = shift ? shift[xl] : 0.f;
    const char* pb = reinterpret_cast<const char*>(patch);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ly = wave * 2 + r;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int lx = f * 16 + xl;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                const char* p = pb + ((ly + s) * kStemPW + lx + 2 * kq) * 16;
                const uint2 h0 = *reinterpret_cast<const uint2*>(p), h1 = *reinterpret_cast<const uint2*>(p + 16);
                const uint2 l0 = *reinterpret_cast<const uint2*>(p + 8), l1 = *reinterpret_cast<const uint2*>(p + 24);
                const u32x4 ah = u32x4{h0.x, h0.y, h1.x, h1.y}, al = u32x4{l0.x, l0.y, l1.x, l1.y};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, wh[s]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, wl[s]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, wh[s]), acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) stage[(kq * 4 + q) * 20 + xl] = apply_act(acc[q] * sc + sh, act, xl);
            __builtin_amdgcn_wave_barrier();
            {
                const int px = lane >> 2, q4 = lane & 3;          // 16 pixels x four 16-byte chunks of the 64-byte fp32 pixel
                const int gy = y0 + ly, gx = x0 + f * 16 + px;
                if (gy < H && gx < W)
                    *reinterpret_cast<f32x4*>(y + (((size_t)b * H + gy) * W + gx) * 16 + q4 * 4) = *reinterpret_cast<const f32x4*>(stage + px * 20 + q4 * 4);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace mfx
using namespace mfx;

extern "C" int mfx_stem_conv7x7_nchw(const float* images, const void* w, const float* scale, const float* shift, void* y,
                                     int B, int H, int W, int Cout, int K_pad, int act, int dtype, void* stream) {
    if (!images || !w || !y) return mfx_fail(MFX_ERR_ARG, "stem_conv7x7: null pointer");
    if ((dtype != MFX_BF16 && dtype != MFX_F16 && dtype != MFX_F16X2) || Cout != 16 || K_pad < 224 || K_pad % 8 != 0)
        return mfx_fail(MFX_ERR_UNSUPPORTED, "stem_conv7x7: bf16 / fp16 / f16x2, 16 output channels, super-tap weights [16][K_pad >= 224] only");
    if (B <= 0 || H <= 0 || W <= 0) return MFX_OK;
    const int tiles = B * ((H + kStemRows - 1) / kStemRows) * ((W + kStemCols - 1) / kStemCols);
    if (dtype == MFX_F16X2)                                   /* w = fp16 [2][16][K_pad]: hi halves, then lo halves; y fp32 */
        hipLaunchKernelGGL(stem_conv7x7_split_kernel, dim3(tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), images,
                           reinterpret_cast<const uint16_t*>(w), K_pad, scale, shift, reinterpret_cast<float*>(y), B, H, W, act);
    else if (dtype == MFX_F16)
        hipLaunchKernelGGL(stem_conv7x7_kernel<half_t>, dim3(tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), images,
                           reinterpret_cast<const half_t*>(w), K_pad, scale, shift, reinterpret_cast<half_t*>(y), B, H, W, act);
    else
        hipLaunchKernelGGL(stem_conv7x7_kernel<bf16_t>, dim3(tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), images,
                           reinterpret_cast<const bf16_t*>(w), K_pad, scale, shift, reinterpret_cast<bf16_t*>(y), B, H, W, act);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

MFX_RANGE_FLAG_ACCESSOR(stem)      // split-precision range sentinel of this translation unit (common.h)
