// DLA F1 in ONE kernel (inference, 16-bit maps): 7x7 stem (3 -> 16) + BN + ReLU -> level0 3x3 (16 -> 16) + BN + ReLU -> level1 3x3 / stride 2
// (16 -> 32) + BN + ReLU  (reference model/backbone/dla_dcn.py:268-276, 312-331).
//
// As three launches these layers move 47 MB (image) + 126 MB x 2 (stem map out / in) + 126 MB x 2 (level0 map out / in) + 63 MB = 614 MB at
// B = 8 for 2.85 GMAC per image of work: HBM-bound, 201 us.  Nothing but level1's half-resolution map is read downstream (DLAUp starts at level
// 2), so here a workgroup owns an 8 x 16 block of LEVEL1 outputs and keeps both full-resolution 16-channel maps of its footprint in LDS:
//
//   image patch   25 x 42 px (fp32 NCHW planes -> 4 x 16-bit per pixel, zero outside the image)                    8.4 KB  (aliased below)
//   stem map      19 x 35 px x 16 ch      <- 42 pixel fragments x 7 MFMAs (kernel rows of 8 "super-tap" columns x 4 channels, pack_stem)   21.3 KB
//   level0 map    17 x 33 px x 16 ch      <- 36 fragments x 5 MFMAs (two taps x 16 channels per k-step, tap 9 = zero weights)            18.0 KB
//   level1 tile    8 x 16 px x 32 ch      <-  8 fragments x 2 x 5 MFMAs, stride 2                                  -> global, 8-byte stores
//
// Halo recompute: 1.30x on the stem, 1.10x on level0 (the r02 attempt with a 5-pixel halo around 8 x 16 FULL-resolution tiles paid 1.5x at
// two workgroups per CU and lost to the three launches); 39 KB of LDS = four workgroups per CU.  Every GEMM runs "transposed" (weights are
// the MFMA A operand, pixels the B operand), so a lane ends up with four consecutive channels of one pixel: BN + ReLU in registers, one
// 8-byte LDS / global write.  Maps positions outside the image are stored as ZERO (they are the next conv's zero padding, not a
// convolution of padded input).  Pixels are addressed per lane (fragment f covers flat pixels 16 f .. 16 f + 15 of the region), so the
// region widths need not be multiples of 16.
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"

namespace mfx {

constexpr int F1_TH = 8, F1_TW = 16;                                     // level1 tile
constexpr int F1_L0H = 2 * F1_TH + 1, F1_L0W = 2 * F1_TW + 1;            // 17 x 33 level0 pixels
constexpr int F1_SH = F1_L0H + 2, F1_SW = F1_L0W + 2;                    // 19 x 35 stem pixels
constexpr int F1_IH = F1_SH + 6, F1_IW = F1_SW + 7;                      // 25 x 42 image pixels (one extra column: pixel pairs)
constexpr int F1_SPX = F1_SH * F1_SW, F1_L0PX = F1_L0H * F1_L0W;         // 665, 561
constexpr int F1_STEM_BYTES = ((F1_SPX * 32 + 63) / 64) * 64;            // 21312
constexpr int F1_L0_BYTES = ((F1_L0PX * 32 + 63) / 64) * 64;             // 17984
constexpr int F1_SMEM = F1_STEM_BYTES + F1_L0_BYTES;                     // image patch (8.4 KB) lives in the level0 region until the stem is done

struct F1Args {
    const float* img;                         // (B,3,H,W) fp32 NCHW
    const void* w_stem; const float* sc_stem; const float* sh_stem;       // [16][224] super-tap order (ops.pack_stem)
    const void* w_l0;   const float* sc_l0;   const float* sh_l0;         // [16][160]: k = tap * 16 + c, tap 9 zero
    const void* w_l1;   const float* sc_l1;   const float* sh_l1;         // [32][160]
    void* y;                                  // (B, H/2, W/2, 32)
    int B, H, W, stem_ld;                     // row length of w_stem in elements (K_pad >= 224)
};

template <typename T> __device__ __forceinline__ f32x4 f1_mfma(const u32x4& a, const u32x4& b, f32x4 acc) {
    mma_chunk<T>(a, b, acc);
    return acc;
}

template <typename T>
__global__ __launch_bounds__(256, 4) void f1_fused_kernel(F1Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* stem_lds = smem;                                                // [F1_SPX][16 ch] 32 B per pixel
    char* l0_lds = smem + F1_STEM_BYTES;                                  // [F1_L0PX][16 ch]
    uint2* img_lds = reinterpret_cast<uint2*>(l0_lds);                    // [F1_IH][F1_IW] 4 x 16-bit per pixel (dead once the stem map exists)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const int tiles_x = (Wo + F1_TW - 1) / F1_TW, tiles_y = (Ho + F1_TH - 1) / F1_TH;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y; const int b = tile / tiles_y;
    const int oy0 = ty * F1_TH, ox0 = tx * F1_TW;                         // level1 tile origin
    const int l0y0 = 2 * oy0 - 1, l0x0 = 2 * ox0 - 1;                     // level0 region origin (image coordinates)
    const int sy0 = l0y0 - 1, sx0 = l0x0 - 1;                             // stem region origin
    const int iy0 = sy0 - 3, ix0 = sx0 - 3;                               // image patch origin

    // ---- weights: A operands, row = output channel l16, k = 8 consecutive values at k-group kq
    const T* ws = reinterpret_cast<const T*>(p.w_stem);
    const T* w0 = reinterpret_cast<const T*>(p.w_l0);
    const T* w1 = reinterpret_cast<const T*>(p.w_l1);
    u32x4 wst[7], wl0[5];
#pragma unroll
    for (int s = 0; s < 7; ++s) wst[s] = *reinterpret_cast<const u32x4*>(ws + (size_t)l16 * p.stem_ld + s * 32 + kq * 8);
#pragma unroll
    for (int s = 0; s < 5; ++s) wl0[s] = *reinterpret_cast<const u32x4*>(w0 + (size_t)l16 * 160 + s * 32 + kq * 8);

    // ---- image patch: three fp32 planes -> [c0 c1 c2 0] in the map's 16-bit type, zero outside the image
    const float* ib = p.img + (size_t)b * 3 * p.H * p.W;
    for (int i = tid; i < F1_IH * F1_IW; i += 256) {
        const int py = i / F1_IW, px = i - py * F1_IW;
        const int gy = iy0 + py, gx = ix0 + px;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
            const size_t o = (size_t)gy * p.W + gx;
            v0 = ib[o]; v1 = ib[o + (size_t)p.H * p.W]; v2 = ib[o + 2 * (size_t)p.H * p.W];
        }
        const float q[8] = {v0, v1, v2, 0.f, 0.f, 0.f, 0.f, 0.f};
        const u32x4 pk = ElemTraits<T>::pack(q);
        img_lds[i] = uint2{pk.x, pk.y};
    }
    __syncthreads();

    // ---- stem: 42 fragments of 16 flat pixels; output channel 4 kq + r of pixel l16 per lane
    {
        f32x4 sc, sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[r] = p.sc_stem[kq * 4 + r]; sh[r] = p.sh_stem[kq * 4 + r]; }
        for (int f = wv; f < (F1_SPX + 15) / 16; f += 4) {
            const int pp = min(f * 16 + l16, F1_SPX - 1);
            const int pr = pp / F1_SW, pc = pp - pr * F1_SW;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 7; ++s) {                             // kernel row s: pixel pair (pc + 2 kq, pc + 2 kq + 1) of image-patch row pr + s
                const uint2* q = &img_lds[(pr + s) * F1_IW + pc + 2 * kq];
                const uint2 a = q[0], c = q[1];
                acc = f1_mfma<T>(wst[s], u32x4{a.x, a.y, c.x, c.y}, acc);
            }
            const int gy = sy0 + pr, gx = sx0 + pc;
            const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = in ? fmaxf(acc[r] * sc[r] + sh[r], 0.f) : 0.f;
            if (f * 16 + l16 < F1_SPX)
                *reinterpret_cast<uint2*>(stem_lds + pp * 32 + kq * 8) = uint2{ElemTraits<T>::pack2(v[0], v[1]), ElemTraits<T>::pack2(v[2], v[3])};
        }
    }
    __syncthreads();                                                      // stem map complete; the image patch is dead

    // ---- level0: 36 fragments; k-step ks = taps 2 ks, 2 ks + 1 x 16 channels: lane k-group kq -> tap 2 ks + (kq >> 1), channels 8 (kq & 1) ..
    {
        f32x4 sc, sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[r] = p.sc_l0[kq * 4 + r]; sh[r] = p.sh_l0[kq * 4 + r]; }
        for (int f = wv; f < (F1_L0PX + 15) / 16; f += 4) {
            const int pp = min(f * 16 + l16, F1_L0PX - 1);
            const int pr = pp / F1_L0W, pc = pp - pr * F1_L0W;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const int tap = min(2 * ks + (kq >> 1), 8);                // (tap 9: zero weights; keep the address inside the map)
                const int th = tap / 3, tw = tap - th * 3;
                const u32x4 bfr = *reinterpret_cast<const u32x4*>(stem_lds + ((pr + th) * F1_SW + pc + tw) * 32 + (kq & 1) * 16);
                acc = f1_mfma<T>(wl0[ks], bfr, acc);
            }
            const int gy = l0y0 + pr, gx = l0x0 + pc;
            const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = in ? fmaxf(acc[r] * sc[r] + sh[r], 0.f) : 0.f;
            if (f * 16 + l16 < F1_L0PX)
                *reinterpret_cast<uint2*>(l0_lds + pp * 32 + kq * 8) = uint2{ElemTraits<T>::pack2(v[0], v[1]), ElemTraits<T>::pack2(v[2], v[3])};
        }
    }
    __syncthreads();                                                      // level0 map complete

    // ---- level1, stride 2: 8 fragments (tile rows) x 2 channel halves; wave wv takes rows wv and wv + 4
    {
        u32x4 wl1[2][5];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 5; ++s) wl1[h][s] = *reinterpret_cast<const u32x4*>(w1 + (size_t)(h * 16 + l16) * 160 + s * 32 + kq * 8);
        T* y = reinterpret_cast<T*>(p.y);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = wv + 4 * rr;                                  // level1 tile row; pixel column l16
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const int tap = min(2 * ks + (kq >> 1), 8);
                const int th = tap / 3, tw = tap - th * 3;
                const u32x4 bfr = *reinterpret_cast<const u32x4*>(l0_lds + ((2 * row + th) * F1_L0W + 2 * l16 + tw) * 32 + (kq & 1) * 16);
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[h] = f1_mfma<T>(wl1[h][ks], bfr, acc[h]);
            }
            const int oy = oy0 + row, ox = ox0 + l16;
            if (oy < Ho && ox < Wo) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = h * 16 + kq * 4 + r;
                        v[r] = fmaxf(acc[h][r] * p.sc_l1[c] + p.sh_l1[c], 0.f);
                    }
                    *reinterpret_cast<uint2*>(y + (((size_t)b * Ho + oy) * Wo + ox) * 32 + h * 16 + kq * 4) =
                        uint2{ElemTraits<T>::pack2(v[0], v[1]), ElemTraits<T>::pack2(v[2], v[3])};
                }
            }
        }
    }
}

}  // namespace mfx
using namespace mfx;

// images (B,3,H,W) fp32 NCHW -> level1 map (B, H/2, W/2, 32) in `dtype` (MFX_BF16 / MFX_F16).  Weights in `dtype`: w_stem [16][224] in
// ops.pack_stem's super-tap order (rows of stem_kpad elements); w_l0 [16][160], w_l1 [32][160] with k = tap * 16 + c (K padded from 144 with zeros); scale / shift = the three
// folded BatchNorms, fp32.  H and W even.
extern "C" int mfx_f1_fused(const float* images, const void* w_stem, const float* sc_stem, const float* sh_stem,
                            const void* w_l0, const float* sc_l0, const float* sh_l0,
                            const void* w_l1, const float* sc_l1, const float* sh_l1,
                            void* y, int B, int H, int W, int stem_kpad, int dtype, void* stream) {
    if (!images || !w_stem || !sc_stem || !sh_stem || !w_l0 || !sc_l0 || !sh_l0 || !w_l1 || !sc_l1 || !sh_l1 || !y)
        return mfx_fail(MFX_ERR_ARG, "f1_fused: null pointer");
    if (dtype != MFX_BF16 && dtype != MFX_F16) return mfx_fail(MFX_ERR_UNSUPPORTED, "f1_fused: bf16 / fp16 maps only");
    if ((H & 1) || (W & 1)) return mfx_fail(MFX_ERR_UNSUPPORTED, "f1_fused: H and W must be even");
    if (stem_kpad < 224 || stem_kpad % 8) return mfx_fail(MFX_ERR_ARG, "f1_fused: w_stem rows are K_pad >= 224 elements (ops.pack_stem)");
    if (B <= 0 || H <= 0 || W <= 0) return MFX_OK;
    F1Args a;
    a.img = images; a.w_stem = w_stem; a.sc_stem = sc_stem; a.sh_stem = sh_stem; a.w_l0 = w_l0; a.sc_l0 = sc_l0; a.sh_l0 = sh_l0;
    a.w_l1 = w_l1; a.sc_l1 = sc_l1; a.sh_l1 = sh_l1; a.y = y; a.B = B; a.H = H; a.W = W; a.stem_ld = stem_kpad;
    const int tiles = B * ((H / 2 + F1_TH - 1) / F1_TH) * ((W / 2 + F1_TW - 1) / F1_TW);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MFX_F16) hipLaunchKernelGGL(f1_fused_kernel<half_t>, dim3(tiles), dim3(256), F1_SMEM, st, a);
    else hipLaunchKernelGGL(f1_fused_kernel<bf16_t>, dim3(tiles), dim3(256), F1_SMEM, st, a);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}
