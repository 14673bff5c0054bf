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
    {   // all 15 loads of a thread in flight together: clamped addresses + select instead of a branch per pixel (the loop form waited for
        // every pixel's three loads in turn: ~5 memory round trips per tile)
        constexpr int NI = (F1_IH * F1_IW + 255) / 256;
        float v[NI][3];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = min(tid + it * 256, F1_IH * F1_IW - 1);
            const int py = i / F1_IW, px = i - py * F1_IW;
            const int gy = iy0 + py, gx = ix0 + px;
            const size_t o = (size_t)min(max(gy, 0), p.H - 1) * p.W + min(max(gx, 0), p.W - 1);
            const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const float a0 = ib[o], a1 = ib[o + (size_t)p.H * p.W], a2 = ib[o + 2 * (size_t)p.H * p.W];
            v[it][0] = in ? a0 : 0.f; v[it][1] = in ? a1 : 0.f; v[it][2] = in ? a2 : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = tid + it * 256;
            const float q[8] = {v[it][0], v[it][1], v[it][2], 0.f, 0.f, 0.f, 0.f, 0.f};
            const u32x4 pk = ElemTraits<T>::pack(q);
            if (i < F1_IH * F1_IW) img_lds[i] = uint2{pk.x, pk.y};
        }
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

// ---- split precision (MFX_F16X2: fp32 image in, fp32 level1 map out, fp16 (hi, lo) operand pairs; csrc/common.h f32s_t) ----------------------
// Same tiling and phases.  A map pixel is 64 bytes in LDS -- four 16-byte slots [hi of channels 0-7 | hi 8-15 | lo 0-7 | lo 8-15], slot index
// XOR-ed with bits 2-3 of the pixel index (16 consecutive pixels, 64 bytes apart, then cover all 64 banks) -- so a lane's 8-channel MFMA operand
// is ONE ds_read_b128 per half, and the lane that owns four consecutive channels of a pixel after a GEMM writes two 8-byte pieces.  An image
// pixel is [h0 h1 h2 0 | l0 l1 l2 0] (csrc/stem.hip's split kernel).  Every k-step is the three products hi.hi + hi.lo + lo.hi.  78.5 KB of LDS:
// two workgroups per CU.  Weights: w_stem = fp16 [2][16][stem_ld] (hi rows, lo rows: ops.pack_stem), w_l0 / w_l1 = the split-chunk matrices
// [Cout][160] of ops.pack_conv (16-byte chunk = 4 elements [h h h h | l l l l]); a lane's 8 elements are two chunks, re-paired in registers.
constexpr int F1S_STEM_BYTES = F1_SPX * 64, F1S_L0_BYTES = F1_L0PX * 64;                                   // 42560 + 35904
constexpr int F1S_SMEM = F1S_STEM_BYTES + F1S_L0_BYTES;
static_assert(F1_IH * F1_IW * 16 <= F1S_L0_BYTES, "image patch must fit the level0 region");

__device__ __forceinline__ int f1s_slot(int pix, int slot) { return pix * 64 + ((slot ^ ((pix >> 2) & 3)) << 4); }
__device__ __forceinline__ f32x4 f1s_mma3(const u32x4& wh, const u32x4& wl, const u32x4& ph, const u32x4& pl, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, ph), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, pl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl), __builtin_bit_cast(f16x8, ph), acc, 0, 0, 0);
    return acc;
}
// the three products into three accumulators (independent MFMA chains; summed by the caller)
__device__ __forceinline__ void f1s_mma3x(const u32x4& wh, const u32x4& wl, const u32x4& ph, const u32x4& pl, f32x4 (&acc)[3]) {
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, ph), acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, pl), acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl), __builtin_bit_cast(f16x8, ph), acc[2], 0, 0, 0);
}
// four fp32 values (consecutive channels 4 kq ..) of map pixel `pix` -> their hi / lo halves in the pixel's record
__device__ __forceinline__ void f1s_store4(char* map, int pix, int kq, const float (&v)[4]) {
    const u32x4 o = lds_operand<f32s_t>(ElemTraits<float>::pack(v));      // [h h | l l] dwords
    *reinterpret_cast<uint2*>(map + f1s_slot(pix, kq >> 1) + (kq & 1) * 8) = uint2{o.x, o.y};
    *reinterpret_cast<uint2*>(map + f1s_slot(pix, 2 + (kq >> 1)) + (kq & 1) * 8) = uint2{o.z, o.w};
}
// a lane's 8 weights (k0 .. k0 + 7 of row n) from the split-chunk matrix -> hi operand, lo operand
__device__ __forceinline__ void f1s_wload(const float* w, int n, int k0, u32x4& h, u32x4& l) {
    const u32x4 c0 = *reinterpret_cast<const u32x4*>(w + (size_t)n * 160 + k0), c1 = *reinterpret_cast<const u32x4*>(w + (size_t)n * 160 + k0 + 4);
    h = u32x4{c0.x, c0.y, c1.x, c1.y}; l = u32x4{c0.z, c0.w, c1.z, c1.w};
}

__global__ __launch_bounds__(256, 2) void f1_fused_split_kernel(F1Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* stem_lds = smem;
    char* l0_lds = smem + F1S_STEM_BYTES;
    u32x4* img_lds = reinterpret_cast<u32x4*>(l0_lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kq = lane >> 4;
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const int tiles_x = (Wo + F1_TW - 1) / F1_TW, tiles_y = (Ho + F1_TH - 1) / F1_TH;
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y; const int b = tile / tiles_y;
    const int oy0 = ty * F1_TH, ox0 = tx * F1_TW;
    const int l0y0 = 2 * oy0 - 1, l0x0 = 2 * ox0 - 1;
    const int sy0 = l0y0 - 1, sx0 = l0x0 - 1;
    const int iy0 = sy0 - 3, ix0 = sx0 - 3;

    const uint16_t* ws = reinterpret_cast<const uint16_t*>(p.w_stem);
    const float* w0 = reinterpret_cast<const float*>(p.w_l0);
    const float* w1 = reinterpret_cast<const float*>(p.w_l1);

    const float* ib = p.img + (size_t)b * 3 * p.H * p.W;
    {
        constexpr int NI = (F1_IH * F1_IW + 255) / 256;
        float v[NI][3];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = min(tid + it * 256, F1_IH * F1_IW - 1);
            const int py = i / F1_IW, px = i - py * F1_IW;
            const int gy = iy0 + py, gx = ix0 + px;
            const size_t o = (size_t)min(max(gy, 0), p.H - 1) * p.W + min(max(gx, 0), p.W - 1);
            const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const float a0 = ib[o], a1 = ib[o + (size_t)p.H * p.W], a2 = ib[o + 2 * (size_t)p.H * p.W];
            v[it][0] = in ? a0 : 0.f; v[it][1] = in ? a1 : 0.f; v[it][2] = in ? a2 : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = tid + it * 256;
            const float q[4] = {v[it][0], v[it][1], v[it][2], 0.f};
            if (i < F1_IH * F1_IW) img_lds[i] = lds_operand<f32s_t>(ElemTraits<float>::pack(q));
        }
    }
    __syncthreads();

    // ---- stem
    {
        u32x4 wh[7], wl[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            wh[s] = *reinterpret_cast<const u32x4*>(ws + (size_t)l16 * p.stem_ld + s * 32 + kq * 8);
            wl[s] = *reinterpret_cast<const u32x4*>(ws + (size_t)(16 + l16) * p.stem_ld + s * 32 + kq * 8);
        }
        f32x4 sc, sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[r] = p.sc_stem[kq * 4 + r]; sh[r] = p.sh_stem[kq * 4 + r]; }
        const char* ibase = reinterpret_cast<const char*>(img_lds);
        // two fragments per pass and one accumulator per product: six independent MFMA chains instead of one chain of 21
        for (int f0 = wv; f0 < (F1_SPX + 15) / 16; f0 += 8) {
            int pp[2], pr[2], pc[2];
            f32x4 acc[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pp[u] = min((f0 + 4 * u) * 16 + l16, F1_SPX - 1);
                pr[u] = pp[u] / F1_SW; pc[u] = pp[u] - pr[u] * F1_SW;
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[u][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s = 0; s < 7; ++s) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const char* q = ibase + ((pr[u] + s) * F1_IW + pc[u] + 2 * kq) * 16;
                    const uint2 h0 = *reinterpret_cast<const uint2*>(q), h1 = *reinterpret_cast<const uint2*>(q + 16);
                    const uint2 l0 = *reinterpret_cast<const uint2*>(q + 8), l1 = *reinterpret_cast<const uint2*>(q + 24);
                    f1s_mma3x(wh[s], wl[s], u32x4{h0.x, h0.y, h1.x, h1.y}, u32x4{l0.x, l0.y, l1.x, l1.y}, acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int gy = sy0 + pr[u], gx = sx0 + pc[u];
                const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = in ? fmaxf((acc[u][0][r] + (acc[u][1][r] + acc[u][2][r])) * sc[r] + sh[r], 0.f) : 0.f;
                if ((f0 + 4 * u) * 16 + l16 < F1_SPX) f1s_store4(stem_lds, pp[u], kq, v);
            }
        }
    }
    __syncthreads();

    // ---- level0
    {
        u32x4 wh[5], wl[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) f1s_wload(w0, l16, s * 32 + kq * 8, wh[s], wl[s]);
        f32x4 sc, sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[r] = p.sc_l0[kq * 4 + r]; sh[r] = p.sh_l0[kq * 4 + r]; }
        for (int f0 = wv; f0 < (F1_L0PX + 15) / 16; f0 += 8) {
            int pp[2], pr[2], pc[2];
            f32x4 acc[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pp[u] = min((f0 + 4 * u) * 16 + l16, F1_L0PX - 1);
                pr[u] = pp[u] / F1_L0W; pc[u] = pp[u] - pr[u] * F1_L0W;
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[u][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const int tap = min(2 * ks + (kq >> 1), 8);
                const int th = tap / 3, tw = tap - th * 3;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int sp = (pr[u] + th) * F1_SW + pc[u] + tw;
                    const u32x4 ph = *reinterpret_cast<const u32x4*>(stem_lds + f1s_slot(sp, kq & 1));
                    const u32x4 pl = *reinterpret_cast<const u32x4*>(stem_lds + f1s_slot(sp, 2 + (kq & 1)));
                    f1s_mma3x(wh[ks], wl[ks], ph, pl, acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int gy = l0y0 + pr[u], gx = l0x0 + pc[u];
                const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = in ? fmaxf((acc[u][0][r] + (acc[u][1][r] + acc[u][2][r])) * sc[r] + sh[r], 0.f) : 0.f;
                if ((f0 + 4 * u) * 16 + l16 < F1_L0PX) f1s_store4(l0_lds, pp[u], kq, v);
            }
        }
    }
    __syncthreads();

    // ---- level1, stride 2
    {
        float* y = reinterpret_cast<float*>(p.y);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 wh[5], wl[5];
#pragma unroll
            for (int s = 0; s < 5; ++s) f1s_wload(w1, h * 16 + l16, s * 32 + kq * 8, wh[s], wl[s]);
            f32x4 sc, sh;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[r] = p.sc_l1[h * 16 + kq * 4 + r]; sh[r] = p.sh_l1[h * 16 + kq * 4 + r]; }
            f32x4 acc[2][3];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[rr][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const int tap = min(2 * ks + (kq >> 1), 8);
                const int th = tap / 3, tw = tap - th * 3;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int sp = (2 * (wv + 4 * rr) + th) * F1_L0W + 2 * l16 + tw;
                    const u32x4 ph = *reinterpret_cast<const u32x4*>(l0_lds + f1s_slot(sp, kq & 1));
                    const u32x4 pl = *reinterpret_cast<const u32x4*>(l0_lds + f1s_slot(sp, 2 + (kq & 1)));
                    f1s_mma3x(wh[ks], wl[ks], ph, pl, acc[rr]);
                }
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int oy = oy0 + wv + 4 * rr, ox = ox0 + l16;
                if (oy < Ho && ox < Wo) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf((acc[rr][0][r] + (acc[rr][1][r] + acc[rr][2][r])) * sc[r] + sh[r], 0.f);
                    *reinterpret_cast<f32x4*>(y + (((size_t)b * Ho + oy) * Wo + ox) * 32 + h * 16 + kq * 4) = v;
                }
            }
        }
    }
}

}  // namespace mfx
using namespace mfx;

// images (B,3,H,W) fp32 NCHW -> level1 map (B, H/2, W/2, 32) in `dtype` (MFX_BF16 / MFX_F16; MFX_F16X2: fp32 map, weights as described at
// f1_fused_split_kernel).  Weights in `dtype`: w_stem [16][224] in
// ops.pack_stem's super-tap order (rows of stem_kpad elements); w_l0 [16][160], w_l1 [32][160] with k = tap * 16 + c (K padded from 144 with zeros); scale / shift = the three
// folded BatchNorms, fp32.  H and W even.
extern "C" int mfx_f1_fused(const float* images, const void* w_stem, const float* sc_stem, const float* sh_stem,
                            const void* w_l0, const float* sc_l0, const float* sh_l0,
                            const void* w_l1, const float* sc_l1, const float* sh_l1,
                            void* y, int B, int H, int W, int stem_kpad, int dtype, void* stream) {
    if (!images || !w_stem || !sc_stem || !sh_stem || !w_l0 || !sc_l0 || !sh_l0 || !w_l1 || !sc_l1 || !sh_l1 || !y)
        return mfx_fail(MFX_ERR_ARG, "f1_fused: null pointer");
    if (dtype != MFX_BF16 && dtype != MFX_F16 && dtype != MFX_F16X2) return mfx_fail(MFX_ERR_UNSUPPORTED, "f1_fused: bf16 / fp16 / f16x2 only");
    if ((H & 1) || (W & 1)) return mfx_fail(MFX_ERR_UNSUPPORTED, "f1_fused: H and W must be even");
    if (stem_kpad < 224 || stem_kpad % 8) return mfx_fail(MFX_ERR_ARG, "f1_fused: w_stem rows are K_pad >= 224 elements (ops.pack_stem)");
    if (B <= 0 || H <= 0 || W <= 0) return MFX_OK;
    F1Args a;
    a.img = images; a.w_stem = w_stem; a.sc_stem = sc_stem; a.sh_stem = sh_stem; a.w_l0 = w_l0; a.sc_l0 = sc_l0; a.sh_l0 = sh_l0;
    a.w_l1 = w_l1; a.sc_l1 = sc_l1; a.sh_l1 = sh_l1; a.y = y; a.B = B; a.H = H; a.W = W; a.stem_ld = stem_kpad;
    const int tiles = B * ((H / 2 + F1_TH - 1) / F1_TH) * ((W / 2 + F1_TW - 1) / F1_TW);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MFX_F16X2) {
        static bool attr_set = false;
        if (!attr_set) {
            MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(f1_fused_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F1S_SMEM));
            attr_set = true;
        }
        hipLaunchKernelGGL(f1_fused_split_kernel, dim3(tiles), dim3(256), F1S_SMEM, st, a);
    } else if (dtype == MFX_F16) hipLaunchKernelGGL(f1_fused_kernel<half_t>, dim3(tiles), dim3(256), F1_SMEM, st, a);
    else hipLaunchKernelGGL(f1_fused_kernel<bf16_t>, dim3(tiles), dim3(256), F1_SMEM, st, a);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

MFX_RANGE_FLAG_ACCESSOR(f1_fused)      // split-precision range sentinel of this translation unit (common.h)
