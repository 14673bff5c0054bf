// Fused detection heads (reference model/head/detector_predictor.py:47-96,125-134).
//
// One workgroup = 64 output pixels x one branch:
//   GEMM1  trunk[64][256] = im2col3x3(feature)[64][576] x W1_b[256][576]^T      (MFMA, shared main loop)
//   epi1   BN (folded) + leaky_relu(0.01) -> element type T, kept in LDS only
//   GEMM2  out[64][<=32] = trunk[64][256] x W2_b[32][256]^T + bias              (MFMA from LDS)
// The nine 256-channel trunk maps (71 M elements per image in the reference) never touch HBM;
// only 3 + 50 fp32 channels per pixel are written.
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "igemm.h"

namespace mfx {

struct HeadGeom { int H, W, M, K_pad, nk, nbranch, ld_out, planar_c; float* planar; };
struct HeadTabs { int ch_off[16]; int c_out[16]; };

template <typename T> struct HeadSmem {
    static constexpr int BM = 64, BN = 256;
    static constexpr int trunk_stride = BN * (int)sizeof(T) + 16;         // bytes; == 16 mod 256 -> conflict-free b128 reads
    static constexpr int main_bytes = TileSmem<BM, BN, 4>::mainloop_bytes;   // 51200
    static constexpr int trunk_bytes = BM * trunk_stride;
    static constexpr int region0 = main_bytes > trunk_bytes ? main_bytes : trunk_bytes;   // main loop stages, then trunk
    static constexpr int w2_bytes = 32 * trunk_stride;
    static constexpr int bytes = region0 + w2_bytes;
};

// minimal conv loader for the 3x3/s1/p1, Cin=64 feature (same scheme as ConvALoader, fixed geometry)
template <typename T> struct HeadALoader {
    static constexpr int ELEMS = ElemTraits<T>::ELEMS;
    const T* x; int H, W, c, r0, oh, ow, pix0; bool ok;
    u32x4 reg;
    __device__ __forceinline__ void init(const T* x_, int H_, int W_, int M, int m0, int tid) {
        x = x_; H = H_; W = W_; c = tid & 3; r0 = tid >> 2;
        const int m = m0 + r0;
        ok = m < M;
        const int pm = ok ? m : 0, hw = H * W, b = pm / hw, rem = pm - b * hw;
        oh = rem / W; ow = rem - oh * W; pix0 = b * hw;
    }
    __device__ __forceinline__ void load(int kiter) {
        const int e = kiter * (4 * ELEMS) + c * ELEMS;
        const int tap = e >> 6, ci = e & 63;
        const int th = (tap * 21846) >> 16, tw = tap - th * 3;
        const int ih = oh - 1 + th, iw = ow - 1 + tw;
        const bool v = ok && tap < 9 && ih >= 0 && ih < H && iw >= 0 && iw < W;
        u32x4 z = {0u, 0u, 0u, 0u};
        if (v) z = *reinterpret_cast<const u32x4*>(x + (size_t)(pix0 + ih * W + iw) * 64 + ci);
        reg = z;
    }
    __device__ __forceinline__ void store(char* As) const { *reinterpret_cast<u32x4*>(As + r0 * RowGeom<4>::bytes + c * 16) = reg; }
};

template <typename T>
__global__ __launch_bounds__(256) void heads_fused_kernel(const T* x, const T* w1, const float* scale1, const float* shift1,
                                                               const T* w2, const float* bias2, float* out, HeadGeom g, HeadTabs tabs) {
    constexpr int BM = 64, BN = 256, WM = 1, WN = 4;
    constexpr int TS = HeadSmem<T>::trunk_stride;
    constexpr int ELEMS = ElemTraits<T>::ELEMS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* trunk = smem;                                  // aliases the main-loop stages (used after them)
    char* w2s = smem + HeadSmem<T>::region0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int br = tile % g.nbranch, tm = tile / g.nbranch;     // branch fastest: 9 neighbours share the A tile in L2
    const int m0 = tm * BM;

    // stage this branch's 1x1 weights [32][256] (read after several barriers)
    {
        constexpr int CPR = BN / ELEMS;                  // 16-byte chunks per row
        const T* src = w2 + (size_t)br * 32 * BN;
        for (int i = tid; i < 32 * CPR; i += 256) {
            const int r = i / CPR, cc = i - r * CPR;
            *reinterpret_cast<u32x4*>(w2s + r * TS + cc * 16) = *reinterpret_cast<const u32x4*>(src + (size_t)r * BN + cc * ELEMS);
        }
    }

    HeadALoader<T> al; al.init(x, g.H, g.W, g.M, m0, tid);
    WeightLoader<T, BN, 256, 4> bl; bl.init(w1, br * BN, g.K_pad, tid);
    f32x4 acc[4][4];
    gemm_mainloop<T, BM, BN, WM, WN, 4>(al, bl, g.nk, smem, acc);

    // epilogue 1: folded BN + leaky -> T -> LDS trunk tile [64][256] (wave `wave` owns columns wave*64..+64)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = wave * 64 + j * 16 + (lane & 15);
        const float sc = scale1[br * BN + n], sh = shift1[br * BN + n];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r] * sc + sh;
                v = v > 0.f ? v : 0.01f * v;
                ElemTraits<T>::store(reinterpret_cast<T*>(trunk + (m + r) * TS) + n, v);
            }
        }
    }
    __syncthreads();

    // GEMM2: wave w -> rows w*16..+16, all 32 output columns, K = 256
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const char* ap = trunk + (wave * 16 + (lane & 15)) * TS + (lane >> 4) * 16;
    const char* bp = w2s + (lane & 15) * TS + (lane >> 4) * 16;
#pragma unroll
    for (int kb = 0; kb < BN * (int)sizeof(T); kb += 64) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(ap + kb);
        const u32x4 b0 = *reinterpret_cast<const u32x4*>(bp + kb);
        const u32x4 b1 = *reinterpret_cast<const u32x4*>(bp + 16 * TS + kb);
        mma_chunk<T>(a, b0, o[0]);
        mma_chunk<T>(a, b1, o[1]);
    }
    const int cn = tabs.c_out[br], co = tabs.ch_off[br];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = j * 16 + (lane & 15);
        if (n < cn) {
            const float bias = bias2[br * 32 + n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wave * 16 + (lane >> 4) * 4 + r;
                if (m < g.M) {
                    out[(size_t)m * g.ld_out + co + n] = o[j][r] + bias;
                    if (br == 0 && g.planar && n < g.planar_c) {      // class-planar copy for the top-K kernel
                        const int hw = g.H * g.W, bi = m / hw;
                        g.planar[((size_t)bi * g.planar_c + n) * hw + (m - bi * hw)] = o[j][r] + bias;
                    }
                }
            }
        }
    }
}

template <typename T> static int launch_heads(const mfx_heads_desc* d, hipStream_t st) {
    HeadGeom g; g.H = d->H; g.W = d->W; g.M = d->B * d->H * d->W; g.K_pad = d->K_pad;
    g.nk = d->K_pad / (4 * ElemTraits<T>::ELEMS); g.nbranch = d->nbranch; g.ld_out = d->ld_out;
    g.planar = d->planar; g.planar_c = d->planar_c;
    HeadTabs t;
    for (int i = 0; i < 16; ++i) { t.ch_off[i] = d->ch_off[i]; t.c_out[i] = d->c_out[i]; }
    const int tiles = ((g.M + 63) / 64) * d->nbranch;
    auto k = heads_fused_kernel<T>;
    constexpr int smem = HeadSmem<T>::bytes;
    static bool attr_set = false;
    if (!attr_set) { MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem)); attr_set = true; }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(256), smem, st, reinterpret_cast<const T*>(d->x), reinterpret_cast<const T*>(d->w1),
                       d->scale1, d->shift1, reinterpret_cast<const T*>(d->w2), d->bias2, d->out, g, t);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

}  // namespace mfx
using namespace mfx;

extern "C" int mfx_heads_fused(const mfx_heads_desc* d, void* stream) {
    if (!d || !d->x || !d->w1 || !d->scale1 || !d->shift1 || !d->w2 || !d->bias2 || !d->out)
        return mfx_fail(MFX_ERR_ARG, "heads_fused: null pointer");
    if (d->nbranch < 1 || d->nbranch > 16) return mfx_fail(MFX_ERR_ARG, "heads_fused: 1..16 branches");
    const int elems = d->dtype == MFX_BF16 ? 8 : 4;
    if (d->K_pad % (4 * elems) != 0 || d->K_pad < 576) return mfx_fail(MFX_ERR_ARG, "heads_fused: K_pad must cover 9*64 and be a multiple of 64 bytes");
    for (int i = 0; i < d->nbranch; ++i)
        if (d->c_out[i] < 1 || d->c_out[i] > 32 || d->ch_off[i] < 0 || d->ch_off[i] + d->c_out[i] > d->ld_out)
            return mfx_fail(MFX_ERR_ARG, "heads_fused: branch output channels out of range");
    if (d->B * d->H * d->W == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MFX_F32) return launch_heads<float>(d, st);
    if (d->dtype == MFX_BF16) return launch_heads<bf16_t>(d, st);
    return mfx_fail(MFX_ERR_ARG, "heads_fused: bad dtype");
}
