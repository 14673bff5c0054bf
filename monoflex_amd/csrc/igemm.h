// LDS-tiled MFMA main loop shared by the conv, concat-1x1 (DLA Root), deformable-conv and head kernels.
//
// GEMM view (NHWC activations, KRSC weights): D[m][n] = sum_k A[m][k] * Wt[n][k]
//   m = output pixel (B*Ho*Wo), n = output channel, k = (tap, input channel)
// Both operands are K-contiguous, so a 16-byte chunk (8 bf16 / 4 f32) is one lane's MFMA
// fragment.  Per k-iteration a tile row is 64 bytes of K (4 chunks), padded to 80 bytes in LDS:
// 16 rows x 80 B land on 16 distinct 16-byte bank slots -> conflict-free ds_read_b128.
//
// 256 threads = 4 waves in a WM x WN grid; register-staged double buffering: global loads of
// k+1 are issued before the MFMAs of k and written to the other LDS stage afterwards, one
// barrier per k-iteration.
#pragma once
#include "common.h"
#include <type_traits>

namespace mfx {

// KC = 16-byte chunks of K per tile row per k-iteration (4 -> 64 B, 8 -> 128 B); a row is padded by one
// chunk so 16 consecutive rows land on 16 distinct 16-byte bank slots (80 B and 144 B strides both do).
template <int KC> struct RowGeom { static constexpr int bytes = KC * 16 + 16; };

template <int BM, int BN, int KC = 4> struct TileSmem {
    static constexpr int stage_bytes = (BM + BN) * RowGeom<KC>::bytes;
    static constexpr int mainloop_bytes = 2 * stage_bytes;
    static constexpr int ldc = BN + 4;
    static constexpr int epilogue_bytes = BM * ldc * 4;
    static constexpr int bytes = mainloop_bytes > epilogue_bytes ? mainloop_bytes : epilogue_bytes;
};

// epilogue / launch arguments shared by the GEMM-shaped kernels
struct EpiArgs {
    const float* scale; const float* shift; const void* res; void* y;
    int ldy, ldres, Cout, act, tiles_n, K_pad, nk;
    // split-K (conv_igemm only): gridDim.y = ksplit workgroups share one output tile, each writes its fp32 partial
    // sums to ws[split][M][ws_ld]; splitk_finalize_kernel adds them and applies the real epilogue
    int ksplit = 1, ws_ld = 0;
    float* ws = nullptr;
    // train-mode BN statistics of the output, accumulated by the epilogue (conv_halo.hip): [stats_ncopy][2*Cout] sums | sums of squares
    float* stats = nullptr;
    int stats_ncopy = 1;
};

// Weight (B operand) loader: rows n0.. of a [N_pad][K_pad] K-contiguous matrix.
template <typename T, int BN, int NT = 256, int KC = 4> struct WeightLoader {
    static constexpr int ELEMS = ElemTraits<T>::ELEMS;
    static constexpr int RPP = NT / KC;                 // rows covered per pass
    static constexpr int R = (BN + RPP - 1) / RPP;
    static constexpr int RB = RowGeom<KC>::bytes;
    const T* w; int ldk; int c, r0;
    u32x4 regs[R];
    __device__ __forceinline__ void init(const T* w_, int n0, int K_pad, int tid) {
        c = tid % KC; r0 = tid / KC; ldk = K_pad;
        w = w_ + (size_t)n0 * K_pad + c * ELEMS;
    }
    __device__ __forceinline__ void load(int kiter) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int n = r0 + RPP * i;
            if (BN % RPP == 0 || n < BN)
                regs[i] = *reinterpret_cast<const u32x4*>(w + (size_t)n * ldk + kiter * (KC * ELEMS));
        }
    }
    __device__ __forceinline__ void store(char* Bs) const {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int n = r0 + RPP * i;
            if (BN % RPP == 0 || n < BN) *reinterpret_cast<u32x4*>(Bs + n * RB + c * 16) = regs[i];
        }
    }
};

// acc[FM][FN] += A-tile x B-tile over nk k-iterations.  ALoader provides load(kiter)/store(As).
template <typename T, int BM, int BN, int WM, int WN, int KC, typename ALoader>
__device__ __forceinline__ void gemm_mainloop(ALoader& al, WeightLoader<T, BN, WM * WN * 64, KC>& bl, int nk, char* smem,
                                              f32x4 (&acc)[BM / WM / 16][BN / WN / 16], int k0 = 0) {
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    constexpr int STAGE = TileSmem<BM, BN, KC>::stage_bytes;
    constexpr int kRowBytes = RowGeom<KC>::bytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int frag_off = (lane & 15) * kRowBytes + (lane >> 4) * 16;
    const int a_off = wm * (BM / WM) * kRowBytes + frag_off;
    const int b_off = BM * kRowBytes + wn * (BN / WN) * kRowBytes + frag_off;

#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    al.load(k0);
    bl.load(k0);
    al.store(smem);
    bl.store(smem + BM * kRowBytes);
    __syncthreads();

    for (int k = 0; k < nk; ++k) {
        char* cur = smem + (k & 1) * STAGE;
        char* nxt = smem + ((k + 1) & 1) * STAGE;
        const bool more = (k + 1) < nk;
        if (more) { al.load(k0 + k + 1); bl.load(k0 + k + 1); }

        if constexpr (std::is_same<T, f32s_t>::value && KC == 8) {
            // split precision, two sub-steps per k-iteration: walk them as ONE pair.  A chunk is [hi hi | lo lo] (dwords), so the hi (lo) halves
            // of the lane's chunks of both sub-steps form one 8-element fp16 operand (two 8-byte LDS reads 64 bytes apart), and the pair costs
            // three products -- hi.hi, lo.hi, hi.lo -- instead of the four of two mma_chunk<f32s_t> calls
            auto rd = [](const char* q) {
                const uint2 a = *reinterpret_cast<const uint2*>(q), b = *reinterpret_cast<const uint2*>(q + 64);
                return u32x4{a.x, a.y, b.x, b.y};
            };
            u32x4 ah[FM], al_[FM], bh[FN], bl_[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) { ah[i] = rd(cur + a_off + i * 16 * kRowBytes); al_[i] = rd(cur + a_off + i * 16 * kRowBytes + 8); }
#pragma unroll
            for (int j = 0; j < FN; ++j) { bh[j] = rd(cur + b_off + j * 16 * kRowBytes); bl_[j] = rd(cur + b_off + j * 16 * kRowBytes + 8); }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[i]), __builtin_bit_cast(f16x8, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al_[i]), __builtin_bit_cast(f16x8, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[i]), __builtin_bit_cast(f16x8, bl_[j]), acc[i][j], 0, 0, 0);
        } else
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {               // 64 bytes of K per sub-step
            u32x4 af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const u32x4*>(cur + a_off + i * 16 * kRowBytes + ks * 64);
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(cur + b_off + j * 16 * kRowBytes + ks * 64);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) mma_chunk<T>(af[i], bf[j], acc[i][j]);
        }

        if (more) { al.store(nxt); bl.store(nxt + BM * kRowBytes); }
        __syncthreads();
    }
}

// Epilogue: acc*scale+shift -> LDS (fp32) -> (+residual) -> activation -> coalesced 16-byte stores.
template <typename T, typename TO, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void epilogue_store(const f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem,
                                               const float* scale, const float* shift, const T* res, int ldres,
                                               TO* y, int ldy, int m0, int n0, int M, int Cout, int act) {
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    constexpr int LDC = TileSmem<BM, BN>::ldc;
    constexpr int kThreads = WM * WN * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    float* Cs = reinterpret_cast<float*>(smem);
    // (the main loop ended with a barrier: every wave is done reading the stage buffers)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = wn * (BN / WN) + j * 16 + (lane & 15);
        const float sc = scale ? scale[n0 + n] : 1.f;
        const float sh = shift ? shift[n0 + n] : 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = wm * (BM / WM) + i * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(m + r) * LDC + n] = acc[i][j][r] * sc + sh;
        }
    }
    __syncthreads();
    constexpr int OE = ElemTraits<TO>::ELEMS;
    constexpr int GPR = BN / OE;
    for (int g = tid; g < BM * GPR; g += kThreads) {
        const int m = g / GPR, ng = g - m * GPR;
        const int gm = m0 + m, gn = n0 + ng * OE;
        if (gm >= M || gn >= Cout) continue;
        float v[OE];
#pragma unroll
        for (int e = 0; e < OE; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(Cs + m * LDC + ng * OE + e);
            v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
        }
        if (res) {
            const T* rp = res + (size_t)gm * ldres + gn;
            if constexpr (ElemTraits<T>::ELEMS == OE) {      // same element type: one 16-byte chunk
                float rv[OE];
                ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(rp), rv);
#pragma unroll
                for (int e = 0; e < OE; ++e) v[e] += rv[e];
            } else {
#pragma unroll
                for (int e = 0; e < OE; ++e) v[e] += ElemTraits<T>::load(rp + e);
            }
        }
apply_act_chunk<OE>(v, act, gn);
        *reinterpret_cast<u32x4*>(y + (size_t)gm * ldy + gn) = ElemTraits<TO>::pack(v);
    }
}

}  // namespace mfx
