// Implicit-GEMM convolution kernels for gfx950: plain conv, concat-1x1 (DLA Root) and the fused
// modulated deformable convolution.  All share igemm.h's LDS-tiled MFMA main loop and epilogue.
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "igemm.h"
#include <string>
#include <utility>
#include <vector>
#include <type_traits>

namespace mfx {

// ------------------------------------------------------------------------------------------------
// A-operand loaders
// ------------------------------------------------------------------------------------------------
struct ConvGeom {
    int H, W, Ho, Wo, x_pixstride, lgC, kh, kw, inv_kw, stride, pad_h, pad_w, dil_w, M;
};

// im2col on the fly: row m = output pixel, chunk -> (tap, channel); out-of-image taps read zeros.
template <typename T, int BM, int NT, int KC> struct ConvALoader {
    static constexpr int ELEMS = ElemTraits<T>::ELEMS;
    static constexpr int RPP = NT / KC;
    static constexpr int R = BM / RPP;
    static constexpr int kRowBytes = RowGeom<KC>::bytes;
    static_assert(BM % RPP == 0, "BM must be a multiple of the rows covered per pass");
    const T* x; ConvGeom g; int c, r0;
    int ih0[R], iw0[R], pix0[R]; bool ok[R];
    u32x4 regs[R];
    __device__ __forceinline__ void init(const T* x_, const ConvGeom& g_, const int* rowmap, int m0, int tid) {
        x = x_; g = g_; c = tid % KC; r0 = tid / KC;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int m = m0 + r0 + RPP * i;
            int pm = (m < g.M) ? (rowmap ? rowmap[m] : m) : -1;
            ok[i] = pm >= 0;
            pm = pm < 0 ? 0 : pm;
            const int hw = g.Ho * g.Wo;
            const int b = pm / hw, rem = pm - b * hw;
            const int oh = rem / g.Wo, ow = rem - oh * g.Wo;
            ih0[i] = oh * g.stride - g.pad_h;
            iw0[i] = ow * g.stride - g.pad_w;
            pix0[i] = b * g.H * g.W;
        }
    }
    __device__ __forceinline__ void load(int kiter) {
        const int e = kiter * (KC * ELEMS) + c * ELEMS;
        const int tap = e >> g.lgC, ci = e & ((1 << g.lgC) - 1);
        const int th = (tap * g.inv_kw) >> 16, tw = tap - th * g.kw;
        const bool tap_ok = tap < g.kh * g.kw;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int ih = ih0[i] + th, iw = iw0[i] + tw * g.dil_w;
            const bool v = ok[i] && tap_ok && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
            // (branch-free: clamped address + select, so the R loads of a k-iteration are all in flight together)
            const int ihc = min(max(ih, 0), g.H - 1), iwc = min(max(iw, 0), g.W - 1);
            const u32x4 z = *reinterpret_cast<const u32x4*>(x + (size_t)(pix0[i] + ihc * g.W + iwc) * g.x_pixstride + ci);
            regs[i] = v ? z : u32x4{0u, 0u, 0u, 0u};
        }
    }
    __device__ __forceinline__ void store(char* As) const {
#pragma unroll
        for (int i = 0; i < R; ++i) *reinterpret_cast<u32x4*>(As + (r0 + RPP * i) * kRowBytes + c * 16) = lds_operand<T>(regs[i]);
    }
};

struct CatSegs {
    const void* src[MFX_MAX_SEG]; int stride[MFX_MAX_SEG]; int off[MFX_MAX_SEG]; int lgC; int M;
};

// virtual channel concat for the 1x1 Root conv: k-iteration -> (segment, channel) is wave-uniform.
template <typename T, int BM, int NT, int KC> struct CatALoader {
    static constexpr int ELEMS = ElemTraits<T>::ELEMS;
    static constexpr int RPP = NT / KC;
    static constexpr int R = BM / RPP;
    static constexpr int kRowBytes = RowGeom<KC>::bytes;
    static_assert(BM % RPP == 0, "BM must be a multiple of the rows covered per pass");
    const CatSegs* s; int c, r0, m0;
    u32x4 regs[R];
    __device__ __forceinline__ void init(const CatSegs* s_, int m0_, int tid) { s = s_; m0 = m0_; c = tid % KC; r0 = tid / KC; }
    __device__ __forceinline__ void load(int kiter) {
        const int e = kiter * (KC * ELEMS);
        const int seg = e >> s->lgC, ci = (e & ((1 << s->lgC) - 1)) + c * ELEMS;
        const T* base = reinterpret_cast<const T*>(s->src[seg]) + s->off[seg] + ci;
        const int st = s->stride[seg];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int m = m0 + r0 + RPP * i;
            const u32x4 z = *reinterpret_cast<const u32x4*>(base + (size_t)min(m, s->M - 1) * st);
            regs[i] = m < s->M ? z : u32x4{0u, 0u, 0u, 0u};
        }
    }
    __device__ __forceinline__ void store(char* As) const {
#pragma unroll
        for (int i = 0; i < R; ++i) *reinterpret_cast<u32x4*>(As + (r0 + RPP * i) * kRowBytes + c * 16) = lds_operand<T>(regs[i]);
    }
};

struct DcnGeom { int H, W, C, lgC, Ho, Wo, kh, kw, inv_kw, stride, pad, dil, M, stride_w, pad_w, dil_w; };      // stride / pad / dil: rows; *_w: columns

// Deformable sampler: A[m][(tap,c)] = mask * bilinear(x[b,:,:,c] at (oh*s-p+th*d+dh, ow*s-p+tw*d+dw)).
// In NHWC the four corners are contiguous channel vectors, so every lane gathers 4 x 16 bytes and
// blends them in fp32; offsets/mask are read once per (pixel, tap) and reused across all channels.
// Sample validity and per-corner zeroing follow src/cuda/dcn_v2_im2col_cuda.cu:25-54,178-189.
template <typename T, int BM, int NT, int KC> struct DcnALoader {
    static constexpr int ELEMS = ElemTraits<T>::ELEMS;
    static constexpr int RPP = NT / KC;
    static constexpr int R = BM / RPP;
    static constexpr int kRowBytes = RowGeom<KC>::bytes;
    static_assert(BM % RPP == 0, "BM must be a multiple of the rows covered per pass");
    const T* x; const float* om; DcnGeom g; int c, r0, cur_tap;
    int oh_[R], ow_[R], pix0[R], mrow[R]; bool ok[R];
    uint32_t cofb[R][4]; float cw[R][4];      // byte offsets of the four (clamped) corner rows' chunk c (the launch checks the tensor is < 4 GB)
    u32x4 regs[R][4];
    __device__ __forceinline__ void init(const T* x_, const float* om_, const DcnGeom& g_, int m0, int tid) {
        x = x_; om = om_; g = g_; c = tid % KC; r0 = tid / KC; cur_tap = -1;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int m = m0 + r0 + RPP * i;
            ok[i] = m < g.M;
            const int pm = ok[i] ? m : 0;
            mrow[i] = pm;
            const int hw = g.Ho * g.Wo;
            const int b = pm / hw, rem = pm - b * hw;
            oh_[i] = rem / g.Wo; ow_[i] = rem - oh_[i] * g.Wo;
            pix0[i] = b * g.H * g.W;
        }
    }
    __device__ __forceinline__ void tap_setup(int tap) {
        const int th = (tap * g.inv_kw) >> 16, tw = tap - th * g.kw;
        // (r05: running the gather TWO k-iterations ahead -- a second register stage, branch-free steady loop, the blend of k+1 under the loads of k+2 --
    // measured 46.3 -> 48.0 us on 128 -> 64 @ 48x160 and worse where the extra 16-32 registers cost a wave per SIMD; tools/r05_call18.sh.  Not kept.)
    // (r05: fetching the NEXT tap's offsets one tap ahead measured neutral in bf16 and -2.8 % in the split-precision instantiation -- six
        // more live registers per row; the offset rows are L2-resident and other workgroups cover the dependent load.  Not kept.)
        const uint32_t rowb = (uint32_t)g.C * sizeof(T), cb = (uint32_t)c * 16;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const float* o = om + (size_t)mrow[i] * 32;
            const float dh = o[2 * tap], dw = o[2 * tap + 1], mk = o[18 + tap];
            const float h = (float)(oh_[i] * g.stride - g.pad + th * g.dil) + dh;
            const float w = (float)(ow_[i] * g.stride_w - g.pad_w + tw * g.dil_w) + dw;
            const bool inside = ok[i] && h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
            const float hf = floorf(h), wf = floorf(w);
            const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
            const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw_ = 1.f - lw;
            const bool t0 = inside && h0 >= 0, t1 = inside && h1 <= g.H - 1;
            const bool l0 = w0 >= 0, l1 = w1 <= g.W - 1;
            const int ch0 = min(max(h0, 0), g.H - 1), ch1 = min(max(h1, 0), g.H - 1);
            const int cw0 = min(max(w0, 0), g.W - 1), cw1 = min(max(w1, 0), g.W - 1);
            cofb[i][0] = (uint32_t)(pix0[i] + ch0 * g.W + cw0) * rowb + cb; cw[i][0] = (t0 && l0) ? hh * hw_ * mk : 0.f;
            cofb[i][1] = (uint32_t)(pix0[i] + ch0 * g.W + cw1) * rowb + cb; cw[i][1] = (t0 && l1) ? hh * lw * mk : 0.f;
            cofb[i][2] = (uint32_t)(pix0[i] + ch1 * g.W + cw0) * rowb + cb; cw[i][2] = (t1 && l0) ? lh * hw_ * mk : 0.f;
            cofb[i][3] = (uint32_t)(pix0[i] + ch1 * g.W + cw1) * rowb + cb; cw[i][3] = (t1 && l1) ? lh * lw * mk : 0.f;
        }
    }
    __device__ __forceinline__ void load(int kiter) {
        const int e = kiter * (KC * ELEMS);
        const uint32_t kb = (uint32_t)(e & (g.C - 1)) * sizeof(T);   // byte offset of this k-iteration's channels inside a row
        const int tap = e >> g.lgC;
        if (tap != cur_tap) { tap_setup(tap); cur_tap = tap; } // wave-uniform: a new tap starts (or a K split starts mid-way)
        const char* xk = reinterpret_cast<const char*>(x) + kb;     // wave-uniform base (scalar registers) + 32-bit lane offset
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                regs[i][q] = *reinterpret_cast<const u32x4*>(xk + cofb[i][q]);
    }
    // blend weights cw[] were (re)computed by the load() that filled regs[]: load(k+1) -> mma(k) -> store(k+1)
    __device__ __forceinline__ void store(char* As) const {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            float v[4][ELEMS], o[ELEMS];
#pragma unroll
            for (int q = 0; q < 4; ++q) ElemTraits<T>::unpack(regs[i][q], v[q]);
            // two channels per instruction (v_pk_mul_f32 / v_pk_fma_f32: packed fp32 runs at twice the scalar rate on CDNA3/4)
            if constexpr (ELEMS == 4) {                        // fp32 / split-precision chunks: scalar FMAs (the packed form measured -0.8 % on the fp16x2 step)
#pragma unroll
                for (int e = 0; e < ELEMS; ++e) o[e] = cw[i][0] * v[0][e] + cw[i][1] * v[1][e] + cw[i][2] * v[2][e] + cw[i][3] * v[3][e];
            } else {
#pragma unroll
                for (int e = 0; e < ELEMS; e += 2) {
                f32x2 t = (f32x2){v[0][e], v[0][e + 1]} * cw[i][0];
                t = __builtin_elementwise_fma((f32x2){v[1][e], v[1][e + 1]}, (f32x2){cw[i][1], cw[i][1]}, t);
                t = __builtin_elementwise_fma((f32x2){v[2][e], v[2][e + 1]}, (f32x2){cw[i][2], cw[i][2]}, t);
                t = __builtin_elementwise_fma((f32x2){v[3][e], v[3][e + 1]}, (f32x2){cw[i][3], cw[i][3]}, t);
                    o[e] = t[0]; o[e + 1] = t[1];
                }
            }
            *reinterpret_cast<u32x4*>(As + (r0 + RPP * i) * kRowBytes + c * 16) = lds_operand<T>(ElemTraits<T>::pack(o));
        }
    }
};

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------

template <typename T, typename TO, int BM, int BN, int WM, int WN, int KC>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_kernel(const T* x, const T* w, ConvGeom g, const int* rowmap, EpiArgs ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % ep.tiles_n, tm = tile / ep.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    ConvALoader<T, BM, WM * WN * 64, KC> al; al.init(x, g, rowmap, m0, threadIdx.x);
    WeightLoader<T, BN, WM * WN * 64, KC> bl; bl.init(w, n0, ep.K_pad, threadIdx.x);
    f32x4 acc[BM / WM / 16][BN / WN / 16];
    if (ep.ksplit > 1) {                                      // this workgroup's slice of K; fp32 partial tile to the workspace
        const int per = (ep.nk + ep.ksplit - 1) / ep.ksplit, k0 = blockIdx.y * per;
        gemm_mainloop<T, BM, BN, WM, WN, KC>(al, bl, min(per, ep.nk - k0), smem, acc, k0);
        epilogue_store<T, float, BM, BN, WM, WN>(acc, smem, nullptr, nullptr, nullptr, 0,
                                                 ep.ws + (size_t)blockIdx.y * g.M * ep.ws_ld, ep.ws_ld, m0, n0, g.M, ep.ws_ld, ACT_NONE);
        return;
    }
    gemm_mainloop<T, BM, BN, WM, WN, KC>(al, bl, ep.nk, smem, acc);
    epilogue_store<T, TO, BM, BN, WM, WN>(acc, smem, ep.scale, ep.shift, reinterpret_cast<const T*>(ep.res), ep.ldres,
                                          reinterpret_cast<TO*>(ep.y), ep.ldy, m0, n0, g.M, ep.Cout, ep.act);
}

// y = act((sum_s ws[s][m][n]) * scale[n] + shift[n] (+ res)): the epilogue of a split-K convolution
template <typename TO>
__global__ void splitk_finalize_kernel(const float* __restrict__ ws, int ksplit, size_t split_stride, int ws_ld,
                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                       const TO* __restrict__ res, int ldres, TO* __restrict__ y, int ldy, int M, int Cout, int act) {
    constexpr int OE = ElemTraits<TO>::ELEMS;
    const int gpr = Cout / OE;
    const long total = (long)M * gpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / gpr), n = (int)(i - (long)m * gpr) * OE;
        float v[OE];
#pragma unroll
        for (int e = 0; e < OE; ++e) v[e] = 0.f;
        for (int s = 0; s < ksplit; ++s) {
            const float* p = ws + s * split_stride + (size_t)m * ws_ld + n;
#pragma unroll
            for (int e = 0; e < OE; e += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p + e);
                v[e] += t[0]; v[e + 1] += t[1]; v[e + 2] += t[2]; v[e + 3] += t[3];
            }
        }
#pragma unroll
        for (int e = 0; e < OE; ++e) v[e] = v[e] * (scale ? scale[n + e] : 1.f) + (shift ? shift[n + e] : 0.f);
        if (res) {
            float rv[OE];
            ElemTraits<TO>::unpack(*reinterpret_cast<const u32x4*>(res + (size_t)m * ldres + n), rv);
#pragma unroll
            for (int e = 0; e < OE; ++e) v[e] += rv[e];
        }
        apply_act_chunk<OE>(v, act, n);
        *reinterpret_cast<u32x4*>(y + (size_t)m * ldy + n) = ElemTraits<TO>::pack(v);
    }
}

template <typename T, int BM, int BN, int WM, int WN, int KC>
__global__ __launch_bounds__(WM * WN * 64) void cat_igemm_kernel(CatSegs segs, const T* w, EpiArgs ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % ep.tiles_n, tm = tile / ep.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    CatALoader<T, BM, WM * WN * 64, KC> al; al.init(&segs, m0, threadIdx.x);
    WeightLoader<T, BN, WM * WN * 64, KC> bl; bl.init(w, n0, ep.K_pad, threadIdx.x);
    f32x4 acc[BM / WM / 16][BN / WN / 16];
    gemm_mainloop<T, BM, BN, WM, WN, KC>(al, bl, ep.nk, smem, acc);
    epilogue_store<T, T, BM, BN, WM, WN>(acc, smem, ep.scale, ep.shift, reinterpret_cast<const T*>(ep.res), ep.ldres,
                                         reinterpret_cast<T*>(ep.y), ep.ldy, m0, n0, segs.M, ep.Cout, ep.act);
}

template <typename T, int BM, int BN, int WM, int WN, int KC>
__global__ __launch_bounds__(WM * WN * 64) void dcn_igemm_kernel(const T* x, const float* om, const T* w, DcnGeom g, EpiArgs ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % ep.tiles_n, tm = tile / ep.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    DcnALoader<T, BM, WM * WN * 64, KC> al; al.init(x, om, g, m0, threadIdx.x);
    WeightLoader<T, BN, WM * WN * 64, KC> bl; bl.init(w, n0, ep.K_pad, threadIdx.x);
    f32x4 acc[BM / WM / 16][BN / WN / 16];
    if (ep.ksplit > 1) {          // split-K: the gather (the expensive part) is split with it, nothing is duplicated
        const int per = (ep.nk + ep.ksplit - 1) / ep.ksplit, k0 = blockIdx.y * per;
        gemm_mainloop<T, BM, BN, WM, WN, KC>(al, bl, min(per, ep.nk - k0), smem, acc, k0);
        epilogue_store<T, float, BM, BN, WM, WN>(acc, smem, nullptr, nullptr, nullptr, 0,
                                                 ep.ws + (size_t)blockIdx.y * g.M * ep.ws_ld, ep.ws_ld, m0, n0, g.M, ep.ws_ld, ACT_NONE);
        return;
    }
    gemm_mainloop<T, BM, BN, WM, WN, KC>(al, bl, ep.nk, smem, acc);
    epilogue_store<T, T, BM, BN, WM, WN>(acc, smem, ep.scale, ep.shift, nullptr, 0,
                                         reinterpret_cast<T*>(ep.y), ep.ldy, m0, n0, g.M, ep.Cout, ep.act);
}

// ------------------------------------------------------------------------------------------------
// launch helpers + tile selection
// ------------------------------------------------------------------------------------------------
static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

int try_conv_halo(const mfx_conv_desc* d, hipStream_t st, int* stats_ran);   // conv_halo.hip
extern int g_opt_halo_cw, g_opt_cw_rows6, g_opt_halo_cws;
extern int g_opt_halo, g_opt_halo_cg, g_opt_halo_pair, g_opt_halo_s2, g_opt_dcn_wave, g_opt_dcn_patch, g_opt_dcn_patch_fn8, g_opt_dcn_wgrad_m;
}
extern long g_cnt_dcn_bt_fused, g_cnt_dcn_bt_fly;
extern int g_opt_ext_bwd_fast;
extern int g_opt_dcn_bt_fly;
extern int g_opt_wgrad_min_m;
extern int g_opt_dcn_bt_gcol_as;
extern int g_opt_dcn_bt_fly_bias;
extern int g_opt_dcn_bt_fuse_min_chunks, g_opt_dcn_bt_fuse_blocks, g_opt_dcn_bt_fuse_wgrad, g_opt_heads_planes, g_opt_heads_persist, g_opt_heads_dbg, g_opt_heads_mfma32, g_opt_dcn_bt_cs, g_opt_dcn_bt_cs_wgs, g_opt_dcn_bt_dbg, g_opt_wgrad_tr, g_opt_wgrad_tr_blocks, g_opt_bn_blocks, g_opt_bn_apply_blocks, g_opt_bn_onepass, g_opt_bn_onepass_grid, g_opt_bn_onepass_min_chunks, g_opt_bn_onepass_fwd_min_chunks, g_opt_wgrad_patch, g_opt_wgrad_patch_blocks, g_opt_wgrad_patch_waves;
extern int g_opt_topk_strips, g_opt_topk_merge_z, g_opt_topk_merge_threads;                                                                                           // decode.hip (global namespace)
extern int g_opt_wgrad_mfma, g_opt_wgrad_blocks, g_opt_wgrad_ws, g_opt_wgrad_ws_blocks;                                   // train_kernels.hip (global namespace)
namespace mfx {
int try_dcn_wave(const mfx_dcn_desc* d, hipStream_t st);      // dcn_wave.hip
int try_dcn_patch(const mfx_dcn_desc* d, hipStream_t st);     // dcn_patch.hip
int try_dcn_lds(const mfx_dcn_desc* d, hipStream_t st);       // dcn_lds.hip
bool dcn_lds_fuses_offset_conv(const mfx_dcn_desc* d);
extern int g_opt_dcn_lds, g_opt_dcn_lds_rows;
bool dcn_patch_fuses_offset_conv(const mfx_dcn_desc* d);
extern int g_opt_dcn_fuse_off;

// tuning overrides (mfx_set_option): 0 = automatic
int g_opt_conv_tile = 0, g_opt_dcn_tile = 0, g_opt_cat_tile = 0, g_opt_kc = 0;
int g_opt_ksplit = 0;        // 0 = automatic, 1 = never split, n = force n splits where legal
int g_opt_dcn_ksplit = 0;    // same for the fused DCN kernel

template <typename K> static int set_smem(K k, int smem) {
    if (smem > 64 * 1024) MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    return MFX_OK;
}

template <typename T, typename TO, int BM, int BN, int WM, int WN, int KC>
static int launch_conv(const mfx_conv_desc* d, const ConvGeom& g, EpiArgs ep, hipStream_t st) {
    ep.tiles_n = d->Cout_pad / BN;
    ep.nk = d->K_pad / (KC * ElemTraits<T>::ELEMS);
    const int tiles = cdiv(d->M, BM) * ep.tiles_n;
    auto k = conv_igemm_kernel<T, TO, BM, BN, WM, WN, KC>;
    constexpr int smem = TileSmem<BM, BN, KC>::bytes;
    static bool attr_set = false;
    if (!attr_set) { int rc = set_smem(k, smem); if (rc) return rc; attr_set = true; }
    // split-K (option "ksplit" = n): shares each output tile among n workgroups.  Built for the level4/level5 layers
    // (M = 15360 / 3840 rows, K = 2304 / 4608: < 2 workgroups per CU), but measured neutral-to-slower there (54 -> 49..58 us,
    // 43 -> 64..80 us): those layers are bound by the L2->LDS operand traffic of the 64-wide tiles, not by occupancy.
    int ksplit = 1;
    if constexpr (std::is_same<T, TO>::value) {
        if (d->workspace && !d->rowmap && g_opt_ksplit != 1 && d->Cout == d->Cout_pad) {
            ksplit = g_opt_ksplit > 1 ? g_opt_ksplit : 1;      // measured (tools/splitk_probe.sh): no gain on DLA level4/5 -> opt-in only
            ksplit = std::min(ksplit, std::min(8, ep.nk / 6));
            while (ksplit > 1 && ((ep.nk + ksplit - 1) / ksplit) * (ksplit - 1) >= ep.nk) --ksplit;     // no empty split
            if ((size_t)ksplit * d->M * d->Cout_pad * sizeof(float) > (size_t)d->workspace_bytes) ksplit = 1;
        }
    }
    if (ksplit > 1) {
        ep.ksplit = ksplit; ep.ws = reinterpret_cast<float*>(d->workspace); ep.ws_ld = d->Cout_pad;
        hipLaunchKernelGGL(k, dim3(tiles, ksplit), dim3(WM * WN * 64), smem, st, reinterpret_cast<const T*>(d->x),
                           reinterpret_cast<const T*>(d->w), g, d->rowmap, ep);
        MFX_HIP_CHECK(hipGetLastError());
        const long chunks = (long)d->M * (d->Cout / ElemTraits<TO>::ELEMS);
        const int blocks = (int)std::min<long>((chunks + 255) / 256, 4096);
        hipLaunchKernelGGL(splitk_finalize_kernel<TO>, dim3(blocks), dim3(256), 0, st, ep.ws, ksplit, (size_t)d->M * d->Cout_pad, d->Cout_pad,
                           d->scale, d->shift, reinterpret_cast<const TO*>(d->res), d->ldres, reinterpret_cast<TO*>(d->y), d->ldy,
                           d->M, d->Cout, d->act);
        MFX_HIP_CHECK(hipGetLastError());
        return MFX_OK;
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(WM * WN * 64), smem, st, reinterpret_cast<const T*>(d->x),
                       reinterpret_cast<const T*>(d->w), g, d->rowmap, ep);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// tile ids (also the values of the "conv_tile" / "dcn_tile" / "cat_tile" options)
enum { T_256x16 = 1, T_256x32 = 2, T_128x64 = 3, T_64x64 = 4, T_128x128 = 5, T_64x128 = 6, T_256x64 = 7, T_256x128_8w = 8, T_128x128_8w = 9 };

template <typename T, typename TO, int KC>
static int dispatch_conv_kc(int tile, const mfx_conv_desc* d, const ConvGeom& g, const EpiArgs& ep, hipStream_t st) {
    switch (tile) {
        case T_256x16: return launch_conv<T, TO, 256, 16, 4, 1, 4>(d, g, ep, st);
        case T_256x32: return launch_conv<T, TO, 256, 32, 4, 1, 4>(d, g, ep, st);
        default: break;
    }
    if constexpr (std::is_same<T, TO>::value) {
        switch (tile) {
            case T_128x64: return launch_conv<T, TO, 128, 64, 4, 1, KC>(d, g, ep, st);
            case T_64x64: return launch_conv<T, TO, 64, 64, 2, 2, KC>(d, g, ep, st);
            case T_128x128: return launch_conv<T, TO, 128, 128, 2, 2, KC>(d, g, ep, st);
            case T_64x128: return launch_conv<T, TO, 64, 128, 2, 2, KC>(d, g, ep, st);
            case T_256x64: return launch_conv<T, TO, 256, 64, 4, 1, KC>(d, g, ep, st);
            case T_256x128_8w: return launch_conv<T, TO, 256, 128, 4, 2, KC>(d, g, ep, st);
            case T_128x128_8w: return launch_conv<T, TO, 128, 128, 2, 4, KC>(d, g, ep, st);
            default: break;
        }
    }
    return mfx_fail(MFX_ERR_UNSUPPORTED, "conv2d: no kernel for this tile / dtype combination");
}

static bool tile_fits(int tile, int N) {
    switch (tile) {
        case T_256x16: return N == 16;
        case T_256x32: return N == 32;
        case T_128x64: case T_64x64: case T_256x64: return N % 64 == 0;
        default: return N % 128 == 0;
    }
}

static int pick_conv_tile(int M, int N) {
    if (N == 16) return T_256x16;
    if (N == 32) return T_256x32;
    if (N == 64) return cdiv(M, 128) >= 512 ? T_128x64 : T_64x64;
    if (N % 128 == 0) {
        const int t128 = cdiv(M, 128) * (N / 128);
        if (t128 >= 384) return T_128x128;
        if (cdiv(M, 64) * (N / 128) >= 384) return T_64x128;
    }
    return T_64x64;
}

template <typename T, typename TO>
static int dispatch_conv(const mfx_conv_desc* d, const ConvGeom& g, const EpiArgs& ep, hipStream_t st) {
    const int N = d->Cout_pad;
    if (N != 16 && N != 32 && N % 64 != 0) return mfx_fail(MFX_ERR_ARG, "conv2d: Cout_pad must be 16, 32 or a multiple of 64");
    int tile = pick_conv_tile(d->M, N);
    if (d->kh * d->kw == 1 && N == 64) tile = T_64x64;          // short-K pointwise layers: more, smaller workgroups (r05 sweep, profiles/r05_pointwise_tiles.md)
    if (g_opt_conv_tile && tile_fits(g_opt_conv_tile, N)) tile = g_opt_conv_tile;
    const int elems = ElemTraits<T>::ELEMS;
    const bool kc8_ok = d->K_pad % (8 * elems) == 0;
    const int kc = (g_opt_kc == 4 || !kc8_ok) ? 4 : 8;
    return kc == 8 ? dispatch_conv_kc<T, TO, 8>(tile, d, g, ep, st) : dispatch_conv_kc<T, TO, 4>(tile, d, g, ep, st);
}

}  // namespace mfx

using namespace mfx;

// option name -> the process-wide switch it sets (nullptr: unknown)
static int* option_slot(const std::string& n) {
    static const std::pair<const char*, int*> table[] = {
        {"conv_tile", &g_opt_conv_tile}, {"dcn_tile", &g_opt_dcn_tile}, {"cat_tile", &g_opt_cat_tile}, {"kc", &g_opt_kc}, {"ksplit", &g_opt_ksplit},
        {"dcn_ksplit", &g_opt_dcn_ksplit}, {"wgrad_mfma", &g_opt_wgrad_mfma}, {"wgrad_blocks", &g_opt_wgrad_blocks}, {"wgrad_ws", &g_opt_wgrad_ws},
        {"wgrad_ws_blocks", &g_opt_wgrad_ws_blocks}, {"dcn_wgrad_m", &g_opt_dcn_wgrad_m}, {"halo", &g_opt_halo}, {"halo_cg", &g_opt_halo_cg},
        {"halo_cw", &g_opt_halo_cw}, {"halo_cws", &g_opt_halo_cws}, {"ext_bwd_fast", &g_opt_ext_bwd_fast}, {"cw_rows6", &g_opt_cw_rows6}, {"halo_pair", &g_opt_halo_pair}, {"halo_s2", &g_opt_halo_s2}, {"dcn_wave", &g_opt_dcn_wave}, {"dcn_patch", &g_opt_dcn_patch}, {"dcn_lds", &g_opt_dcn_lds}, {"dcn_lds_rows", &g_opt_dcn_lds_rows},
        {"dcn_patch_fn8", &g_opt_dcn_patch_fn8}, {"dcn_fuse_off", &g_opt_dcn_fuse_off}, {"topk_strips", &g_opt_topk_strips}, {"topk_merge_z", &g_opt_topk_merge_z}, {"topk_merge_threads", &g_opt_topk_merge_threads},
#ifdef MFX_PROBES
        {"dcn_bt_dbg", &g_opt_dcn_bt_dbg}, {"heads_dbg", &g_opt_heads_dbg},
#endif
        {"wgrad_tr", &g_opt_wgrad_tr}, {"bn_blocks", &g_opt_bn_blocks}, {"heads_planes", &g_opt_heads_planes}, {"heads_mfma32", &g_opt_heads_mfma32}, {"heads_persist", &g_opt_heads_persist},
        {"dcn_bt_fuse_wgrad", &g_opt_dcn_bt_fuse_wgrad}, {"dcn_bt_fly", &g_opt_dcn_bt_fly}, {"dcn_bt_gcol_as", &g_opt_dcn_bt_gcol_as}, {"dcn_bt_fly_bias", &g_opt_dcn_bt_fly_bias}, {"wgrad_min_m", &g_opt_wgrad_min_m}, {"dcn_bt_fuse_blocks", &g_opt_dcn_bt_fuse_blocks},
        {"deterministic", &g_opt_det}, {"dcn_bt_fuse_min_chunks", &g_opt_dcn_bt_fuse_min_chunks}, {"dcn_bt_cs", &g_opt_dcn_bt_cs},
        {"dcn_bt_cs_wgs", &g_opt_dcn_bt_cs_wgs}, {"bn_apply_blocks", &g_opt_bn_apply_blocks}, {"bn_onepass", &g_opt_bn_onepass}, {"bn_onepass_grid", &g_opt_bn_onepass_grid}, {"bn_onepass_min_chunks", &g_opt_bn_onepass_min_chunks}, {"bn_onepass_fwd_min_chunks", &g_opt_bn_onepass_fwd_min_chunks}, {"wgrad_patch", &g_opt_wgrad_patch},
        {"wgrad_patch_waves", &g_opt_wgrad_patch_waves}, {"wgrad_patch_blocks", &g_opt_wgrad_patch_blocks}, {"wgrad_tr_blocks", &g_opt_wgrad_tr_blocks}};
    for (const auto& e : table)
        if (n == e.first) return e.second;
    return nullptr;
}

// the value every switch had before anyone touched it (recorded at the first mfx_set_option of that switch): mfx_reset_options() puts them back
static std::vector<std::pair<int*, int>>& option_defaults() { static std::vector<std::pair<int*, int>> v; return v; }

extern "C" int mfx_set_option(const char* name, int value) {
    if (!name) return mfx_fail(MFX_ERR_ARG, "set_option: null name");
    const std::string n(name);
#ifndef MFX_PROBES
    if (n == "dcn_bt_dbg" || n == "heads_dbg")
        return mfx_fail(MFX_ERR_UNSUPPORTED, "set_option: timing-probe switches (wrong results by design) exist in probe builds only: MFX_PROBES=1 python -m monoflex_amd.build");
#endif
    int* slot = option_slot(n);
    if (!slot) return mfx_fail(MFX_ERR_ARG, "set_option: unknown option");
    auto& defs = option_defaults();
    bool seen = false;
    for (const auto& e : defs) seen = seen || e.first == slot;
    if (!seen) defs.emplace_back(slot, *slot);
    if (n == "dcn_wgrad_m") value = value < 64 ? 64 : value;
    else if (n == "dcn_bt_fuse_blocks") value = value > 0 ? value : 170;
    else if (n == "deterministic") value = value ? 1 : 0;
    else if (n == "dcn_bt_fuse_min_chunks" || n == "wgrad_patch_blocks" || n == "wgrad_tr_blocks") value = value < 1 ? 1 : value;
    *slot = value;
    return MFX_OK;
}

// Every tuning / debug switch back to the value it had at load time.  The switches are process-wide (one host thread drives the library:
// see the threading note in the header), so a test that forces a kernel variant and fails half-way would otherwise leak it into the next test
// (tests/conftest.py calls this after every GPU test).
extern "C" int mfx_reset_options(void) {
    for (const auto& e : option_defaults()) *e.first = e.second;
    return MFX_OK;
}

// The CURRENT values become the ones mfx_reset_options() restores.  The host calls this once after it has applied the process's own switches
// (MFX_OPTIONS in the environment, monoflex_amd/lib.py), so that "load-time value" means "after the environment" and an `MFX_OPTIONS=... pytest`
// sweep keeps its switches across the per-test resets.
extern "C" int mfx_commit_options(void) {
    for (auto& e : option_defaults()) e.second = *e.first;
    return MFX_OK;
}

extern "C" long mfx_get_counter(const char* name) {
    if (!name) return mfx_fail(MFX_ERR_ARG, "get_counter: null name");
    const std::string n(name);
    if (n == "dcn_bt_fused") return g_cnt_dcn_bt_fused;
    if (n == "dcn_bt_fly") return g_cnt_dcn_bt_fly;
    return mfx_fail(MFX_ERR_ARG, "get_counter: unknown counter");
}

extern "C" int mfx_conv2d_nhwc(const mfx_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->y) return mfx_fail(MFX_ERR_ARG, "conv2d: null pointer");
    const bool f32like = d->dtype == MFX_F32 || d->dtype == MFX_F16X2;      // fp32 storage (F16X2: split-precision MFMA operands)
    const int elems = f32like ? 4 : 8;
    if (!f32like && d->dtype != MFX_BF16 && d->dtype != MFX_F16) return mfx_fail(MFX_ERR_ARG, "conv2d: bad dtype");
    if (d->dtype == MFX_F16 && d->out_dtype != MFX_F16 && d->out_dtype != MFX_F32)
        return mfx_fail(MFX_ERR_UNSUPPORTED, "conv2d: fp16 input needs fp16 or fp32 output");
    if (!is_pow2(d->Ck) || d->Ck < elems) return mfx_fail(MFX_ERR_ARG, "conv2d: Ck must be a power of two >= one 16-byte chunk");
    if (d->K_pad % (4 * elems) != 0 || d->K_pad < d->kh * d->kw * d->Ck) return mfx_fail(MFX_ERR_ARG, "conv2d: bad K_pad");
    const int oe = d->out_dtype == MFX_F32 ? 4 : 8;
    if (d->Cout % oe != 0 || d->Cout > d->Cout_pad) return mfx_fail(MFX_ERR_ARG, "conv2d: Cout must be a multiple of the output chunk and <= Cout_pad");
    if (d->kh * d->kw > 64 || d->kw > 8) return mfx_fail(MFX_ERR_UNSUPPORTED, "conv2d: kernel too large");
    if (d->M <= 0) return MFX_OK;
    if (!f32like && d->out_dtype == MFX_F32 && d->res) return mfx_fail(MFX_ERR_UNSUPPORTED, "conv2d: residual needs out_dtype == dtype");
    if (d->dtype == MFX_BF16 && d->out_dtype == MFX_F16) return mfx_fail(MFX_ERR_UNSUPPORTED, "conv2d: bf16 input needs bf16 or fp32 output");
    if (f32like && d->out_dtype != MFX_F32) return mfx_fail(MFX_ERR_UNSUPPORTED, "conv2d: f32 input needs f32 output");
    if (d->stats && (!d->stats_done || d->res || d->act != MFX_ACT_NONE || d->rowmap))
        return mfx_fail(MFX_ERR_ARG, "conv2d: output statistics need stats_done and a plain (no residual / activation / row map) epilogue");
    if (d->stats_done) *d->stats_done = 0;
    mfx_conv_desc det_copy;
    if (g_opt_det && d->stats) {                            // epilogue statistics are one atomic per column and wave: the BN runs its own (ordered) pass
        det_copy = *d; det_copy.stats = nullptr; det_copy.stats_done = nullptr; d = &det_copy;
    }
    {
        int ran = 0;
        const int h = try_conv_halo(d, reinterpret_cast<hipStream_t>(stream), &ran);   // 3x3/s1: LDS-staged halo kernel
        if (h != 0) {
            if (h > 0 && ran && d->stats_done) *d->stats_done = 1;
            return h < 0 ? h : MFX_OK;
        }
    }
    ConvGeom g;
    g.H = d->H; g.W = d->W; g.Ho = d->Ho; g.Wo = d->Wo; g.x_pixstride = d->x_pixstride; g.lgC = ilog2(d->Ck);
    g.kh = d->kh; g.kw = d->kw; g.inv_kw = (65536 + d->kw - 1) / d->kw; g.stride = d->stride;
    g.pad_h = d->pad_h; g.pad_w = d->pad_w; g.dil_w = d->dil_w; g.M = d->M;
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = d->res; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = d->ldres;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MFX_F32) {
        if (d->out_dtype != MFX_F32) return mfx_fail(MFX_ERR_UNSUPPORTED, "conv2d: f32 input needs f32 output");
        return dispatch_conv<float, float>(d, g, ep, st);
    }
    if (d->dtype == MFX_F16X2) return dispatch_conv<f32s_t, f32s_t>(d, g, ep, st);
    if (d->dtype == MFX_F16) {
        if (d->out_dtype == MFX_F16) return dispatch_conv<half_t, half_t>(d, g, ep, st);
        return dispatch_conv<half_t, float>(d, g, ep, st);
    }
    if (d->out_dtype == MFX_BF16) return dispatch_conv<bf16_t, bf16_t>(d, g, ep, st);
    if (d->res) return mfx_fail(MFX_ERR_UNSUPPORTED, "conv2d: residual needs out_dtype == dtype");
    return dispatch_conv<bf16_t, float>(d, g, ep, st);
}

namespace mfx {
template <typename T, int BM, int BN, int WM, int WN, int KC>
static int launch_cat(const mfx_cat_desc* d, const CatSegs& s, EpiArgs ep, hipStream_t st) {
    ep.tiles_n = d->Cout_pad / BN;
    ep.nk = d->K_pad / (KC * ElemTraits<T>::ELEMS);
    const int tiles = cdiv(d->M, BM) * ep.tiles_n;
    auto k = cat_igemm_kernel<T, BM, BN, WM, WN, KC>;
    constexpr int smem = TileSmem<BM, BN, KC>::bytes;
    static bool attr_set = false;
    if (!attr_set) { int rc = set_smem(k, smem); if (rc) return rc; attr_set = true; }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(WM * WN * 64), smem, st, s, reinterpret_cast<const T*>(d->w), ep);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}
template <typename T> static int dispatch_cat(const mfx_cat_desc* d, const CatSegs& s, const EpiArgs& ep, hipStream_t st) {
    const int N = d->Cout_pad;
    if (N % 64 != 0) return mfx_fail(MFX_ERR_ARG, "cat_conv1x1: Cout_pad must be a multiple of 64");
    int tile = pick_conv_tile(d->M, N);
    if (N == 64) tile = T_64x64;                                 // (level-2 Root: 24.5 -> 20.3 us)
    if (g_opt_cat_tile && tile_fits(g_opt_cat_tile, N)) tile = g_opt_cat_tile;
    switch (tile) {
        case T_128x64: return launch_cat<T, 128, 64, 4, 1, 8>(d, s, ep, st);
        case T_128x128: return launch_cat<T, 128, 128, 2, 2, 8>(d, s, ep, st);
        case T_64x128: return launch_cat<T, 64, 128, 2, 2, 8>(d, s, ep, st);
        default: return launch_cat<T, 64, 64, 2, 2, 8>(d, s, ep, st);
    }
}
}  // namespace mfx

extern "C" int mfx_cat_conv1x1_nhwc(const mfx_cat_desc* d, void* stream) {
    if (!d || !d->w || !d->y) return mfx_fail(MFX_ERR_ARG, "cat_conv1x1: null pointer");
    if (d->nseg < 1 || d->nseg > MFX_MAX_SEG) return mfx_fail(MFX_ERR_ARG, "cat_conv1x1: 1..9 segments");
    const int elems = (d->dtype == MFX_F32 || d->dtype == MFX_F16X2) ? 4 : 8;
    if (!is_pow2(d->Cseg) || d->Cseg < 8 * elems) return mfx_fail(MFX_ERR_ARG, "cat_conv1x1: Cseg must be a power of two >= 128 bytes");
    if (d->K_pad != d->nseg * d->Cseg) return mfx_fail(MFX_ERR_ARG, "cat_conv1x1: K_pad != nseg*Cseg");
    if (d->Cout % elems != 0) return mfx_fail(MFX_ERR_ARG, "cat_conv1x1: Cout must be a multiple of the chunk");
    if (d->M <= 0) return MFX_OK;
    CatSegs s;
    for (int i = 0; i < MFX_MAX_SEG; ++i) {
        s.src[i] = i < d->nseg ? d->src[i] : nullptr; s.stride[i] = i < d->nseg ? d->stride[i] : 0; s.off[i] = i < d->nseg ? d->off[i] : 0;
        if (i < d->nseg && !d->src[i]) return mfx_fail(MFX_ERR_ARG, "cat_conv1x1: null segment");
    }
    s.lgC = ilog2(d->Cseg); s.M = d->M;
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = d->res; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = d->ldres;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MFX_F32) return dispatch_cat<float>(d, s, ep, st);
    if (d->dtype == MFX_BF16) return dispatch_cat<bf16_t>(d, s, ep, st);
    if (d->dtype == MFX_F16) return dispatch_cat<half_t>(d, s, ep, st);
    if (d->dtype == MFX_F16X2) return dispatch_cat<f32s_t>(d, s, ep, st);
    return mfx_fail(MFX_ERR_ARG, "cat_conv1x1: bad dtype");
}

namespace mfx {
template <typename T, int BM, int BN, int WM, int WN, int KC>
static int launch_dcn(const mfx_dcn_desc* d, const DcnGeom& g, EpiArgs ep, hipStream_t st) {
    ep.tiles_n = d->Cout_pad / BN;
    ep.nk = d->K_pad / (KC * ElemTraits<T>::ELEMS);
    const int tiles = cdiv(g.M, BM) * ep.tiles_n;
    auto k = dcn_igemm_kernel<T, BM, BN, WM, WN, KC>;
    constexpr int smem = TileSmem<BM, BN, KC>::bytes;
    static bool attr_set = false;
    if (!attr_set) { int rc = set_smem(k, smem); if (rc) return rc; attr_set = true; }
    // split-K on small maps (12x40 .. 24x80: 60 .. 480 tiles with K = 2304 .. 4608): more workgroups, same gather work
    int ksplit = 1;
    if (d->workspace && g_opt_dcn_ksplit != 1 && d->Cout == d->Cout_pad) {
        ksplit = g_opt_dcn_ksplit > 1 ? g_opt_dcn_ksplit : (tiles <= 256 ? 3 : 1);   // layer_bench: 512->256@12x40 83 -> 58 us, 256->64@24x80 49 -> 39; 480-tile layers lose
        ksplit = std::min(ksplit, std::min(9, ep.nk / 4));
        while (ksplit > 1 && ((ep.nk + ksplit - 1) / ksplit) * (ksplit - 1) >= ep.nk) --ksplit;
        if ((size_t)ksplit * g.M * d->Cout_pad * sizeof(float) > (size_t)d->workspace_bytes) ksplit = 1;
    }
    if (ksplit > 1) {
        ep.ksplit = ksplit; ep.ws = reinterpret_cast<float*>(d->workspace); ep.ws_ld = d->Cout_pad;
        hipLaunchKernelGGL(k, dim3(tiles, ksplit), dim3(WM * WN * 64), smem, st, reinterpret_cast<const T*>(d->x), d->offmask,
                           reinterpret_cast<const T*>(d->w), g, ep);
        MFX_HIP_CHECK(hipGetLastError());
        const long chunks = (long)g.M * (d->Cout / ElemTraits<T>::ELEMS);
        const int blocks = (int)std::min<long>((chunks + 255) / 256, 4096);
        hipLaunchKernelGGL(splitk_finalize_kernel<T>, dim3(blocks), dim3(256), 0, st, ep.ws, ksplit, (size_t)g.M * d->Cout_pad, d->Cout_pad,
                           d->scale, d->shift, (const T*)nullptr, 0, reinterpret_cast<T*>(d->y), d->ldy, g.M, d->Cout, d->act);
        MFX_HIP_CHECK(hipGetLastError());
        return MFX_OK;
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(WM * WN * 64), smem, st, reinterpret_cast<const T*>(d->x), d->offmask,
                       reinterpret_cast<const T*>(d->w), g, ep);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}
template <typename T> static int dispatch_dcn(const mfx_dcn_desc* d, const DcnGeom& g, const EpiArgs& ep, hipStream_t st) {
    const int N = d->Cout_pad;
    if (N % 64 != 0) return mfx_fail(MFX_ERR_ARG, "dcn: Cout_pad must be a multiple of 64");
    int tile = N == 64 ? (cdiv(g.M, 128) >= 512 ? T_128x64 : T_64x64)
                       : (N % 128 == 0 && cdiv(g.M, 64) * (N / 128) >= 256 ? T_64x128 : T_64x64);
    if (g_opt_dcn_tile && tile_fits(g_opt_dcn_tile, N)) tile = g_opt_dcn_tile;
    const bool kc8 = g_opt_kc != 4 && d->C >= 8 * ElemTraits<T>::ELEMS;
    switch (tile) {
        case T_128x64: return kc8 ? launch_dcn<T, 128, 64, 4, 1, 8>(d, g, ep, st) : launch_dcn<T, 128, 64, 4, 1, 4>(d, g, ep, st);
        case T_64x128: return kc8 ? launch_dcn<T, 64, 128, 2, 2, 8>(d, g, ep, st) : launch_dcn<T, 64, 128, 2, 2, 4>(d, g, ep, st);
        case T_128x128: return kc8 ? launch_dcn<T, 128, 128, 2, 2, 8>(d, g, ep, st) : launch_dcn<T, 128, 128, 2, 2, 4>(d, g, ep, st);
        default: return kc8 ? launch_dcn<T, 64, 64, 2, 2, 8>(d, g, ep, st) : launch_dcn<T, 64, 64, 2, 2, 4>(d, g, ep, st);
    }
}
}  // namespace mfx

static bool dcn_fuses_offset_conv(const mfx_dcn_desc* d) { return dcn_lds_fuses_offset_conv(d) || dcn_patch_fuses_offset_conv(d); }
extern "C" int mfx_dcn_fuses_offset_conv(const mfx_dcn_desc* d) { return (d && d->x && dcn_fuses_offset_conv(d)) ? 1 : 0; }

extern "C" int mfx_dcn_nhwc(const mfx_dcn_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->y) return mfx_fail(MFX_ERR_ARG, "dcn: null pointer");
    if (!d->offmask && !dcn_fuses_offset_conv(d)) return mfx_fail(MFX_ERR_ARG, "dcn: offmask is NULL (the offset conv is computed inside the kernel only where mfx_dcn_fuses_offset_conv says so)");
    const int elems = (d->dtype == MFX_F32 || d->dtype == MFX_F16X2) ? 4 : 8;
    if (d->dtype != MFX_F32 && d->dtype != MFX_BF16 && d->dtype != MFX_F16 && d->dtype != MFX_F16X2) return mfx_fail(MFX_ERR_ARG, "dcn: bad dtype");
    if (!is_pow2(d->C) || d->C < 4 * elems) return mfx_fail(MFX_ERR_ARG, "dcn: C must be a power of two >= 64 bytes of channels");
    if (d->kh * d->kw > 9) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn: at most 9 taps (offmask row is 32 floats)");
    if (d->K_pad != d->kh * d->kw * d->C) return mfx_fail(MFX_ERR_ARG, "dcn: K_pad != kh*kw*C");
    if (d->Cout % elems != 0 || d->Cout > d->Cout_pad) return mfx_fail(MFX_ERR_ARG, "dcn: bad Cout");
    DcnGeom g;
    g.H = d->H; g.W = d->W; g.C = d->C; g.lgC = ilog2(d->C); g.Ho = d->Ho; g.Wo = d->Wo; g.kh = d->kh; g.kw = d->kw;
    g.inv_kw = (65536 + d->kw - 1) / d->kw; g.stride = d->stride; g.pad = d->pad; g.dil = d->dil; g.M = d->B * d->Ho * d->Wo;
    g.stride_w = d->nonsquare ? d->stride_w : d->stride; g.pad_w = d->nonsquare ? d->pad_w : d->pad; g.dil_w = d->nonsquare ? d->dil_w : d->dil;
    if (d->nonsquare && (d->stride_w < 1 || d->dil_w < 1 || d->pad_w < 0)) return mfx_fail(MFX_ERR_ARG, "dcn: bad per-axis geometry");
    if (g.M <= 0) return MFX_OK;
    if ((size_t)d->B * d->H * d->W * d->C * (elems == 8 ? 2 : 4) >= ((size_t)1 << 32))
        return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn: input tensor of 4 GB or more (the gather kernels address it with 32-bit byte offsets)");
    if (!d->nonsquare) {                                      // (the LDS-patch / wave kernels are built for square geometry)
        int h = try_dcn_lds(d, reinterpret_cast<hipStream_t>(stream));
        if (h != 0) return h < 0 ? h : MFX_OK;
        h = try_dcn_patch(d, reinterpret_cast<hipStream_t>(stream));
        if (h != 0) return h < 0 ? h : MFX_OK;
        h = try_dcn_wave(d, reinterpret_cast<hipStream_t>(stream));
        if (h != 0) return h < 0 ? h : MFX_OK;
    }
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = nullptr; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = 0;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MFX_F16) return dispatch_dcn<half_t>(d, g, ep, st);
    if (d->dtype == MFX_F16X2) return dispatch_dcn<f32s_t>(d, g, ep, st);
    return d->dtype == MFX_F32 ? dispatch_dcn<float>(d, g, ep, st) : dispatch_dcn<bf16_t>(d, g, ep, st);
}

MFX_RANGE_FLAG_ACCESSOR(conv_kernels)      // split-precision range sentinel of this translation unit (common.h)
