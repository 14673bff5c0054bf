// DCNv2 backward, second generation (training path, NHWC, 3x3 / stride 1 / pad 1 / dilation 1, C a multiple of 64).
// Reference math: model/backbone/DCNv2/src/cuda/dcn_v2_cuda.cu:206-335 and dcn_v2_im2col_cuda.cu:197-327
// (col2im for grad_input, col2im_coord for grad_offset / grad_mask, im2col + GEMM for grad_weight).
//
// The first generation (dcn_bwd.hip) scattered every (pixel, tap) sample's four corners into grad_input with global fp32
// atomics -- 36 atomic bursts per input element, 2.5 ms for one 64->64 @ 96x320 layer at B=8, 100x its HBM floor -- and
// re-sampled the columns a second time on the vector ALUs for grad_weight.  Here (launch order):
//
//   1. d(columns) = dy x W^T as a 1x1 implicit GEMM on the matrix cores (gcol, activation dtype);
//   2. dcn_bwd_sample_kernel: one lane group per (pixel, tap) sample blends the four corners once and produces grad_offset /
//      grad_mask (the sigmoid derivative of the mask logit folded in: the gradient of the raw 27-channel offset/mask conv
//      output) and the modulated columns col[m][tap*C + c], which turn grad_weight into a plain MFMA GEMM (step 5);
//   3. dcn_bwd_tile_kernel: the INPUT gradient is owned by tiles.  One workgroup owns an 8 x 16 tile of grad_input pixels
//      (x a slice of <= 128 channels).  Phase 1 walks the samples of the tile grown by D = 8 pixels and BINS every
//      (sample, corner) pair that lands in the tile into that pixel's LDS list (one integer LDS atomic per pair); phase 3
//      lets each pixel's lane group walk its list, gather the d(columns) rows, accumulate in registers and store the pixel
//      ONCE in the activation dtype -- no global atomics, no fp32 grad_input buffer, no zero-fill, no narrowing pass;
//   4. dcn_bwd_far_kernel: a corner further than D pixels from its sample's tile ("far": offsets beyond ~8 px), or one that
//      found its pixel's list full, was appended to a global list instead; those are added into the finished grad_input with
//      atomics, so every offset stays exact: a corner (sample m, pixel p) is handled by the owner of tile(p) iff m lies in
//      tile(p) grown by D, and by the far list otherwise -- exactly once;
//   5. grad_weight = dy^T x col (conv_wgrad_mfma_kernel, "direct" operand mode), grad_bias = column sums of dy.
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"
#include "fill.h"
#include <algorithm>
#include <type_traits>

// train_kernels.hip: weight gradient with a dense [M][K] A operand (direct = 1), written as (Cout, Cin, kh, kw)
int mfx_internal_conv_wgrad(const void* x, const void* dy, float* dw, int B, int H, int W, int x_pixstride, int Ck,
                            int kh, int kw, int stride, int pad_h, int pad_w, int Ho, int Wo, int Cout, int ldy,
                            int dtype, int oihw, int Cin_out, int Cout_out, void* stream, int dil_w,
                            void* workspace, size_t workspace_bytes, int direct);

int mfx_internal_wgrad_slab_sum(const float* ws, int nslab, int Cout, int Ck, int kh, int kw, float* dw_oihw, void* stream);
int mfx_internal_colsum_add(const void* x, float* out, long M, int C, int ld, int dtype, void* stream);      // train_kernels.hip: sums ADDED into a zeroed `out`

int g_opt_dcn_bt_fuse_blocks = 170; // option "dcn_bt_fuse_blocks": workgroups per tap group of the fused kernel
int g_opt_dcn_bt_fly_bias = 1;     // option "dcn_bt_fly_bias": the gcol-free sample kernel also sums grad_bias from the dy rows it loads (0: a separate column-sum pass)
int g_opt_dcn_bt_gcol_as = 1;      // option "dcn_bt_gcol_as": d(columns) = dy . W^T of the 16-bit layers on the activation-stationary GEMM (gemm_as.hip); 0: the tiled 1x1 kernel
int g_opt_dcn_bt_fly = 1;          // option "dcn_bt_fly": 64 -> 64 16-bit layers rebuild d(columns) from dy inside both consumers (no [M][9C] matrix in memory)
long g_cnt_dcn_bt_fly = 0;         // counter "dcn_bt_fly": backward calls that took the gcol-free form
int g_opt_dcn_bt_fuse_wgrad = 1;   // option "dcn_bt_fuse_wgrad": 64 -> 64 bf16 layers accumulate grad_weight inside the sample kernel (no columns in memory)
int g_opt_dcn_bt_fuse_min_chunks = 1024; // option "dcn_bt_fuse_min_chunks": fewer 32-pixel chunks than this keep the unfused kernels (tests lower it)
long g_cnt_dcn_bt_fused = 0;   // counter "dcn_bt_fused": launches of dcn_bwd_sample_wgrad_kernel since process start
int g_opt_dcn_bt_cs = 0;       // option "dcn_bt_cs": channel slice of the tile kernel for C >= 128 (0 = by workgroup count, 64, 128)
int g_opt_dcn_bt_cs_wgs = 1000; // option "dcn_bt_cs_wgs": below this many 128-channel workgroups the tile kernel takes 64-channel slices
int g_opt_dcn_bt_dbg = 0;      // option "dcn_bt_dbg": experiment switches of dcn_bwd_tile_kernel (0 in production)

// experiment switches (skip a phase, drop an operand stream: timing probes with WRONG results) exist in probe builds only
#ifdef MFX_PROBES
#define BT_PROBE(g, bits) ((g).dbg & (bits))
#else
#define BT_PROBE(g, bits) false
#endif

namespace mfx {

#ifndef BT_D_VAL
#define BT_D_VAL 8            // ring of the candidate window (pixels): a corner further than this from its sample's tile goes to the far list
#endif
constexpr int BT_TH = 8, BT_TW = 16, BT_D = BT_D_VAL, BT_NPIX = BT_TH * BT_TW;
constexpr int BT_FCAP = 512;   // far corners staged per workgroup before one global reservation
#ifndef BT_ROWS_IN_FLIGHT
#define BT_ROWS_IN_FLIGHT 8
#endif
#ifndef BT_LCAP_BF16
#define BT_LCAP_BF16 62       // 4-byte list entries per target pixel (mean 36, sigma ~6): 32 KB of lists = four workgroups per CU
#endif
constexpr int BT_CH = BT_TH + 2 * BT_D, BT_CW = BT_TW + 2 * BT_D;        // candidate window (24 x 32 pixels)
constexpr int BT_WIN = BT_CH * BT_CW, BT_ROUNDS = (BT_WIN + 255) / 256;  // window pixels, and how many of them a thread takes (3 at D = 8)
static_assert(BT_CH <= 32 && BT_CW <= 32, "list entries keep window coordinates in 5 + 5 bits");

struct BtGeom { int B, H, W, C, tiles_x, tiles_y, Kp, CS, nslices, dbg;
                int raw_mask;      // 1: d_raw[18 + tap] = d loss / d mask (the `_ext` contract: the mask is an INPUT there); 0: through the sigmoid of the mask logit
                int raw16; };      // 1: d_raw rows are written in the activation type T (16-bit layers: what the offset conv's backward consumes), 0: fp32

// one channel of a pixel's d_raw row (32 channels per pixel)
template <typename T> __device__ __forceinline__ void raw_put(float* graw, int raw16, size_t idx, float v) {
    if (raw16) ElemTraits<T>::store(reinterpret_cast<T*>(graw) + idx, v); else graw[idx] = v;
}

struct SampGeo { int h0, w0; float lh, lw, mask; int inside; };

// sampling geometry of (pixel (my,mx), tap): same rules as the forward sampler (dcn_v2_im2col_cuda.cu:25-54,178-189)
__device__ __forceinline__ SampGeo samp_geo(const BtGeom& g, const float* __restrict__ om_row, int my, int mx, int tap) {
    SampGeo s;
    const int th = (tap * 11) >> 5, tw = tap - th * 3;        // tap / 3 for 0..8
    const float h = (float)(my - 1 + th) + om_row[2 * tap];
    const float w = (float)(mx - 1 + tw) + om_row[2 * tap + 1];
    s.inside = (h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W) ? 1 : 0;
    const float hf = floorf(h), wf = floorf(w);
    s.lh = h - hf; s.lw = w - wf;
    // clamp before the int conversion: a wild (or NaN) offset must not overflow; such a sample is not `inside` anyway
    s.h0 = (int)fminf(fmaxf(hf, -4.f), 32000.f);
    s.w0 = (int)fminf(fmaxf(wf, -4.f), 32000.f);
    s.mask = om_row[18 + tap];
    return s;
}

// two transposed 8-byte LDS reads (ds_read_b64_tr_b16: the 16-bit elements of a 16-lane group's rows come back transposed), at `a` and `a + 512`
__device__ __forceinline__ u32x4 bt_tr2(uint32_t a) {
    uint64_t l, h;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512\n\ts_waitcnt lgkmcnt(0)" : "=&v"(l), "=&v"(h) : "v"(a) : "memory");
    return u32x4{(uint32_t)l, (uint32_t)(l >> 32), (uint32_t)h, (uint32_t)(h >> 32)};
}

// sum over the LPS (8 or 16) consecutive lanes that share one sample, on the VALU's DPP path (no LDS crossbar):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
template <int LPS> __device__ __forceinline__ float bt_group_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    if (LPS == 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

// the primary templates serve both 16-bit activation types (bf16_t, half_t: eight elements per 16-byte chunk); float is specialised
template <typename T> __device__ __forceinline__ void bt_load8(const T* p, float (&v)[8]) {
    ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(p), v);
}
template <> __device__ __forceinline__ void bt_load8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <typename T> __device__ __forceinline__ void bt_store8(T* p, const float (&v)[8]) {
    *reinterpret_cast<u32x4*>(p) = ElemTraits<T>::pack(v);
}
template <> __device__ __forceinline__ void bt_store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

template <typename T> __device__ __forceinline__ void bt_store4(T* p, const f32x4& v) {
    *reinterpret_cast<uint2*>(p) = uint2{ElemTraits<T>::pack2(v[0], v[1]), ElemTraits<T>::pack2(v[2], v[3])};
}
template <> __device__ __forceinline__ void bt_store4<float>(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

// LPS = lanes per pixel / sample: every lane owns 8 consecutive channels (one 16-byte chunk in bf16), so a channel slice is
// CS = 8 * LPS channels and a wavefront works on 64 / LPS samples (or target pixels) at once.
//
// Measured dead ends (MI355X, 64 -> 64 @ 96x320, B=8; profiles/r02_dcn_bwd.md): one sample per wavefront with lanes over
// channels is latency-bound (five 128-byte loads in flight per wave: 4.7 ms); accumulating the tile with ds_add_f32 costs
// ~140 cycles per wave instruction (the LDS float-atomic unit retires about one lane per clock: 2.0 ms for the 566 M lane
// adds of this layer).  So the tile is not accumulated by atomics at all: phase 1 BINS every (sample, corner) pair that
// lands in the tile into a per-pixel list (one integer LDS atomic per pair, 64x fewer than per channel), and phase 3 lets
// each target pixel's lane group walk its list and accumulate in registers.
template <typename T> struct Raw8 {          // 16-bit activations
    u32x4 v;
    __device__ __forceinline__ void load(const T* p) { v = *reinterpret_cast<const u32x4*>(p); }
    __device__ __forceinline__ void zero() { v = u32x4{0u, 0u, 0u, 0u}; }
    __device__ __forceinline__ void unpack(float (&f)[8]) const { ElemTraits<T>::unpack(v, f); }
};
template <> struct Raw8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
    __device__ __forceinline__ void zero() { a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a; }
    __device__ __forceinline__ void unpack(float (&f)[8]) const { f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3]; }
};

// List entries.  fp32 maps keep (sample, fp32 weight) pairs: 8 bytes, 47 per pixel in the 48 KB that leave three workgroups
// per CU.  bf16 maps pack the sample as window coordinates (5 + 5 bits) + tap (4) and the weight's exponent and top ten
// mantissa bits (18 bits, the weight is non-negative; 2^-11 relative: below the bf16 operands it multiplies, at the rounding of fp16 ones) into 4 bytes:
// 94 per pixel, so a list practically never overflows (mean 36 entries, sigma ~6) and the far path is left to far offsets.
template <typename T> struct BtEntry {        // 16-bit activations
    typedef uint32_t type;
    static constexpr int LCAP = BT_LCAP_BF16;
    static __device__ __forceinline__ uint32_t make(int wy, int wx, int tap, float w) {
        return ((uint32_t)wy << 27) | ((uint32_t)wx << 22) | ((uint32_t)tap << 18) | (((__float_as_uint(w) + 0x1000u) >> 13) & 0x3ffffu);
    }
    static __device__ __forceinline__ void read(uint32_t e, int& wy, int& wx, int& tap, float& w) {
        wy = (int)(e >> 27); wx = (int)((e >> 22) & 31); tap = (int)((e >> 18) & 15); w = __uint_as_float((e & 0x3ffffu) << 13);
    }
};
template <> struct BtEntry<float> {
    typedef uint2 type;
    static constexpr int LCAP = 47;
    static __device__ __forceinline__ uint2 make(int wy, int wx, int tap, float w) { return uint2{((uint32_t)wy << 16) | ((uint32_t)wx << 4) | (uint32_t)tap, __float_as_uint(w)}; }
    static __device__ __forceinline__ void read(const uint2& e, int& wy, int& wx, int& tap, float& w) {
        wy = (int)(e.x >> 16); wx = (int)((e.x >> 4) & 0xfff); tap = (int)(e.x & 15); w = __uint_as_float(e.y);
    }
};

template <typename T, int LPS>
__global__ __launch_bounds__(256) void dcn_bwd_tile_kernel(const float* __restrict__ om, const T* __restrict__ gcol, BtGeom g,
                                                          T* __restrict__ dx, int* __restrict__ far_count,
                                                          u32x4* __restrict__ far_list, int far_cap) {
    using EN = BtEntry<T>;
    typedef typename EN::type entry_t;
    constexpr int CS = 8 * LPS, SPW = 64 / LPS, LCAP = EN::LCAP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    entry_t* lists = reinterpret_cast<entry_t*>(smem);                         // [BT_NPIX][LCAP]
    int* counts = reinterpret_cast<int*>(smem + BT_NPIX * LCAP * sizeof(entry_t));     // [BT_NPIX]
    uint2* fstage = reinterpret_cast<uint2*>(smem + BT_NPIX * LCAP * sizeof(entry_t) + BT_NPIX * 4);   // [BT_FCAP] (sample | corner << 28, weight)
    __shared__ int fcount, fbase;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = lane / LPS, cl = lane % LPS;                               // pixel slot in the wave, 8-channel group
    int tile = blockIdx.x;
    const int tx = tile % g.tiles_x; tile /= g.tiles_x;
    const int ty = tile % g.tiles_y; const int b = tile / g.tiles_y;
    const int ty0 = ty * BT_TH, tx0 = tx * BT_TW;
    const int cs0 = blockIdx.y * CS, c0 = cs0 + cl * 8;                        // slice start, this lane's first channel
    const int HW = g.H * g.W;
    const long mb = (long)b * HW;                                              // first sample index of image b

    for (int i = tid; i < BT_NPIX; i += 256) counts[i] = 0;
    if (tid == 0) fcount = 0;
    __syncthreads();

    // a corner that cannot go through the tile lists (far from its sample's tile, or a full list) is staged in LDS and, after
    // phase 1, appended to a global list with ONE counter reservation per workgroup (a per-corner atomic on the single global
    // counter serialised at ~12 ns each: +1 ms at 10 % far corners); dcn_bwd_far_kernel adds that list into dx afterwards.
    // A full stage reserves list slots one by one; the list holds every (sample, corner) pair of the map, so it cannot fill.
    auto far_entry = [&](int slot, size_t m, int tap, int hc, int wc, uint32_t wbits) {
        if (slot < far_cap) far_list[slot] = u32x4{(uint32_t)m, ((uint32_t)hc << 16) | ((uint32_t)wc << 4) | (uint32_t)tap, wbits, (uint32_t)cs0};
    };
    auto to_far = [&](size_t m, int my, int mx, int tap, int q, int hc, int wc, float w) {
        const int fs = atomicAdd(&fcount, 1);
        if (fs < BT_FCAP) fstage[fs] = uint2{((uint32_t)q << 28) | ((uint32_t)my << 16) | ((uint32_t)mx << 4) | (uint32_t)tap, __float_as_uint(w)};
        else far_entry(atomicAdd(far_count, 1), m, tap, hc, wc, __float_as_uint(w));
    };

    // ---------------- phase 1: bin the (sample, corner) pairs of the candidate window by target pixel ----------------
    // window = tile grown by D pixels (BT_CH x BT_CW = 3 x 256 pixels): one thread per pixel, its 9 taps from registers.
    // Own samples (window pixel inside the tile): corners in the tile are listed; corners elsewhere are left to the owner of
    // their tile unless this pixel lies outside that tile's window ("far").  Ring samples: only corners inside the tile.
    for (int round = 0; round < (BT_PROBE(g, 4) ? 0 : BT_ROUNDS); ++round) {
        const int cp = round * 256 + tid;
        if (BT_WIN % 256 != 0 && cp >= BT_WIN) continue;
        const int wy = cp / BT_CW, wx = cp - wy * BT_CW;
        const int my = ty0 - BT_D + wy, mx = tx0 - BT_D + wx;
        const bool own = wy >= BT_D && wy < BT_D + BT_TH && wx >= BT_D && wx < BT_D + BT_TW;
        if (my < 0 || my >= g.H || mx < 0 || mx >= g.W) continue;
        const size_t m = (size_t)(mb + (long)my * g.W + mx);
        const float* r = om + m * 32;
        float o[28];
#pragma unroll
        for (int q4 = 0; q4 < 28; q4 += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(r + q4);
            o[q4] = t[0]; o[q4 + 1] = t[1]; o[q4 + 2] = t[2]; o[q4 + 3] = t[3];
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int th = tap / 3, tw = tap - th * 3;
            const float h = (float)(my - 1 + th) + o[2 * tap], w = (float)(mx - 1 + tw) + o[2 * tap + 1];
            const float mask = o[18 + tap];
            if (!(h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W) || mask == 0.f) continue;
            const float hf = floorf(h), wf = floorf(w);
            const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
            const int h0 = (int)hf, w0 = (int)wf;                             // inside => -1 <= h0 < H: no overflow
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int hc = h0 + (q >> 1), wc = w0 + (q & 1);
                const float wq = ((q >> 1) ? lh : hh) * ((q & 1) ? lw : hw) * mask;
                if (hc < 0 || hc >= g.H || wc < 0 || wc >= g.W || wq == 0.f) continue;
                const int ly = hc - ty0, lx = wc - tx0;
                if (ly >= 0 && ly < BT_TH && lx >= 0 && lx < BT_TW) {
                    const int p = ly * BT_TW + lx;
                    const int slot = atomicAdd(&counts[p], 1);
                    if (slot < LCAP) lists[p * LCAP + slot] = EN::make(wy, wx, tap, wq);
                    else to_far(m, my, mx, tap, q, hc, wc, wq);               // list full: exact fallback
                } else if (own) {
                    const int cty0 = (hc / BT_TH) * BT_TH, ctx0 = (wc / BT_TW) * BT_TW;
                    const bool near = my >= cty0 - BT_D && my < cty0 + BT_TH + BT_D && mx >= ctx0 - BT_D && mx < ctx0 + BT_TW + BT_D;
                    if (!near) to_far(m, my, mx, tap, q, hc, wc, wq);
                }
            }
        }
    }
    __syncthreads();
    {   // flush the staged far corners: one reservation on the global counter for the whole workgroup
        const int nf = min(fcount, BT_FCAP);
        if (nf > 0) {
            if (tid == 0) fbase = atomicAdd(far_count, nf);
            __syncthreads();
            for (int i = tid; i < nf; i += 256) {
                const uint2 en = fstage[i];
                const int q = (int)(en.x >> 28), emy = (int)((en.x >> 16) & 0xfff), emx = (int)((en.x >> 4) & 0xfff), etap = (int)(en.x & 15);
                const size_t m = (size_t)(mb + (long)emy * g.W + emx);
                const SampGeo sg = samp_geo(g, om + m * 32, emy, emx, etap);
                far_entry(fbase + i, m, etap, sg.h0 + (q >> 1), sg.w0 + (q & 1), en.y);
            }
        }
    }

    // ---------------- phase 3: every target pixel's lane group walks its list, accumulates in registers, stores once ----------------
    for (int p = wv * SPW + sl; p < (BT_PROBE(g, 1) ? 0 : BT_NPIX); p += 4 * SPW) {
        const int y = ty0 + p / BT_TW, xx = tx0 + (p % BT_TW);
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int n = min(counts[p], LCAP);
        const entry_t* lp = lists + p * LCAP;
        auto row = [&](const entry_t& en, float& w) {
            int wy, wx, tap;
            EN::read(en, wy, wx, tap, w);
            const size_t m = (size_t)(mb + (long)(ty0 - BT_D + wy) * g.W + (tx0 - BT_D + wx));
            return gcol + m * g.Kp + tap * g.C + c0;
        };
        int e = 0;
        constexpr int NF = BT_ROWS_IN_FLIGHT;                                 // d(columns) rows in flight per lane group
        for (; e + NF <= n; e += NF) {
            entry_t en[NF]; Raw8<T> gr[NF]; float w[NF];
#pragma unroll
            for (int u = 0; u < NF; ++u) en[u] = lp[e + u];
#pragma unroll
            for (int u = 0; u < NF; ++u) gr[u].load(row(en[u], w[u]));
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                float gq[8];
                gr[u].unpack(gq);
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += w[u] * gq[k];
            }
        }
        for (; e < n; ++e) {
            float gq[8], w;
            bt_load8<T>(row(lp[e], w), gq);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += w * gq[k];
        }
        if (y < g.H && xx < g.W) bt_store8<T>(dx + ((size_t)b * HW + (size_t)y * g.W + xx) * g.C + c0, a);
    }
}

// far corners (rare: offsets beyond the 8-pixel ring, or an over-full tile list): entry = (sample m, corner pixel, tap,
// weight, first channel of the slice); 8 lanes per entry x 8 channels per lane per pass, added into the finished dx with
// atomics (fp32 atomics, or packed bf16 atomics: each add rounds to bf16, which the few far contributions of a pixel can afford)
__device__ __forceinline__ void bt_atomic_add8(float* p, const float (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) unsafeAtomicAdd(p + k, v[k]);
}
__device__ __forceinline__ void bt_atomic_add8(bf16_t* p, const float (&v)[8]) {
    typedef short s2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const bf16x2 t = __builtin_convertvector((f32x2){v[k], v[k + 1]}, bf16x2);
        __builtin_amdgcn_global_atomic_fadd_v2bf16((s2_t __attribute__((address_space(1)))*)(p + k), __builtin_bit_cast(s2_t, t));
    }
}
__device__ __forceinline__ void bt_atomic_add8(half_t* p, const float (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const f16x2 t = __builtin_convertvector((f32x2){v[k], v[k + 1]}, f16x2);
        __builtin_amdgcn_global_atomic_fadd_v2f16((f16x2 __attribute__((address_space(1)))*)(p + k), t);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dcn_bwd_far_kernel(const T* __restrict__ gcol, const u32x4* __restrict__ far_list,
                                                         const int* __restrict__ far_count, int far_cap, BtGeom g,
                                                         T* __restrict__ dx) {
    const int n = min(*far_count, far_cap);
    const int HW = g.H * g.W, cl = threadIdx.x & 7;
    for (long e = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 3; e < n; e += ((long)gridDim.x * blockDim.x) >> 3) {
        const u32x4 en = far_list[e];
        const size_t m = en.x;
        const int hc = (int)(en.y >> 16), wc = (int)((en.y >> 4) & 0xfff), tap = (int)(en.y & 15), cs0 = (int)en.w;
        const float w = __uint_as_float(en.z);
        const size_t bimg = m / HW;
        const T* gp = gcol + m * g.Kp + tap * g.C + cs0;
        T* f = dx + (bimg * HW + (size_t)hc * g.W + wc) * g.C + cs0;
        for (int c = cl * 8; c < g.CS; c += 64) {
            float gq[8];
            bt_load8<T>(gp + c, gq);
#pragma unroll
            for (int k = 0; k < 8; ++k) gq[k] *= w;
            bt_atomic_add8(f + c, gq);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Deterministic mode (option "deterministic"): grad_input in 64-bit FIXED POINT.  The tile lists above are filled in the order
// LDS atomics happen to retire and the far corners arrive through float atomics, so the floating-point sum of a pixel depends
// on the run.  Integer addition commutes: every (sample, tap, corner) contribution w * d(columns) is converted to a signed
// 28.36 fixed-point number (round to nearest of an exactly scaled double: |value| < 1.3e8, resolution 1.5e-11) and added into
// an int64 map with integer atomics; a second pass rounds the map to the activation dtype.  Any arrival order gives the same
// bits.  One lane group of eight per (pixel, tap), eight channels per lane and pass (the first generation's scatter shape:
// ~2.5 ms for 64 -> 64 @ 96x320, B = 8 -- a debugging / testing mode, not the fast path).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr double BT_FIX_SCALE = 68719476736.0;            // 2^36

template <typename T>
__global__ __launch_bounds__(256) void dcn_bwd_dx_fixed_kernel(const float* __restrict__ om, const T* __restrict__ gcol, BtGeom g,
                                                              unsigned long long* __restrict__ acc) {
    const int HW = g.H * g.W, cl = threadIdx.x & 7;
    const long nsamp = (long)g.B * HW * 9;
    for (long s = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 3; s < nsamp; s += ((long)gridDim.x * blockDim.x) >> 3) {
        const long m = s / 9;
        const int tap = (int)(s - m * 9);
        const int b = (int)(m / HW), rem = (int)(m - (long)b * HW), my = rem / g.W, mx = rem - my * g.W;
        const SampGeo sg = samp_geo(g, om + m * 32, my, mx, tap);
        if (!sg.inside || sg.mask == 0.f) continue;
        const float hh = 1.f - sg.lh, hw = 1.f - sg.lw;
        for (int c = cl * 8; c < g.C; c += 64) {
            float gq[8];
            bt_load8<T>(gcol + m * g.Kp + tap * g.C + c, gq);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int hc = sg.h0 + (q >> 1), wc = sg.w0 + (q & 1);
                const float wq = ((q >> 1) ? sg.lh : hh) * ((q & 1) ? sg.lw : hw) * sg.mask;
                if (hc < 0 || hc >= g.H || wc < 0 || wc >= g.W || wq == 0.f) continue;
                unsigned long long* ap = acc + ((size_t)b * HW + (size_t)hc * g.W + wc) * g.C + c;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const double v = fmin(fmax((double)(wq * gq[k]) * BT_FIX_SCALE, -9.0e18), 9.0e18);
                    atomicAdd(ap + k, (unsigned long long)__double2ll_rn(v));
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dcn_bwd_dx_unfix_kernel(const unsigned long long* __restrict__ acc, T* __restrict__ dx, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        ElemTraits<T>::store(dx + i, (float)((double)(long long)acc[i] * (1.0 / BT_FIX_SCALE)));
}

// weight (Cout,C,9) fp32 -> wT[K][Cout] (k = tap*C + c), compute dtype: the B operand of d(columns) = dy x W
template <typename T>
__global__ void bt_pack_weight_t(const float* __restrict__ w, T* __restrict__ wT, int Cout, int C, int* __restrict__ far_count, float* __restrict__ dbias) {
    if (blockIdx.x == 0 && threadIdx.x < 2) far_count[threadIdx.x] = 0;       // the far-corner counter of this call (was a separate 8-byte fill launch)
    if (blockIdx.x == 0) for (int o = threadIdx.x; o < Cout; o += blockDim.x) dbias[o] = 0.f;      // ... and the zeros the bias-gradient sums are added into (another one)
    const long total = (long)9 * C * Cout;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i % Cout);
        const int k = (int)(i / Cout);
        const int tap = k / C, c = k - tap * C;
        ElemTraits<T>::store(wT + i, w[((size_t)o * C + c) * 9 + tap]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The per-sample half of the backward pass as its own kernel: grad_offset / grad_mask (gradient of the raw offset/mask conv
// output) and the modulated columns for the weight-gradient GEMM.  Inside dcn_bwd_tile_kernel this phase ran at the tile
// kernel's occupancy (three workgroups per CU: 52 KB of corner lists each) and took 400 of its 610 us on 64 -> 64 @ 96x320,
// B = 8 -- five dependent 16-byte gathers per sample with three waves per SIMD to hide them.  Here there is no LDS, the raw
// offset/mask values of the NEXT sample are prefetched while the current one is blended, every sample's channel slices are
// looped inside the lane group (no atomics on the offset gradients, so no zero-fill either), and the 27 gradient channels of
// a pixel are written whether or not the sample lies inside the image.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T, int LPS>
__global__ __launch_bounds__(256) void dcn_bwd_sample_kernel(const T* __restrict__ x, const float* __restrict__ om,
                                                            const T* __restrict__ gcol, BtGeom g, int xsplit,
                                                            float* __restrict__ graw, T* __restrict__ col) {
    constexpr int CS = 8 * LPS, SPW = 64 / LPS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = lane / LPS, cl = lane % LPS;
    const int row = blockIdx.x;                                                // (image, output row)
    const int b = row / g.H, my = row - b * g.H;
    const int xw = (g.W + xsplit - 1) / xsplit, x_begin = blockIdx.y * xw, x_end = min(x_begin + xw, g.W);
    const int nsamp = (x_end - x_begin) * 9;
    const int HW = g.H * g.W;
    const T* xb = x + (size_t)b * HW * g.C;
    const size_t mrow = (size_t)b * HW + (size_t)my * g.W;
    const int nsl = g.C / CS;

    struct Pre { float oh, ow, mk; };
    auto prefetch = [&](int j) {
        Pre p = {0.f, 0.f, 0.f};
        if (j < nsamp) {
            const int px = j / 9, tap = j - px * 9;
            const float* r = om + (mrow + x_begin + px) * 32;
            p.oh = r[2 * tap]; p.ow = r[2 * tap + 1]; p.mk = r[18 + tap];
        }
        return p;
    };
    struct Geo { bool ok, inside; int tap, h0, w0; size_t m; float lh, lw, mask; };
    auto geo = [&](int j, const Pre& p) {
        Geo q;
        q.ok = j < nsamp;
        const int px = j / 9, mx = x_begin + px;
        q.tap = j - px * 9;
        const int th = (q.tap * 11) >> 5, tw = q.tap - th * 3;
        const float h = (float)(my - 1 + th) + p.oh, w = (float)(mx - 1 + tw) + p.ow;
        q.inside = q.ok && h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
        const float hf = floorf(h), wf = floorf(w);
        q.lh = h - hf; q.lw = w - wf; q.mask = p.mk;
        q.h0 = (int)fminf(fmaxf(hf, -4.f), 32000.f); q.w0 = (int)fminf(fmaxf(wf, -4.f), 32000.f);
        q.m = mrow + mx;
        return q;
    };
    struct Raw5 { Raw8<T> g, v[4]; };
    auto issue = [&](const Geo& q, int s) {
        Raw5 r;
        const int c0 = s * CS + cl * 8;
        if (q.inside) {
            r.g.load(gcol + q.m * g.Kp + q.tap * g.C + c0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int hc = q.h0 + (c >> 1), wc = q.w0 + (c & 1);
                if (hc >= 0 && hc < g.H && wc >= 0 && wc < g.W) r.v[c].load(xb + ((size_t)hc * g.W + wc) * g.C + c0);
                else r.v[c].zero();
            }
        }
        return r;
    };
    auto finish = [&](const Geo& q, const Raw5& r, int s, float& gh, float& gw, float& gm) {
        float cv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (q.inside) {
            float gc[8], v0[8], v1[8], v2[8], v3[8];
            r.g.unpack(gc); r.v[0].unpack(v0); r.v[1].unpack(v1); r.v[2].unpack(v2); r.v[3].unpack(v3);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // top / bottom row blends share the horizontal differences (dmcn_get_coordinate_weight, dcn_v2_im2col_cuda.cu:82-122)
                const float d10 = v1[k] - v0[k], d32 = v3[k] - v2[k];
                const float top = v0[k] + q.lw * d10, bot = v2[k] + q.lw * d32;
                const float dh = bot - top, val = top + q.lh * dh, dw = d10 + q.lh * (d32 - d10);
                cv[k] = q.mask * val;
                gm += gc[k] * val; gh += gc[k] * dh; gw += gc[k] * dw;
            }
        }
        if (q.ok) bt_store8<T>(col + q.m * g.Kp + q.tap * g.C + s * CS + cl * 8, cv);
    };
    // software pipeline: the raw offset/mask values of sample i+2 and the five gathers of sample i+1 (first channel slice) are
    // in flight while sample i is blended
    int j = wv * SPW + sl;
    const int step = 4 * SPW;
    Pre pre1 = prefetch(j + step);
    Geo q0 = geo(j, prefetch(j));
    Raw5 r0 = issue(q0, 0);
    for (int base = wv * SPW; base < nsamp; base += step, j += step) {
        const Pre pre2 = prefetch(j + 2 * step);
        const Geo q1 = geo(j + step, pre1);
        const Raw5 r1 = issue(q1, 0);
        float gh = 0.f, gw = 0.f, gm = 0.f;
        finish(q0, r0, 0, gh, gw, gm);
        for (int s = 1; s < nsl; ++s) {
            const Raw5 rs = issue(q0, s);
            finish(q0, rs, s, gh, gw, gm);
        }
        gh = bt_group_sum<LPS>(gh); gw = bt_group_sum<LPS>(gw); gm = bt_group_sum<LPS>(gm);
        if (cl == 0 && q0.ok) {
            const size_t o = (size_t)q0.m * 32;
            raw_put<T>(graw, g.raw16, o + 2 * q0.tap, gh * q0.mask); raw_put<T>(graw, g.raw16, o + 2 * q0.tap + 1, gw * q0.mask);
            raw_put<T>(graw, g.raw16, o + 18 + q0.tap, g.raw_mask ? gm : gm * q0.mask * (1.f - q0.mask));      // through the sigmoid of the mask logit
            if (q0.tap == 0) {
#pragma unroll
                for (int z = 27; z < 32; ++z) raw_put<T>(graw, g.raw16, o + z, 0.f);
            }
        }
        q0 = q1; r0 = r1; pre1 = pre2;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 64 -> 64 channel bf16 layers (the five 96x320 layers: 283 MB of columns each): the sample kernel with the weight gradient
// folded in.  A workgroup walks chunks of 32 consecutive pixels of a row for ONE group of three taps (grid.y = 0..2): its four
// waves blend the chunk's 96 samples as dcn_bwd_sample_kernel does, but the modulated columns go into an LDS tile
// [tap][16-channel sub][pixel][32 B] next to the chunk's dy rows, and after a barrier the tile is multiplied on the matrix cores:
// grad_weight[o][tap, c] += sum_px dy[px][o] * col[px][tap, c], both operands fetched with transposed LDS reads (the pixel is the
// MFMA k index; wgrad_tr.hip explains the lane pattern), twelve 16x16 accumulator blocks per wave kept across all chunks.
// The columns never reach memory: per launch 283 MB written + 283 MB re-read + the separate GEMM (103 us) are replaced by
// 3 x 31 MB of dy reads and 75 MB of partial gradient blocks.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int SF_PX = 32, SF_TAPS = 3, SF_BT = SF_TAPS * 4;                   // chunk pixels, taps per workgroup, (tap, sub) tiles
constexpr int SF_TILE = SF_PX * 32;                                          // bytes of one 16-channel tile of the chunk
constexpr int SF_STAGE = (SF_BT + 4) * SF_TILE;                              // columns + dy of one chunk: 16 KB

template <typename T>                                                       // T = bf16_t or half_t
__global__ __launch_bounds__(256) void dcn_bwd_sample_wgrad_kernel(const T* __restrict__ x, const float* __restrict__ om,
                                                                  const T* __restrict__ gcol, const T* __restrict__ dy, BtGeom g,
                                                                  int chunks_per_block, int nchunks, float* __restrict__ graw,
                                                                  float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) char lds[2 * SF_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = lane >> 3, cl = lane & 7;                                  // sample slot in the wave, 8-channel group
    const int tg = blockIdx.y;                                                // taps 3*tg .. 3*tg+2
    const int HW = g.H * g.W, cpr = g.W / SF_PX;                              // chunks per row (W is a multiple of 32)
    const int c_begin = blockIdx.x * chunks_per_block, c_end = min(c_begin + chunks_per_block, nchunks);
    const int c0 = cl * 8;

    f32x4 acc[4][SF_TAPS];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < SF_TAPS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int l16 = lane & 15, gq = lane >> 4;
    const uint32_t lane_off = (uint32_t)((4 * gq + (l16 >> 2)) * 32 + (l16 & 3) * 8);
    const uint32_t lds_a = (uint32_t)(uintptr_t)lds;
    auto tr2 = [](uint32_t a) { return bt_tr2(a); };

    // The lane's samples form one stream over (chunk, it = 0..2), software-pipelined as in dcn_bwd_sample_kernel: while sample i
    // is blended, the five gathers of sample i+1 are in flight and the raw offsets of sample i+2 are being fetched -- also across
    // the chunk's barrier and MFMAs.  (Issuing all three samples of a chunk at once cost 194 VGPRs next to the 48 accumulators.)
    struct Loc { bool ok; int my, px, tap; size_t m; const T* xb; int mx; };
    auto locate = [&](int ch, int it) {
        Loc q;
        q.ok = ch < c_end;
        const int chc = q.ok ? ch : c_begin;
        const int row = chc / cpr, x_begin = (chc - row * cpr) * SF_PX;
        const int b = row / g.H;
        q.my = row - b * g.H;
        const int s = it * 32 + wv * 8 + sl;
        q.px = s / 3; q.tap = tg * SF_TAPS + (s - q.px * 3);
        q.mx = x_begin + q.px;
        q.m = (size_t)b * HW + (size_t)q.my * g.W + q.mx;
        q.xb = x + (size_t)b * HW * g.C;
        return q;
    };
    struct Pre { float oh, ow, mk; };
    auto prefetch = [&](const Loc& q) {
        Pre p = {0.f, 0.f, 0.f};
        if (q.ok) { const float* r = om + q.m * 32; p.oh = r[2 * q.tap]; p.ow = r[2 * q.tap + 1]; p.mk = r[18 + q.tap]; }
        return p;
    };
    struct Geo { bool inside; int h0, w0; float lh, lw, mask; };
    auto geo = [&](const Loc& q, const Pre& p) {
        Geo e;
        const int th = (q.tap * 11) >> 5, tw = q.tap - th * 3;
        const float h = (float)(q.my - 1 + th) + p.oh, w = (float)(q.mx - 1 + tw) + p.ow;
        e.mask = p.mk;
        e.inside = q.ok && h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
        const float hf = floorf(h), wf = floorf(w);
        e.lh = h - hf; e.lw = w - wf;
        e.h0 = (int)fminf(fmaxf(hf, -4.f), 32000.f); e.w0 = (int)fminf(fmaxf(wf, -4.f), 32000.f);
        return e;
    };
    struct Raw5 { Raw8<T> g, v[4]; };
    auto issue = [&](const Loc& q, const Geo& e) {
        Raw5 r;
        if (e.inside) {
            r.g.load(gcol + q.m * g.Kp + q.tap * g.C + c0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int hc = e.h0 + (c >> 1), wc = e.w0 + (c & 1);
                if (hc >= 0 && hc < g.H && wc >= 0 && wc < g.W) r.v[c].load(q.xb + ((size_t)hc * g.W + wc) * g.C + c0);
                else r.v[c].zero();
            }
        }
        return r;
    };
    Loc l0 = locate(c_begin, 0), l1 = locate(c_begin, 1);
    Geo e0 = geo(l0, prefetch(l0));
    Raw5 r0 = issue(l0, e0);
    Pre p1 = prefetch(l1);
    for (int ch = c_begin; ch < c_end; ++ch) {
        char* stage = lds + ((ch - c_begin) & 1) * SF_STAGE;
        const u32x4 dyv = *reinterpret_cast<const u32x4*>(dy + (l0.m - l0.px + (tid >> 3)) * g.C + (tid & 7) * 8);      // dy rows of the chunk (Cout = C = 64)
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const Loc l2 = it == 0 ? locate(ch, 2) : locate(ch + 1, it - 1);          // the sample after next
            const Pre p2 = prefetch(l2);
            const Geo e1 = geo(l1, p1);
            const Raw5 r1 = issue(l1, e1);
            {   // blend sample l0 (always inside this chunk)
                const int tl = l0.tap - tg * SF_TAPS;
                float cv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                float gh = 0.f, gw = 0.f, gm = 0.f;
                if (e0.inside) {
                    float gc[8], v0[8], v1[8], v2[8], v3[8];
                    r0.g.unpack(gc); r0.v[0].unpack(v0); r0.v[1].unpack(v1); r0.v[2].unpack(v2); r0.v[3].unpack(v3);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float d10 = v1[k] - v0[k], d32 = v3[k] - v2[k];
                        const float top = v0[k] + e0.lw * d10, bot = v2[k] + e0.lw * d32;
                        const float dh = bot - top, val = top + e0.lh * dh, dw = d10 + e0.lh * (d32 - d10);
                        cv[k] = e0.mask * val;
                        gm += gc[k] * val; gh += gc[k] * dh; gw += gc[k] * dw;
                    }
                }
                *reinterpret_cast<u32x4*>(stage + (tl * 4 + (cl >> 1)) * SF_TILE + l0.px * 32 + (cl & 1) * 16) = ElemTraits<T>::pack(cv);
                gh = bt_group_sum<8>(gh); gw = bt_group_sum<8>(gw); gm = bt_group_sum<8>(gm);
                if (cl == 0) {
                    const size_t o = (size_t)l0.m * 32;
                    raw_put<T>(graw, g.raw16, o + 2 * l0.tap, gh * e0.mask); raw_put<T>(graw, g.raw16, o + 2 * l0.tap + 1, gw * e0.mask);
                    raw_put<T>(graw, g.raw16, o + 18 + l0.tap, g.raw_mask ? gm : gm * e0.mask * (1.f - e0.mask));
                    if (l0.tap == 0) {
#pragma unroll
                        for (int z = 27; z < 32; ++z) raw_put<T>(graw, g.raw16, o + z, 0.f);
                    }
                }
            }
            l0 = l1; e0 = e1; r0 = r1; l1 = l2; p1 = p2;
        }
        // [o sub][pixel][32 B]
        *reinterpret_cast<u32x4*>(stage + (SF_BT + ((tid & 7) >> 1)) * SF_TILE + (tid >> 3) * 32 + (tid & 1) * 16) = dyv;
        __syncthreads();                                                      // tile complete (the other stage is free: two chunks ago)
        {
            const uint32_t sb = lds_a + (uint32_t)(((ch - c_begin) & 1) * SF_STAGE);
            u32x4 df[4], cf[SF_TAPS];
#pragma unroll
            for (int i = 0; i < 4; ++i) df[i] = tr2(sb + (uint32_t)((SF_BT + i) * SF_TILE) + lane_off);
#pragma unroll
            for (int j = 0; j < SF_TAPS; ++j) cf[j] = tr2(sb + (uint32_t)((wv * SF_TAPS + j) * SF_TILE) + lane_off);      // wave wv: tiles 3wv..3wv+2 of 12
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < SF_TAPS; ++j) mma_chunk<T>(df[i], cf[j], acc[i][j]);
        }
        // (two stages: the next chunk writes the other one; the barrier after ITS writes orders this chunk's reads before the
        // writes of the chunk after next)
    }
    // partial block -> ws[blockIdx.x][o][k], k = tap*64 + sub*16 + (lane & 15); tile t = wv*3 + j = tl*4 + sub
    float* slab = ws + (size_t)blockIdx.x * (64 * 576);
#pragma unroll
    for (int j = 0; j < SF_TAPS; ++j) {
        const int t = wv * SF_TAPS + j, tl = t >> 2, sub = t & 3;
        const int k = (tg * SF_TAPS + tl) * 64 + sub * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(size_t)(i * 16 + (lane >> 4) * 4 + r) * 576 + k] = acc[i][j][r];
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Third generation for the 64 -> 64 layers (16-bit maps): d(columns) is never written.  (Round 3: the [M][9C] matrix dy x W^T
// -- 283 MB per 64 -> 64 @ 8x96x320 layer -- was written once and read by both consumers, 2.05 GB of traffic for 189 MB of
// algorithmic bytes.)  Both consumers now rebuild what they need from dy (31 MB, cache-resident) on the matrix cores:
//
//   * grad_input:  dx[p][c] = sum_{(m,t,corner) on p} w * sum_o dy[m][o] W[o][t,c]  =  sum_t  z_t[p][:] . W_t,
//     z_t[p][o] = sum_{(m,t,corner) on p} w * dy[m][o]:  the bilinear SPLAT of dy (same per-pixel corner lists as before, gathered rows
//     are 128-byte dy rows instead of d(columns) rows), then nine small GEMMs (128 pixels x 64 o) . (64 o x 64 c) per tile on the MFMA.
//     The lists are filled tap by tap (one barrier per tap), so a pixel's list is tap-sorted and its lane group walks one segment per tap.
//   * grad_offset / grad_mask / grad_weight (dcn_bwd_sample_wgrad_fly_kernel): the 32-pixel chunk's d(columns) for the workgroup's
//     three taps = W_tg (192 x 64) . dy_chunk^T (64 x 32) -- 12 MFMAs per wave -- goes to a 12 KB LDS tile the samples read.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int FZ_LCAP = 62;

template <typename T>                                         // bf16_t / half_t, C = Cout = 64
__global__ __launch_bounds__(256) void dcn_bwd_tile_fly_kernel(const float* __restrict__ om, const T* __restrict__ dy, const T* __restrict__ wT,
                                                              BtGeom g, T* __restrict__ dx, int* __restrict__ far_count,
                                                              u32x4* __restrict__ far_list, int far_cap) {
    using EN = BtEntry<T>;
    constexpr int LCAP = FZ_LCAP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* lists = reinterpret_cast<uint32_t*>(smem);                       // [BT_NPIX][LCAP], tap-sorted per pixel
    int* counts = reinterpret_cast<int*>(smem + BT_NPIX * LCAP * 4);           // [BT_NPIX]
    uint2* fstage = reinterpret_cast<uint2*>(smem + BT_NPIX * LCAP * 4 + BT_NPIX * 4);   // [BT_FCAP]
    __shared__ int fcount, fbase;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = lane >> 3, cl = lane & 7;                                   // pixel slot in the wave, 8-channel group
    const int l16 = lane & 15, kq = lane >> 4;
    int tile = blockIdx.x;
    const int tx = tile % g.tiles_x; tile /= g.tiles_x;
    const int ty = tile % g.tiles_y; const int b = tile / g.tiles_y;
    const int ty0 = ty * BT_TH, tx0 = tx * BT_TW;
    const int HW = g.H * g.W;
    const long mb = (long)b * HW;

    for (int i = tid; i < BT_NPIX; i += 256) counts[i] = 0;
    if (tid == 0) fcount = 0;
    __syncthreads();

    auto far_entry = [&](int slot, size_t m, int tap, int hc, int wc, uint32_t wbits) {
        if (slot < far_cap) far_list[slot] = u32x4{(uint32_t)m, ((uint32_t)hc << 16) | ((uint32_t)wc << 4) | (uint32_t)tap, wbits, 0u};
    };
    auto to_far = [&](size_t m, int my, int mx, int tap, int q, int hc, int wc, float w) {
        const int fs = atomicAdd(&fcount, 1);
        if (fs < BT_FCAP) fstage[fs] = uint2{((uint32_t)q << 28) | ((uint32_t)my << 16) | ((uint32_t)mx << 4) | (uint32_t)tap, __float_as_uint(w)};
        else far_entry(atomicAdd(far_count, 1), m, tap, hc, wc, __float_as_uint(w));
    };

    // ---------------- phase 1: bin the (sample, corner) pairs of the candidate window by target pixel, ONE TAP AT A TIME ----------------
    {
        // the window's raw offset / mask rows stay in registers across the nine taps (re-reading three values per (pixel, tap) from L1
        // measured slower: 176 vs 111 us for this phase alone)
        float o[BT_ROUNDS][28];
        bool valid[BT_ROUNDS];
#pragma unroll
        for (int round = 0; round < BT_ROUNDS; ++round) {
            const int cp = round * 256 + tid;
            const int wy = cp / BT_CW, wx = cp - wy * BT_CW;
            const int my = ty0 - BT_D + wy, mx = tx0 - BT_D + wx;
            valid[round] = (BT_WIN % 256 == 0 || cp < BT_WIN) && my >= 0 && my < g.H && mx >= 0 && mx < g.W && !BT_PROBE(g, 4);
            const float* r = om + (size_t)(mb + (long)(valid[round] ? my : 0) * g.W + (valid[round] ? mx : 0)) * 32;
#pragma unroll
            for (int q4 = 0; q4 < 28; q4 += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(r + q4);
                o[round][q4] = t[0]; o[round][q4 + 1] = t[1]; o[round][q4 + 2] = t[2]; o[round][q4 + 3] = t[3];
            }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int th = tap / 3, tw = tap - th * 3;
#pragma unroll
            for (int round = 0; round < BT_ROUNDS; ++round) {
                if (!valid[round]) continue;
                const int cp = round * 256 + tid;
                const int wy = cp / BT_CW, wx = cp - wy * BT_CW;
                const int my = ty0 - BT_D + wy, mx = tx0 - BT_D + wx;
                const bool own = wy >= BT_D && wy < BT_D + BT_TH && wx >= BT_D && wx < BT_D + BT_TW;
                const size_t m = (size_t)(mb + (long)my * g.W + mx);
                const float h = (float)(my - 1 + th) + o[round][2 * tap], w = (float)(mx - 1 + tw) + o[round][2 * tap + 1];
                const float mask = o[round][18 + tap];
                if (!(h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W) || mask == 0.f) continue;
                const float hf = floorf(h), wf = floorf(w);
                const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw = 1.f - lw;
                const int h0 = (int)hf, w0 = (int)wf;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int hc = h0 + (q >> 1), wc = w0 + (q & 1);
                    const float wq = ((q >> 1) ? lh : hh) * ((q & 1) ? lw : hw) * mask;
                    if (hc < 0 || hc >= g.H || wc < 0 || wc >= g.W || wq == 0.f) continue;
                    const int ly = hc - ty0, lx = wc - tx0;
                    if (ly >= 0 && ly < BT_TH && lx >= 0 && lx < BT_TW) {
                        const int p = ly * BT_TW + lx;
                        const int slot = atomicAdd(&counts[p], 1);
                        if (slot < LCAP) lists[p * LCAP + slot] = EN::make(wy, wx, tap, wq);
                        else to_far(m, my, mx, tap, q, hc, wc, wq);
                    } else if (own) {
                        const int cty0 = (hc / BT_TH) * BT_TH, ctx0 = (wc / BT_TW) * BT_TW;
                        const bool near = my >= cty0 - BT_D && my < cty0 + BT_TH + BT_D && mx >= ctx0 - BT_D && mx < ctx0 + BT_TW + BT_D;
                        if (!near) to_far(m, my, mx, tap, q, hc, wc, wq);
                    }
                }
            }
            __syncthreads();                                                  // every entry of tap `tap` is in its list before the first of tap + 1
        }
    }
    {   // flush the staged far corners
        const int nf = min(fcount, BT_FCAP);
        if (nf > 0) {
            if (tid == 0) fbase = atomicAdd(far_count, nf);
            __syncthreads();
            for (int i = tid; i < nf; i += 256) {
                const uint2 en = fstage[i];
                const int q = (int)(en.x >> 28), emy = (int)((en.x >> 16) & 0xfff), emx = (int)((en.x >> 4) & 0xfff), etap = (int)(en.x & 15);
                const size_t m = (size_t)(mb + (long)emy * g.W + emx);
                const SampGeo sg = samp_geo(g, om + m * 32, emy, emx, etap);
                far_entry(fbase + i, m, etap, sg.h0 + (q >> 1), sg.w0 + (q & 1), en.y);
            }
        }
    }

    // ---------------- phase 3: per tap, splat dy into z_t (registers -> LDS), then z_t . W_t on the matrix cores ----------------
    // lane group (wv, sl) owns pixels p = pass * 32 + wv * 8 + sl, pass 0..3; cursor[pass] walks that pixel's tap-sorted list
    int cur[4], cnt[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) { cur[ps] = 0; cnt[ps] = min(counts[ps * 32 + wv * 8 + sl], LCAP); }
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const T* dyb = dy + (size_t)mb * 64 + cl * 8;
    const int wbase_y = ty0 - BT_D, wbase_x = tx0 - BT_D;

    for (int tap = 0; tap < (BT_PROBE(g, 1) ? 0 : 9); ++tap) {
        // W_t fragments (B operand: row n = input channel c, k = output channel o): L2 -> registers, in flight during the splat
        u32x4 wf[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[j][ks] = BT_PROBE(g, 16) ? u32x4{0u, 0u, 0u, 0u} : *reinterpret_cast<const u32x4*>(wT + (size_t)(tap * 64 + j * 16 + l16) * 64 + ks * 32 + kq * 8);
        // half-round h = passes 2h, 2h + 1 = the 16 pixels of MFMA fragment h: the first four entries of both pixels are gathered together
        // (eight independent dy rows in flight per lane group; a segment has ~4 entries, so this is usually all of it)
#pragma unroll
        for (int hf_ = 0; hf_ < 2; ++hf_) {
            float z[2][8];
            bool more[2];
            {
                uint32_t en[2][4]; bool ok[2][4]; Raw8<T> gr[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int ps = 2 * hf_ + q;
                    const uint32_t* lp = lists + (ps * 32 + wv * 8 + sl) * LCAP;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = cur[ps] + u;
                        en[q][u] = e < cnt[ps] ? lp[e] : 0xffffffffu;
                        ok[q][u] = e < cnt[ps] && (int)((en[q][u] >> 18) & 15) == tap;
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int wy = (int)(en[q][u] >> 27), wx = (int)((en[q][u] >> 22) & 31);
                        if (ok[q][u]) gr[q][u].load(dyb + ((size_t)(wbase_y + wy) * g.W + (wbase_x + wx)) * 64);
                        else gr[q][u].zero();
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int ps = 2 * hf_ + q;
#pragma unroll
                    for (int k = 0; k < 8; ++k) z[q][k] = 0.f;
                    int nok = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (ok[q][u]) {
                            int wy, wx, t_; float w;
                            EN::read(en[q][u], wy, wx, t_, w);
                            float gq[8];
                            gr[q][u].unpack(gq);
#pragma unroll
                            for (int k = 0; k < 8; ++k) z[q][k] += w * gq[k];
                            ++nok;
                        }
                    }
                    cur[ps] += nok;
                    more[q] = nok == 4;
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ps = 2 * hf_ + q;
                const uint32_t* lp = lists + (ps * 32 + wv * 8 + sl) * LCAP;
                while (more[q]) {                                             // longer segments: four more entries per round
                    uint32_t en[4]; bool ok[4]; float w[4]; Raw8<T> gr[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = cur[ps] + u;
                        en[u] = e < cnt[ps] ? lp[e] : 0xffffffffu;
                        ok[u] = e < cnt[ps] && (int)((en[u] >> 18) & 15) == tap;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int wy, wx, t_;
                        EN::read(en[u], wy, wx, t_, w[u]);
                        if (ok[u]) gr[u].load(dyb + ((size_t)(wbase_y + wy) * g.W + (wbase_x + wx)) * 64);
                        else gr[u].zero();
                    }
                    int nok = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (ok[u]) {
                            float gq[8];
                            gr[u].unpack(gq);
#pragma unroll
                            for (int k = 0; k < 8; ++k) z[q][k] += w[u] * gq[k];
                            ++nok;
                        }
                    }
                    cur[ps] += nok;
                    more[q] = nok == 4;
                }
            }
            // z_t of the fragment's 16 pixels is in this wave's registers as (pixel slot sl, 8-output group cl) x two passes; the MFMA wants
            // (pixel l16, k-group kq) per k-step: lane (l16, kq) of k-step ks takes the packed dwords of lane (l16 & 7) * 8 + 4 ks + kq, from the
            // first pass for l16 < 8 and from the second otherwise.  Two ds_bpermute per dword (one per pass register, the destination
            // picks) -- no LDS tile, no barrier: the waves of a workgroup never wait for each other in this phase.
            const u32x4 pk0 = ElemTraits<T>::pack(z[0]), pk1 = ElemTraits<T>::pack(z[1]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int src = (((l16 & 7) << 3) + ks * 4 + kq) << 2;
                u32x4 af;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if (BT_PROBE(g, 8)) { af[d] = pk0[d] ^ pk1[d]; continue; }
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)pk0[d]);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)pk1[d]);
                    af[d] = (l16 & 8) ? hi : lo;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) mma_chunk<T>(af, wf[j][ks], acc[hf_][j]);
            }
        }
    }

    // ---------------- epilogue: acc (row = pixel, col = channel) -> wave-private stage -> one 16-byte store per 8 channels ----------------
    // accumulator row 16 i + q of this wave = pixel (pass 2 i + (q >> 3), slot q & 7) = tile pixel (2 i + (q >> 3)) * 32 + wv * 8 + (q & 7)
    __syncthreads();                                                          // every wave is done with the lists (the stages below overlay them)
    float* stage = reinterpret_cast<float*>(smem) + wv * (32 * 68);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(i * 16 + kq * 4 + r) * 68 + j * 16 + l16] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int item = it * 64 + lane, row = item >> 3, c8 = item & 7;
        const int p = (row >> 3) * 32 + wv * 8 + (row & 7);
        const int y = ty0 + p / BT_TW, xx = tx0 + (p % BT_TW);
        const f32x4 a = *reinterpret_cast<const f32x4*>(stage + row * 68 + c8 * 8), bq = *reinterpret_cast<const f32x4*>(stage + row * 68 + c8 * 8 + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], bq[0], bq[1], bq[2], bq[3]};
        if (y < g.H && xx < g.W) bt_store8<T>(dx + ((size_t)b * HW + (size_t)y * g.W + xx) * 64 + c8 * 8, v);
    }
}

// far corners of the gcol-free form: the d(columns) row of the entry's (sample, tap) is rebuilt from dy and W (64 x 64 MACs per entry,
// eight lanes x eight channels) and added into the finished dx with (packed 16-bit) atomics
template <typename T>
__global__ __launch_bounds__(256) void dcn_bwd_far_fly_kernel(const T* __restrict__ dy, const T* __restrict__ wT, const u32x4* __restrict__ far_list,
                                                             const int* __restrict__ far_count, int far_cap, BtGeom g, T* __restrict__ dx) {
    const int n = min(*far_count, far_cap);
    const int HW = g.H * g.W, cl = threadIdx.x & 7;
    for (long e = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 3; e < n; e += ((long)gridDim.x * blockDim.x) >> 3) {
        const u32x4 en = far_list[e];
        const size_t m = en.x;
        const int hc = (int)(en.y >> 16), wc = (int)((en.y >> 4) & 0xfff), tap = (int)(en.y & 15);
        const float w = __uint_as_float(en.z);
        const size_t bimg = m / HW;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const T* dr = dy + m * 64;
        for (int o8 = 0; o8 < 64; o8 += 8) {
            float d[8];
            bt_load8<T>(dr + o8, d);
#pragma unroll
            for (int k = 0; k < 8; ++k) {                                    // channel c = cl * 8 + k: row (tap, c) of wT, eight consecutive o
                float wv8[8];
                bt_load8<T>(wT + (size_t)(tap * 64 + cl * 8 + k) * 64 + o8, wv8);
#pragma unroll
                for (int q = 0; q < 8; ++q) a[k] += d[q] * wv8[q];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] *= w;
        bt_atomic_add8(dx + (bimg * HW + (size_t)hc * g.W + wc) * 64 + cl * 8, a);
    }
}


// dcn_bwd_sample_wgrad_kernel without the d(columns) operand: per 32-pixel chunk the workgroup first multiplies
// g^T = W_tg (192 (tap, c) rows x 64 o) . dy_chunk^T (64 o x 32 px) on the matrix cores -- wave wv owns rows 48 wv .. +48: three
// A fragments of W (held in registers for the whole launch) x two B fragments of the chunk's dy rows (plain 16-byte reads of the tile
// the weight-gradient MFMAs read transposed) -- and writes it to an LDS tile [pixel][192] in the activation dtype (the rounding the
// d(columns) GEMM applied); the samples then take their 8-channel piece of it with one ds_read_b128.  Three barriers per chunk
// instead of one; no 283 MB read.
constexpr int SG_ROW = 3 * 64 * 2 + 16;                                      // bytes of one pixel row of the g tile (400: conflict-free 16-byte reads)
constexpr int SG_TILE = SF_PX * SG_ROW;                                      // 12.5 KB

// WG = true: the chunk's d(columns) tile is also written out (coalesced 16-byte stores, 12 KB per chunk and tap group) for the tile kernel
// that gathers d(columns) rows -- the separate d(columns) GEMM (111 us for 64 -> 64 @ 8x96x320) and its re-read here disappear, the
// tile kernel's read stays.
template <typename T, bool WG>                                              // T = bf16_t or half_t
__global__ __launch_bounds__(256) void dcn_bwd_sample_wgrad_fly_kernel(const T* __restrict__ x, const float* __restrict__ om,
                                                                      const T* __restrict__ wT, const T* __restrict__ dy, BtGeom g,
                                                                      int chunks_per_block, int nchunks, float* __restrict__ graw,
                                                                      float* __restrict__ ws, T* __restrict__ gcol, float* __restrict__ dbias) {
    __shared__ __attribute__((aligned(16))) char lds[2 * SF_STAGE + SG_TILE];
    char* gt = lds + 2 * SF_STAGE;
    // grad_bias[o] = sum of dy over the pixels (r06): the tap-group-0 workgroups see every dy row once (their `dyv` loads), so they sum it on the way
    // -- eight channels per thread, folded over the chunk's 32 pixels at the end -- instead of a separate column-sum pass over dy (24 us per layer)
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool bias_here = dbias != nullptr && blockIdx.y == 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl = lane >> 3, cl = lane & 7;                                  // sample slot in the wave, 8-channel group
    const int tg = blockIdx.y;                                                // taps 3*tg .. 3*tg+2
    const int HW = g.H * g.W, cpr = g.W / SF_PX;                              // chunks per row (W is a multiple of 32)
    const int c_begin = blockIdx.x * chunks_per_block, c_end = min(c_begin + chunks_per_block, nchunks);
    const int c0 = cl * 8;

    f32x4 acc[4][SF_TAPS];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < SF_TAPS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int l16 = lane & 15, gq = lane >> 4;
    const uint32_t lane_off = (uint32_t)((4 * gq + (l16 >> 2)) * 32 + (l16 & 3) * 8);
    const uint32_t lds_a = (uint32_t)(uintptr_t)lds;
    auto tr2 = [](uint32_t a) { return bt_tr2(a); };
    // W fragments of this wave's 48 (tap, c) rows: A operand, row n = 48 wv + 16 j + l16 of the tap group, k = o
    u32x4 wfr[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            wfr[j][ks] = *reinterpret_cast<const u32x4*>(wT + (size_t)(tg * 192 + wv * 48 + j * 16 + l16) * 64 + ks * 32 + gq * 8);

    struct Loc { bool ok; int my, px, tap; size_t m; const T* xb; int mx; };
    auto locate = [&](int ch, int it) {
        Loc q;
        q.ok = ch < c_end;
        const int chc = q.ok ? ch : c_begin;
        const int row = chc / cpr, x_begin = (chc - row * cpr) * SF_PX;
        const int b = row / g.H;
        q.my = row - b * g.H;
        const int s = it * 32 + wv * 8 + sl;
        q.px = s / 3; q.tap = tg * SF_TAPS + (s - q.px * 3);
        q.mx = x_begin + q.px;
        q.m = (size_t)b * HW + (size_t)q.my * g.W + q.mx;
        q.xb = x + (size_t)b * HW * g.C;
        return q;
    };
    struct Pre { float oh, ow, mk; };
    auto prefetch = [&](const Loc& q) {
        Pre p = {0.f, 0.f, 0.f};
        if (q.ok) { const float* r = om + q.m * 32; p.oh = r[2 * q.tap]; p.ow = r[2 * q.tap + 1]; p.mk = r[18 + q.tap]; }
        return p;
    };
    struct Geo { bool inside; int h0, w0; float lh, lw, mask; };
    auto geo = [&](const Loc& q, const Pre& p) {
        Geo e;
        const int th = (q.tap * 11) >> 5, tw = q.tap - th * 3;
        const float h = (float)(q.my - 1 + th) + p.oh, w = (float)(q.mx - 1 + tw) + p.ow;
        e.mask = p.mk;
        e.inside = q.ok && h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
        const float hf = floorf(h), wf = floorf(w);
        e.lh = h - hf; e.lw = w - wf;
        e.h0 = (int)fminf(fmaxf(hf, -4.f), 32000.f); e.w0 = (int)fminf(fmaxf(wf, -4.f), 32000.f);
        return e;
    };
    struct Raw4 { Raw8<T> v[4]; };
    auto issue = [&](const Loc& q, const Geo& e) {
        Raw4 r;
        if (e.inside) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int hc = e.h0 + (c >> 1), wc = e.w0 + (c & 1);
                if (hc >= 0 && hc < g.H && wc >= 0 && wc < g.W) r.v[c].load(q.xb + ((size_t)hc * g.W + wc) * g.C + c0);
                else r.v[c].zero();
            }
        }
        return r;
    };
    auto dy_rows = [&](int ch) {                                             // this thread's 16 bytes of the chunk's dy rows (Cout = 64)
        const int chc = ch < c_end ? ch : c_begin;
        const int row = chc / cpr, x_begin = (chc - row * cpr) * SF_PX;
        const int b = row / g.H, my = row - b * g.H;
        const size_t m0 = (size_t)b * HW + (size_t)my * g.W + x_begin;
        return *reinterpret_cast<const u32x4*>(dy + (m0 + (tid >> 3)) * 64 + (tid & 7) * 8);
    };
    Loc l0 = locate(c_begin, 0), l1 = locate(c_begin, 1);
    Geo e0 = geo(l0, prefetch(l0));
    Raw4 r0 = issue(l0, e0);
    Pre p1 = prefetch(l1);
    u32x4 dyv = dy_rows(c_begin);
    for (int ch = c_begin; ch < c_end; ++ch) {
        char* stage = lds + ((ch - c_begin) & 1) * SF_STAGE;
        // dy rows of the chunk: [o sub][pixel][32 B]  (this stage's previous readers -- the MFMAs of chunk ch - 2 -- are two barriers back)
        *reinterpret_cast<u32x4*>(stage + (SF_BT + ((tid & 7) >> 1)) * SF_TILE + (tid >> 3) * 32 + (tid & 1) * 16) = dyv;
        if (bias_here) {
            float f8[8];
            ElemTraits<T>::unpack(dyv, f8);
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[e] += f8[e];
        }
        dyv = dy_rows(ch + 1);                                                // next chunk's rows: in flight through this one
        __syncthreads();                                                      // dy rows visible; every wave is done reading the g tile of the previous chunk
        {   // g^T rows 48 wv .. +48 = W . dy^T  ->  gt[pixel][192]
            f32x4 ga[3][2];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) ga[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 df[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)                                  // B operand: pixel i*16 + l16, o = ks*32 + gq*8 .. +8 -> sub-tile 2ks + (gq>>1), half gq&1
                    df[i] = *reinterpret_cast<const u32x4*>(stage + (SF_BT + 2 * ks + (gq >> 1)) * SF_TILE + (i * 16 + l16) * 32 + (gq & 1) * 16);
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) mma_chunk<T>(wfr[j][ks], df[i], ga[j][i]);
            }
            // D: row = (tap, c) index 48 wv + 16 j + 4 gq + r, col = pixel i*16 + l16: four consecutive channels of one pixel per lane
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    bt_store4<T>(reinterpret_cast<T*>(gt + (i * 16 + l16) * SG_ROW + (wv * 48 + j * 16 + gq * 4) * 2), ga[j][i]);
        }
        __syncthreads();                                                      // g tile complete
        if constexpr (WG) {                                                   // 32 pixels x 24 chunks of 16 bytes = three per thread
            const int row = ch / cpr, x_begin = (ch - row * cpr) * SF_PX;
            const int b = row / g.H, my = row - b * g.H;
            const size_t m0 = (size_t)b * HW + (size_t)my * g.W + x_begin;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int idx = i * 256 + tid, px = idx / 24, c16 = idx - px * 24;
                *reinterpret_cast<u32x4*>(gcol + (m0 + px) * g.Kp + tg * 192 + c16 * 8) = *reinterpret_cast<const u32x4*>(gt + px * SG_ROW + c16 * 16);
            }
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const Loc l2 = it == 0 ? locate(ch, 2) : locate(ch + 1, it - 1);          // the sample after next
            const Pre p2 = prefetch(l2);
            const Geo e1 = geo(l1, p1);
            const Raw4 r1 = issue(l1, e1);
            {   // blend sample l0 (always inside this chunk)
                const int tl = l0.tap - tg * SF_TAPS;
                float cv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                float gh = 0.f, gw = 0.f, gm = 0.f;
                if (e0.inside) {
                    float gc[8], v0[8], v1[8], v2[8], v3[8];
                    bt_load8<T>(reinterpret_cast<const T*>(gt + l0.px * SG_ROW + (tl * 64 + c0) * 2), gc);
                    r0.v[0].unpack(v0); r0.v[1].unpack(v1); r0.v[2].unpack(v2); r0.v[3].unpack(v3);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float d10 = v1[k] - v0[k], d32 = v3[k] - v2[k];
                        const float top = v0[k] + e0.lw * d10, bot = v2[k] + e0.lw * d32;
                        const float dh = bot - top, val = top + e0.lh * dh, dw = d10 + e0.lh * (d32 - d10);
                        cv[k] = e0.mask * val;
                        gm += gc[k] * val; gh += gc[k] * dh; gw += gc[k] * dw;
                    }
                }
                *reinterpret_cast<u32x4*>(stage + (tl * 4 + (cl >> 1)) * SF_TILE + l0.px * 32 + (cl & 1) * 16) = ElemTraits<T>::pack(cv);
                gh = bt_group_sum<8>(gh); gw = bt_group_sum<8>(gw); gm = bt_group_sum<8>(gm);
                if (cl == 0) {
                    const size_t o = (size_t)l0.m * 32;
                    raw_put<T>(graw, g.raw16, o + 2 * l0.tap, gh * e0.mask); raw_put<T>(graw, g.raw16, o + 2 * l0.tap + 1, gw * e0.mask);
                    raw_put<T>(graw, g.raw16, o + 18 + l0.tap, g.raw_mask ? gm : gm * e0.mask * (1.f - e0.mask));
                    if (l0.tap == 0) {
#pragma unroll
                        for (int z = 27; z < 32; ++z) raw_put<T>(graw, g.raw16, o + z, 0.f);
                    }
                }
            }
            l0 = l1; e0 = e1; r0 = r1; l1 = l2; p1 = p2;
        }
        __syncthreads();                                                      // column tile complete
        {
            const uint32_t sb = lds_a + (uint32_t)(((ch - c_begin) & 1) * SF_STAGE);
            u32x4 df[4], cf[SF_TAPS];
#pragma unroll
            for (int i = 0; i < 4; ++i) df[i] = tr2(sb + (uint32_t)((SF_BT + i) * SF_TILE) + lane_off);
#pragma unroll
            for (int j = 0; j < SF_TAPS; ++j) cf[j] = tr2(sb + (uint32_t)((wv * SF_TAPS + j) * SF_TILE) + lane_off);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < SF_TAPS; ++j) mma_chunk<T>(df[i], cf[j], acc[i][j]);
        }
    }
    float* slab = ws + (size_t)blockIdx.x * (64 * 576);
#pragma unroll
    for (int j = 0; j < SF_TAPS; ++j) {
        const int t = wv * SF_TAPS + j, tl = t >> 2, sub = t & 3;
        const int k = (tg * SF_TAPS + tl) * 64 + sub * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(size_t)(i * 16 + (lane >> 4) * 4 + r) * 576 + k] = acc[i][j][r];
    }
    if (bias_here) {                                                          // thread (pixel tid >> 3, channel group tid & 7) -> [32][64] floats, 64 column sums, one atomic each
        __syncthreads();                                                      // (every wave is done with the stages)
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int e = 0; e < 8; e += 4) *reinterpret_cast<f32x4*>(red + (tid >> 3) * 64 + (tid & 7) * 8 + e) = f32x4{bsum[e], bsum[e + 1], bsum[e + 2], bsum[e + 3]};
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) t += red[r * 64 + tid];
            unsafeAtomicAdd(dbias + tid, t);
        }
    }
}

static inline size_t bt_al(size_t v) { return (v + 255) & ~(size_t)255; }

struct BtLayout { size_t wT, gcol, col, cnt, flist, wg, total; long far_cap; };
static BtLayout bt_layout(int B, int C, int H, int W, int Cout, int es) {
    BtLayout L; size_t o = 0;
    const size_t M = (size_t)B * H * W, K = (size_t)9 * C;
    L.wT = o;   o += bt_al(K * Cout * es);
    L.gcol = o; o += bt_al(M * K * es);
    L.col = o;  o += bt_al(M * K * es);
    L.cnt = o;  o += 256;
    L.far_cap = (long)M * 36 * (C / 64);                      // every (sample, corner) pair of every (64-channel) slice: the list cannot fill
    L.flist = o; o += bt_al((size_t)L.far_cap * 16);
    L.wg = o;   o += bt_al((size_t)80 * 1024 * 1024);         // partial blocks of the weight gradient (512 x 64 x 576 fp32 for the fused 64 -> 64 form)
    L.total = o;
    return L;
}

template <typename T>
static int dcn_backward_v2_impl(const T* x, const float* offmask, const float* weight, const T* dy, T* dx, float* d_raw,
                                float* dweight, float* dbias, int B, int C, int H, int W, int Cout, void* workspace,
                                size_t workspace_bytes, void* stream, int raw_mask = 0, int raw16 = 0) {
    constexpr int es = (int)sizeof(T);
    constexpr int dt = ElemTraits<T>::DT;                    // MFX_F32 / MFX_BF16 / MFX_F16
    const BtLayout L = bt_layout(B, C, H, W, Cout, es);
    if (!workspace || workspace_bytes < L.total) return mfx_fail(MFX_ERR_WORKSPACE, "dcn_backward_v2: workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(workspace);
    T* wT = (T*)(ws + L.wT); T* gcol = (T*)(ws + L.gcol); T* col = (T*)(ws + L.col);
    int* cnt = (int*)(ws + L.cnt);
    u32x4* flist = (u32x4*)(ws + L.flist);
    if (L.far_cap >= (1L << 31)) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_backward_v2: map too large for the far-corner list");
    const int far_cap = (int)L.far_cap;
    const long M = (long)B * H * W;
    const int K = 9 * C;
    {
        const long total = (long)K * Cout;
        hipLaunchKernelGGL(bt_pack_weight_t<T>, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, st, weight, wT, Cout, C, cnt, dbias);
    }
    BtGeom g;
    g.B = B; g.H = H; g.W = W; g.C = C; g.tiles_x = (W + BT_TW - 1) / BT_TW; g.tiles_y = (H + BT_TH - 1) / BT_TH; g.Kp = K;
    g.CS = C >= 128 ? 128 : 64; g.nslices = C / g.CS; g.dbg = g_opt_dcn_bt_dbg; g.raw_mask = raw_mask; g.raw16 = (raw16 && es == 2) ? 1 : 0;
    if constexpr (!std::is_same<T, float>::value) {
        // gcol-free form (third generation): 64 -> 64, 16-bit, the shapes the fused sample + weight-gradient kernel takes
        const long nchunks = M / SF_PX;
        const size_t slab_bytes = (size_t)64 * 576 * sizeof(float);
        if (g_opt_dcn_bt_fly && g_opt_dcn_bt_fuse_wgrad && !g_opt_det && C == 64 && Cout == 64 && W % SF_PX == 0 &&
            nchunks >= g_opt_dcn_bt_fuse_min_chunks && L.total - L.wg >= 128 * slab_bytes) {
            int nblk = (int)std::min<long>(g_opt_dcn_bt_fuse_blocks, (long)((L.total - L.wg) / slab_bytes));
            const int cpb = (int)((nchunks + nblk - 1) / nblk);
            nblk = (int)((nchunks + cpb - 1) / cpb);
            float* slabs = reinterpret_cast<float*>(ws + L.wg);
            const bool bias_in_kernel = g_opt_dcn_bt_fly_bias != 0;          // (dbias was zeroed by bt_pack_weight_t)
            const bool write_gcol = g_opt_dcn_bt_fly == 2;                   // 2: d(columns) written by the sample kernel, gathered by the second-generation tile kernel
            if (write_gcol)
                hipLaunchKernelGGL((dcn_bwd_sample_wgrad_fly_kernel<T, true>), dim3((unsigned)nblk, 3), dim3(256), 0, st, x, offmask, (const T*)wT, dy, g, cpb, (int)nchunks,
                                   d_raw, slabs, gcol, bias_in_kernel ? dbias : (float*)nullptr);
            else
                hipLaunchKernelGGL((dcn_bwd_sample_wgrad_fly_kernel<T, false>), dim3((unsigned)nblk, 3), dim3(256), 0, st, x, offmask, (const T*)wT, dy, g, cpb, (int)nchunks,
                                   d_raw, slabs, (T*)nullptr, bias_in_kernel ? dbias : (float*)nullptr);
            MFX_HIP_CHECK(hipGetLastError());
            ++g_cnt_dcn_bt_fused; ++g_cnt_dcn_bt_fly;
            int rc2 = mfx_internal_wgrad_slab_sum(slabs, nblk, Cout, C, 3, 3, dweight, stream);
            if (rc2) return rc2;
            if (write_gcol) {
                const dim3 grid((unsigned)(g.tiles_x * g.tiles_y * B), 1u);
                const size_t smem = (size_t)BT_NPIX * BtEntry<T>::LCAP * sizeof(typename BtEntry<T>::type) + (size_t)BT_NPIX * 4 + (size_t)BT_FCAP * 8;
                hipLaunchKernelGGL((dcn_bwd_tile_kernel<T, 8>), grid, dim3(256), smem, st, offmask, (const T*)gcol, g, dx, cnt, flist, far_cap);
                hipLaunchKernelGGL(dcn_bwd_far_kernel<T>, dim3(1024), dim3(256), 0, st, (const T*)gcol, (const u32x4*)flist, (const int*)cnt, far_cap, g, dx);
            } else {
            const size_t smem = (size_t)BT_NPIX * FZ_LCAP * 4 + (size_t)BT_NPIX * 4 + (size_t)BT_FCAP * 8;      // 36 KB: four workgroups per CU (the epilogue's 4 x 8.7 KB stages overlay it)
            hipLaunchKernelGGL(dcn_bwd_tile_fly_kernel<T>, dim3((unsigned)(g.tiles_x * g.tiles_y * B)), dim3(256), smem, st, offmask, dy, (const T*)wT, g, dx, cnt, flist, far_cap);
            hipLaunchKernelGGL(dcn_bwd_far_fly_kernel<T>, dim3(1024), dim3(256), 0, st, dy, (const T*)wT, (const u32x4*)flist, (const int*)cnt, far_cap, g, dx);
            }
            MFX_HIP_CHECK(hipGetLastError());
            return bias_in_kernel ? MFX_OK : mfx_internal_colsum_add(dy, dbias, M, Cout, Cout, dt, stream);
        }
    }
    // d(columns)[m][k] = sum_o dy[m][o] * W[o][k]   (dcn_v2_cuda.cu:273) as a 1x1 implicit GEMM on the matrix cores
    mfx_conv_desc cd = {};
    cd.x = dy; cd.w = wT; cd.y = gcol;
    cd.B = 1; cd.H = 1; cd.W = (int)M; cd.x_pixstride = Cout; cd.Ck = Cout; cd.kh = 1; cd.kw = 1; cd.stride = 1; cd.dil_w = 1;
    cd.Ho = 1; cd.Wo = (int)M; cd.M = (int)M; cd.Cout = K; cd.Cout_pad = K; cd.K_pad = Cout; cd.ldy = K;
    cd.act = MFX_ACT_NONE; cd.dtype = dt; cd.out_dtype = dt;
    int rc;
    // 16-bit layers: the activation-stationary GEMM (gemm_as.hip) -- K = Cout is short (64 .. 256) and the M x 9C result is bound by its own stores,
    // the shape that kernel was written for (profiles/r06_dcn_ps.md: 1.4 - 1.8x the tiled kernel on such maps); option "dcn_bt_gcol_as" = 0: the tiled kernel
    if (es == 2 && g_opt_dcn_bt_gcol_as && (Cout == 64 || Cout == 128 || Cout == 256) && K % 64 == 0)
        rc = mfx_project_nhwc(dy, wT, gcol, (int)M, Cout, K, Cout, K, dt, stream);
    else
        rc = mfx_conv2d_nhwc(&cd, stream);
    if (rc) return rc;
    BtGeom gs = g;                                             // the sample kernel loops its slices inside a lane group: widest slice
    // tile kernel: one workgroup per (tile, slice).  On the small maps 128-channel slices leave the chip under-filled
    // (256ch@24x80: 240 workgroups), so those take 64-channel slices (the binning is repeated per slice, the gathers are not)
    {
        const long wgs128 = (long)g.tiles_x * g.tiles_y * B * g.nslices;
        const int force = g_opt_dcn_bt_cs;
        if (C >= 128 && (force == 64 || (force == 0 && wgs128 < g_opt_dcn_bt_cs_wgs))) { g.CS = 64; g.nslices = C / 64; }
    }
    bool fused_wgrad = false;
    if constexpr (!std::is_same<T, float>::value) {
        const long nchunks = M / SF_PX;
        const size_t slab_bytes = (size_t)64 * 576 * sizeof(float);
        if (g_opt_dcn_bt_fuse_wgrad && C == 64 && Cout == 64 && W % SF_PX == 0 && nchunks >= g_opt_dcn_bt_fuse_min_chunks && L.total - L.wg >= 128 * slab_bytes) {
            // two workgroups fit a CU (210 registers per lane): 170 x 3 tap groups = one resident round of the chip, no tail, 170 slabs to sum
            int nblk = (int)std::min<long>(g_opt_dcn_bt_fuse_blocks, (long)((L.total - L.wg) / slab_bytes));
            const int cpb = (int)((nchunks + nblk - 1) / nblk);
            nblk = (int)((nchunks + cpb - 1) / cpb);
            float* slabs = reinterpret_cast<float*>(ws + L.wg);
            hipLaunchKernelGGL(dcn_bwd_sample_wgrad_kernel<T>, dim3((unsigned)nblk, 3), dim3(256), 0, st, x, offmask, (const T*)gcol, dy, gs, cpb, (int)nchunks,
                               d_raw, slabs);
            MFX_HIP_CHECK(hipGetLastError());
            ++g_cnt_dcn_bt_fused;
            rc = mfx_internal_wgrad_slab_sum(slabs, nblk, Cout, C, 3, 3, dweight, stream);
            if (rc) return rc;
            fused_wgrad = true;
        }
    }
    if (!fused_wgrad) {   // grad_offset / grad_mask and the modulated columns: one lane group per (pixel, tap), every d_raw channel written
        const int rows = B * H, xsplit = std::max(1, std::min(W / 16, (2048 + rows - 1) / rows));
        const dim3 sgrid((unsigned)rows, (unsigned)xsplit);
        if (gs.CS == 64) hipLaunchKernelGGL((dcn_bwd_sample_kernel<T, 8>), sgrid, dim3(256), 0, st, x, offmask, (const T*)gcol, gs, xsplit, d_raw, col);
        else hipLaunchKernelGGL((dcn_bwd_sample_kernel<T, 16>), sgrid, dim3(256), 0, st, x, offmask, (const T*)gcol, gs, xsplit, d_raw, col);
    }
    if (g_opt_det) {
        // grad_input through the fixed-point map (the far-corner list's memory: 9 * M * C bytes >= 8 * M * C); see dcn_bwd_dx_fixed_kernel
        unsigned long long* acc = reinterpret_cast<unsigned long long*>(flist);
        const long n = M * C;
        MFX_HIP_CHECK(mfx::zero_async(acc, (size_t)n * 8, st));
        hipLaunchKernelGGL(dcn_bwd_dx_fixed_kernel<T>, dim3(2048), dim3(256), 0, st, offmask, (const T*)gcol, g, acc);
        hipLaunchKernelGGL(dcn_bwd_dx_unfix_kernel<T>, dim3((unsigned)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, st, acc, dx, n);
        MFX_HIP_CHECK(hipGetLastError());
    } else {
    const dim3 grid((unsigned)(g.tiles_x * g.tiles_y * B), (unsigned)g.nslices);
    const size_t smem = (size_t)BT_NPIX * BtEntry<T>::LCAP * sizeof(typename BtEntry<T>::type) + (size_t)BT_NPIX * 4 + (size_t)BT_FCAP * 8;
    if (g.CS == 64) hipLaunchKernelGGL((dcn_bwd_tile_kernel<T, 8>), grid, dim3(256), smem, st, offmask, (const T*)gcol, g, dx, cnt, flist, far_cap);
    else hipLaunchKernelGGL((dcn_bwd_tile_kernel<T, 16>), grid, dim3(256), smem, st, offmask, (const T*)gcol, g, dx, cnt, flist, far_cap);
    hipLaunchKernelGGL(dcn_bwd_far_kernel<T>, dim3(1024), dim3(256), 0, st, (const T*)gcol, (const u32x4*)flist, (const int*)cnt, far_cap, g, dx);
    MFX_HIP_CHECK(hipGetLastError());
    }
    // grad_weight[o][c][tap] = sum_m dy[m][o] * col[m][tap*C + c]: MFMA GEMM over the pixels, written as (Cout, C, 3, 3)
    if (!fused_wgrad) rc = mfx_internal_conv_wgrad(col, dy, dweight, 1, 1, (int)M, K, C, 3, 3, 1, 0, 0, 1, (int)M, Cout, Cout, dt, 1, C, Cout, stream, 1,
                                 ws + L.wg, L.total - L.wg, 1);
    if (rc) return rc;
    return mfx_internal_colsum_add(dy, dbias, M, Cout, Cout, dt, stream);        // grad_bias[o] = sum_m dy[m][o] (dbias zeroed by bt_pack_weight_t)
}

}  // namespace mfx
using namespace mfx;

// library-internal (dcn_bwd.hip, the `_ext` boundary): the fp32 second-generation backward with d_raw's mask channels as the gradient of the MASK itself
int mfx_internal_dcn_backward_v2_f32_rawmask(const float* x, const float* offmask, const float* weight_oihw, const float* dy, float* dx, float* d_raw,
                                             float* dweight, float* dbias, int B, int C, int H, int W, int Cout, void* workspace, size_t workspace_bytes, void* stream) {
    if (C < 64 || (C & (C - 1)) || Cout < 64 || (Cout & (Cout - 1)) || H >= 4096 || W >= 4096 || (long)B * H * W >= (1L << 31) / 32)
        return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_backward_v2 (raw mask): shape outside the tile-owned kernels' range");
    return dcn_backward_v2_impl<float>(x, offmask, weight_oihw, dy, dx, d_raw, dweight, dbias, B, C, H, W, Cout, workspace, workspace_bytes, stream, 1);
}

extern "C" size_t mfx_dcn_backward_v2_workspace_bytes(int B, int C, int H, int W, int Cout, int dtype) {
    return bt_layout(B, C, H, W, Cout, dtype == MFX_F32 ? 4 : 2).total;
}

static int dcn_backward_v2_entry(const void* x, const float* offmask, const float* weight_oihw, const void* dy, void* dx, void* d_raw, int raw16,
                                 float* dweight, float* dbias, int B, int C, int H, int W, int Cout, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !offmask || !weight_oihw || !dy || !dx || !d_raw || !dweight || !dbias) return mfx_fail(MFX_ERR_ARG, "dcn_backward_v2: null pointer");
    if (C < 64 || (C & (C - 1)) || Cout < 64 || (Cout & (Cout - 1))) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_backward_v2: C and Cout must be powers of two >= 64");
    if (H >= 4096 || W >= 4096 || (long)B * H * W >= (1L << 31) / 32) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_backward_v2: map too large");
    if (B * H * W == 0) return MFX_OK;
    float* dr = reinterpret_cast<float*>(d_raw);
    if (dtype == MFX_F32)
        return dcn_backward_v2_impl<float>((const float*)x, offmask, weight_oihw, (const float*)dy, (float*)dx, dr, dweight, dbias, B, C, H, W, Cout, workspace, workspace_bytes, stream);
    if (dtype == MFX_BF16)
        return dcn_backward_v2_impl<bf16_t>((const bf16_t*)x, offmask, weight_oihw, (const bf16_t*)dy, (bf16_t*)dx, dr, dweight, dbias, B, C, H, W, Cout, workspace, workspace_bytes, stream, 0, raw16);
    if (dtype == MFX_F16)
        return dcn_backward_v2_impl<half_t>((const half_t*)x, offmask, weight_oihw, (const half_t*)dy, (half_t*)dx, dr, dweight, dbias, B, C, H, W, Cout, workspace, workspace_bytes, stream, 0, raw16);
    return mfx_fail(MFX_ERR_ARG, "dcn_backward_v2: bad dtype");
}

extern "C" int mfx_dcn_backward_v2(const void* x, const float* offmask, const float* weight_oihw, const void* dy, void* dx,
                                   float* d_raw, float* dweight, float* dbias, int B, int C, int H, int W, int Cout, int dtype,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    return dcn_backward_v2_entry(x, offmask, weight_oihw, dy, dx, d_raw, 0, dweight, dbias, B, C, H, W, Cout, dtype, workspace, workspace_bytes, stream);
}

extern "C" int mfx_dcn_backward_v2_rt(const void* x, const float* offmask, const float* weight_oihw, const void* dy, void* dx,
                                      void* d_raw, int raw_in_act_dtype, float* dweight, float* dbias, int B, int C, int H, int W, int Cout, int dtype,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    return dcn_backward_v2_entry(x, offmask, weight_oihw, dy, dx, d_raw, raw_in_act_dtype != 0, dweight, dbias, B, C, H, W, Cout, dtype, workspace, workspace_bytes, stream);
}
