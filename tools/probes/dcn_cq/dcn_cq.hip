// Fused modulated deformable convolution (DCNv2 forward + bias/BN + activation), third gather kernel: the four-corner bilinear
// blend is done by the MATRIX CORES' ACCUMULATION instead of by vector instructions.
//
//   y[m][n] = act( scale[n] * sum_tap sum_q  w_q(m,tap) * ( sum_c W[n][tap,c] * x[corner q of (m,tap)][c] )  + shift[n] )
//   (reference: model/backbone/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:125-195 + dcn_v2_cuda.cu:139-163; `columns` never exists)
//
// The bilinear weight (with the modulation mask and the corner's validity folded in) is a scalar per (pixel, tap, corner), so it
// commutes with the channel contraction: the products  P_q = W_tap . x_q  of the four UN-BLENDED corner rows are formed on the matrix
// cores and combined as  acc += w_q * P_q  once per (tap, corner).  What that buys (r05 ISA of dcn_igemm_kernel: 279 vector
// instructions per 8 MFMAs -- bf16 unpack 64, fp32 blend 80, addresses 32 ...): a gathered 16-byte chunk IS an MFMA operand -- no
// unpack, no blend, no re-pack, no LDS round trip of the sampled tile; per (tap, corner) and 16 output channels 4 fused
// multiply-adds.  What it costs: 4x the matrix work of the blended form, on cores these layers used to 7-13 %.  It also drops one
// rounding: the blended sample is never rounded to 16 bits (products of 16-bit values are exact in fp32, sums in fp32).
//
// The GEMM is run transposed, output channels on the MFMA M axis, pixels on N: lane l holds D[channel 4*(l>>4)+r][pixel l&15], so every
// value of a lane belongs to ONE pixel and the scalar w_q is the lane's own (each lane computes the sampling geometry of pixel l&15;
// no cross-lane traffic), and the B operand of lane l -- 8 consecutive channels (l>>4) of pixel l&15's corner row -- is one 16-byte
// global load (a wave instruction reads 16 pixels x 64 contiguous bytes).
//
// Workgroup = 4 waves = 64 pixels x 64 output channels; the weights of one tap (<= 128 input channels at a time) are staged in LDS,
// double-buffered, one barrier per slab; corner rows are fetched two k-steps ahead through a three-slot register ring (the tap loop is
// unrolled by three so that ring slots, LDS offsets and channel offsets are instruction immediates).
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "igemm.h"
#include <type_traits>

namespace mfx {

int g_opt_dcn_cq = 0;          // 0: never (default: measured slower than the blend-first kernels, see the table in DESIGN.md section 8); 2: first form, 3: LDS-DMA form, wherever an instantiation exists

struct DcnQGeom { int H, W, Ho, Wo, stride, pad, dil, M, tiles_n, K_pad; };

template <int C> struct CqCfg {
    static constexpr int SLAB = C < 128 ? C : 128;         // input channels of one tap per LDS weight slab
    static constexpr int SPT = C / SLAB;                   // slabs per tap
    static constexpr int KCS = SLAB / 32;                  // 32-channel MFMA steps per slab
    static constexpr int STEPS = C / 32;                   // ... per tap
    static constexpr int RB = SLAB * 2 + 16;               // LDS row: one output channel's slab + 16 B (16 rows on 16 distinct 16-byte bank slots)
    static constexpr int STAGE = 64 * RB;
    static constexpr int SMEM = 2 * STAGE;
    static constexpr int CPR = SLAB / 8;                   // 16-byte chunks per row
    static constexpr int WCH = 64 * CPR / 256;             // chunks per thread and slab
    static constexpr int NSLAB = 9 * SPT;
};

// LDS-only workgroup barrier: outstanding GLOBAL loads (the corner ring, the next weight slab) stay in flight across it
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <typename T, int C>
__global__ __launch_bounds__(256, 2) void dcn_cq_kernel(const T* __restrict__ x, const float* __restrict__ om, const T* __restrict__ w,
                                                         DcnQGeom g, EpiArgs ep) {
    using K = CqCfg<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, xl = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    // two roles per lane.  LOADING: four adjacent lanes fetch the four 16-byte chunks of one pixel's corner row (a wave instruction reads
    // 16 pixels x 64 contiguous bytes; with the MFMA's own lane order -- pixel = l & 15 -- adjacent lanes hit 16 different rows and the
    // texture path splits the instruction into 64 requests instead of 16: measured 1.8x SLOWER than the kernel this one replaces), so a
    // lane computes the sampling geometry of pixel gp = l >> 2.  COMPUTING: the MFMA wants (pixel l & 15, chunk l >> 4) in lane l: a fixed
    // lane permutation of every loaded chunk (four ds_bpermute_b32, no LDS memory), and of the four corner weights once per tap.
    const int gp = lane >> 2, gc = lane & 3;
    const int m = tm * 64 + wave * 16 + xl;                   // the pixel this lane accumulates and stores
    const int ml = tm * 64 + wave * 16 + gp;                  // the pixel this lane loads for
    const bool ok = ml < g.M;
    const int pm = ok ? ml : 0;
    const int bp_x = (4 * xl + kq) * 4, bp_w = (4 * xl) * 4;  // ds_bpermute source lanes (byte addresses)
    const int hw = g.Ho * g.Wo;
    const int b = pm / hw, rem = pm - b * hw, oh = rem / g.Wo, ow = rem - oh * g.Wo;
    // every address below is a wave-uniform base (scalar registers) + a 32-bit byte offset per lane: the tensors of this path are < 4 GB
    const char* xc = reinterpret_cast<const char*>(x);
    const char* oc = reinterpret_cast<const char*>(om);
    const uint32_t xo = ((uint32_t)(b * g.H * g.W) * C + gc * 8) * 2;   // the lane's 8-channel chunk of its image
    const uint32_t oo = (uint32_t)pm * 128;                             // the pixel's 32-float offset / mask row
    const float hb = (float)(oh * g.stride - g.pad), wb = (float)(ow * g.stride - g.pad);

    // sampling geometry of (pixel, tap): byte offsets of the four (clamped) corner rows and their weights; validity and per-corner
    // zeroing as src/cuda/dcn_v2_im2col_cuda.cu:25-54,178-189
    // (the three offset / mask values of a tap are fetched one tap before its geometry is computed: a load consumed right after it was
    // issued drains the whole in-order load queue -- corner ring and weight slab -- once per tap)
    float omv[3];
    auto om_load = [&](int tap) {
        const char* ot = oc + tap * 8;
        omv[0] = *reinterpret_cast<const float*>(ot + oo); omv[1] = *reinterpret_cast<const float*>(ot + oo + 4);
        omv[2] = *reinterpret_cast<const float*>(oc + 72 + tap * 4 + oo);
    };
    auto tap_setup = [&](int tap, uint32_t (&off)[4], float (&cw)[4]) {
        const int th = (tap * 11) >> 5, tw = tap - 3 * th;
        const float dh = omv[0], dw = omv[1], mk = omv[2];
        const float h = hb + (float)(th * g.dil) + dh, wv = wb + (float)(tw * g.dil) + dw;
        const bool inside = ok && h > -1.f && wv > -1.f && h < (float)g.H && wv < (float)g.W;
        const float hf = floorf(h), wf = floorf(wv);
        const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
        const float lh = h - hf, lw = wv - wf, hh = 1.f - lh, hw_ = 1.f - lw;
        const bool t0 = inside && h0 >= 0, t1 = inside && h1 <= g.H - 1;
        const bool l0 = w0 >= 0, l1 = w1 <= g.W - 1;
        const int ch0 = min(max(h0, 0), g.H - 1), ch1 = min(max(h1, 0), g.H - 1);
        const int cw0 = min(max(w0, 0), g.W - 1), cw1 = min(max(w1, 0), g.W - 1);
        off[0] = xo + (uint32_t)(ch0 * g.W + cw0) * (C * 2); cw[0] = (t0 && l0) ? hh * hw_ * mk : 0.f;
        off[1] = xo + (uint32_t)(ch0 * g.W + cw1) * (C * 2); cw[1] = (t0 && l1) ? hh * lw * mk : 0.f;
        off[2] = xo + (uint32_t)(ch1 * g.W + cw0) * (C * 2); cw[2] = (t1 && l0) ? lh * hw_ * mk : 0.f;
        off[3] = xo + (uint32_t)(ch1 * g.W + cw1) * (C * 2); cw[3] = (t1 && l1) ? lh * lw * mk : 0.f;
    };

    // weight slab sidx = (tap, 128-channel part): rows = this tile's 64 output channels, K offset sidx * SLAB of the [Cout][9*C] matrix
    const char* wc = reinterpret_cast<const char*>(w) + (size_t)tn * 64 * g.K_pad * 2;
    const uint32_t wo = ((uint32_t)(tid / K::CPR) * g.K_pad + (tid % K::CPR) * 8) * 2;
    const int wl = (tid / K::CPR) * K::RB + (tid % K::CPR) * 16;         // LDS offset of the thread's first chunk
    constexpr int RPJ = 256 / K::CPR;                                    // rows between a thread's chunks
    u32x4 wreg[K::WCH];
    auto wload = [&](int sidx) {
        const char* ws = wc + sidx * (K::SLAB * 2);
#pragma unroll
        for (int j = 0; j < K::WCH; ++j) wreg[j] = *reinterpret_cast<const u32x4*>(ws + (size_t)j * RPJ * g.K_pad * 2 + wo);
    };
    auto wstore = [&](char* st) {
#pragma unroll
        for (int j = 0; j < K::WCH; ++j) *reinterpret_cast<u32x4*>(st + wl + j * RPJ * K::RB) = wreg[j];
    };

    f32x4 acc[4], P[4][4];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc[ob] = zero;

    uint32_t off[4], offN[4]; float cw[4], cwN[4];
    om_load(0);
    tap_setup(0, off, cw);
    om_load(1);
#pragma unroll
    for (int q = 0; q < 4; ++q) cw[q] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_w, __builtin_bit_cast(int, cw[q])));
    u32x4 gx[3][4];
    auto gload = [&](u32x4 (&dst)[4], const uint32_t (&o)[4], int step) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const u32x4*>(xc + step * 64 + o[q]);
    };
    wload(0);
    gload(gx[0], off, 0);
    gload(gx[1], off, 1);
    wstore(smem);
    lds_barrier();
    char* cur = smem;
    char* nxt = smem + K::STAGE;
    const int frag_off = xl * K::RB + kq * 16;

    for (int tb = 0; tb < 3; ++tb) {
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) {
            const int tap = tb * 3 + tl;
#pragma unroll
            for (int s = 0; s < K::STEPS; ++s) {
                const int t = tl * K::STEPS + s;                 // static: ring slots are immediates
                const int h = s / K::KCS, kc = s % K::KCS;
                __builtin_amdgcn_sched_barrier(0);
                if (s == 0) {
                    tap_setup(min(tap + 1, 8), offN, cwN); om_load(min(tap + 2, 8));
#pragma unroll
                    for (int q = 0; q < 4; ++q) cwN[q] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_w, __builtin_bit_cast(int, cwN[q])));
                }
                // corner rows of step t + 2 (past the last tap: a harmless repeat of tap 8's rows -- no branch in the loop)
#ifndef MFX_CQ_NOLOAD
                if (s + 2 < K::STEPS) gload(gx[(t + 2) % 3], off, s + 2);
                else gload(gx[(t + 2) % 3], offN, s + 2 - K::STEPS);
#endif
                if (kc == 0) wload(min(tap * K::SPT + h + 1, K::NSLAB - 1));
                __builtin_amdgcn_sched_barrier(0);
                u32x4 wf[4], gq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
#ifdef MFX_CQ_NOPERM
                    for (int d = 0; d < 4; ++d) gq[q][d] = gx[t % 3][q][d];
#else
                    for (int d = 0; d < 4; ++d) gq[q][d] = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_x, (int)gx[t % 3][q][d]);
#endif
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) wf[ob] = *reinterpret_cast<const u32x4*>(cur + frag_off + ob * 16 * K::RB + kc * 64);
#pragma unroll
#ifdef MFX_CQ_ONEQ
                for (int q = 0; q < 1; ++q)
#else
                for (int q = 0; q < 4; ++q)
#endif
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob) {
                        if (s == 0) P[q][ob] = zero;
                        mma_chunk<T>(wf[ob], gq[q], P[q][ob]);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (kc == K::KCS - 1) {                          // slab done: publish the next one
                    wstore(nxt);
                    lds_barrier();
                    char* tmp = cur; cur = nxt; nxt = tmp;
                }
                if (s == K::STEPS - 1) {                         // tap done: acc += w_q * P_q
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[ob][r] += cw[q] * P[q][ob][r];
                    // (pins the sums here: left alone, the optimiser sinks all three taps' scaling to the end of the unrolled body and
                    // keeps 128 more accumulator registers alive)
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) { off[q] = offN[q]; cw[q] = cwN[q]; }
                }
            }
        }
    }

    // epilogue: lane holds channels n0 + 16*ob + 4*kq + r of pixel m
    if (m >= g.M) return;
    T* yrow = reinterpret_cast<T*>(ep.y) + (size_t)m * ep.ldy;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        const int n = tn * 64 + ob * 16 + kq * 4;
        if (n >= ep.Cout) continue;
        float v[4];
        if (ep.scale) { const f32x4 sc = *reinterpret_cast<const f32x4*>(ep.scale + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ob][r] * sc[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ob][r];
        }
        if (ep.shift) { const f32x4 sh = *reinterpret_cast<const f32x4*>(ep.shift + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += sh[r];
        }
        apply_act_chunk<4>(v, ep.act, n);
        uint2 o; o.x = ElemTraits<T>::pack2(v[0], v[1]); o.y = ElemTraits<T>::pack2(v[2], v[3]);
        *reinterpret_cast<uint2*>(yrow + n) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// second form: the corner rows go global -> LDS by the DMA path (`global_load_lds_dwordx4`: no registers, no ds_write, no lane
// permutation) into a wave-private ring of three k-steps, and the MFMA reads its B operand from there.  A wave's 64 lanes write 64
// consecutive 16-byte slots (wave-uniform base + 16 * lane: tools/probes/glds_probe.hip), so a corner's 1 KB image is [pixel][4 chunks],
// 64-byte rows without padding; bank conflicts of the fragment read (16 pixels, same chunk) are avoided by rotating which SOURCE chunk a
// lane fetches: chunk c of pixel p lands in slot (c + (p >> 2)) & 3.  Workgroup = 8 waves = 128 pixels x 64 output channels.
typedef __attribute__((address_space(3))) void cq_lds_void;
typedef __attribute__((address_space(1))) const void cq_glb_void;

template <int C> struct Cq2Cfg {
    static constexpr int SLAB = C < 128 ? C : 128;
    static constexpr int SPT = C / SLAB, KCS = SLAB / 32, STEPS = C / 32, NSLAB = 9 * SPT;
    static constexpr int RB = SLAB * 2 + 16, STAGE = 64 * RB;
    static constexpr int CPR = SLAB / 8, WCH = 64 * CPR / 512;
    static constexpr int D = 3;                                 // ring depth in k-steps
    static constexpr int RING = D * 4096;                       // per wave: D steps x 4 corners x 1 KB
    static constexpr int RING_OFF = (2 * STAGE + 1023) / 1024 * 1024;
    static constexpr int SMEM = RING_OFF + 8 * RING;
};

template <typename T, int C>
__global__ __launch_bounds__(512, 1) void dcn_cq2_kernel(const T* __restrict__ x, const float* __restrict__ om, const T* __restrict__ w,
                                                          DcnQGeom g, EpiArgs ep) {
    using K = Cq2Cfg<C>;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, xl = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    const int gp = lane >> 2, gc = ((lane & 3) - (gp >> 2)) & 3;        // loading role: pixel gp, source chunk gc (lands in slot lane & 3)
    const int m = tm * 128 + wave * 16 + xl;                  // the pixel this lane accumulates and stores
    const int ml = tm * 128 + wave * 16 + gp;                 // the pixel this lane loads for
    const bool ok = ml < g.M;
    const int pm = ok ? ml : 0;
    const int bp_w = (4 * xl) * 4;
    const int hw = g.Ho * g.Wo;
    const int b = pm / hw, rem = pm - b * hw, oh = rem / g.Wo, ow = rem - oh * g.Wo;
    const char* xc = reinterpret_cast<const char*>(x);
    const char* oc = reinterpret_cast<const char*>(om);
    const uint32_t xo = ((uint32_t)(b * g.H * g.W) * C + gc * 8) * 2;
    const uint32_t oo = (uint32_t)pm * 128;
    const float hb = (float)(oh * g.stride - g.pad), wb = (float)(ow * g.stride - g.pad);

    float omv[3];
    auto om_load = [&](int tap) {
        const char* ot = oc + tap * 8;
        omv[0] = *reinterpret_cast<const float*>(ot + oo); omv[1] = *reinterpret_cast<const float*>(ot + oo + 4);
        omv[2] = *reinterpret_cast<const float*>(oc + 72 + tap * 4 + oo);
    };
    auto tap_setup = [&](int tap, uint32_t (&off)[4], float (&cw)[4]) {
        const int th = (tap * 11) >> 5, tw = tap - 3 * th;
        const float dh = omv[0], dw = omv[1], mk = omv[2];
        const float h = hb + (float)(th * g.dil) + dh, wv = wb + (float)(tw * g.dil) + dw;
        const bool inside = ok && h > -1.f && wv > -1.f && h < (float)g.H && wv < (float)g.W;
        const float hf = floorf(h), wf = floorf(wv);
        const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
        const float lh = h - hf, lw = wv - wf, hh = 1.f - lh, hw_ = 1.f - lw;
        const bool t0 = inside && h0 >= 0, t1 = inside && h1 <= g.H - 1;
        const bool l0 = w0 >= 0, l1 = w1 <= g.W - 1;
        const int ch0 = min(max(h0, 0), g.H - 1), ch1 = min(max(h1, 0), g.H - 1);
        const int cw0 = min(max(w0, 0), g.W - 1), cw1 = min(max(w1, 0), g.W - 1);
        off[0] = xo + (uint32_t)(ch0 * g.W + cw0) * (C * 2); cw[0] = (t0 && l0) ? hh * hw_ * mk : 0.f;
        off[1] = xo + (uint32_t)(ch0 * g.W + cw1) * (C * 2); cw[1] = (t0 && l1) ? hh * lw * mk : 0.f;
        off[2] = xo + (uint32_t)(ch1 * g.W + cw0) * (C * 2); cw[2] = (t1 && l0) ? lh * hw_ * mk : 0.f;
        off[3] = xo + (uint32_t)(ch1 * g.W + cw1) * (C * 2); cw[3] = (t1 && l1) ? lh * lw * mk : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) cw[q] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp_w, __builtin_bit_cast(int, cw[q])));
    };

    const char* wc = reinterpret_cast<const char*>(w) + (size_t)tn * 64 * g.K_pad * 2;
    const uint32_t wo = ((uint32_t)(tid / K::CPR) * g.K_pad + (tid % K::CPR) * 8) * 2;
    const int wl = (tid / K::CPR) * K::RB + (tid % K::CPR) * 16;
    constexpr int RPJ = 512 / K::CPR;
    u32x4 wreg[K::WCH];
    auto wload = [&](int sidx) {
        const char* ws = wc + sidx * (K::SLAB * 2);
#pragma unroll
        for (int j = 0; j < K::WCH; ++j) wreg[j] = *reinterpret_cast<const u32x4*>(ws + (size_t)j * RPJ * g.K_pad * 2 + wo);
    };
    auto wstore = [&](char* st) {
#pragma unroll
        for (int j = 0; j < K::WCH; ++j) *reinterpret_cast<u32x4*>(st + wl + j * RPJ * K::RB) = wreg[j];
    };

    char* ring = smem + K::RING_OFF + wave * K::RING;                     // wave-uniform
    const uint32_t brd = (uint32_t)(K::RING_OFF + wave * K::RING) + xl * 64 + ((kq + (xl >> 2)) & 3) * 16;   // the lane's fragment read address (LDS byte offset)
    auto dma = [&](int slot, const uint32_t (&o)[4], int step) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((cq_glb_void*)(xc + step * 64 + o[q]), (cq_lds_void*)(ring + slot * 4096 + q * 1024), 16, 0, 0);
    };

    f32x4 acc[4], P[4][4];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc[ob] = zero;
    uint32_t off[4], offN[4]; float cw[4], cwN[4];
    om_load(0);
    tap_setup(0, off, cw);
    om_load(1);
    wload(0);
    dma(0, off, 0);
    dma(1, off, 1);
    wstore(smem);
    lds_barrier();
    char* cur = smem;
    char* nxt = smem + K::STAGE;
    const int frag_off = xl * K::RB + kq * 16;

    for (int tb = 0; tb < 3; ++tb) {
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) {
            const int tap = tb * 3 + tl;
#pragma unroll
            for (int s = 0; s < K::STEPS; ++s) {
                const int t = tl * K::STEPS + s;
                const int h = s / K::KCS, kc = s % K::KCS;
                __builtin_amdgcn_sched_barrier(0);
                if (s == 0) { tap_setup(min(tap + 1, 8), offN, cwN); om_load(min(tap + 2, 8)); }
                if (kc == 0) wload(min(tap * K::SPT + h + 1, K::NSLAB - 1));
                // corner rows of step t + 2 into ring slot (t + 2) % 3 (its last readers were this wave's MFMAs of step t - 1)
                if (s + 2 < K::STEPS) dma((t + 2) % 3, off, s + 2);
                else dma((t + 2) % 3, offN, s + 2 - K::STEPS);
                // step t's rows have landed when at most the 8 DMA instructions of steps t + 1, t + 2 are outstanding (VMEM ops retire in order;
                // other loads issued in between only make this wait longer, never shorter)
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                u32x4 bx[4];
                asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                             : "=&v"(bx[0]), "=&v"(bx[1]), "=&v"(bx[2]), "=&v"(bx[3])
                             : "v"(brd), "n"((t % 3) * 4096), "n"((t % 3) * 4096 + 1024), "n"((t % 3) * 4096 + 2048), "n"((t % 3) * 4096 + 3072)
                             : "memory");
                u32x4 wf[4];
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) wf[ob] = *reinterpret_cast<const u32x4*>(cur + frag_off + ob * 16 * K::RB + kc * 64);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bx[0]), "+v"(bx[1]), "+v"(bx[2]), "+v"(bx[3]) :: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob) {
                        if (s == 0) P[q][ob] = zero;
                        mma_chunk<T>(wf[ob], bx[q], P[q][ob]);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (kc == K::KCS - 1) {
                    wstore(nxt);
                    lds_barrier();
                    char* tmp = cur; cur = nxt; nxt = tmp;
                }
                if (s == K::STEPS - 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[ob][r] += cw[q] * P[q][ob][r];
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) { off[q] = offN[q]; cw[q] = cwN[q]; }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the two look-ahead DMA groups past the last tap must not outlive the workgroup's LDS

    if (m >= g.M) return;
    T* yrow = reinterpret_cast<T*>(ep.y) + (size_t)m * ep.ldy;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        const int n = tn * 64 + ob * 16 + kq * 4;
        if (n >= ep.Cout) continue;
        float v[4];
        if (ep.scale) { const f32x4 sc = *reinterpret_cast<const f32x4*>(ep.scale + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ob][r] * sc[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ob][r];
        }
        if (ep.shift) { const f32x4 sh = *reinterpret_cast<const f32x4*>(ep.shift + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += sh[r];
        }
        apply_act_chunk<4>(v, ep.act, n);
        uint2 o; o.x = ElemTraits<T>::pack2(v[0], v[1]); o.y = ElemTraits<T>::pack2(v[2], v[3]);
        *reinterpret_cast<uint2*>(yrow + n) = o;
    }
}

template <typename T, int C> static int launch_cq2(const mfx_dcn_desc* d, hipStream_t st) {
    using K = Cq2Cfg<C>;
    DcnQGeom g;
    g.H = d->H; g.W = d->W; g.Ho = d->Ho; g.Wo = d->Wo; g.stride = d->stride; g.pad = d->pad; g.dil = d->dil;
    g.M = d->B * d->Ho * d->Wo; g.tiles_n = d->Cout_pad / 64; g.K_pad = d->K_pad;
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = nullptr; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = 0;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = g.tiles_n;
    auto k = dcn_cq2_kernel<T, C>;
    static bool attr_set = false;
    if (!attr_set) {
        MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, K::SMEM));
        attr_set = true;
    }
    const int tiles = ((g.M + 127) / 128) * g.tiles_n;
    hipLaunchKernelGGL(k, dim3(tiles), dim3(512), K::SMEM, st, reinterpret_cast<const T*>(d->x), d->offmask, reinterpret_cast<const T*>(d->w), g, ep);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

template <typename T, int C> static int launch_cq(const mfx_dcn_desc* d, hipStream_t st) {
    using K = CqCfg<C>;
    DcnQGeom g;
    g.H = d->H; g.W = d->W; g.Ho = d->Ho; g.Wo = d->Wo; g.stride = d->stride; g.pad = d->pad; g.dil = d->dil;
    g.M = d->B * d->Ho * d->Wo; g.tiles_n = d->Cout_pad / 64; g.K_pad = d->K_pad;
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = nullptr; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = 0;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = g.tiles_n;
    auto k = dcn_cq_kernel<T, C>;
    static bool attr_set = false;
    if (!attr_set) {
        MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, K::SMEM));
        attr_set = true;
    }
    const int tiles = ((g.M + 63) / 64) * g.tiles_n;
    hipLaunchKernelGGL(k, dim3(tiles), dim3(256), K::SMEM, st, reinterpret_cast<const T*>(d->x), d->offmask, reinterpret_cast<const T*>(d->w), g, ep);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

template <typename T> static int cq_by_c(const mfx_dcn_desc* d, hipStream_t st) {
    if (g_opt_dcn_cq != 2)
        switch (d->C) {
            case 64: return launch_cq2<T, 64>(d, st);
            case 128: return launch_cq2<T, 128>(d, st);
            case 256: return launch_cq2<T, 256>(d, st);
            case 512: return launch_cq2<T, 512>(d, st);
        }
    switch (d->C) {
        case 64: return launch_cq<T, 64>(d, st);
        case 128: return launch_cq<T, 128>(d, st);
        case 256: return launch_cq<T, 256>(d, st);
        case 512: return launch_cq<T, 512>(d, st);
    }
    return 1;
}

// 0: not taken, 1: ran, < 0: error
int try_dcn_cq(const mfx_dcn_desc* d, hipStream_t st) {
    if (g_opt_dcn_cq == 0 || !d->offmask || !d->w) return 0;
    if (d->dtype != MFX_BF16 && d->dtype != MFX_F16) return 0;
    if (d->kh != 3 || d->kw != 3 || d->K_pad != 9 * d->C || d->Cout_pad % 64 != 0 || d->Cout % 4 != 0) return 0;
    if (d->C != 64 && d->C != 128 && d->C != 256 && d->C != 512) return 0;
    if ((size_t)d->H * d->W * d->C >= (1u << 31)) return 0;
    const int rc = d->dtype == MFX_F16 ? cq_by_c<half_t>(d, st) : cq_by_c<bf16_t>(d, st);
    return rc == MFX_OK ? 1 : (rc == 1 ? 0 : rc);
}

}  // namespace mfx
