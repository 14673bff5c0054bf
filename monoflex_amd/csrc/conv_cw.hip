// 3x3 / stride-1 / pad-1 convolution on 16-bit maps, second form of the LDS-halo kernel (conv_halo.hip): the SAME decomposition
// (8 x 16-pixel tile, halo patch staged once in LDS, wave-private output slices, fragment-major weights L2 -> registers, K-split
// waves, wave-private epilogue) with every piece of geometry a COMPILE-TIME constant and the K loop written as an explicit
// software pipeline.
//
// Why (round 5, ISA of conv3x3_wave_kernel<bf16, bf16, 2, 2, 1>, `hipcc --save-temps`): with the channel group, patch width and
// weight row length as run-time values, one K step of that kernel is 16 MFMAs inside ~60 other instructions -- per step three
// v_mul_lo_u32 and a dozen VALU to rebuild the lane's patch address from (tap, channel), two exec-masked branch diamonds around the
// weight fetch (null-pointer / K-padding tests), 64-bit VALU address arithmetic per weight load -- and two scheduling accidents
// that follow from the control flow: `s_waitcnt vmcnt(0)` at the top of EVERY step (the branches hide the load count from the
// waitcnt pass, so the five-deep weight ring drains to empty each step and every step exposes one full L2 round trip) and
// `s_waitcnt lgkmcnt(1)` in front of every MFMA pair (pixel fragments are read one pair = 34 cycles ahead of their use against an
// LDS round trip of 64-128).  The kernel was instruction-issue- and latency-bound by its own address code, not by a hardware unit
// (profiles/r03_conv_halo_probes.md: no unit above 25 % busy).
//
// Here <CG, CT, WN, FN, WK> are template parameters, so
//   * a pixel-fragment read is `ds_read_b128 v, v_lane offset:imm` -- the lane's patch base is computed once per workgroup;
//   * a weight-fragment fetch is `global_load_dwordx4 v, v_lane16, s[base] offset:imm` against a wave-uniform base advanced by SALU;
//   * the K loop is fully unrolled and branch-free (the waitcnt pass counts exactly: vmcnt(2 FN) / lgkmcnt(4) waits);
//   * pixel fragments are read HALF A STEP (8 FN MFMAs = 136+ cycles) ahead of their use into two 4-row register sets, weights two
//     steps ahead through a 3-deep ring; `sched_barrier`s pin that order;
//   * BN scale / shift are fetched before the K loop instead of at the head of the epilogue.
// Same K order, same accumulation order, same epilogue arithmetic as conv3x3_wave_kernel: outputs are bit-identical
// (tests/test_gpu_ops.py::test_conv_cw_kernel_is_bit_identical_to_the_halo_kernel).
//
// Reference layers: model/backbone/dla_dcn.py:84-98 (BasicBlock conv1 / conv2 of levels 2-5), 246-259 (Tree).
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "igemm.h"
#include <type_traits>

namespace mfx {

struct CwGeom { int B, H, W, Ho, Wo, tiles_x, tiles_y, tiles_n; };      // H x W: input map, Ho x Wo: output map (= H x W at stride 1)

constexpr int kCwRows = 8;

// patch of an 8 x 16 output tile at stride S: (7 S + 3) x (15 S + 3) input pixels (10 x 18, or 17 x 33 at stride 2)
template <int CG, int WN, int FN, int WK, int S = 1, int ROWS = kCwRows> struct CwSmem {
    static constexpr int PS = CG * 2 + 16;                                    // patch pixel stride: 16 consecutive pixels on 16 distinct 16-byte bank slots
    static constexpr int PH = (ROWS - 1) * S + 3, PW = 15 * S + 3;
    static constexpr int patch_bytes = PH * PW * PS;
    static constexpr int stage_ld = FN * 16 + 4;
    static constexpr int stage_bytes = 16 * stage_ld * 4;
    static constexpr int reduce_bytes = WN * (WK - 1) * ROWS * FN * 64 * 16;
    static constexpr int main_bytes = patch_bytes > reduce_bytes ? patch_bytes : reduce_bytes;
    static constexpr int total = main_bytes + WN * WK * stage_bytes;
};

// CG = channels per patch pass, CT = input channels of the layer (a multiple of CG), WN x FN x 16 = output channels per workgroup,
// WK = waves sharing an output slice (they take the K steps round-robin); OCC = waves per SIMD the register allocation must admit
// ROWS = output rows of a tile (8; 6 for maps whose height is a multiple of 6 but not of 8 -- the 12 x 40 level-5 maps: 2 x 3 tiles of 6 x 16 instead of
// 2 x 3 tiles of 8 x 16, a quarter less matrix work)
// ST: train-mode BN statistics of the stored values accumulated by the epilogue (mfx_conv_desc.stats; conv_halo.hip's arithmetic)
template <typename T, typename TO, int CG, int CT, int WN, int FN, int WK, int OCC, int S = 1, int ROWS = kCwRows, bool ST = false>
__global__ __launch_bounds__(WN * WK * 64, OCC) void conv3x3_cw_kernel(const T* __restrict__ x, const u32x4* __restrict__ wfm, CwGeom g, EpiArgs ep) {
    static_assert(sizeof(T) == 2, "16-bit maps");
    static_assert(ROWS % 2 == 0, "the K loop reads the pixel fragments in two half-tiles");
    using SM = CwSmem<CG, WN, FN, WK, S, ROWS>;
    constexpr int NT = WN * WK * 64, FM = ROWS, HR = ROWS / 2, PW = SM::PW, PH = SM::PH;
    constexpr int PS = SM::PS, CPP = CG / 8;
    constexpr int KS = CG / 32;                       // 64-byte K steps per tap and channel group
    constexpr int NG = CT / CG, FSTEPS = (9 * CT + 63) / 64 * 2;   // channel groups; steps per fragment row of the fragment-major weights (K padded to 128 bytes)
    constexpr int SPG = 9 * KS, NL = SPG / WK;        // steps per group; steps per wave and group
    static_assert(KS % WK == 0 && CT % CG == 0 && FM % WK == 0, "K split must divide the steps of a tap and the rows");
    constexpr int BN = WN * FN * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wk = wave / WN;
    const int xl = lane & 15, kq = lane >> 4;

    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % g.tiles_n; tile /= g.tiles_n;
    const int tx = tile % g.tiles_x; tile /= g.tiles_x;
    const int ty = tile % g.tiles_y; const int b = tile / g.tiles_y;
    const int x0 = tx * 16, y0 = ty * ROWS, n0 = tn * BN + wn * (FN * 16);

    char* patch = smem;
    float* stage = reinterpret_cast<float*>(smem + SM::main_bytes + wave * SM::stage_bytes);

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // BN scale / shift of this lane's output columns: in flight during the whole K loop
    float sc[FN], sh[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        sc[j] = ep.scale ? ep.scale[n0 + j * 16 + xl] : 1.f;
        sh[j] = ep.shift ? ep.shift[n0 + j * 16 + xl] : 0.f;
    }

    // lane's patch base: pixel column xl, K chunk kq of the wave's first step (wave wk starts at step wk of every tap)
    const char* abase = patch + xl * S * PS + kq * 16 + wk * 64;
    const uint32_t loff = (uint32_t)lane * 16u;
    const T* xb = x + (size_t)b * g.H * g.W * CT;

    for (int grp = 0; grp < NG; ++grp) {
        if (grp > 0) __syncthreads();                         // every wave is done reading the previous patch
        // ---- halo patch: 10 x 18 input pixels x CG channels, zero outside the image.  Branch-free: every lane loads from a CLAMPED (always
        // valid) address and the out-of-image chunks are zeroed by a select, so all of a lane's loads are in flight together (as exec-masked
        // branch diamonds the compiler serialised some of them behind `s_waitcnt vmcnt(0)`); 32-bit element offsets against the image base
        {
            constexpr int nchunks = PH * PW * CPP;
            constexpr int PU = (nchunks + NT - 1) / NT < 12 ? (nchunks + NT - 1) / NT : 12;
            const T* xg = xb + grp * CG;
#pragma unroll
            for (int base = 0; base < nchunks; base += NT * PU) {
                u32x4 pr[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    int idx = base + u * NT + tid;
                    if (base + u * NT + NT > nchunks) idx = idx < nchunks ? idx : nchunks - 1;
                    const int pix = idx / CPP, ch = idx % CPP;
                    const int py = pix / PW, px = pix - py * PW;
                    const int iy = y0 * S - 1 + py, ix = x0 * S - 1 + px;
                    const bool in = iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                    const int cy = min(max(iy, 0), g.H - 1), cx = min(max(ix, 0), g.W - 1);
                    const u32x4 z = *reinterpret_cast<const u32x4*>(xg + (uint32_t)((cy * g.W + cx) * CT + ch * 8));
                    pr[u] = in ? z : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int idx = base + u * NT + tid;
                    if (base + u * NT + NT <= nchunks || idx < nchunks) *reinterpret_cast<u32x4*>(patch + (idx / CPP) * PS + (idx % CPP) * 16) = pr[u];
                }
            }
        }

        // wave-uniform base of this wave's weight fragments: fragment row n0/16, step (grp * KS + wk) of tap 0
        const char* wbase = reinterpret_cast<const char*>(wfm) + ((size_t)(n0 >> 4) * FSTEPS + grp * KS + wk) * 1024;
        // step m of this wave: q = m * WK (+ wk, folded into the bases) -> tap q / KS, K step q % KS inside the tap
        auto bload = [&](int m, u32x4 (&bf)[FN]) {
            const int q = m * WK, tap = q / KS, ks = q % KS;
            const int st = tap * (CT / 32) + ks;
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(wbase + ((size_t)j * FSTEPS + st) * 1024 + loff);
        };
        auto aread = [&](int m, int i) -> u32x4 {
            const int q = m * WK, tap = q / KS, ks = q % KS;
            const int th = tap / 3, tw = tap - th * 3;
            return *reinterpret_cast<const u32x4*>(abase + ((th + i * S) * PW + tw) * PS + ks * 64);
        };

        constexpr int RING = 3;
        u32x4 wb[RING][FN];
        bload(0, wb[0]);
        if (NL > 1) bload(1, wb[1]);
        __syncthreads();                                      // patch visible to all waves

        u32x4 a0[HR], a1[HR];
#pragma unroll
        for (int i = 0; i < HR; ++i) a0[i] = aread(0, i);
#pragma unroll
        for (int m = 0; m < NL; ++m) {
            if (m + 2 < NL) bload(m + 2, wb[(m + 2) % RING]);
#pragma unroll
            for (int i = 0; i < HR; ++i) a1[i] = aread(m, HR + i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < HR; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) mma_chunk<T>(a0[i], wb[m % RING][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (m + 1 < NL) {
#pragma unroll
                for (int i = 0; i < HR; ++i) a0[i] = aread(m + 1, i);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < HR; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) mma_chunk<T>(a1[i], wb[m % RING][j], acc[HR + i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- K-split: partial accumulators -> LDS (the patch is dead), wave wk sums and finishes rows wk*RW .. +RW
    constexpr int RW = FM / WK;
    if constexpr (WK > 1) {
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
        for (int dst = 0; dst < WK; ++dst) {
            if (dst == wk) continue;
            const int slot = wk < dst ? wk : wk - 1;
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    red[((((wn * WK + dst) * (WK - 1) + slot) * RW + r) * FN + j) * 64 + lane] = acc[dst * RW + r][j];
        }
        __syncthreads();
        f32x4 own[RW][FN];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < WK; ++q) if (q == wk) t = acc[q * RW + r][j];
#pragma unroll
                for (int slot = 0; slot < WK - 1; ++slot) t += red[((((wn * WK + wk) * (WK - 1) + slot) * RW + r) * FN + j) * 64 + lane];
                own[r][j] = t;
            }
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[r][j] = own[r][j];
    }

    // ---- epilogue, wave-private: accumulator row-fragment -> stage -> 16-byte stores (conv_halo.hip's arithmetic, bit for bit).
    // The residual chunks of ALL the wave's rows are fetched up front (the K loop's fragment registers are dead by now, so they cost no
    // occupancy): one exposed memory round trip per tile instead of one per output row; 32-bit element offsets.
    constexpr int LDS_ = SM::stage_ld;
    constexpr int OE = ElemTraits<TO>::ELEMS;                 // output elements per 16-byte store: 8 (16-bit maps) / 4 (fp32: the DCN offset / mask conv)
    constexpr int GPR = FN * 16 / OE;
    constexpr int RITEMS = (16 * GPR + 63) / 64;
    constexpr bool kRes = std::is_same<T, TO>::value;         // residual operand: same type as the output (the launcher rejects the other case)
    const T* res = kRes ? reinterpret_cast<const T*>(ep.res) : nullptr;
    TO* y = reinterpret_cast<TO*>(ep.y);
    const uint32_t pix0 = (uint32_t)((b * g.Ho + y0) * g.Wo + x0);            // first pixel of the tile (32-bit: M * ld < 2^31 checked by the launcher)
    f32x2 st_s[FN], st_q[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) { st_s[j] = f32x2{0.f, 0.f}; st_q[j] = f32x2{0.f, 0.f}; }
    u32x4 rres[RW][RITEMS];
    if (res) {
#pragma unroll
        for (int ii = 0; ii < RW; ++ii) {
            const int i = WK > 1 ? wk * RW + ii : ii;
#pragma unroll
            for (int q = 0; q < RITEMS; ++q) {
                const int it = q * 64 + lane;
                const int px = it / GPR, ng = it - px * GPR;
                const bool ok = (16 * GPR % 64 == 0 || it < 16 * GPR) && y0 + i < g.Ho && x0 + px < g.Wo && n0 + ng * OE < ep.Cout;
                u32x4 z = {0u, 0u, 0u, 0u};
                if (ok) z = *reinterpret_cast<const u32x4*>(res + (pix0 + (uint32_t)(i * g.Wo + px)) * (uint32_t)ep.ldres + (uint32_t)(n0 + ng * OE));
                rres[ii][q] = z;
            }
        }
    }
#pragma unroll
    for (int ii = 0; ii < RW; ++ii) {
        const int i = WK > 1 ? wk * RW + ii : ii;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[ii][j][r] * sc[j] + sh[j];
                stage[((lane >> 4) * 4 + r) * LDS_ + j * 16 + xl] = v[r];
            }
            if constexpr (ST) {
                if (y0 + i < g.Ho) {                          // every lane sums its column over its pixels, of the values AS STORED
                    const int px = x0 + (lane >> 4) * 4;
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        f32x2 vr = {ElemTraits<TO>::round(v[r]), ElemTraits<TO>::round(v[r + 1])};
                        if (px + r + 1 >= g.Wo) { if (px + r >= g.Wo) vr[0] = 0.f; vr[1] = 0.f; }      // ragged right edge only
                        st_s[j] += vr; st_q[j] += vr * vr;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < RITEMS; ++q) {
            const int it = q * 64 + lane;
            const int px = it / GPR, ng = it - px * GPR;
            const int gn = n0 + ng * OE;
            const bool ok = (16 * GPR % 64 == 0 || it < 16 * GPR) && y0 + i < g.Ho && x0 + px < g.Wo && gn < ep.Cout;
            if (ok) {
                float v[OE];
#pragma unroll
                for (int e = 0; e < OE; e += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(stage + px * LDS_ + ng * OE + e);
                    v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
                }
                if constexpr (kRes) {
                    if (res) {
                        float rv[OE];
                        ElemTraits<T>::unpack(rres[ii][q], rv);
#pragma unroll
                        for (int e = 0; e < OE; ++e) v[e] += rv[e];
                    }
                }
                apply_act_chunk<OE>(v, ep.act, gn);
                *reinterpret_cast<u32x4*>(y + (pix0 + (uint32_t)(i * g.Wo + px)) * (uint32_t)ep.ldy + (uint32_t)gn) = ElemTraits<TO>::pack(v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (ST) {                                       // the four pixel groups of the wave meet by shuffles, lanes 0..15 add into the layer's scratch
        float* sp = ep.stats + (size_t)(blockIdx.x % ep.stats_ncopy) * 2 * ep.Cout;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            float a = st_s[j][0] + st_s[j][1], q = st_q[j][0] + st_q[j][1];
            a += __shfl_xor(a, 16); q += __shfl_xor(q, 16);
            a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);
            const int n = n0 + j * 16 + xl;
            if (lane < 16 && n < ep.Cout) { unsafeAtomicAdd(sp + n, a); unsafeAtomicAdd(sp + ep.Cout + n, q); }
        }
    }
}

int g_opt_halo_cw = 1;       // option "halo_cw": 0 = conv3x3_wave_kernel only, 1 = this kernel where an instantiation exists

template <typename T, typename TO, int CG, int CT, int WN, int FN, int WK, int OCC, int S = 1, int ROWS = kCwRows, bool ST = false>
static int launch_cw(const mfx_conv_desc* d, hipStream_t st) {
    using SM = CwSmem<CG, WN, FN, WK, S, ROWS>;
    constexpr int BN = WN * FN * 16;
    CwGeom g;
    g.B = d->B; g.H = d->H; g.W = d->W; g.Ho = d->Ho; g.Wo = d->Wo;
    g.tiles_x = (d->Wo + 15) / 16; g.tiles_y = (d->Ho + ROWS - 1) / ROWS; g.tiles_n = d->Cout_pad / BN;
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = d->res; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = d->ldres;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = g.tiles_n;
    ep.stats = ST ? d->stats : nullptr; ep.stats_ncopy = d->stats_ncopy > 0 ? d->stats_ncopy : 1;
    auto k = conv3x3_cw_kernel<T, TO, CG, CT, WN, FN, WK, OCC, S, ROWS, ST>;
    constexpr int smem = SM::total;
    static bool attr_done = false;
    if (!attr_done && smem > 64 * 1024) {
        MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    const int tiles = g.tiles_n * g.tiles_x * g.tiles_y * d->B;
    hipLaunchKernelGGL(k, dim3(tiles), dim3(WN * WK * 64), smem, st, reinterpret_cast<const T*>(d->x), reinterpret_cast<const u32x4*>(d->w_frag), g, ep);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

int g_opt_cw_rows6 = 1;      // option "cw_rows6": 0 = 8-row tiles everywhere
static inline bool cw_rows6(const mfx_conv_desc* d) { return g_opt_cw_rows6 && d->Ho % 6 == 0 && d->Ho % 8 != 0 && d->Ho <= 48; }

template <typename T, int OCC> static int cw_shape(const mfx_conv_desc* d, int v, hipStream_t st) {
    const int C = d->Ck;
    if (d->stats) {                                           // training forward: statistics epilogue (outputs up to 128 channels wide)
        if (v == 6 && C == 64) return launch_cw<T, T, 64, 64, 2, 2, 1, OCC, 1, kCwRows, true>(d, st);
        if (v == 7 && C == 128) return launch_cw<T, T, 128, 128, 4, 2, 1, OCC, 1, kCwRows, true>(d, st);
        return 1;
    }
    // 32 input channels: the data gradients of the DCN modules' 27-channel offset / mask convs (training)
    if (v == 6 && C == 32) return launch_cw<T, T, 32, 32, 2, 2, 1, OCC>(d, st);
    if ((v == 7 || v == 11) && C == 32) return launch_cw<T, T, 32, 32, 4, 2, 1, OCC>(d, st);
    if (v == 12 && C == 128) return launch_cw<T, T, 128, 128, 2, 2, 2, OCC>(d, st);
    if (v == 13 && C == 256) return launch_cw<T, T, 256, 256, 2, 2, 4, OCC>(d, st);
    if (v == 5 && C == 64) return launch_cw<T, T, 64, 64, 4, 4, 1, OCC>(d, st);
    if (v == 6 && C == 64) return launch_cw<T, T, 64, 64, 2, 2, 1, OCC>(d, st);
    if (v == 7 && C == 128) return launch_cw<T, T, 128, 128, 4, 2, 1, OCC>(d, st);
    if (v == 7 && C == 64) return launch_cw<T, T, 64, 64, 4, 2, 1, OCC>(d, st);
    if (v == 11 && C == 256) return launch_cw<T, T, 256, 256, 4, 2, 2, OCC>(d, st);
    if (v == 11 && C == 512) return cw_rows6(d) ? launch_cw<T, T, 256, 512, 4, 2, 2, OCC, 1, 6>(d, st) : launch_cw<T, T, 256, 512, 4, 2, 2, OCC>(d, st);
    return 1;                                                 // no instantiation: the caller falls back
}

// the four stride-2 convs that open DLA levels 2-5 (dla_dcn.py:84-98 conv1 of the first BasicBlock): 17 x 33 patches; 32-channel groups keep a
// patch at 45 KB where no wave splits K, the K-split variant needs two steps per tap (64-channel groups, 81 KB: one workgroup per CU)
template <typename T> static int cw_shape_s2(const mfx_conv_desc* d, int v, hipStream_t st) {
    const int C = d->Ck;
    if (d->stats) {
        if (v == 6 && C == 32) return launch_cw<T, T, 32, 32, 2, 2, 1, 2, 2, kCwRows, true>(d, st);
        if (v == 7 && C == 64) return launch_cw<T, T, 32, 64, 4, 2, 1, 2, 2, kCwRows, true>(d, st);
        return 1;
    }
    if (v == 6 && C == 32) return launch_cw<T, T, 32, 32, 2, 2, 1, 2, 2>(d, st);
    if (v == 7 && C == 64) return launch_cw<T, T, 32, 64, 4, 2, 1, 2, 2>(d, st);
    if (v == 11 && C == 128) return launch_cw<T, T, 64, 128, 4, 2, 2, 2, 2>(d, st);
    if (v == 11 && C == 256) return cw_rows6(d) ? launch_cw<T, T, 64, 256, 4, 2, 2, 2, 2, 6>(d, st) : launch_cw<T, T, 64, 256, 4, 2, 2, 2, 2>(d, st);
    return 1;
}

// the 27-channel DCN offset / mask convs of the wide layers (fp32 out, sigmoid on the mask channels, N = 32): one output slice per workgroup,
// four waves share it and split K (conv_halo.hip's variants 8 = 32 channels per workgroup, 10 = 16)
template <typename T> static int cw_shape_f32out(const mfx_conv_desc* d, int v, hipStream_t st) {
    const int C = d->Ck;
    if (v == 8 && C == 128) return launch_cw<T, float, 128, 128, 1, 2, 4, 2>(d, st);
    if (v == 8 && C == 256) return launch_cw<T, float, 256, 256, 1, 2, 4, 2>(d, st);
    if (v == 8 && C == 512) return launch_cw<T, float, 256, 512, 1, 2, 4, 2>(d, st);      // (8 rows x 4-way K split: 6 rows do not divide)
    if (v == 10 && C == 128) return launch_cw<T, float, 128, 128, 1, 1, 4, 2>(d, st);
    if (v == 10 && C == 256) return launch_cw<T, float, 256, 256, 1, 1, 4, 2>(d, st);
    if (v == 10 && C == 512) return launch_cw<T, float, 256, 512, 1, 1, 4, 2>(d, st);
    return 1;
}

// returns MFX_OK (0) if this kernel ran, 1 if there is no instantiation for the shape / variant (caller runs conv3x3_wave_kernel), < 0 on error.
// `v` is conv_halo.hip's variant number (6: 2 waves x 32 channels, 7: 4 x 32, 11: 4 x 32 with a 2-way K split; 8 / 10: one slice, 4-way K split)
int try_conv_cw(const mfx_conv_desc* d, int v, hipStream_t st) {
    if (!g_opt_halo_cw || !d->w_frag || (d->stride != 1 && d->stride != 2)) return 1;
    if (d->stats && (d->out_dtype != d->dtype || d->Cout_pad > 128)) return 1;
    if (d->dtype != MFX_BF16 && d->dtype != MFX_F16) return 1;
    if (d->K_pad != (9 * d->Ck + 63) / 64 * 64) return 1;
    if ((long long)d->M * (d->ldy > d->ldres ? d->ldy : d->ldres) >= (1ll << 31) || (long long)d->H * d->W * d->Ck >= (1ll << 31)) return 1;      // 32-bit element offsets
    if (d->out_dtype == MFX_F32) {
        if (d->stride != 1 || d->res || d->Cout % 4 != 0 || d->Cout_pad != 32 || (v != 8 && v != 10)) return 1;
        return d->dtype == MFX_F16 ? cw_shape_f32out<half_t>(d, v, st) : cw_shape_f32out<bf16_t>(d, v, st);
    }
    if (d->out_dtype != d->dtype || d->Cout % 8 != 0 || d->act == MFX_ACT_DCN_OFFMASK) return 1;
    if ((v == 6 || v == 12 || v == 13) && d->Cout_pad % 64 != 0) return 1;
    if ((v == 7 || v == 11) && d->Cout_pad % 128 != 0) return 1;
    if (v == 5 && d->Cout_pad % 256 != 0) return 1;
    if (d->stride == 2) return d->dtype == MFX_F16 ? cw_shape_s2<half_t>(d, v, st) : cw_shape_s2<bf16_t>(d, v, st);
    if (d->dtype == MFX_F16) return cw_shape<half_t, 2>(d, v, st);
    return cw_shape<bf16_t, 2>(d, v, st);
}

}  // namespace mfx
