// 3x3 / stride-1 / pad-1 convolution in SPLIT PRECISION (MFX_F16X2: fp32 maps, fp16 (hi, lo) MFMA operand pairs -- the mode that carries the
// north-star gate), compile-time-geometry form (r06).  conv_cw.hip did this for the 16-bit maps in r05 (-18 ... -44 % against the run-time-geometry
// kernel of conv_halo.hip); the split mode kept running conv3x3_wave_kernel<f32s_t, .., PR>, 2.3 ms of its 6.4 ms step (r06 timeline).  Same decomposition
// and the same arithmetic per output as that kernel's pair-walking K loop (conv_halo.hip:64-67): 8 x 16-pixel tile, 64-channel halo patch staged once
// in LDS as [4 hi | 4 lo] fp16 chunks (lds_operand<f32s_t>: range sentinel included), wave-private output slices, the K loop over step PAIRS -- 32
// channels of one tap: the hi (lo) halves of two consecutive 16-element steps form one 8-element fp16 operand, three products per pair (hi.hi, lo.hi,
// hi.lo) -- weights from mfx_conv_desc.w_frag_pair (ops.pair_steps) L2 -> registers, K-split waves, wave-private epilogue.  What is new is what conv_cw.hip
// brought: every LDS offset an immediate (a wave-uniform select where a 4-way K split does not divide the two pairs of a tap), the K loop fully unrolled and
// branch-free with exact waitcnt counts, pixel fragments read half a step ahead into two register sets, weights two pairs ahead through a 3-deep ring,
// BN scale / shift fetched before the loop, residual chunks before the first epilogue row.
//
// Reference layers: model/backbone/dla_dcn.py:84-98 (BasicBlock conv1 / conv2 of levels 2-5), DCNv2/dcn_v2.py:118-122 (the 27-channel offset / mask convs).
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "igemm.h"
#include <type_traits>

namespace mfx {

struct CwsGeom { int B, H, W, tiles_x, tiles_y, tiles_n; };

template <int WN, int FN, int WK, int ROWS> struct CwsSmem {
    static constexpr int CG = 64, PS = CG * 4 + 16, PH = ROWS + 2, PW = 18;
    static constexpr int patch_bytes = PH * PW * PS;
    static constexpr int stage_ld = FN * 16 + 4;
    static constexpr int stage_bytes = 16 * stage_ld * 4;
    static constexpr int reduce_bytes = WN * (WK - 1) * ROWS * FN * 64 * 16;
    static constexpr int main_bytes = patch_bytes > reduce_bytes ? patch_bytes : reduce_bytes;
    static constexpr int total = main_bytes + WN * WK * stage_bytes;
};

// CT = input channels (a multiple of 64), WN x FN x 16 = output channels per workgroup, WK = waves sharing an output slice (they take the pairs round-robin)
template <int CT, int WN, int FN, int WK, int ROWS = 8>
__global__ __launch_bounds__(WN * WK * 64, 2) void conv3x3_cws_kernel(const float* __restrict__ x, const u32x4* __restrict__ wfm, CwsGeom g, EpiArgs ep) {
    using SM = CwsSmem<WN, FN, WK, ROWS>;
    constexpr int CG = 64, NT = WN * WK * 64, FM = ROWS, HR = ROWS / 2, PW = SM::PW, PH = SM::PH, PS = SM::PS, CPP = CG / 4;
    constexpr int KP = CG / 32;                       // pairs per tap and channel group (2)
    constexpr int NG = CT / CG, FSTEPS = 9 * CT / 16; // channel groups; 16-element steps per fragment row of the weights (K = 9 CT, a multiple of 32)
    constexpr int NP = 9 * KP;                        // pairs per group (18)
    constexpr int NLMAX = (NP + WK - 1) / WK;         // pairs per wave and group (the last one only for waves wk < NP % WK when WK does not divide NP)
    constexpr bool EVEN = NP % WK == 0;
    constexpr bool IMM = KP % WK == 0;                // the wave's pair offset folds into its bases: every LDS / weight offset is an immediate
    static_assert(ROWS % 2 == 0 && CT % CG == 0 && FM % WK == 0, "shape");
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

    float sc[FN], sh[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        sc[j] = ep.scale ? ep.scale[n0 + j * 16 + xl] : 1.f;
        sh[j] = ep.shift ? ep.shift[n0 + j * 16 + xl] : 0.f;
    }

    // lane's patch base: pixel column xl, 4-element chunk kq of the pair's first step (its second step is 64 bytes on); IMM: + the wave's pair inside a tap
    const char* abase = patch + xl * PS + kq * 16 + (IMM ? wk * 128 : 0);
    const uint32_t loff = (uint32_t)lane * 16u;
    const float* xb = x + (size_t)b * g.H * g.W * CT;

    for (int grp = 0; grp < NG; ++grp) {
        if (grp > 0) __syncthreads();
        {   // ---- halo patch: (ROWS + 2) x 18 input pixels x 64 channels, zero outside the image; branch-free batches (see conv_cw.hip)
            constexpr int nchunks = PH * PW * CPP;
            constexpr int PU = (nchunks + NT - 1) / NT < 12 ? (nchunks + NT - 1) / NT : 12;
            const float* xg = xb + grp * CG;
#pragma unroll
            for (int base = 0; base < nchunks; base += NT * PU) {
                u32x4 pr[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    int idx = base + u * NT + tid;
                    if (base + u * NT + NT > nchunks) idx = idx < nchunks ? idx : nchunks - 1;
                    const int pix = idx / CPP, ch = idx % CPP;
                    const int py = pix / PW, px = pix - py * PW;
                    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                    const bool in = iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                    const int cy = min(max(iy, 0), g.H - 1), cx = min(max(ix, 0), g.W - 1);
                    const u32x4 z = *reinterpret_cast<const u32x4*>(xg + (uint32_t)((cy * g.W + cx) * CT + ch * 4));
                    pr[u] = in ? z : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int idx = base + u * NT + tid;
                    if (base + u * NT + NT <= nchunks || idx < nchunks)
                        *reinterpret_cast<u32x4*>(patch + (idx / CPP) * PS + (idx % CPP) * 16) = lds_operand<f32s_t>(pr[u]);
                }
            }
        }

        // pair m of this wave: q = m * WK + wk -> tap q / KP, pair q % KP inside the tap.  IMM: wk is folded into the bases (q = m * WK).
        // fragment of (tap, group, pair kp): 16-element step (tap * CT + grp * 64 + kp * 32) / 16 (even): hi operand there, lo operand at the next one
        const char* wbase = reinterpret_cast<const char*>(wfm) + ((size_t)(n0 >> 4) * FSTEPS + grp * (CG / 16) + (IMM ? wk * 2 : 0)) * 1024;
        auto tap_of = [&](int m) { return IMM ? (m * WK) / KP : 0; };
        auto bload = [&](int m, u32x4 (&bh)[FN], u32x4 (&bl)[FN]) {
            size_t st;
            if constexpr (IMM) st = (size_t)(tap_of(m) * (CT / 16) + ((m * WK) % KP) * 2);
            else { const int q = m * WK + wk; st = (size_t)((q / KP) * (CT / 16) + (q % KP) * 2); }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                bh[j] = *reinterpret_cast<const u32x4*>(wbase + ((size_t)j * FSTEPS + st) * 1024 + loff);
                bl[j] = *reinterpret_cast<const u32x4*>(wbase + ((size_t)j * FSTEPS + st + 1) * 1024 + loff);
            }
        };
        // pixel operand pair of row i: hi = [hi(step 2p) | hi(step 2p + 1)], lo likewise: four 8-byte LDS reads
        auto aoff = [&](int m) -> int {
            if constexpr (IMM) { const int q = m * WK, tap = q / KP, kp = q % KP, th = tap / 3, tw = tap - th * 3; return (th * PW + tw) * PS + kp * 128; }
            else { const int q = m * WK + wk, tap = q / KP, kp = q % KP, th = tap / 3, tw = tap - th * 3; return (th * PW + tw) * PS + kp * 128; }
        };
        auto aread = [&](int off, int i, u32x4& ph, u32x4& pl) {
            const char* p = abase + off + i * (PW * PS);
            const uint2 h0 = *reinterpret_cast<const uint2*>(p), h1 = *reinterpret_cast<const uint2*>(p + 64);
            const uint2 l0 = *reinterpret_cast<const uint2*>(p + 8), l1 = *reinterpret_cast<const uint2*>(p + 72);
            ph = u32x4{h0.x, h0.y, h1.x, h1.y}; pl = u32x4{l0.x, l0.y, l1.x, l1.y};
        };
        const int nl = EVEN ? NLMAX : (NP - wk + WK - 1) / WK;      // this wave's pairs in the group (wave-uniform)

        constexpr int RING = 3;
        u32x4 wh[RING][FN], wl[RING][FN];
        bload(0, wh[0], wl[0]);
        if (NLMAX > 1) bload(1, wh[1], wl[1]);
        __syncthreads();                                      // patch visible to all waves

        u32x4 a0h[HR], a0l[HR], a1h[HR], a1l[HR];
        {
            const int o0 = aoff(0);
#pragma unroll
            for (int i = 0; i < HR; ++i) aread(o0, i, a0h[i], a0l[i]);
        }
#pragma unroll
        for (int m = 0; m < NLMAX; ++m) {
            if (!EVEN && m == NLMAX - 1 && m >= nl) break;    // (wave-uniform: the last pair exists only for the first NP % WK waves)
            if (m + 2 < NLMAX) bload(EVEN || m + 2 < NLMAX - 1 ? m + 2 : (m + 2 < nl ? m + 2 : m + 1), wh[(m + 2) % RING], wl[(m + 2) % RING]);
            const int om_ = aoff(m);
#pragma unroll
            for (int i = 0; i < HR; ++i) aread(om_, HR + i, a1h[i], a1l[i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < HR; ++i) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0h[i]), __builtin_bit_cast(f16x8, wh[m % RING][j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0l[i]), __builtin_bit_cast(f16x8, wh[m % RING][j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0h[i]), __builtin_bit_cast(f16x8, wl[m % RING][j]), acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (m + 1 < NLMAX) {
                const int on = aoff(EVEN || m + 1 < NLMAX - 1 ? m + 1 : (m + 1 < nl ? m + 1 : m));
#pragma unroll
                for (int i = 0; i < HR; ++i) aread(on, i, a0h[i], a0l[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < HR; ++i) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[HR + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1h[i]), __builtin_bit_cast(f16x8, wh[m % RING][j]), acc[HR + i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[HR + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1l[i]), __builtin_bit_cast(f16x8, wh[m % RING][j]), acc[HR + i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[HR + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1h[i]), __builtin_bit_cast(f16x8, wl[m % RING][j]), acc[HR + i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- K-split: partial accumulators -> LDS (the patch is dead), wave wk sums and finishes rows wk*RW .. +RW   (conv_cw.hip's scheme)
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

    // ---- epilogue, wave-private: accumulator row-fragment -> stage -> 16-byte fp32 stores; residual chunks of all rows fetched up front
    constexpr int LDS_ = SM::stage_ld;
    constexpr int OE = 4, GPR = FN * 16 / OE, RITEMS = (16 * GPR + 63) / 64;
    const float* res = reinterpret_cast<const float*>(ep.res);
    float* y = reinterpret_cast<float*>(ep.y);
    const uint32_t pix0 = (uint32_t)((b * g.H + y0) * g.W + x0);
    f32x4 rres[RW][RITEMS];
    if (res) {
#pragma unroll
        for (int ii = 0; ii < RW; ++ii) {
            const int i = WK > 1 ? wk * RW + ii : ii;
#pragma unroll
            for (int q = 0; q < RITEMS; ++q) {
                const int it = q * 64 + lane;
                const int px = it / GPR, ng = it - px * GPR;
                const bool ok = (16 * GPR % 64 == 0 || it < 16 * GPR) && y0 + i < g.H && x0 + px < g.W && n0 + ng * OE < ep.Cout;
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                if (ok) z = *reinterpret_cast<const f32x4*>(res + (pix0 + (uint32_t)(i * g.W + px)) * (uint32_t)ep.ldres + (uint32_t)(n0 + ng * OE));
                rres[ii][q] = z;
            }
        }
    }
#pragma unroll
    for (int ii = 0; ii < RW; ++ii) {
        const int i = WK > 1 ? wk * RW + ii : ii;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[((lane >> 4) * 4 + r) * LDS_ + j * 16 + xl] = acc[ii][j][r] * sc[j] + sh[j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < RITEMS; ++q) {
            const int it = q * 64 + lane;
            const int px = it / GPR, ng = it - px * GPR;
            const int gn = n0 + ng * OE;
            const bool ok = (16 * GPR % 64 == 0 || it < 16 * GPR) && y0 + i < g.H && x0 + px < g.W && gn < ep.Cout;
            if (ok) {
                float v[OE];
                const f32x4 t = *reinterpret_cast<const f32x4*>(stage + px * LDS_ + ng * OE);
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
                if (res) { v[0] += rres[ii][q][0]; v[1] += rres[ii][q][1]; v[2] += rres[ii][q][2]; v[3] += rres[ii][q][3]; }
                apply_act_chunk<OE>(v, ep.act, gn);
                *reinterpret_cast<f32x4*>(y + (pix0 + (uint32_t)(i * g.W + px)) * (uint32_t)ep.ldy + (uint32_t)gn) = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int g_opt_halo_cws = 1;      // option "halo_cws": 0 = conv3x3_wave_kernel<f32s_t> only, 1 = this kernel where an instantiation exists

template <int CT, int WN, int FN, int WK, int ROWS = 8>
static int launch_cws(const mfx_conv_desc* d, hipStream_t st) {
    using SM = CwsSmem<WN, FN, WK, ROWS>;
    constexpr int BN = WN * FN * 16;
    CwsGeom g;
    g.B = d->B; g.H = d->H; g.W = d->W;
    g.tiles_x = (d->W + 15) / 16; g.tiles_y = (d->H + ROWS - 1) / ROWS; g.tiles_n = d->Cout_pad / BN;
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = d->res; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = d->ldres;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = g.tiles_n;
    auto k = conv3x3_cws_kernel<CT, WN, FN, WK, ROWS>;
    constexpr int smem = SM::total;
    static bool attr_done = false;
    if (!attr_done && smem > 64 * 1024) {
        MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    const int tiles = g.tiles_n * g.tiles_x * g.tiles_y * d->B;
    hipLaunchKernelGGL(k, dim3(tiles), dim3(WN * WK * 64), smem, st, reinterpret_cast<const float*>(d->x), reinterpret_cast<const u32x4*>(d->w_frag_pair), g, ep);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// returns MFX_OK (0) if this kernel ran, 1 if there is no instantiation for the shape / variant (caller runs conv3x3_wave_kernel), < 0 on error.
// `v` is conv_halo.hip's variant number (6: 2 waves x 32 channels, 7: 4 x 32, 11: 4 x 32 with a 2-way K split, 8: one 32-channel slice, 4-way K split)
int try_conv_cws(const mfx_conv_desc* d, int v, hipStream_t st) {
    if (!g_opt_halo_cws || d->dtype != MFX_F16X2 || d->out_dtype != MFX_F32 || !d->w_frag_pair || d->stride != 1 || d->stats) return 1;
    if (d->K_pad != 9 * d->Ck || d->Ck % 64 != 0 || d->Cout % 4 != 0 || d->Ho != d->H || d->Wo != d->W) return 1;
    if ((long long)d->M * (d->ldy > d->ldres ? d->ldy : d->ldres) >= (1ll << 31) || (long long)d->H * d->W * d->Ck >= (1ll << 31)) return 1;      // 32-bit element offsets
    const int C = d->Ck;
    if (v == 8 && d->Cout_pad == 32 && !d->res) {
        if (C == 64) return launch_cws<64, 1, 2, 4>(d, st);
        if (C == 128) return launch_cws<128, 1, 2, 4>(d, st);
        if (C == 256) return launch_cws<256, 1, 2, 4>(d, st);
        if (C == 512) return launch_cws<512, 1, 2, 4>(d, st);
        return 1;
    }
    if (d->act == MFX_ACT_DCN_OFFMASK) return 1;
    if (v == 6 && d->Cout_pad % 64 == 0 && C == 64) return launch_cws<64, 2, 2, 1>(d, st);
    if (v == 7 && d->Cout_pad % 128 == 0) {
        if (C == 64) return launch_cws<64, 4, 2, 1>(d, st);
        if (C == 128) return launch_cws<128, 4, 2, 1>(d, st);
    }
    if (v == 11 && d->Cout_pad % 128 == 0) {
        if (C == 128) return launch_cws<128, 4, 2, 2>(d, st);
        if (C == 256) return launch_cws<256, 4, 2, 2>(d, st);
        if (C == 512) return launch_cws<512, 4, 2, 2>(d, st);
    }
    return 1;
}

}  // namespace mfx

MFX_RANGE_FLAG_ACCESSOR(conv_cws)      // split-precision range sentinel of this translation unit (common.h)
