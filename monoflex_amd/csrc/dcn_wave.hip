// Fused modulated deformable convolution (DCNv2 forward + bias/BN + activation), second-generation kernel.
//
//   y[m][n] = act( scale[n] * sum_{tap,c} W[n][tap,c] * mask[m,tap] * bilinear(x[b,:,:,c] @ p(m,tap)) + shift[n] )
//   (reference: model/backbone/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:125-195 + dcn_v2_cuda.cu:139-163, never
//    materialising the `columns` buffer)
//
// Workgroup = WN waves, tile = 64 output pixels x (WN*FN*16) output channels:
//   * gather: four lanes per pixel, one 16-byte chunk each, so a wave instruction reads 16 pixels x 64 contiguous
//     bytes (NHWC: a bilinear corner is a channel vector).  A lane-per-pixel mapping (each lane streaming its own
//     64 bytes in four instructions) touches 64 cache lines per instruction and measured SLOWER: this path is
//     texture/L1-bandwidth bound (every input element is fetched ~36x: 9 taps x 4 corners), so line efficiency
//     is what matters.  Per K step (64 bytes of K = one tap, 32 bf16 / 16 f32 channels) the four corners are
//     blended in fp32 with weights that already contain the modulation mask and written to a double-buffered
//     64 x 80 B LDS A tile; sampling geometry is recomputed only when the tap changes;
//   * corner loads of step s+1 are issued before the MFMAs of step s and blended after them;
//   * weights are fragment-major (one contiguous KiB per MFMA fragment) and stream L2 -> registers through a
//     3-deep ring; each weight fragment feeds 4 MFMAs, each A fragment FN;
//   * WN = 1 (Cout <= 128) has no workgroup barrier at all; WN = 2/4 share the gathered tile through LDS with one
//     barrier per step, so the (expensive) gather is never duplicated across output-channel tiles;
//   * epilogue through a wave-private LDS buffer: 16-byte stores along channels.
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "igemm.h"
#include <type_traits>

namespace mfx {

struct DcnWGeom { int H, W, C, lgC, Ho, Wo, kh, kw, inv_kw, stride, pad, dil, M, tiles_n, fsteps, spt; };

constexpr int kDcnFM = 4;                                    // 64 pixels per workgroup
constexpr int kDcnRow = 80;                                  // A-tile row: 64 B of K + 16 B pad (conflict-free b128 reads/writes)

template <int WN, int FN> struct DcnWSmem {
    static constexpr int a_bytes = 2 * 64 * kDcnRow;                           // double-buffered A tile
    static constexpr int stage_ld = FN * 16 + 4;
    static constexpr int stage_bytes = 16 * stage_ld * 4;                       // per wave
    static constexpr int bytes = a_bytes + WN * stage_bytes;
};

template <typename T, int WN, int FN>
__global__ __launch_bounds__(WN * 64, 2) void dcn_wave_kernel(const T* __restrict__ x, const float* __restrict__ om,
                                                             const u32x4* __restrict__ wfm, DcnWGeom g, EpiArgs ep) {
    constexpr int FM = kDcnFM, ELEMS = ElemTraits<T>::ELEMS;
    using SM = DcnWSmem<WN, FN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* stage = reinterpret_cast<float*>(smem + SM::a_bytes + wn * SM::stage_bytes);
    const int xl = lane & 15, kq = lane >> 4;

    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    const int m0 = tm * 64, n0 = tn * (WN * FN * 16) + wn * (FN * 16);

    // ---- gather assignment: 4 lanes per pixel (one 16-byte chunk each -> a wave instruction reads 16 pixels x 64
    // contiguous bytes: full half-lines for the texture path); this wave covers pixel groups wn*RPW .. +RPW of 16
    constexpr int RPW = 4 / WN;
    const int gc = lane & 3, gp = lane >> 2;
    int pix0[RPW], oh_[RPW], ow_[RPW]; bool mok[RPW]; const float* omr[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int m = m0 + (wn * RPW + r) * 16 + gp;
        mok[r] = m < g.M;
        const int pm = mok[r] ? m : 0;
        const int hw = g.Ho * g.Wo, b = pm / hw, rem = pm - b * hw;
        oh_[r] = rem / g.Wo; ow_[r] = rem - oh_[r] * g.Wo;
        pix0[r] = b * g.H * g.W;
        omr[r] = om + (size_t)pm * 32;
    }
    uint32_t cofb[RPW][4]; float cw[RPW][4];      // byte offsets of the corner rows' chunk gc (tensor < 4 GB: checked at launch)
    const uint32_t rowb = (uint32_t)g.C * sizeof(T), chb = (uint32_t)gc * 16;
    auto geom = [&](int tap) {
        const int th = (tap * g.inv_kw) >> 16, tw = tap - th * g.kw;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float dh = omr[r][2 * tap], dw = omr[r][2 * tap + 1], mk = omr[r][18 + tap];
            const float h = (float)(oh_[r] * g.stride - g.pad + th * g.dil) + dh;
            const float w = (float)(ow_[r] * g.stride - g.pad + tw * g.dil) + dw;
            const bool inside = mok[r] && h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
            const float hf = floorf(h), wf = floorf(w);
            const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
            const float lh = h - hf, lw = w - wf, hh = 1.f - lh, hw_ = 1.f - lw;
            const bool t0 = inside && h0 >= 0, t1 = inside && h1 <= g.H - 1, l0 = w0 >= 0, l1 = w1 <= g.W - 1;
            const int ch0 = min(max(h0, 0), g.H - 1), ch1 = min(max(h1, 0), g.H - 1);
            const int cw0 = min(max(w0, 0), g.W - 1), cw1 = min(max(w1, 0), g.W - 1);
            cofb[r][0] = (uint32_t)(pix0[r] + ch0 * g.W + cw0) * rowb + chb; cw[r][0] = (t0 && l0) ? hh * hw_ * mk : 0.f;
            cofb[r][1] = (uint32_t)(pix0[r] + ch0 * g.W + cw1) * rowb + chb; cw[r][1] = (t0 && l1) ? hh * lw * mk : 0.f;
            cofb[r][2] = (uint32_t)(pix0[r] + ch1 * g.W + cw0) * rowb + chb; cw[r][2] = (t1 && l0) ? lh * hw_ * mk : 0.f;
            cofb[r][3] = (uint32_t)(pix0[r] + ch1 * g.W + cw1) * rowb + chb; cw[r][3] = (t1 && l1) ? lh * lw * mk : 0.f;
        }
    };
    u32x4 gr[RPW][4];
    auto gload = [&](int s) {                                 // corners of step s (tap = s / spt, channel block s % spt)
        const char* xk = reinterpret_cast<const char*>(x) + (s % g.spt) * 64;      // wave-uniform base + 32-bit lane offset
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) gr[r][q] = *reinterpret_cast<const u32x4*>(xk + cofb[r][q]);
    };
    auto gstore = [&](int buf) {                              // blend (fp32) and write this lane's chunk of its A-tile rows
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            float v[4][ELEMS], o[ELEMS];
#pragma unroll
            for (int q = 0; q < 4; ++q) ElemTraits<T>::unpack(gr[r][q], v[q]);
            if constexpr (ELEMS == 4) {                        // fp32 / split-precision chunks: scalar FMAs (the packed form measured -0.8 % on the fp16x2 step)
#pragma unroll
                for (int e = 0; e < ELEMS; ++e) o[e] = cw[r][0] * v[0][e] + cw[r][1] * v[1][e] + cw[r][2] * v[2][e] + cw[r][3] * v[3][e];
            } else {
#pragma unroll
                for (int e = 0; e < ELEMS; e += 2) {               // two channels per instruction (v_pk_mul_f32 / v_pk_fma_f32)
                f32x2 t = (f32x2){v[0][e], v[0][e + 1]} * cw[r][0];
                t = __builtin_elementwise_fma((f32x2){v[1][e], v[1][e + 1]}, (f32x2){cw[r][1], cw[r][1]}, t);
                t = __builtin_elementwise_fma((f32x2){v[2][e], v[2][e + 1]}, (f32x2){cw[r][2], cw[r][2]}, t);
                t = __builtin_elementwise_fma((f32x2){v[3][e], v[3][e + 1]}, (f32x2){cw[r][3], cw[r][3]}, t);
                    o[e] = t[0]; o[e + 1] = t[1];
                }
            }
            *reinterpret_cast<u32x4*>(As + buf * (64 * kDcnRow) + ((wn * RPW + r) * 16 + gp) * kDcnRow + gc * 16) = lds_operand<T>(ElemTraits<T>::pack(o));
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const u32x4* wfl = wfm + (size_t)(n0 >> 4) * g.fsteps * 64 + lane;
    auto wfetch = [&](int s, u32x4 (&wf)[FN]) {
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = wfl[((size_t)j * g.fsteps + s) * 64];
    };
    auto compute = [&](int s, const u32x4 (&wf)[FN]) {
        const char* ap = As + (s & 1) * (64 * kDcnRow) + xl * kDcnRow + kq * 16;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const u32x4 af = *reinterpret_cast<const u32x4*>(ap + i * 16 * kDcnRow);
#pragma unroll
            for (int j = 0; j < FN; ++j) mma_chunk<T>(af, wf[j], acc[i][j]);
        }
    };
    // one pipeline step: prefetch weights of s+2, gather corners of s+1, MFMAs of s, blend/store s+1
    auto step = [&](int s, const u32x4 (&wcur)[FN], u32x4 (&wnext2)[FN], int ns) {
        if (s + 2 < ns) wfetch(s + 2, wnext2);
        const bool more = s + 1 < ns;
        if (more) {
            if ((s + 1) % g.spt == 0) geom((s + 1) / g.spt);
            gload(s + 1);
        }
        compute(s, wcur);
        if (more) gstore((s + 1) & 1);
        if (WN > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    };

    const int ns = g.fsteps;                                  // kh*kw*C / (4*ELEMS)
    u32x4 wb[3][FN];
    geom(0);
    gload(0);
    wfetch(0, wb[0]);
    if (ns > 1) wfetch(1, wb[1]);
    gstore(0);
    if (WN > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    int s = 0;
    for (; s + 3 <= ns; s += 3) {
        step(s, wb[0], wb[2], ns);
        step(s + 1, wb[1], wb[0], ns);
        step(s + 2, wb[2], wb[1], ns);
    }
    if (s < ns) { step(s, wb[0], wb[2], ns); if (s + 1 < ns) step(s + 1, wb[1], wb[0], ns); }

    // ---- epilogue, wave-private staging: fragment i = pixels m0+16i .. +16
    constexpr int LDS_ = SM::stage_ld;
    constexpr int OE = ElemTraits<T>::ELEMS;
    constexpr int GPR = FN * 16 / OE;
    T* y = reinterpret_cast<T*>(ep.y);
    float sc[FN], sh[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        sc[j] = ep.scale ? ep.scale[n0 + j * 16 + xl] : 1.f;
        sh[j] = ep.shift ? ep.shift[n0 + j * 16 + xl] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(kq * 4 + r) * LDS_ + j * 16 + xl] = acc[i][j][r] * sc[j] + sh[j];
        __builtin_amdgcn_wave_barrier();
        for (int it = lane; it < 16 * GPR; it += 64) {
            const int px = it / GPR, ng = it - px * GPR;
            const int gm = m0 + i * 16 + px, gn = n0 + ng * OE;
            if (gm < g.M && gn < ep.Cout) {
                float v[OE];
#pragma unroll
                for (int e = 0; e < OE; e += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(stage + px * LDS_ + ng * OE + e);
                    v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
                }
apply_act_chunk<OE>(v, ep.act, gn);
                *reinterpret_cast<u32x4*>(y + (size_t)gm * ep.ldy + gn) = ElemTraits<T>::pack(v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int g_opt_dcn_wave = 1;      // 0 = first-generation kernel only, 1 = automatic, 2.. = force variant

template <typename T, int WN, int FN>
static int launch_dcn_wave(const mfx_dcn_desc* d, hipStream_t st) {
    constexpr int ELEMS = ElemTraits<T>::ELEMS;
    DcnWGeom g;
    g.H = d->H; g.W = d->W; g.C = d->C; g.lgC = 0; while ((1 << g.lgC) < d->C) ++g.lgC;
    g.Ho = d->Ho; g.Wo = d->Wo; g.kh = d->kh; g.kw = d->kw; g.inv_kw = (65536 + d->kw - 1) / d->kw;
    g.stride = d->stride; g.pad = d->pad; g.dil = d->dil; g.M = d->B * d->Ho * d->Wo;
    g.tiles_n = d->Cout_pad / (WN * FN * 16); g.fsteps = d->K_pad / (4 * ELEMS); g.spt = d->C / (4 * ELEMS);
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = nullptr; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = 0;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = g.tiles_n;
    const int tiles = ((g.M + 63) / 64) * g.tiles_n;
    constexpr int smem = DcnWSmem<WN, FN>::bytes;
    hipLaunchKernelGGL((dcn_wave_kernel<T, WN, FN>), dim3(tiles), dim3(WN * 64), smem, st, reinterpret_cast<const T*>(d->x), d->offmask,
                       reinterpret_cast<const u32x4*>(d->w_frag), g, ep);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// variants: 1: 1 wave x FN4 (BN 64)   2: 1 x FN8 (BN 128)   3: 2 x FN4 (BN 128)   4: 2 x FN8 (BN 256)   5: 4 x FN4 (BN 256)
//           6: 2 x FN2 (BN 64)        7: 4 x FN2 (BN 128)
template <typename T> static int dcn_wave_variant(int v, const mfx_dcn_desc* d, hipStream_t st) {
    switch (v) {
        case 1: return launch_dcn_wave<T, 1, 4>(d, st);
        case 2: return launch_dcn_wave<T, 1, 8>(d, st);
        case 3: return launch_dcn_wave<T, 2, 4>(d, st);
        case 4: return launch_dcn_wave<T, 2, 8>(d, st);
        case 5: return launch_dcn_wave<T, 4, 4>(d, st);
        case 6: return launch_dcn_wave<T, 2, 2>(d, st);
        case 7: return launch_dcn_wave<T, 4, 2>(d, st);
        default: return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn wave: no such variant");
    }
}

// returns 1 if handled, 0 to fall back to the first-generation kernel, < 0 on error
int try_dcn_wave(const mfx_dcn_desc* d, hipStream_t st) {
    if (g_opt_dcn_wave == 0 || !d->w_frag) return 0;
    const int elems = (d->dtype == MFX_F32 || d->dtype == MFX_F16X2) ? 4 : 8;
    if (d->C < 4 * elems || d->K_pad != d->kh * d->kw * d->C) return 0;
    const int N = d->Cout_pad;
    if (N % 64 != 0) return 0;
    const int bn[8] = {0, 64, 128, 128, 256, 256, 64, 128};
    // measured (tools/conv_probe.py, MI355X): every variant of both kernel generations lands within ~10 % -- the gather is
    // L1/texture-bandwidth bound (each input element is fetched ~36x; 64->64 @ B=8 moves 1.13 GB through L1 in 90 us =
    // 77 % of 64 B/clk/CU).  The 1-wave x 64-channel variant spills, so Cout = 64 stays on the first-generation kernel.
    // Small maps (M/64 workgroups < 2 per CU) keep the first generation too: its 64x64 tiling has 2-4x more waves.
    if (g_opt_dcn_wave < 2 && (N % 128 != 0 || (d->B * d->Ho * d->Wo / 64) * (N / 128) < 512)) return 0;
    int v = N % 256 == 0 ? 5 : N % 128 == 0 ? 3 : 6;
    if (g_opt_dcn_wave >= 2 && g_opt_dcn_wave - 1 <= 7 && N % bn[g_opt_dcn_wave - 1] == 0) v = g_opt_dcn_wave - 1;
    const int rc = d->dtype == MFX_F16X2 ? dcn_wave_variant<f32s_t>(v, d, st) : d->dtype == MFX_F32 ? dcn_wave_variant<float>(v, d, st)
                 : (d->dtype == MFX_F16 ? dcn_wave_variant<half_t>(v, d, st) : dcn_wave_variant<bf16_t>(v, d, st));
    return rc == MFX_OK ? 1 : rc;
}

}  // namespace mfx

MFX_RANGE_FLAG_ACCESSOR(dcn_wave)      // split-precision range sentinel of this translation unit (common.h)
