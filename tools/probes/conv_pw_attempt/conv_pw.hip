// Streaming pointwise convolution: 1x1 conv (+ BN scale / shift + activation) over one source or a virtual channel concat (DLA "project" convs and
// Root nodes, dla_dcn.py:195-203, 268-276), for the LARGE-map / SHORT-K layers of the trunk (levels 2 - 4: M = 15 k .. 246 k pixels, K = 32 .. 448).
//
// Why a second form next to the LDS-tiled kernel (conv_igemm / cat_igemm): these layers are HBM-bound (1 - 5 flop / byte) and their whole K is one to seven
// k-iterations of the tiled loop, so a tiled launch is all prologue and epilogue -- load, LDS, barrier, a handful of MFMAs, fp32 tile through LDS, barrier,
// store -- and ran at 2.1 - 3.2 TB/s (r05 timeline: ten launches, 158 us, against 57 us of HBM time).  Here
//   * the GEMM is run transposed (output channels on the MFMA M axis, pixels on N): lane l holds D[channel 4*(l>>4)+r][pixel l&15], four consecutive output
//     channels of one pixel -- stored straight from the accumulators as 8 bytes, no LDS transposition, no barrier;
//   * both operands are K-contiguous in memory, so an MFMA fragment is one 16-byte global load per lane: the pixel fragment (8 channels l>>4 of pixel l&15) comes
//     straight from the activation tensor (a wave reads 16 pixels x 64 contiguous bytes per k-step; the texture path sees 16-byte pieces, which an HBM-bound layer
//     can afford), the weight fragments are loaded ONCE per wave and stay in registers (<= 112 VGPRs: wide outputs are split over the workgroup's waves);
//   * a wave walks BPW consecutive 16-pixel blocks with the next block's loads in flight under the current block's MFMAs and stores.
// No LDS, no workgroup barrier anywhere.
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "igemm.h"

namespace mfx {

int g_opt_conv_pw = 1;         // 0: never (LDS-tiled kernels), 1: where an instantiation exists and M >= 8192

struct PwSegs { const void* src[MFX_MAX_SEG]; int stride[MFX_MAX_SEG]; int off[MFX_MAX_SEG]; int lgC; int M; int K_pad; int bpw; };

// KS = K / 32 (k-steps), FN = 16-channel output fragments per wave, NSPLIT = waves of a workgroup that share a pixel block and split the output channels
template <typename T, int KS, int FN, int NSPLIT>
__global__ __launch_bounds__(256) void conv_pw_kernel(PwSegs s, const T* __restrict__ w, EpiArgs ep) {
    const int tid = threadIdx.x, lane = tid & 63, xl = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ng = wave % NSPLIT, pg = blockIdx.x * (4 / NSPLIT) + wave / NSPLIT;
    const int n0 = ng * (FN * 16);
    const int nblk = (s.M + 15) >> 4;
    const int b0 = pg * s.bpw, b1 = min(b0 + s.bpw, nblk);
    if (b0 >= nblk) return;

    // weight fragments (A operand): lane = (output channel n0 + 16*ob + xl, k-group kq)
    u32x4 wf[FN][KS];
#pragma unroll
    for (int ob = 0; ob < FN; ++ob)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wf[ob][ks] = *reinterpret_cast<const u32x4*>(w + (size_t)(n0 + ob * 16 + xl) * s.K_pad + ks * 32 + kq * 8);

    // pixel fragments (B operand) of one 16-pixel block: lane = (pixel xl, k-group kq); rows past M repeat the last pixel (never stored)
    auto xload = [&](u32x4 (&x)[KS], int blk) {
        const int m = min(blk * 16 + xl, s.M - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int seg = (ks * 32) >> s.lgC, ci = (ks * 32) & ((1 << s.lgC) - 1);          // wave-uniform
            const char* base = reinterpret_cast<const char*>(reinterpret_cast<const T*>(s.src[seg]) + s.off[seg] + ci);   // wave-uniform base ...
            x[ks] = *reinterpret_cast<const u32x4*>(base + ((uint32_t)m * (uint32_t)s.stride[seg] * 2u + (uint32_t)kq * 16u));  // ... + 32-bit byte offset
        }
    };
    T* y = reinterpret_cast<T*>(ep.y);
    auto block = [&](const u32x4 (&x)[KS], int blk) {
        f32x4 acc[FN];
#pragma unroll
        for (int ob = 0; ob < FN; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int ob = 0; ob < FN; ++ob) mma_chunk<T>(wf[ob][ks], x[ks], acc[ob]);
        const int m = blk * 16 + xl;
        if (m >= s.M) return;
        T* yrow = y + (size_t)m * ep.ldy;
#pragma unroll
        for (int ob = 0; ob < FN; ++ob) {
            const int n = n0 + ob * 16 + kq * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ob][r];
            if (ep.scale) { const f32x4 sc = *reinterpret_cast<const f32x4*>(ep.scale + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= sc[r];
            }
            if (ep.shift) { const f32x4 sh = *reinterpret_cast<const f32x4*>(ep.shift + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += sh[r];
            }
            apply_act_chunk<4>(v, ep.act, n);
            uint2 o; o.x = ElemTraits<T>::pack2(v[0], v[1]); o.y = ElemTraits<T>::pack2(v[2], v[3]);
            *reinterpret_cast<uint2*>(yrow + n) = o;
        }
    };

    if constexpr (KS <= 8) {                                  // next block's loads in flight under this block's MFMAs and stores
        u32x4 xa[KS], xb[KS];
        xload(xa, b0);
        for (int b = b0; b < b1; b += 2) {
            if (b + 1 < b1) xload(xb, b + 1);
            block(xa, b);
            if (b + 1 < b1) {
                if (b + 2 < b1) xload(xa, b + 2);
                block(xb, b + 1);
            }
        }
    } else {                                                  // long K: one buffer (a second one would halve the waves per SIMD); the other waves cover the loads
        u32x4 xa[KS];
        for (int b = b0; b < b1; ++b) { xload(xa, b); block(xa, b); }
    }
}

template <typename T, int KS, int FN, int NSPLIT>
static int launch_pw(const PwSegs& s0, const void* w, const EpiArgs& ep, hipStream_t st) {
    PwSegs s = s0;
    const int nblk = (s.M + 15) / 16;
    // pixel blocks per wave: enough that the one-time weight load amortises, few enough that the launch has >= ~4 workgroups per CU
    const int groups_per_wg = 4 / NSPLIT;
    int bpw = 16;                                            // ~4096 waves (16 per CU) when the map is large enough
    while (bpw > 2 && (long)((nblk + bpw - 1) / bpw) * NSPLIT < 4096) bpw >>= 1;
    s.bpw = bpw;
    const int ngroups = (nblk + bpw - 1) / bpw;
    const int wgs = (ngroups + groups_per_wg - 1) / groups_per_wg;
    hipLaunchKernelGGL((conv_pw_kernel<T, KS, FN, NSPLIT>), dim3(wgs), dim3(256), 0, st, s, reinterpret_cast<const T*>(w), ep);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// (K, N) -> instantiation; 1: none
template <typename T> static int pw_by_shape(int K, int N, const PwSegs& s, const void* w, const EpiArgs& ep, hipStream_t st) {
    if (K == 32 && N == 64) return launch_pw<T, 1, 4, 1>(s, w, ep, st);
    if (K == 64 && N == 64) return launch_pw<T, 2, 4, 1>(s, w, ep, st);
    if (K == 128 && N == 64) return launch_pw<T, 4, 4, 1>(s, w, ep, st);
    if (K == 64 && N == 128) return launch_pw<T, 2, 4, 2>(s, w, ep, st);
    if (K == 128 && N == 128) return launch_pw<T, 4, 4, 2>(s, w, ep, st);
    if (K == 256 && N == 128) return launch_pw<T, 8, 2, 4>(s, w, ep, st);
    if (K == 448 && N == 128) return launch_pw<T, 14, 2, 4>(s, w, ep, st);
    if (K == 128 && N == 256) return launch_pw<T, 4, 4, 4>(s, w, ep, st);
    return 1;
}

static bool pw_common_ok(int dtype, int M, int Cout, int Cout_pad, const void* res) {
    if (g_opt_conv_pw == 0 || res) return false;
    if (dtype != MFX_BF16 && dtype != MFX_F16) return false;
    if (Cout != Cout_pad || M < 8192) return false;
    return true;
}

// 0: not taken, 1: ran, < 0: error
int try_conv_pw_cat(const mfx_cat_desc* d, hipStream_t st) {
    if (!pw_common_ok(d->dtype, d->M, d->Cout, d->Cout_pad, d->res)) return 0;
    PwSegs s;
    for (int i = 0; i < MFX_MAX_SEG; ++i) {
        s.src[i] = i < d->nseg ? d->src[i] : d->src[0]; s.stride[i] = i < d->nseg ? d->stride[i] : 0; s.off[i] = i < d->nseg ? d->off[i] : 0;
        if (i < d->nseg && ((size_t)d->M * d->stride[i] * 2 >= ((size_t)1 << 32))) return 0;
    }
    s.lgC = 0; while ((1 << s.lgC) < d->Cseg) ++s.lgC;
    s.M = d->M; s.K_pad = d->K_pad; s.bpw = 0;
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = nullptr; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = 0;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = 1;
    const int rc = d->dtype == MFX_F16 ? pw_by_shape<half_t>(d->K_pad, d->Cout, s, d->w, ep, st) : pw_by_shape<bf16_t>(d->K_pad, d->Cout, s, d->w, ep, st);
    return rc == MFX_OK ? 1 : (rc == 1 ? 0 : rc);
}

int try_conv_pw_1x1(const mfx_conv_desc* d, hipStream_t st) {
    if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad_h != 0 || d->pad_w != 0 || d->rowmap || d->stats) return 0;
    if (d->out_dtype != d->dtype || d->K_pad != d->Ck) return 0;
    if (!pw_common_ok(d->dtype, d->M, d->Cout, d->Cout_pad, d->res)) return 0;
    if ((size_t)d->M * d->x_pixstride * 2 >= ((size_t)1 << 32)) return 0;
    PwSegs s;
    for (int i = 0; i < MFX_MAX_SEG; ++i) { s.src[i] = d->x; s.stride[i] = d->x_pixstride; s.off[i] = 0; }
    s.lgC = 30;                                              // one segment: every k-step maps to segment 0 at channel offset 32 * ks
    s.M = d->M; s.K_pad = d->K_pad; s.bpw = 0;
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = nullptr; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = 0;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = 1;
    const int rc = d->dtype == MFX_F16 ? pw_by_shape<half_t>(d->K_pad, d->Cout, s, d->w, ep, st) : pw_by_shape<bf16_t>(d->K_pad, d->Cout, s, d->w, ep, st);
    return rc == MFX_OK ? 1 : (rc == 1 ? 0 : rc);
}

}  // namespace mfx
