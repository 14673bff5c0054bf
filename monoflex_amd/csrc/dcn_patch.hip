// Fused modulated deformable convolution, third generation: bilinear corners are gathered from an LDS-resident input
// patch instead of global memory.
//
//   y[m][n] = act( scale[n] * sum_{tap,c} W[n][tap,c] * mask[m,tap] * bilinear(x[b,:,:,c] @ p(m,tap)) + shift[n] )
//   (reference: model/backbone/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:125-195 + dcn_v2_cuda.cu:139-163)
//
// Why: the first two generations fetch every input element ~36x (9 taps x 4 corners) through the texture/L1 path and sit
// on its 64 B/clk/CU limit (MFMA busy 11 %).  Learned DCN offsets are a few pixels, so almost every corner of a 16-wide
// output tile lies inside the tile grown by 4 pixels; that patch is loaded ONCE per 64-channel slice with coalesced
// 16-byte loads, and the 36 corner reads per output element become ds_read_b128 (128 B/clk/CU, no address coalescing
// stage).  Samples that leave the patch take the old global path, lane by lane, so any offset stays exact.
//
// Workgroup = 4 waves, tile = (4*FM) rows x 16 columns of output pixels x FN*16 output channels; wave w owns tile rows
// w*FM .. +FM (one MFMA M-block = 16 horizontally adjacent pixels).  3x3 / stride 1 / pad 1 / dilation 1, bf16.
//   * patch: (4*FM+8) x 24 pixels x 64 channels, zero-filled outside the image (so in-patch corners need no validity
//     tests: the zero rows/columns ARE the reference's out-of-image corners), 16-byte columns XOR-swizzled by pixel;
//   * the patch is held in LDS as fp16 (bf16 -> fp16 is exact) and the conv runs on the fp16 MFMA with an fp16 copy of the
//     weights: PMC on the first version showed the kernel VALU-bound (107 VALU instructions per MFMA fragment: bf16
//     unpack + fp32 blend + repack, MFMA pipe 7 % busy).  In fp16 the four-corner blend is 1 v_pk_mul + 3 v_pk_fma per
//     channel pair (weights carry the modulation mask, 11-bit mantissa) and its result IS the MFMA A fragment
//     (lane = pixel (lane&15), k-group (lane>>4)): no unpack, no repack, no A tile in LDS, no barrier inside a slice;
//   * sampling geometry (floor, weights, patch address) is computed once per pixel by an owner lane and handed to the
//     pixel's four k-group lanes with ds_bpermute instead of being recomputed four times;
//   * weights: fragment-major, L2 -> registers through a 3-deep ring, every fragment feeds FM MFMAs;
//   * epilogue staged through the (now free) patch memory: 16-byte stores along channels.
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "igemm.h"

namespace mfx {

struct DcnPGeom { int B, H, W, C, tiles_x, tiles_y, tiles_n, fsteps, cpt; };   // cpt = C/32: k-steps per tap

// patch = output tile grown by R+1 pixels on every side (1 for the 3x3 taps, R for the offsets), CS channels per slice
// PD = padded layout: pixels CS*2 + 16 bytes apart and no XOR swizzle, so the four corners of a sample are ONE base address plus
// compile-time offsets (0, PB, PW*PB, PW*PB + PB: the ds_read offset field) and the owner lane hands out that base instead of
// coordinates every consumer lane turns into four swizzled addresses (20 VALU per fragment and tap: PMC counted 18 VALU
// instructions per MFMA in the swizzled form, a good part of them this address arithmetic).  64->64 @ 8x96x320: 80.8 -> 75.0 us, same
// bits (tools/probes/dcn_patch_probe.py); the automatic choice for those layers, option dcn_patch = 8 forces it, 5 = the swizzled form.
template <int FM, int R, int CS, bool PD = false> struct DcnPSmem {
    static constexpr int PW = 16 + 2 * (R + 1);
    static constexpr int rows = 4 * FM + 2 * (R + 1);
    static constexpr int pix = rows * PW;
    static constexpr int PB = CS * 2 + (PD ? 16 : 0);        // bytes per patch pixel (fp16)
    static constexpr int NC = CS / 8;                        // 16-byte columns per pixel
    static constexpr int bytes = pix * PB;
};

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

// bf16 pair (one dword) -> fp16 pair; exact for every bf16 inside fp16's range (bf16 carries 7 mantissa bits)
__device__ __forceinline__ uint32_t bf2_to_h2(uint32_t d) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)));
}
__device__ __forceinline__ u32x4 bf8_to_h8(const u32x4& v) { return u32x4{bf2_to_h2(v.x), bf2_to_h2(v.y), bf2_to_h2(v.z), bf2_to_h2(v.w)}; }
// 16-byte chunk of the input map as fp16: bf16 maps are converted (exact), fp16 maps are copied
template <typename TX> __device__ __forceinline__ u32x4 to_h8(const u32x4& v);
template <> __device__ __forceinline__ u32x4 to_h8<bf16_t>(const u32x4& v) { return bf8_to_h8(v); }
template <> __device__ __forceinline__ u32x4 to_h8<half_t>(const u32x4& v) { return v; }

// OF = true (r04): the module's 27-channel offset/mask conv (3x3 / stride 1 / pad 1 on the SAME input, reference dcn_v2.py:118-122) runs
// inside this kernel.  Phase 0 stages the tile's 18 x 18 x 64-channel neighbourhood in the (still unused) patch memory, every wave
// multiplies its own 64 pixels x 32 channels x K = 576 on the matrix cores (144 MFMAs, fp16 weights fragment-major from L2), adds the
// bias, applies the sigmoid to the nine mask channels and transposes the accumulators through a 2.3 KB wave-private LDS buffer so that
// each pixel's owner lane holds its 27 values in registers -- exactly what the separate conv's output row gave it.  That removes
// the 27.5 us offset conv launch, its 31 MB output and this kernel's row loads; the rows are written out only when a caller needs
// them (training: the backward pass).
struct DcnOffArgs { const u32x4* wfm; const float* shift; float* om_out; };

template <int FN, int FM, int R = 3, int CS = 64, bool PD = false, typename TX = bf16_t, bool OF = false>
__global__ __launch_bounds__(256, 2) void dcn_patch_kernel(const TX* __restrict__ x, const float* __restrict__ om,
                                                          const u32x4* __restrict__ wfm, DcnPGeom g, EpiArgs ep, DcnOffArgs oa) {
    using SM = DcnPSmem<FM, R, CS, PD>;
    constexpr int kPW = SM::PW, PB = SM::PB, NC = SM::NC, KS = CS / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xl = lane & 15, kq = lane >> 4;

    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % g.tiles_n; tile /= g.tiles_n;
    const int tx = tile % g.tiles_x; tile /= g.tiles_x;
    const int ty = tile % g.tiles_y, b = tile / g.tiles_y;
    const int ty0 = ty * (4 * FM), tx0 = tx * 16, n0 = tn * (FN * 16);
    const int py0 = ty0 - (R + 1), px0 = tx0 - (R + 1);        // image coordinates of patch pixel (0,0)
    const TX* xb = x + (size_t)b * g.H * g.W * g.C;

    // ---- sampling geometry is computed ONCE per pixel: lane l owns pixel (tile row wv*FM + (l>>4) % FM, column l&15) and
    // hands the result to the four k-group lanes of that pixel through ds_bpermute
    const int gi = (lane >> 4) % FM;
    const int yo = ty0 + wv * FM + gi, xo = tx0 + xl;
    const bool own_ok = yo < g.H && xo < g.W;
    float omv[27];
    if constexpr (!OF) {
        const float* r = om + ((size_t)(b * g.H + min(yo, g.H - 1)) * g.W + min(xo, g.W - 1)) * 32;
#pragma unroll
        for (int q = 0; q < 24; q += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(r + q);
            omv[q] = t[0]; omv[q + 1] = t[1]; omv[q + 2] = t[2]; omv[q + 3] = t[3];
        }
        omv[24] = r[24]; omv[25] = r[25]; omv[26] = r[26];
    } else {
        static_assert(FM == 4, "owner lanes: one pixel per lane needs FM == 4");
        constexpr int ZW = 18, ZPS = 144;                                     // neighbourhood width, bytes per pixel (64 x fp16 + 16: conflict-free b128)
        constexpr int ZBYTES = (4 * FM + 2) * ZW * ZPS, TLD = 36;
        float* tbuf = reinterpret_cast<float*>(smem + ((ZBYTES + 63) & ~63)) + wv * (16 * TLD);
        // ---- phase 0a: tile + 1 pixel, all 64 channels, zero outside the image (the conv's zero padding)
        // (r05: branch-free -- clamped, always valid addresses, out-of-image chunks zeroed by a select -- and unrolled in batches of ZU loads:
        // as `if (inside) load` in a rolled loop every iteration was load -> s_waitcnt vmcnt(0) -> convert -> ds_write, eleven exposed
        // memory round trips per thread)
        {
            constexpr int ZN = (4 * FM + 2) * ZW * 8, ZU = 6;
#pragma unroll
            for (int base = 0; base < ZN; base += 256 * ZU) {
                u32x4 zr[ZU];
#pragma unroll
                for (int u = 0; u < ZU; ++u) {
                    int idx = base + u * 256 + tid;
                    if (base + u * 256 + 256 > ZN) idx = idx < ZN ? idx : ZN - 1;
                    const int p = idx >> 3, col = idx & 7;
                    const int ry = p / ZW, rx = p - ry * ZW;
                    const int gy = ty0 - 1 + ry, gx = tx0 - 1 + rx;
                    const bool in = gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
                    const int cy = min(max(gy, 0), g.H - 1), cx = min(max(gx, 0), g.W - 1);
                    const u32x4 v = *reinterpret_cast<const u32x4*>(xb + (uint32_t)((cy * g.W + cx) * g.C + col * 8));
                    zr[u] = in ? v : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < ZU; ++u) {
                    const int idx = base + u * 256 + tid;
                    if (base + u * 256 + 256 <= ZN || idx < ZN) *reinterpret_cast<u32x4*>(smem + (idx >> 3) * ZPS + ((idx & 7) << 4)) = to_h8<TX>(zr[u]);
                }
            }
        }
        f32x4 ao[FM][2];
#pragma unroll
        for (int i = 0; i < FM; ++i) { ao[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ao[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // weight fragments: ring of six steps, five in flight (a step is only 8 MFMAs = 53 ns of matrix-pipe time per wave against an L2 round
        // trip of ~0.5 us: one step of look-ahead left every step waiting on its weights, 17 of the phase's ~25 us)
        constexpr int OR = 6;
        u32x4 ow[OR][2];
        const u32x4* owl = oa.wfm + lane;                                     // [nf 2][step 18][lane 64]
#pragma unroll
        for (int u = 0; u < OR - 1; ++u) { ow[u][0] = owl[(0 * 18 + u) * 64]; ow[u][1] = owl[(1 * 18 + u) * 64]; }
        __syncthreads();
        // ---- phase 0b: 18 k-steps of 32 (tap = s / 2, channels 32 (s & 1) + 8 kq ..)
#pragma unroll
        for (int s_ = 0; s_ < 18; ++s_) {
            if (s_ + OR - 1 < 18) { ow[(s_ + OR - 1) % OR][0] = owl[(0 * 18 + s_ + OR - 1) * 64]; ow[(s_ + OR - 1) % OR][1] = owl[(1 * 18 + s_ + OR - 1) * 64]; }
            const int tap = s_ >> 1, th = tap / 3, tw = tap - th * 3;
            const char* ap = smem + ((wv * FM + th) * ZW + xl + tw) * ZPS + ((s_ & 1) * 32 + kq * 8) * 2;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const u32x4 af = *reinterpret_cast<const u32x4*>(ap + i * ZW * ZPS);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    ao[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, af), __builtin_bit_cast(h8_t, ow[s_ % OR][j]), ao[i][j], 0, 0, 0);
            }
        }
        // ---- phase 0c: + bias, sigmoid on the mask channels, accumulator fragment -> the pixel's owner lane
        const float b0 = oa.shift[xl], b1 = oa.shift[16 + xl];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v0 = ao[i][0][r] + b0, v1 = ao[i][1][r] + b1;           // channels xl and 16 + xl of pixel 4 kq + r
                if (xl >= 2) v1 = 1.f / (1.f + __expf(-v1));                    // 18 .. 26 (27 .. 31 are padding)
                tbuf[(kq * 4 + r) * TLD + xl] = v0;
                tbuf[(kq * 4 + r) * TLD + 16 + xl] = (xl < 11) ? v1 : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            if (kq == i) {                                                    // gi == kq for FM == 4: these sixteen lanes own fragment i's pixels
#pragma unroll
                for (int q = 0; q < 24; q += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(tbuf + xl * TLD + q);
                    omv[q] = t[0]; omv[q + 1] = t[1]; omv[q + 2] = t[2]; omv[q + 3] = t[3];
                }
                omv[24] = tbuf[xl * TLD + 24]; omv[25] = tbuf[xl * TLD + 25]; omv[26] = tbuf[xl * TLD + 26];
                if (oa.om_out && own_ok) {
                    float* o = oa.om_out + ((size_t)(b * g.H + yo) * g.W + xo) * 32;
#pragma unroll
                    for (int q = 0; q < 32; q += 4) *reinterpret_cast<f32x4*>(o + q) = *reinterpret_cast<const f32x4*>(tbuf + xl * TLD + q);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

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

    // per-fragment sampling state of the current tap (consumer side)
    int cb[FM][4]; uint32_t cwa[FM], cwb[FM]; int chw[FM]; bool inp[FM];
    auto geom = [&](int tap) {
        const int th = tap / 3, tw = tap - th * 3;
        const float dh = omv[2 * tap], dw = omv[2 * tap + 1], mk = own_ok ? omv[18 + tap] : 0.f;
        const float h = (float)(yo - 1 + th) + dh, w = (float)(xo - 1 + tw) + dw;
        const bool inside = h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
        const float hf = floorf(h), wf_ = floorf(w);
        const float lh = h - hf, lw = w - wf_, hh = 1.f - lh, hw_ = 1.f - lw;
        const float m_ = inside ? mk : 0.f;
        // clamp before the int conversion: a wild offset must not overflow (the sample is outside the image then: weight 0)
        const int h0 = (int)fminf(fmaxf(hf, -24.f), 30000.f), w0 = (int)fminf(fmaxf(wf_, -24.f), 30000.f);
        const uint32_t wa = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(hh * hw_ * m_, hh * lw * m_));
        const uint32_t wb_ = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(lh * hw_ * m_, lh * lw * m_));
        const int hw = (h0 + 32) | ((w0 + 32) << 16);
        int own_base = 0;
        if constexpr (PD) {                                   // patch byte offset of the top-left corner; bit 31 = sample leaves the patch
            const int ry = h0 - py0, rx = w0 - px0;
            const bool in_patch = ry >= 0 && ry + 1 < SM::rows && rx >= 0 && rx + 1 < kPW;
            own_base = in_patch ? (ry * kPW + rx) * PB : (int)0x80000000;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int src = (i * 16 + xl) << 2;
            cwa[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)wa);
            cwb[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)wb_);
            chw[i] = __builtin_amdgcn_ds_bpermute(src, hw);
            if constexpr (PD) {
                cb[i][0] = __builtin_amdgcn_ds_bpermute(src, own_base);
                inp[i] = cb[i][0] >= 0;
            }
        }
        if constexpr (!PD) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int ry = (chw[i] & 0xffff) - 32 - py0, rx = (chw[i] >> 16) - 32 - px0;
                inp[i] = ry >= 0 && ry + 1 < SM::rows && rx >= 0 && rx + 1 < kPW;
                const int p = ry * kPW + rx;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int pq = p + (q >> 1) * kPW + (q & 1);
                    cb[i][q] = pq * PB + ((pq & (NC - 1)) << 4);
                }
            }
        }
    };
    // corners of fragment i, 16-byte channel column col of the current 64-channel slice (fp16x8 each)
    auto corners = [&](int i, int col, int c0, u32x4 (&v)[4]) {
        if (inp[i]) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[q] = PD ? *reinterpret_cast<const u32x4*>(smem + cb[i][0] + (col << 4) + ((q >> 1) * kPW + (q & 1)) * PB)
                          : *reinterpret_cast<const u32x4*>(smem + (cb[i][q] ^ (col << 4)));
        } else {                                              // rare: sample left the patch -> exact global gather
            const int h0 = (chw[i] & 0xffff) - 32, w0 = (chw[i] >> 16) - 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int hc = h0 + (q >> 1), wc = w0 + (q & 1);
                const bool ok = hc >= 0 && hc < g.H && wc >= 0 && wc < g.W;
                v[q] = ok ? to_h8<TX>(*reinterpret_cast<const u32x4*>(xb + ((size_t)hc * g.W + wc) * g.C + c0 + col * 8)) : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    // bilinear blend in packed fp16 (weights carry the modulation mask): 1 v_pk_mul + 3 v_pk_fma per channel pair
    auto blend = [&](int i, const u32x4 (&v)[4]) -> u32x4 {
        const h2_t wa = __builtin_bit_cast(h2_t, cwa[i]), wb_ = __builtin_bit_cast(h2_t, cwb[i]);
        const h2_t w0 = {wa[0], wa[0]}, w1 = {wa[1], wa[1]}, w2 = {wb_[0], wb_[0]}, w3 = {wb_[1], wb_[1]};
        u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // (copy the elements out first: __builtin_bit_cast applied to a vector-element lvalue is miscompiled by this
            // clang -- every d aliases element 0)
            const uint32_t a0 = v[0][d], a1 = v[1][d], a2 = v[2][d], a3 = v[3][d];
            const h2_t r = __builtin_bit_cast(h2_t, a0) * w0 + __builtin_bit_cast(h2_t, a1) * w1 +
                           __builtin_bit_cast(h2_t, a2) * w2 + __builtin_bit_cast(h2_t, a3) * w3;
            o[d] = __builtin_bit_cast(uint32_t, r);
        }
        return o;
    };

    const int nslice = g.C / CS;
    for (int sl = 0; sl < nslice; ++sl) {
        const int c0 = sl * CS;
        if (OF || sl) __syncthreads();                        // previous slice (or the offset conv's neighbourhood) fully consumed
        // ---- patch load: pix x 8 columns of 16 bytes (bf16 -> fp16), zero outside the image
        // (r05: branch-free and in batches of PLU loads in flight -- see phase 0a; the rolled `if (inside) load` loop waited for every single
        // load: 16 exposed round trips per thread and slice, the 18 us "patch load" line of profiles/r03_dcn_patch_probes.md)
        {
            constexpr int PN = SM::pix * NC, PLU = 8;
#pragma unroll
            for (int base = 0; base < PN; base += 256 * PLU) {
                u32x4 pr[PLU];
#pragma unroll
                for (int u = 0; u < PLU; ++u) {
                    int idx = base + u * 256 + tid;
                    if (base + u * 256 + 256 > PN) idx = idx < PN ? idx : PN - 1;
                    const int p = idx / NC, col = idx % NC;
                    const int ry = p / kPW, rx = p - ry * kPW;
                    const int gy = py0 + ry, gx = px0 + rx;
                    const bool in = gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
                    const int cy = min(max(gy, 0), g.H - 1), cx = min(max(gx, 0), g.W - 1);
                    const u32x4 v = *reinterpret_cast<const u32x4*>(xb + (uint32_t)((cy * g.W + cx) * g.C + c0 + col * 8));
                    pr[u] = in ? v : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < PLU; ++u) {
                    const int idx = base + u * 256 + tid;
                    const int p = idx / NC, col = idx % NC;
                    if (base + u * 256 + 256 <= PN || idx < PN)
                        *reinterpret_cast<u32x4*>(smem + (PD ? p * PB + (col << 4) : ((p * PB + ((p & (NC - 1)) << 4)) ^ (col << 4)))) = to_h8<TX>(pr[u]);
                }
            }
        }
        // weights of the slice's first two steps while the patch lands
        constexpr int RD = FN >= 8 ? 2 : 3;                   // weight ring depth (registers: RD*FN*4)
        u32x4 wb[RD][FN];
        auto sidx = [&](int t, int ks) { return t * g.cpt + sl * KS + ks; };      // fragment step of (tap, k-step)
        wfetch(sidx(0, 0), wb[0]);
        if (RD > 2) wfetch(sidx(1 / KS, 1 % KS), wb[1]);
        __syncthreads();

        // 9*KS steps per slice: (tap, ks), fully unrolled (static ring slots and static indices into the offset registers)
        constexpr int NS = 9 * KS;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int tap = u / KS, ks = u % KS;
            if (ks == 0) geom(tap);
            if (u + RD - 1 < NS) wfetch(sidx((u + RD - 1) / KS, (u + RD - 1) % KS), wb[(u + RD - 1) % RD]);
            const int col = ks * 4 + kq;
            u32x4 v[2][4];
            corners(0, col, c0, v[0]);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if (i + 1 < FM) corners(i + 1, col, c0, v[(i + 1) & 1]);
                const u32x4 af = blend(i, v[i & 1]);
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, af), __builtin_bit_cast(h8_t, wb[u % RD][j]),
                                                                       acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: per-wave staging in the patch memory (all waves are done reading it)
    __syncthreads();
    constexpr int LDS_ = FN * 16 + 4;
    constexpr int GPR = FN * 16 / 8;
    float* stage = reinterpret_cast<float*>(smem) + wv * (16 * LDS_);
    TX* y = reinterpret_cast<TX*>(ep.y);
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
        const int yg = ty0 + wv * FM + i;
        for (int it = lane; it < 16 * GPR; it += 64) {
            const int px = it / GPR, ng = it - px * GPR;
            const int gx = tx0 + px, gn = n0 + ng * 8;
            if (yg < g.H && gx < g.W && gn < ep.Cout) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; e += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(stage + px * LDS_ + ng * 8 + e);
                    v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
                }
                apply_act_chunk<8>(v, ep.act, gn);
                *reinterpret_cast<u32x4*>(y + ((size_t)(b * g.H + yg) * g.W + gx) * ep.ldy + gn) = ElemTraits<TX>::pack(v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int g_opt_dcn_patch_fn8 = 1;
int g_opt_dcn_patch = 1;     // 0 = off, 1 = automatic, 2 = force (FM 4), 3 = force FM 2, 4 = force FM 1, 5-7 = wide margin FM 4/2/1, 8 = padded layout

template <int FN, int FM, int R = 3, int CS = 64, bool PD = false, typename TX = bf16_t, bool OF = false>
static int launch_dcn_patch_t(const mfx_dcn_desc* d, hipStream_t st);

int g_opt_dcn_fuse_off = 1;  // option "dcn_fuse_off": 1 = the LDS-patch kernel computes the offset/mask conv itself where the caller supplies its weights

template <int FN, int FM, int R = 3, int CS = 64, bool PD = false>
static int launch_dcn_patch(const mfx_dcn_desc* d, hipStream_t st) {
    if constexpr (FN == 4 && FM == 4 && R == 7 && CS == 32 && PD) {      // the production variant exists for fp16 maps too, and with the offset conv inside
        const bool of = g_opt_dcn_fuse_off && d->off_w_frag_f16 && d->off_shift && d->C == 64;
        if (d->dtype == MFX_F16) return of ? launch_dcn_patch_t<FN, FM, R, CS, PD, half_t, true>(d, st) : launch_dcn_patch_t<FN, FM, R, CS, PD, half_t>(d, st);
        if (d->dtype == MFX_BF16 && of) return launch_dcn_patch_t<FN, FM, R, CS, PD, bf16_t, true>(d, st);
    }
    if (d->dtype != MFX_BF16) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn patch: this variant is built for bf16 maps only");
    return launch_dcn_patch_t<FN, FM, R, CS, PD, bf16_t>(d, st);
}

template <int FN, int FM, int R, int CS, bool PD, typename TX, bool OF>
static int launch_dcn_patch_t(const mfx_dcn_desc* d, hipStream_t st) {
    DcnPGeom g;
    g.B = d->B; g.H = d->H; g.W = d->W; g.C = d->C;
    g.tiles_x = (d->W + 15) / 16; g.tiles_y = (d->H + 4 * FM - 1) / (4 * FM); g.tiles_n = d->Cout_pad / (FN * 16);
    g.fsteps = d->K_pad / 32; g.cpt = d->C / 32;
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = nullptr; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = 0;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = g.tiles_n;
    const int tiles = d->B * g.tiles_y * g.tiles_x * g.tiles_n;
    constexpr int smem = DcnPSmem<FM, R, CS, PD>::bytes;
    static bool attr_done = false;
    if (!attr_done) {
        MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcn_patch_kernel<FN, FM, R, CS, PD, TX, OF>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    DcnOffArgs oa;
    oa.wfm = reinterpret_cast<const u32x4*>(d->off_w_frag_f16); oa.shift = d->off_shift; oa.om_out = d->offmask_out;
    if (!OF && !d->offmask) return mfx_fail(MFX_ERR_ARG, "dcn: offmask is NULL and this kernel does not compute the offsets itself");
    hipLaunchKernelGGL((dcn_patch_kernel<FN, FM, R, CS, PD, TX, OF>), dim3(tiles), dim3(256), smem, st, reinterpret_cast<const TX*>(d->x), d->offmask,
                       reinterpret_cast<const u32x4*>(d->w_frag_f16), g, ep, oa);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// the automatic choice of the production variant (64 -> 64 on large maps, 16-bit): also where the offset conv can be fused
static bool dcn_patch_auto(const mfx_dcn_desc* d) {
    if (g_opt_dcn_patch != 1 || !d->w_frag_f16 || (d->dtype != MFX_BF16 && d->dtype != MFX_F16)) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1 || d->Ho != d->H || d->Wo != d->W) return false;
    if (d->C % 64 != 0 || d->K_pad != 9 * d->C || d->Cout_pad % 64 != 0) return false;
    return d->C == 64 && d->Cout_pad == 64 && (long)d->B * d->H * d->W >= 65536;
}

bool dcn_patch_fuses_offset_conv(const mfx_dcn_desc* d) {
    // (per-axis geometry goes to the generic gather kernel, which reads `offmask`: never claim the offset conv for it)
    return g_opt_dcn_fuse_off && !d->nonsquare && d->off_w_frag_f16 && d->off_shift && dcn_patch_auto(d);
}

// returns 1 if handled, 0 to fall through to the older kernels, < 0 on error
int try_dcn_patch(const mfx_dcn_desc* d, hipStream_t st) {
    if (g_opt_dcn_patch == 0 || !d->w_frag_f16 || (d->dtype != MFX_BF16 && d->dtype != MFX_F16)) return 0;
    if (d->dtype == MFX_F16 && g_opt_dcn_patch != 1 && g_opt_dcn_patch != 8) return 0;      // fp16 maps: the production variant only
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1 || d->Ho != d->H || d->Wo != d->W) return 0;
    if (d->C % 64 != 0 || d->K_pad != 9 * d->C || d->Cout_pad % 64 != 0) return 0;
    const long px = (long)d->B * d->H * d->W;
    if (g_opt_dcn_patch >= 5 && g_opt_dcn_patch <= 7) {       // wide-margin variants: +-7 pixel offsets in range, 32-channel slices
        const int rc = g_opt_dcn_patch == 5 ? launch_dcn_patch<4, 4, 7, 32>(d, st)
                     : g_opt_dcn_patch == 6 ? launch_dcn_patch<4, 2, 7, 32>(d, st) : launch_dcn_patch<4, 1, 7, 32>(d, st);
        return rc == MFX_OK ? 1 : rc;
    }
    if (g_opt_dcn_patch == 8) {                               // padded patch, owner-computed corner base (see DcnPSmem)
        const int rc = launch_dcn_patch<4, 4, 7, 32, true>(d, st);
        return rc == MFX_OK ? 1 : rc;
    }
    int fm = g_opt_dcn_patch == 2 ? 4 : g_opt_dcn_patch == 3 ? 2 : g_opt_dcn_patch == 4 ? 1 : 0;
    if (!fm) {
        // automatic choice, measured in the full network at B=8 (tools/layer_bench.py --opts dcn_patch=..).  The synthetic
        // benchmark weights produce offsets of std 2.2 .. 7 px: with the +-3 px patch 15 .. 70 % of the samples take the global
        // path and the gain over the first generation is gone (64->64 @ 96x320: 73 us at std 0, 147 us at std 2.5).  The
        // +-7 px patch over 32-channel slices keeps ~99 % of the samples of the 64-channel layers in LDS (std 2.5: 84 us;
        // in the network 86-88 us vs 102-117) and is what those layers use; the multi-slice layers (C >= 128, smaller maps,
        // larger offsets) stay on the first-generation kernel, which measured faster there.
        if (d->C == 64 && d->Cout_pad == 64 && px >= 65536) {
            // padded layout (one corner base + immediate offsets): bit-identical, 80.8 -> 75.0 us (tools/probes/dcn_patch_probe.py)
            const int rc0 = launch_dcn_patch<4, 4, 7, 32, true>(d, st);
            return rc0 == MFX_OK ? 1 : rc0;
        }
        return 0;
    }
    int rc;
    const bool wide = d->Cout_pad % 128 == 0 && g_opt_dcn_patch_fn8;      // one workgroup covers 128 output channels
    if (fm == 4) rc = launch_dcn_patch<4, 4>(d, st);
    else if (fm == 2) rc = wide ? launch_dcn_patch<8, 2>(d, st) : launch_dcn_patch<4, 2>(d, st);
    else rc = wide ? launch_dcn_patch<8, 1>(d, st) : launch_dcn_patch<4, 1>(d, st);
    return rc == MFX_OK ? 1 : rc;
}

}  // namespace mfx
