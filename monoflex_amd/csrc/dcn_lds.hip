// Fused modulated deformable convolution, fourth generation (r06): LDS patch + LDS geometry table + a branch-free, software-pipelined
// sampling loop.
//
//   y[m][n] = act( scale[n] * sum_{tap,c} W[n][tap,c] * mask[m,tap] * bilinear(x[b,:,:,c] @ p(m,tap)) + shift[n] )
//   (reference: model/backbone/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:125-195 + dcn_v2_cuda.cu:139-163; module: dcn_v2.py:118-128)
//
// What the third generation (dcn_patch.hip) left on the table, from its ISA and the r03 probes (profiles/r03_dcn_patch_probes.md): its sampling
// loop is a per-fragment branch diamond (in-patch ds_reads | exact global gather, both fully unrolled: 437 branches, 253 global loads in the
// code of one workgroup), the geometry of every tap is recomputed per channel slice behind 16 ds_bpermutes that nothing overlaps, the patch of
// the next slice is loaded only after the previous one is consumed, and the kernel needs all 256 VGPRs.  MFMA pipe 12 % busy with no unit
// above 50 %: a latency chain.  Here:
//   * GEOMETRY TABLE: the bilinear weights (4 x fp16, mask folded in) and the patch byte offset of every (tap, pixel) of the tile are
//     computed ONCE per workgroup by the pixel's owner lane and kept in LDS (12 bytes each, 27 KB); the sampling loop reads them back with
//     plain ds_read_b64 / ds_read_b32 two fragments ahead -- no ds_bpermute, no per-slice recomputation, no offset registers live in the loop;
//   * 16-CHANNEL SLICES: patch pixel = 32 bytes of fp16 + 16 pad (48 B: 16 consecutive pixels cover the 64 banks once); 32 x 32 pixels
//     (16 x 16 tile grown by 8: samples up to +-7 px away are in range) = 48 KB, + the table = 75 KB: two workgroups per CU.  One MFMA k-step
//     (K = 32) is TWO taps x 16 channels: lanes of k-groups 0-1 sample tap 2j, k-groups 2-3 tap 2j + 1; five steps per slice, the second half
//     of the fifth multiplies zero weights (its lanes read a zero table entry);  the weights arrive pre-packed in that K order
//     (mfx_dcn_desc.w_pair_f16, ops.dcn_pair_fragments);
//   * the sampling loop is straight-line code: geometry of fragment f + 2, corner reads of fragment f + 1 (4 x ds_read_b128 against ONE base
//     + immediates), packed-fp16 blend (1 v_pk_mul + 3 v_pk_fma per channel pair, its result IS the MFMA A fragment) and 4 MFMAs of
//     fragment f, pinned in that order by sched_barriers;
//   * samples that leave the patch ("far") get weight zero in the loop; their (tap, tile row) pairs are remembered in a wave-uniform
//     bit mask and a FAR PASS after the last slice adds their exact contribution from global memory (extra MFMAs whose A rows are zero
//     for every other pixel).  Rare by construction (0.1 % of the samples of the 96 x 320 layers), so any offset stays exact;
//   * the next slice's patch chunks are fetched into registers while the current slice is sampled (one L2 round trip per slice hidden);
//   * OF: the module's 27-channel offset / mask conv runs inside (phase 0, as in dcn_patch.hip: 18 x 18 x 64 neighbourhood in the still
//     unused patch memory, 144 MFMAs per wave, bias + sigmoid, transposition to the owner lanes).
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "igemm.h"

namespace mfx {

struct DcnLGeom { int B, H, W, C, tiles_x, tiles_y, nslice, fsteps_pair, fsteps_far, cpt_far; };
struct DcnLOffArgs { const u32x4* wfm; const float* shift; float* om_out; };

typedef _Float16 lh2_t __attribute__((ext_vector_type(2)));
typedef _Float16 lh8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t l_bf2_to_h2(uint32_t d) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)));
}
template <typename TX> __device__ __forceinline__ u32x4 l_to_h8(const u32x4& v);
template <> __device__ __forceinline__ u32x4 l_to_h8<bf16_t>(const u32x4& v) { return u32x4{l_bf2_to_h2(v.x), l_bf2_to_h2(v.y), l_bf2_to_h2(v.z), l_bf2_to_h2(v.w)}; }
template <> __device__ __forceinline__ u32x4 l_to_h8<half_t>(const u32x4& v) { return v; }

#ifdef MFX_PROBES     /* probe build (MFX_PROBES=1 python -m monoflex_amd.build): per-wave time stamps of the phases, read back by tools/probes/dcn_lds_probe.py */
__device__ unsigned long long mfx_dcn_lds_probe[1024 * 4 * 16];
#define LPROBE(k) do { if (lane == 0 && blockIdx.x < 1024) mfx_dcn_lds_probe[(blockIdx.x * 4 + wv) * 16 + (k)] = (unsigned long long)clock64(); } while (0)
#else
#define LPROBE(k) do { } while (0)
#endif

// TR = tile rows (16: four fragments per wave, two workgroups per CU; 8: two fragments per wave, 51 KB and <= 168 VGPRs: THREE workgroups per CU --
// every phase of this kernel is a latency chain, and a third resident workgroup is what covers them)
template <int R, int TR> struct DcnLSmem {
    static constexpr int PW = 16 + 2 * (R + 1), PH = TR + 2 * (R + 1);
    static constexpr int PB = 48;                                  // bytes per patch pixel: 16 x fp16 + 16
    static constexpr int patch_bytes = PW * PH * PB;
    static constexpr int NPIX = TR * 16;
    static constexpr int NG = 9 * NPIX;                            // (tap, pixel) entries; entries NG + 16 i (i < FM) are all zero
    static constexpr int GW_OFF = patch_bytes;                     // 8-byte entries: [w00 w01 | w10 w11] fp16
    static constexpr int GB_OFF = GW_OFF + (NG + 64) * 8;          // 4-byte entries: patch byte offset of the top-left corner, or bit 31 + coordinates (far)
    static constexpr int bytes = GB_OFF + (NG + 64) * 4;
    // fused offset conv: the (TR + 2) x 18 x 64-channel neighbourhood lives in the patch memory; the four 2304-byte transposition buffers behind it where
    // the patch leaves room (TR = 8), else in the offset half of the table (written only after barrier A)
    static constexpr int ZBYTES = (TR + 2) * 18 * 144;
    static constexpr int TBUF_OFF = (ZBYTES + 4 * 2304 <= patch_bytes) ? ZBYTES : GB_OFF;
    static_assert(ZBYTES <= patch_bytes && (TBUF_OFF != GB_OFF || 4 * 2304 <= (NG + 64) * 4), "offset-conv scratch fits");
};

template <int R, int TR, typename TX, bool OF>
__global__ __launch_bounds__(256, TR == 8 ? 3 : 2) void dcn_lds_kernel(const TX* __restrict__ x, const float* __restrict__ om, const u32x4* __restrict__ wpair,
                                                        const u32x4* __restrict__ wfar, DcnLGeom g, EpiArgs ep, DcnLOffArgs oa) {
    using SM = DcnLSmem<R, TR>;
    constexpr int PW = SM::PW, PH = SM::PH, PB = SM::PB, FM = TR / 4, FN = 4, NPIX = SM::NPIX;
    constexpr int OWN = 4 / FM;                                    // lanes per tile pixel in the owner phases (1 | 2): each takes TPT of the nine taps
    constexpr int TPT = (9 + OWN - 1) / OWN;
    constexpr int ROWB = PW * PB;                                  // bytes per patch row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xl = lane & 15, kq = lane >> 4;

    LPROBE(0);
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = tile % g.tiles_x; tile /= g.tiles_x;
    const int ty = tile % g.tiles_y, b = tile / g.tiles_y;
    const int ty0 = ty * TR, tx0 = tx * 16;
    const int py0 = ty0 - (R + 1), px0 = tx0 - (R + 1);           // image coordinates of patch pixel (0,0)
    const TX* xb = x + (size_t)b * g.H * g.W * g.C;
    const char* xbb = reinterpret_cast<const char*>(xb);

    // this thread's pixel of the tile (owner lane of its geometry): tile row wv*FM + kq % FM, column xl; lanes kq / FM = tg share a pixel and split its taps
    const int oi = kq % FM, tg = kq / FM;
    const int opix = (wv * FM + oi) * 16 + xl;
    const int yo = ty0 + wv * FM + oi, xo = tx0 + xl;
    const bool own_ok = yo < g.H && xo < g.W;

    // ---- patch chunk bookkeeping: 2048 chunks of 16 bytes per slice, 8 per thread.  chunk idx = u*256 + tid: pixel idx >> 1, column idx & 1
    constexpr int PU = PW * PH * 2 / 256;
    static_assert(PW * PH * 2 % 256 == 0, "patch chunks per thread");
    uint32_t poff[PU]; uint32_t pin = 0;
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int idx = u * 256 + tid, p = idx >> 1, col = idx & 1;
        const int ry = p / PW, rx = p - ry * PW;
        const int gy = py0 + ry, gx = px0 + rx;
        const bool in = gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
        const int cy = min(max(gy, 0), g.H - 1), cx = min(max(gx, 0), g.W - 1);
        poff[u] = (uint32_t)((cy * g.W + cx) * g.C + col * 8) * 2u;
        pin |= in ? (1u << u) : 0u;
    }
    const int pw_addr = (tid >> 1) * PB + (tid & 1) * 16;         // LDS address of chunk u: pw_addr + u * 128 * PB
    u32x4 pr[PU];
    auto patch_fetch = [&](int sl) {
#pragma unroll
        for (int u = 0; u < PU; ++u) pr[u] = *reinterpret_cast<const u32x4*>(xbb + poff[u] + (uint32_t)sl * 32u);
    };
    auto patch_write = [&]() {
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const u32x4 v = (pin >> u) & 1u ? pr[u] : u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(smem + pw_addr + u * (128 * PB)) = l_to_h8<TX>(v);
        }
    };

    // offsets / mask of this lane's taps (tap tt + TPT * tg; a lane group without a fifth tap re-reads tap 8 and ignores it)
    float odh[TPT], odw[TPT], omk[TPT];
    const int tap0 = TPT * tg;
    if constexpr (!OF) {
        const float* r = om + ((size_t)(b * g.H + min(yo, g.H - 1)) * g.W + min(xo, g.W - 1)) * 32;
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
            const int tap = min(tap0 + tt, 8);
            odh[tt] = r[2 * tap]; odw[tt] = r[2 * tap + 1]; omk[tt] = r[18 + tap];
        }
        patch_fetch(0);
    } else {
        // ---- phase 0: the offset / mask conv of the tile (dcn_patch.hip's, C = 64): tile + 1 pixel, all 64 channels, in the patch memory
        constexpr int ZW = 18, ZPS = 144, TLD = 36;
        float* tbuf = reinterpret_cast<float*>(smem + SM::TBUF_OFF) + wv * (16 * TLD);
        {
            // (all of a thread's loads in ONE batch: two batches were two exposed memory round trips at the head of every workgroup, 14 % of its life)
            constexpr int ZN = (TR + 2) * ZW * 8, ZU = (ZN + 255) / 256;
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
                    if (base + u * 256 + 256 <= ZN || idx < ZN) *reinterpret_cast<u32x4*>(smem + (idx >> 3) * ZPS + ((idx & 7) << 4)) = l_to_h8<TX>(zr[u]);
                }
            }
        }
        f32x4 ao[FM][2];
#pragma unroll
        for (int i = 0; i < FM; ++i) { ao[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ao[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        constexpr int OR = 6;
        u32x4 ow[OR][2];
        const u32x4* owl = oa.wfm + lane;                                     // [nf 2][step 18][lane 64]
#pragma unroll
        for (int u = 0; u < OR - 1; ++u) { ow[u][0] = owl[(0 * 18 + u) * 64]; ow[u][1] = owl[(1 * 18 + u) * 64]; }
        patch_fetch(0);                                                       // slice 0 of the sampling patch rides under the conv
        __syncthreads();
        LPROBE(1);
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
                    ao[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lh8_t, af), __builtin_bit_cast(lh8_t, ow[s_ % OR][j]), ao[i][j], 0, 0, 0);
            }
        }
        LPROBE(2);
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
            if (oi == i) {                                                    // these lanes own fragment i's pixels
#pragma unroll
                for (int tt = 0; tt < TPT; ++tt) {
                    const int tap = min(tap0 + tt, 8);
                    const float2 d2 = *reinterpret_cast<const float2*>(tbuf + xl * TLD + 2 * tap);
                    odh[tt] = d2.x; odw[tt] = d2.y; omk[tt] = tbuf[xl * TLD + 18 + tap];
                }
                if (oa.om_out && own_ok && tg == 0) {
                    float* o = oa.om_out + ((size_t)(b * g.H + yo) * g.W + xo) * 32;
#pragma unroll
                    for (int q = 0; q < 32; q += 4) *reinterpret_cast<f32x4*>(o + q) = *reinterpret_cast<const f32x4*>(tbuf + xl * TLD + q);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    LPROBE(3);
    // ---- phase G: geometry of the nine taps of this thread's pixel -> table
    uint32_t gb[TPT];
    unsigned long long farm[TPT];
#pragma unroll
    for (int tt = 0; tt < TPT; ++tt) {
        // this lane's tt-th tap: tt + TPT * tg (OWN == 2: lanes of the second group start at tap 5 and have no fifth one)
        const int tap = tap0 + tt;
        const bool has = tap < 9;
        const float dh = odh[tt], dw = odw[tt];
        const float mk = (own_ok && has) ? omk[tt] : 0.f;
        const int th = tap / 3, tw = tap - th * 3;
        const float h = (float)(yo - 1 + th) + dh, w = (float)(xo - 1 + tw) + dw;
        const bool inside = h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
        const float hf = floorf(h), wf_ = floorf(w);
        const float lh = h - hf, lw = w - wf_, hh = 1.f - lh, hw_ = 1.f - lw;
        const float m_ = inside ? mk : 0.f;
        // clamp before the int conversion: a wild offset must not overflow (the sample is outside the image then: weight 0)
        const int h0 = (int)fminf(fmaxf(hf, -24.f), 30000.f), w0 = (int)fminf(fmaxf(wf_, -24.f), 30000.f);
        uint32_t wa = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(hh * hw_ * m_, hh * lw * m_));
        uint32_t wb_ = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(lh * hw_ * m_, lh * lw * m_));
        const int ry = h0 - py0, rx = w0 - px0;
        const bool in_patch = ry >= 0 && ry + 1 < PH && rx >= 0 && rx + 1 < PW;
        const bool live = m_ != 0.f;                                          // (a sample outside the image or under a zero mask adds nothing: never far)
        const bool far = live && !in_patch;
        uint32_t base = in_patch ? (uint32_t)((ry * PW + rx) * PB) : 0u;
        if (far) { base = 0x80000000u | (uint32_t)(h0 + 32) | ((uint32_t)(w0 + 32) << 16); wa |= 0x80008000u; wb_ |= 0x80008000u; }   // sign bits: the loop clamps these weights to zero
        if (!live) { wa = 0u; wb_ = 0u; }
        if (has) *reinterpret_cast<uint2*>(smem + SM::GW_OFF + (tap * NPIX + opix) * 8) = uint2{wa, wb_};
        gb[tt] = base;
        farm[tt] = __builtin_amdgcn_ballot_w64(far);         // (mk = 0 without a tap: never far)
    }
    if (tid < 4) *reinterpret_cast<uint2*>(smem + SM::GW_OFF + (SM::NG + 16 * tid) * 8) = uint2{0u, 0u};
    LPROBE(4);
    __syncthreads();                                          // barrier A: phase 0 (neighbourhood, transposition buffers) fully consumed
    LPROBE(5);
#pragma unroll
    for (int tt = 0; tt < TPT; ++tt)
        if (tt + TPT * tg < 9) *reinterpret_cast<uint32_t*>(smem + SM::GB_OFF + ((tt + TPT * tg) * NPIX + opix) * 4) = gb[tt];
    if (tid < 4) *reinterpret_cast<uint32_t*>(smem + SM::GB_OFF + (SM::NG + 16 * tid) * 4) = 0u;
    patch_write();
    __syncthreads();                                          // barrier B
    LPROBE(6);

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- sampling loop
    // lane constants: table entry of (step j, tile row i) = gent + 2 NPIX j + 16 i  (tap 2j + (kq >> 1), pixel (wv*FM + i)*16 + xl); in the
    // fifth step k-groups 2-3 have no tap: they read the zero entries
    const int gent = (kq >> 1) * NPIX + wv * (FM * 16) + xl;
    const int gw_a = SM::GW_OFF + gent * 8, gb_a = SM::GB_OFF + gent * 4;
    const int gw_z = (kq >> 1) ? SM::GW_OFF + (SM::NG - 8 * NPIX) * 8 : gw_a;          // + (8 NPIX + 16 i) * 8 below = zero entry NG + 16 i
    const int gb_z = (kq >> 1) ? SM::GB_OFF + (SM::NG - 8 * NPIX) * 4 : gb_a;
    const int ccol = (kq & 1) * 16;
    const u32x4* wpl = wpair + lane;                         // [nf 4][nslice*5][64 lanes]
    auto wfetch = [&](int step, u32x4 (&wf)[FN]) {
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = wpl[((size_t)j * g.fsteps_pair + step) * 64];
    };
    u32x4 wq[2][FN];
    wfetch(0, wq[0]);

    constexpr int NF = 5 * FM;                                // fragment-steps per slice
    for (int sl = 0; sl < g.nslice; ++sl) {
        if (sl + 1 < g.nslice) patch_fetch(sl + 1);
        uint2 gwv[3]; uint32_t gbv[3];                        // geometry in flight (ring of three)
        u32x4 cv[2][4];                                       // corners in flight (ring of two)
        auto geo_read = [&](int f, int slot) {
            const int j = f / FM, i = f % FM;
            if (j < 4) {
                gwv[slot] = *reinterpret_cast<const uint2*>(smem + gw_a + (2 * NPIX * j + 16 * i) * 8);
                gbv[slot] = *reinterpret_cast<const uint32_t*>(smem + gb_a + (2 * NPIX * j + 16 * i) * 4);
            } else {
                gwv[slot] = *reinterpret_cast<const uint2*>(smem + gw_z + (8 * NPIX + 16 * i) * 8);
                gbv[slot] = *reinterpret_cast<const uint32_t*>(smem + gb_z + (8 * NPIX + 16 * i) * 4);
            }
        };
        auto corner_read = [&](int slot_g, int slot_c) {
            const int base = max((int)gbv[slot_g], 0) + ccol;
            cv[slot_c][0] = *reinterpret_cast<const u32x4*>(smem + base);
            cv[slot_c][1] = *reinterpret_cast<const u32x4*>(smem + base + PB);
            cv[slot_c][2] = *reinterpret_cast<const u32x4*>(smem + base + ROWB);
            cv[slot_c][3] = *reinterpret_cast<const u32x4*>(smem + base + ROWB + PB);
        };
        auto blend = [&](int slot_g, int slot_c) -> u32x4 {
            // far samples carry negative weights: zero here (integer max on the halves: a negative fp16 is a negative int16 -- one v_pk_max_i16, no
            // canonicalisation of the operand as the fp16 max needs)
            typedef short ls2_t __attribute__((ext_vector_type(2)));
            const ls2_t z2 = {0, 0};
            const lh2_t wa = __builtin_bit_cast(lh2_t, __builtin_elementwise_max(__builtin_bit_cast(ls2_t, gwv[slot_g].x), z2));
            const lh2_t wb_ = __builtin_bit_cast(lh2_t, __builtin_elementwise_max(__builtin_bit_cast(ls2_t, gwv[slot_g].y), z2));
            const lh2_t w0 = {wa[0], wa[0]}, w1 = {wa[1], wa[1]}, w2 = {wb_[0], wb_[0]}, w3 = {wb_[1], wb_[1]};
            u32x4 o;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t a0 = cv[slot_c][0][d], a1 = cv[slot_c][1][d], a2 = cv[slot_c][2][d], a3 = cv[slot_c][3][d];
                const lh2_t r = __builtin_bit_cast(lh2_t, a0) * w0 + __builtin_bit_cast(lh2_t, a1) * w1 +
                                __builtin_bit_cast(lh2_t, a2) * w2 + __builtin_bit_cast(lh2_t, a3) * w3;
                o[d] = __builtin_bit_cast(uint32_t, r);
            }
            return o;
        };
        geo_read(0, 0);
        geo_read(1, 1);
        corner_read(0, 0);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int j = f / FM, i = f % FM;
            if (i == 0) {                                     // a new k-step: weights of the next one (the next slice's first after the last)
                const int nstep = sl * 5 + j + 1;
                wfetch(min(nstep, g.fsteps_pair - 1), wq[(j + 1) & 1]);
            }
            if (f + 2 < NF) geo_read(f + 2, (f + 2) % 3);
            if (f + 1 < NF) corner_read((f + 1) % 3, (f + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 af = blend(f % 3, f & 1);
#pragma unroll
            for (int n = 0; n < FN; ++n)
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lh8_t, af), __builtin_bit_cast(lh8_t, wq[j & 1][n]), acc[i][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ring parity: five steps per slice, so the slot the next slice starts from alternates -- keep it at slot 0
        // (step 5 of this slice == step 0 of the next was fetched into wq[1])
#pragma unroll
        for (int n = 0; n < FN; ++n) wq[0][n] = wq[1][n];
        if (sl < 4) LPROBE(7 + 2 * sl);
        if (sl + 1 < g.nslice) {
            __syncthreads();                                  // every wave is done with this slice's patch
            patch_write();
            __syncthreads();
        }
        if (sl < 3) LPROBE(8 + 2 * sl);
    }

    // The patch is dead from here on: the far pass reads only the table (behind the patch), the epilogue stages each wave's rows in the wave's own
    // slice of the patch memory -- ONE barrier here instead of one after the far pass, whose length differs from wave to wave
    // (r06 probe: the epilogue's barrier waited 4-5 k cycles for the slowest wave's far pass)
    __syncthreads();
    float sc[FN], sh[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        sc[j] = ep.scale ? ep.scale[j * 16 + xl] : 1.f;
        sh[j] = ep.shift ? ep.shift[j * 16 + xl] : 0.f;
    }
    // ---- far pass: exact global gather for the samples that left the patch (wave-uniform masks from phase G)
    {
        unsigned long long any = 0;
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) any |= farm[tt];
        if (any) {
            const u32x4* wfl = wfar + lane;                  // standard fragment-major fp16 weights [nf][9 * C/32][64 lanes]
            for (int tap = 0; tap < 9; ++tap) {
                const int ftt = tap % TPT, ftg = tap / TPT;  // which lane group computed this tap, as its ftt-th
                unsigned long long fm_ = 0;
#pragma unroll
                for (int t = 0; t < TPT; ++t) fm_ = (t == ftt) ? farm[t] : fm_;
                fm_ >>= (ftg * FM) * 16;
                if (!(fm_ & (FM == 4 ? ~0ull : 0xffffffffull))) continue;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    if (!((fm_ >> (16 * i)) & 0xffffull)) continue;
                    const int ent = tap * NPIX + (wv * FM + i) * 16 + xl;
                    const uint2 gw = *reinterpret_cast<const uint2*>(smem + SM::GW_OFF + ent * 8);
                    const uint32_t gbase = *reinterpret_cast<const uint32_t*>(smem + SM::GB_OFF + ent * 4);
                    const bool isfar = (gbase >> 31) != 0u;
                    const int h0 = (int)(gbase & 0xffffu) - 32, w0 = (int)((gbase >> 16) & 0x7fffu) - 32;
                    const uint32_t wau = isfar ? (gw.x & 0x7fff7fffu) : 0u, wbu = isfar ? (gw.y & 0x7fff7fffu) : 0u;
                    const lh2_t wa = __builtin_bit_cast(lh2_t, wau), wb_ = __builtin_bit_cast(lh2_t, wbu);
                    const lh2_t w0_ = {wa[0], wa[0]}, w1_ = {wa[1], wa[1]}, w2_ = {wb_[0], wb_[0]}, w3_ = {wb_[1], wb_[1]};
                    for (int ks = 0; ks < g.cpt_far; ++ks) {
                        u32x4 v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int hc = h0 + (q >> 1), wc = w0 + (q & 1);
                            const bool ok = isfar && hc >= 0 && hc < g.H && wc >= 0 && wc < g.W;
                            v[q] = u32x4{0u, 0u, 0u, 0u};
                            if (ok) v[q] = l_to_h8<TX>(*reinterpret_cast<const u32x4*>(xb + ((size_t)hc * g.W + wc) * g.C + ks * 32 + kq * 8));
                        }
                        u32x4 af;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const uint32_t a0 = v[0][d], a1 = v[1][d], a2 = v[2][d], a3 = v[3][d];
                            const lh2_t r = __builtin_bit_cast(lh2_t, a0) * w0_ + __builtin_bit_cast(lh2_t, a1) * w1_ +
                                            __builtin_bit_cast(lh2_t, a2) * w2_ + __builtin_bit_cast(lh2_t, a3) * w3_;
                            af[d] = __builtin_bit_cast(uint32_t, r);
                        }
#pragma unroll
                        for (int n = 0; n < FN; ++n) {
                            const u32x4 wf = wfl[((size_t)n * g.fsteps_far + tap * g.cpt_far + ks) * 64];
                            acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lh8_t, af), __builtin_bit_cast(lh8_t, wf), acc[i][n], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue: per-wave staging in the (dead) patch memory
    LPROBE(14);
    constexpr int LDS_ = FN * 16 + 4;
    constexpr int GPR = FN * 16 / 8;
    float* stage = reinterpret_cast<float*>(smem) + wv * (16 * LDS_);
    TX* y = reinterpret_cast<TX*>(ep.y);
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
            const int gx = tx0 + px, gn = ng * 8;
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
    LPROBE(15);
}

#ifdef MFX_PROBES
}  // namespace mfx
extern "C" int mfx_dcn_lds_probe_read(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(mfx::mfx_dcn_lds_probe), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
namespace mfx {
#endif

// ------------------------------------------------------------------------------------------------------------------------------------
// Split precision (MFX_F16X2: fp32 maps, fp16 (hi, lo) MFMA operand pairs -- the mode that carries the north-star gate): the same kernel with an fp32
// patch.  What changes against the 16-bit form above:
//   * patch pixel = 16 channels x 4 bytes + 16 pad = 80 B; 8 x 16 tile grown by 8 = 24 x 32 pixels = 60 KB; the table holds (lh, lw, mask, base) as
//     fp32 (16 bytes, ONE ds_read_b128 per fragment) -- fp16 blend weights would cap the result at 11 bits; 79 KB: two workgroups per CU;
//   * a lane's 8 k-slots are 32 bytes per corner (two ds_read_b128); the four-corner blend runs in fp32 (v_pk_fma_f32), its result is split into
//     hi = fp16(s), lo = fp16(s - hi) by common.h's lds_operand<f32s_t> (range sentinel included) and multiplied as hi.Whi + hi.Wlo + lo.Whi;
//   * weights: w_pair_f16 / w_frag_f16 each hold TWO consecutive arrays -- the hi halves, then the lo halves of the weights times the pack's power-of-two
//     scale (ops.split_weight_scale; the epilogue's `scale` already carries its inverse);
//   * offsets come from the module's separate (split-precision) offset conv; no fused phase 0; the next slice's patch is loaded after the current one is
//     consumed (no register prefetch: 12 chunks per thread would not fit next to two corner sets of 32 registers).
// Reference arithmetic is fp32 (src/cuda/dcn_v2_cuda.cu:58); measured parity: tests/test_gpu_ops.py::test_dcn_lds_split_kernel_matches_the_fp32_kernel.
template <int R>
__global__ __launch_bounds__(256, 2) void dcn_lds_split_kernel(const float* __restrict__ x, const float* __restrict__ om, const u32x4* __restrict__ wpair,
                                                              const u32x4* __restrict__ wfar, size_t pair_lo, size_t far_lo, DcnLGeom g, EpiArgs ep) {
    constexpr int TR = 8, FM = 2, FN = 4, NPIX = TR * 16, OWN = 2, TPT = 5;
    constexpr int PW = 16 + 2 * (R + 1), PH = TR + 2 * (R + 1), PB = 80, ROWB = PW * PB;
    constexpr int patch_bytes = PW * PH * PB, NG = 9 * NPIX, GT_OFF = patch_bytes;       // table: 16-byte entries {lh, lw, mask, base}
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xl = lane & 15, kq = lane >> 4;

    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = tile % g.tiles_x; tile /= g.tiles_x;
    const int ty = tile % g.tiles_y, b = tile / g.tiles_y;
    const int ty0 = ty * TR, tx0 = tx * 16;
    const int py0 = ty0 - (R + 1), px0 = tx0 - (R + 1);
    const float* xb = x + (size_t)b * g.H * g.W * g.C;
    const char* xbb = reinterpret_cast<const char*>(xb);

    const int oi = kq % FM, tg = kq / FM;
    const int opix = (wv * FM + oi) * 16 + xl;
    const int yo = ty0 + wv * FM + oi, xo = tx0 + xl;
    const bool own_ok = yo < g.H && xo < g.W;

    // patch chunks: 768 pixels x 4 chunks of 16 bytes per slice, 12 per thread: chunk idx = u*256 + tid: pixel idx >> 2, column idx & 3
    constexpr int PU = PW * PH * 4 / 256;
    static_assert(PW * PH * 4 % 256 == 0, "patch chunks per thread");
    auto patch_load = [&](int sl) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 pr[PU / 2];
#pragma unroll
            for (int u = 0; u < PU / 2; ++u) {
                const int idx = (half * (PU / 2) + u) * 256 + tid, p = idx >> 2, col = idx & 3;
                const int ry = p / PW, rx = p - ry * PW;
                const int gy = py0 + ry, gx = px0 + rx;
                const bool in = gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
                const int cy = min(max(gy, 0), g.H - 1), cx = min(max(gx, 0), g.W - 1);
                const u32x4 v = *reinterpret_cast<const u32x4*>(xbb + (uint32_t)(((cy * g.W + cx) * g.C + sl * 16 + col * 4) * 4));
                pr[u] = in ? v : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < PU / 2; ++u) {
                const int idx = (half * (PU / 2) + u) * 256 + tid;
                *reinterpret_cast<u32x4*>(smem + (idx >> 2) * PB + (idx & 3) * 16) = pr[u];
            }
        }
    };

    // ---- phase G: geometry of this lane's taps -> table
    {
        const float* r = om + ((size_t)(b * g.H + min(yo, g.H - 1)) * g.W + min(xo, g.W - 1)) * 32;
        const int tap0 = TPT * tg;
        float odh[TPT], odw[TPT], omk[TPT];
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
            const int tap = min(tap0 + tt, 8);
            odh[tt] = r[2 * tap]; odw[tt] = r[2 * tap + 1]; omk[tt] = r[18 + tap];
        }
        patch_load(0);
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
            const int tap = tap0 + tt;
            const bool has = tap < 9;
            const float mk = (own_ok && has) ? omk[tt] : 0.f;
            const int th = tap / 3, tw = tap - th * 3;
            const float h = (float)(yo - 1 + th) + odh[tt], w = (float)(xo - 1 + tw) + odw[tt];
            const bool inside = h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
            const float hf = floorf(h), wf_ = floorf(w);
            const float m_ = inside ? mk : 0.f;
            const int h0 = (int)fminf(fmaxf(hf, -24.f), 30000.f), w0 = (int)fminf(fmaxf(wf_, -24.f), 30000.f);
            const int ry = h0 - py0, rx = w0 - px0;
            const bool in_patch = ry >= 0 && ry + 1 < PH && rx >= 0 && rx + 1 < PW;
            const bool far = m_ != 0.f && !in_patch;
            uint32_t base = in_patch ? (uint32_t)((ry * PW + rx) * PB) : 0u;
            if (far) base = 0x80000000u | (uint32_t)(h0 + 32) | ((uint32_t)(w0 + 32) << 16);
            // the mask of a far sample is stored NEGATED: the loop clamps it to zero, the far pass takes its magnitude
            if (has) *reinterpret_cast<f32x4*>(smem + GT_OFF + (tap * NPIX + opix) * 16) = f32x4{h - hf, w - wf_, far ? -m_ : m_, __uint_as_float(base)};
        }
        if (tid < FM) *reinterpret_cast<f32x4*>(smem + GT_OFF + (NG + 16 * tid) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    unsigned long long farm[TPT];
    __syncthreads();
    {   // far masks from the table (the ballot needs every lane's own entries: re-read them -- after the barrier they are all there)
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) {
            const int tap = TPT * tg + tt;
            const float mk = tap < 9 ? reinterpret_cast<const float*>(smem + GT_OFF + (min(tap, 8) * NPIX + opix) * 16)[2] : 0.f;
            farm[tt] = __builtin_amdgcn_ballot_w64(mk < 0.f);
        }
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int gent = (kq >> 1) * NPIX + wv * (FM * 16) + xl;
    const int gt_a = GT_OFF + gent * 16;
    const int gt_z = (kq >> 1) ? GT_OFF + (NG - 8 * NPIX) * 16 : gt_a;
    const int ccol = (kq & 1) * 32;
    const u32x4* wph = wpair + lane;
    const u32x4* wpl = wpair + pair_lo + lane;
    auto wfetch = [&](int step, u32x4 (&wh)[FN], u32x4 (&wl)[FN]) {
#pragma unroll
        for (int j = 0; j < FN; ++j) { wh[j] = wph[((size_t)j * g.fsteps_pair + step) * 64]; wl[j] = wpl[((size_t)j * g.fsteps_pair + step) * 64]; }
    };
    u32x4 wqh[2][FN], wql[2][FN];
    wfetch(0, wqh[0], wql[0]);

    constexpr int NF = 5 * FM;
    for (int sl = 0; sl < g.nslice; ++sl) {
        f32x4 gv[3];
        u32x4 cv[2][4][2];
        auto geo_read = [&](int f, int slot) {
            const int j = f / FM, i = f % FM;
            gv[slot] = j < 4 ? *reinterpret_cast<const f32x4*>(smem + gt_a + (2 * NPIX * j + 16 * i) * 16)
                             : *reinterpret_cast<const f32x4*>(smem + gt_z + (8 * NPIX + 16 * i) * 16);
        };
        auto corner_read = [&](int slot_g, int slot_c) {
            const int base = max((int)__float_as_uint(gv[slot_g][3]), 0) + ccol;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cv[slot_c][q][0] = *reinterpret_cast<const u32x4*>(smem + base + (q >> 1) * ROWB + (q & 1) * PB);
                cv[slot_c][q][1] = *reinterpret_cast<const u32x4*>(smem + base + (q >> 1) * ROWB + (q & 1) * PB + 16);
            }
        };
        // fp32 blend of the lane's 8 k-slots, then the (hi, lo) split: -> the two MFMA A operands
        auto blend = [&](int slot_g, int slot_c, u32x4& ah, u32x4& al) {
            const float lh = gv[slot_g][0], lw = gv[slot_g][1], m_ = fmaxf(gv[slot_g][2], 0.f);
            const float hh = 1.f - lh, hw_ = 1.f - lw;
            const float w0 = hh * hw_ * m_, w1 = hh * lw * m_, w2 = lh * hw_ * m_, w3 = lh * lw * m_;
            u32x4 sp[2];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    f32x2 t = f32x2{__uint_as_float(cv[slot_c][0][c2][e]), __uint_as_float(cv[slot_c][0][c2][e + 1])} * w0;
                    t = __builtin_elementwise_fma(f32x2{__uint_as_float(cv[slot_c][1][c2][e]), __uint_as_float(cv[slot_c][1][c2][e + 1])}, f32x2{w1, w1}, t);
                    t = __builtin_elementwise_fma(f32x2{__uint_as_float(cv[slot_c][2][c2][e]), __uint_as_float(cv[slot_c][2][c2][e + 1])}, f32x2{w2, w2}, t);
                    t = __builtin_elementwise_fma(f32x2{__uint_as_float(cv[slot_c][3][c2][e]), __uint_as_float(cv[slot_c][3][c2][e + 1])}, f32x2{w3, w3}, t);
                    o[e] = t[0]; o[e + 1] = t[1];
                }
                sp[c2] = lds_operand<f32s_t>(ElemTraits<float>::pack(o));       // [h0 h1 | h2 h3 | l0 l1 | l2 l3]
            }
            ah = u32x4{sp[0].x, sp[0].y, sp[1].x, sp[1].y};
            al = u32x4{sp[0].z, sp[0].w, sp[1].z, sp[1].w};
        };
        geo_read(0, 0);
        geo_read(1, 1);
        corner_read(0, 0);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int j = f / FM, i = f % FM;
            if (i == 0) wfetch(min(sl * 5 + j + 1, g.fsteps_pair - 1), wqh[(j + 1) & 1], wql[(j + 1) & 1]);
            if (f + 2 < NF) geo_read(f + 2, (f + 2) % 3);
            if (f + 1 < NF) corner_read((f + 1) % 3, (f + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            u32x4 ah, al;
            blend(f % 3, f & 1, ah, al);
#pragma unroll
            for (int n = 0; n < FN; ++n) {
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lh8_t, ah), __builtin_bit_cast(lh8_t, wqh[j & 1][n]), acc[i][n], 0, 0, 0);
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lh8_t, ah), __builtin_bit_cast(lh8_t, wql[j & 1][n]), acc[i][n], 0, 0, 0);
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lh8_t, al), __builtin_bit_cast(lh8_t, wqh[j & 1][n]), acc[i][n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int n = 0; n < FN; ++n) { wqh[0][n] = wqh[1][n]; wql[0][n] = wql[1][n]; }
        __syncthreads();                                      // every wave is done with this slice's patch
        if (sl + 1 < g.nslice) {
            patch_load(sl + 1);
            __syncthreads();
        }
    }

    float sc[FN], sh[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        sc[j] = ep.scale ? ep.scale[j * 16 + xl] : 1.f;
        sh[j] = ep.shift ? ep.shift[j * 16 + xl] : 0.f;
    }
    // ---- far pass (exact global gather, same arithmetic): wave-uniform masks from phase G
    {
        unsigned long long any = 0;
#pragma unroll
        for (int tt = 0; tt < TPT; ++tt) any |= farm[tt];
        if (any) {
            const u32x4* wfh = wfar + lane;
            const u32x4* wfl = wfar + far_lo + lane;
            for (int tap = 0; tap < 9; ++tap) {
                const int ftt = tap % TPT, ftg = tap / TPT;
                unsigned long long fm_ = 0;
#pragma unroll
                for (int t = 0; t < TPT; ++t) fm_ = (t == ftt) ? farm[t] : fm_;
                fm_ >>= (ftg * FM) * 16;
                if (!(fm_ & 0xffffffffull)) continue;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    if (!((fm_ >> (16 * i)) & 0xffffull)) continue;
                    const f32x4 ge = *reinterpret_cast<const f32x4*>(smem + GT_OFF + (tap * NPIX + (wv * FM + i) * 16 + xl) * 16);
                    const uint32_t gbase = __float_as_uint(ge[3]);
                    const bool isfar = ge[2] < 0.f;
                    const int h0 = (int)(gbase & 0xffffu) - 32, w0 = (int)((gbase >> 16) & 0x7fffu) - 32;
                    const float m_ = isfar ? -ge[2] : 0.f, lh = ge[0], lw = ge[1], hh = 1.f - lh, hw_ = 1.f - lw;
                    const float wq4[4] = {hh * hw_ * m_, hh * lw * m_, lh * hw_ * m_, lh * lw * m_};
                    for (int ks = 0; ks < g.cpt_far; ++ks) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int hc = h0 + (q >> 1), wc = w0 + (q & 1);
                            const bool ok = isfar && hc >= 0 && hc < g.H && wc >= 0 && wc < g.W;
                            if (ok) {
                                const float* p = xb + ((size_t)hc * g.W + wc) * g.C + ks * 32 + kq * 8;
                                const f32x4 a = *reinterpret_cast<const f32x4*>(p), c4 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { o[e] = fmaf(wq4[q], a[e], o[e]); o[4 + e] = fmaf(wq4[q], c4[e], o[4 + e]); }
                            }
                        }
                        const float o0[4] = {o[0], o[1], o[2], o[3]}, o1[4] = {o[4], o[5], o[6], o[7]};
                        const u32x4 s0 = lds_operand<f32s_t>(ElemTraits<float>::pack(o0)), s1 = lds_operand<f32s_t>(ElemTraits<float>::pack(o1));
                        const u32x4 ah = {s0.x, s0.y, s1.x, s1.y}, al = {s0.z, s0.w, s1.z, s1.w};
#pragma unroll
                        for (int n = 0; n < FN; ++n) {
                            const size_t wi = ((size_t)n * g.fsteps_far + tap * g.cpt_far + ks) * 64;
                            const u32x4 wh = wfh[wi], wl = wfl[wi];
                            acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lh8_t, ah), __builtin_bit_cast(lh8_t, wh), acc[i][n], 0, 0, 0);
                            acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lh8_t, ah), __builtin_bit_cast(lh8_t, wl), acc[i][n], 0, 0, 0);
                            acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lh8_t, al), __builtin_bit_cast(lh8_t, wh), acc[i][n], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue: per-wave staging in the (dead) patch memory, fp32 rows out
    constexpr int LDS_ = FN * 16 + 4;
    float* stage = reinterpret_cast<float*>(smem) + wv * (16 * LDS_);
    float* y = reinterpret_cast<float*>(ep.y);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(kq * 4 + r) * LDS_ + j * 16 + xl] = acc[i][j][r] * sc[j] + sh[j];
        __builtin_amdgcn_wave_barrier();
        const int yg = ty0 + wv * FM + i;
        for (int it = lane; it < 16 * 16; it += 64) {                        // 16 pixels x 16 chunks of 4 channels
            const int px = it >> 4, ng = it & 15;
            const int gx = tx0 + px, gn = ng * 4;
            if (yg < g.H && gx < g.W && gn < ep.Cout) {
                float v[4];
                const f32x4 t = *reinterpret_cast<const f32x4*>(stage + px * LDS_ + gn);
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
                apply_act_chunk<4>(v, ep.act, gn);
                *reinterpret_cast<f32x4*>(y + ((size_t)(b * g.H + yg) * g.W + gx) * ep.ldy + gn) = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int g_opt_dcn_lds = 1;       // option "dcn_lds": 0 = off, 1 = automatic (64 -> 64 on large 16-bit maps), 2 = wherever the kernel applies
int g_opt_dcn_lds_rows = 16; // option "dcn_lds_rows": tile rows, 16 | 8 (three workgroups per CU; measured slower: 2.570 vs 2.551 ms per step, profiles/r06_dcn_lds.md)

static bool dcn_lds_shape_ok(const mfx_dcn_desc* d) {
    if (!d->w_pair_f16 || !d->w_frag_f16 || (d->dtype != MFX_BF16 && d->dtype != MFX_F16)) return false;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1 || d->Ho != d->H || d->Wo != d->W || d->nonsquare) return false;
    if (d->C % 32 != 0 || d->K_pad != 9 * d->C || d->Cout_pad != 64) return false;
    return true;
}
static bool dcn_lds_auto(const mfx_dcn_desc* d) {
    if (!g_opt_dcn_lds || !dcn_lds_shape_ok(d)) return false;
    if (g_opt_dcn_lds >= 2) return true;
    return d->C == 64 && (long)d->B * d->H * d->W >= 65536;
}
extern int g_opt_dcn_fuse_off;
bool dcn_lds_fuses_offset_conv(const mfx_dcn_desc* d) {
    return g_opt_dcn_fuse_off && d->off_w_frag_f16 && d->off_shift && d->C == 64 && dcn_lds_auto(d);
}

template <int TR, typename TX, bool OF> static int launch_dcn_lds(const mfx_dcn_desc* d, hipStream_t st) {
    constexpr int R = 7;
    DcnLGeom g;
    g.B = d->B; g.H = d->H; g.W = d->W; g.C = d->C;
    g.tiles_x = (d->W + 15) / 16; g.tiles_y = (d->H + TR - 1) / TR;
    g.nslice = d->C / 16; g.fsteps_pair = g.nslice * 5; g.fsteps_far = d->K_pad / 32; g.cpt_far = d->C / 32;
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = nullptr; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = 0;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = 1;
    const int tiles = d->B * g.tiles_y * g.tiles_x;
    constexpr int smem = DcnLSmem<R, TR>::bytes;
    static bool attr_done = false;
    if (!attr_done) {
        MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcn_lds_kernel<R, TR, TX, OF>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    DcnLOffArgs oa;
    oa.wfm = reinterpret_cast<const u32x4*>(d->off_w_frag_f16); oa.shift = d->off_shift; oa.om_out = d->offmask_out;
    if (!OF && !d->offmask) return mfx_fail(MFX_ERR_ARG, "dcn: offmask is NULL and this kernel does not compute the offsets itself");
    hipLaunchKernelGGL((dcn_lds_kernel<R, TR, TX, OF>), dim3(tiles), dim3(256), smem, st, reinterpret_cast<const TX*>(d->x), d->offmask,
                       reinterpret_cast<const u32x4*>(d->w_pair_f16), reinterpret_cast<const u32x4*>(d->w_frag_f16), g, ep, oa);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

static int launch_dcn_lds_split(const mfx_dcn_desc* d, hipStream_t st) {
    constexpr int R = 7, TR = 8;
    DcnLGeom g;
    g.B = d->B; g.H = d->H; g.W = d->W; g.C = d->C;
    g.tiles_x = (d->W + 15) / 16; g.tiles_y = (d->H + TR - 1) / TR;
    g.nslice = d->C / 16; g.fsteps_pair = g.nslice * 5; g.fsteps_far = d->K_pad / 32; g.cpt_far = d->C / 32;
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = nullptr; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = 0;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = 1;
    constexpr int smem = (16 + 2 * (R + 1)) * (TR + 2 * (R + 1)) * 80 + (9 * TR * 16 + 32) * 16;
    static bool attr_done = false;
    if (!attr_done) {
        MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcn_lds_split_kernel<R>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    if (!d->offmask) return mfx_fail(MFX_ERR_ARG, "dcn: offmask is NULL (the split-precision LDS kernel does not compute the offsets itself)");
    const size_t pair_lo = (size_t)(d->Cout_pad / 16) * g.fsteps_pair * 64, far_lo = (size_t)(d->Cout_pad / 16) * g.fsteps_far * 64;   // 16-byte units to the lo arrays
    hipLaunchKernelGGL((dcn_lds_split_kernel<R>), dim3(d->B * g.tiles_y * g.tiles_x), dim3(256), smem, st, reinterpret_cast<const float*>(d->x), d->offmask,
                       reinterpret_cast<const u32x4*>(d->w_pair_f16), reinterpret_cast<const u32x4*>(d->w_frag_f16), pair_lo, far_lo, g, ep);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// returns 1 if handled, 0 to fall through to the older kernels, < 0 on error
int try_dcn_lds(const mfx_dcn_desc* d, hipStream_t st) {
    if (d->dtype == MFX_F16X2) {                              // split precision: fp32 maps, [hi | lo] weight arrays (see dcn_lds_split_kernel)
        if (!g_opt_dcn_lds || !d->w_pair_f16 || !d->w_frag_f16 || d->nonsquare) return 0;
        if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1 || d->Ho != d->H || d->Wo != d->W) return 0;
        if (d->C % 32 != 0 || d->K_pad != 9 * d->C || d->Cout_pad != 64 || d->Cout % 4 != 0) return 0;
        if (g_opt_dcn_lds < 2 && !(d->C == 64 && (long)d->B * d->H * d->W >= 65536)) return 0;
        const int rc = launch_dcn_lds_split(d, st);
        return rc == MFX_OK ? 1 : rc;
    }
    if (!dcn_lds_auto(d)) return 0;
    const bool of = dcn_lds_fuses_offset_conv(d);
    int rc;
    if (g_opt_dcn_lds_rows == 16) {
        if (d->dtype == MFX_F16) rc = of ? launch_dcn_lds<16, half_t, true>(d, st) : launch_dcn_lds<16, half_t, false>(d, st);
        else rc = of ? launch_dcn_lds<16, bf16_t, true>(d, st) : launch_dcn_lds<16, bf16_t, false>(d, st);
    } else {
        if (d->dtype == MFX_F16) rc = of ? launch_dcn_lds<8, half_t, true>(d, st) : launch_dcn_lds<8, half_t, false>(d, st);
        else rc = of ? launch_dcn_lds<8, bf16_t, true>(d, st) : launch_dcn_lds<8, bf16_t, false>(d, st);
    }
    return rc == MFX_OK ? 1 : rc;
}

}  // namespace mfx

MFX_RANGE_FLAG_ACCESSOR(dcn_lds)      // split-precision range sentinel of this translation unit (common.h)
