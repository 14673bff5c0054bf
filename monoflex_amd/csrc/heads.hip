// Fused detection heads (reference model/head/detector_predictor.py:47-96,125-134):
// nine branches of  conv3x3 64->256 (no bias) -> BN + leaky_relu(0.01) -> conv1x1 256->c_k (+bias).
//
// One workgroup (4 waves) owns an 8 x 16 block of output pixels of one image and runs ALL branches:
//   * the 10 x 18 x 64-channel input halo patch is staged in LDS once (pixel stride 64*sizeof(T)+16 bytes:
//     conflict-free ds_read_b128) and reused by the 9 branches x 9 taps x 4 waves;
//   * wave wn owns trunk channels [64wn, 64wn+64) of the current branch for all 128 pixels.  GEMM1 runs
//     "transposed" -- weights are the MFMA A operand, pixels the B operand -- so a lane ends up holding 4
//     consecutive trunk channels of one pixel;  weights are pre-packed fragment-major (each wave-load is one
//     contiguous KiB) and streamed L2 -> registers through a 3-deep ring: no LDS, no barrier in the K loop;
//   * BN + leaky are applied in registers and the accumulators are fed STRAIGHT into GEMM2 as its B operand
//     (the K order inside a k-block is permuted identically in the pre-packed 1x1 weights), so the nine
//     256-channel trunk maps (71 M elements per image in the reference) touch neither HBM nor LDS;
//   * the four waves' partial 1x1 sums (over their 64 trunk channels each) are written to per-wave LDS slices
//     with conflict-free ds_write_b128 and summed by all waves (each takes 32 pixels): two barriers per branch,
//     no atomics (ds_add_f32 with this access pattern kept the LDS 72 % busy, profiles/r01_*).
// Only 3 + 50 fp32 channels per pixel are written.
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "igemm.h"
#include <type_traits>

int g_opt_heads_persist = 1;   // option "heads_persist": 1 = one workgroup per resident slot (n > 1: n workgroups), each a contiguous range of (tile, branch)
                               // units; 0 = one workgroup per tile.  B=8 bf16: 516 -> 503 us (tools/probes/heads_probe.py), bit-identical output
int g_opt_heads_mfma32 = 0;    // option "heads_mfma32": 1 = the v_mfma_f32_32x32x16 form of the kernel where the caller supplies its packs (mfx_heads_desc.w1_32 / w2_32)
int g_opt_heads_dbg = 0;       // option "heads_dbg": timing probes of the bf16 kernel (see DBG below); results are wrong -- compiled in with -DMFX_PROBES only

namespace mfx {

constexpr int kHeadC = 64, kHeadTrunk = 256, kHeadRows = 8, kHeadWaves = 4, kHeadFN = 4;

struct HeadGeom {
    int B, H, W, tiles_x, tiles_y, nbranch, ld_out, planar_c, steps, persist;
    float* planar;
};
struct HeadTabs { int ch_off[16]; int c_out[16]; float w2s[16]; };

// Halo patch in LDS: FOUR PLANES, one per MFMA k-group (lane >> 4).  A lane of k-group q only ever reads the 16-byte channel
// chunks q, q+4, (q+8, q+12) of a pixel, so plane q holds exactly those, pixels PS = (chunks + 1) * 16 bytes apart (48 / 80:
// an odd number of 16-byte slots, so 16 consecutive pixels cover all 64 banks once) and the planes a multiple of 256 bytes
// apart.  ds_read_b128 serves the lanes in groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: eight lanes of one k-group and
// eight of the next -- with every k-group's chunks inside ONE 144-byte pixel record (the first layout) the two halves of a
// group collided on 7 of 8 slots (PMC: bank conflicts on 48 % of the LDS cycles of this kernel); with planes they read the
// same slots as 16 consecutive pixels of one plane would.  PL = false keeps the single-record layout (option "heads_planes" = 0).
template <typename T, bool PL> struct HeadSmem {
    static constexpr int CPL = kHeadC * (int)sizeof(T) / 16 / 4;        // 16-byte chunks of a pixel per plane: 2 (bf16) / 4 (f32)
    static constexpr int PS = PL ? (CPL + 1) * 16 : kHeadC * (int)sizeof(T) + 16;    // pixel stride (inside a plane / of the single record)
    static constexpr int PLANE = PL ? ((kHeadRows + 2) * 18 * PS + 255) / 256 * 256 : 0;
    static constexpr int patch_bytes = PL ? 4 * PLANE : (kHeadRows + 2) * 18 * PS;
    static constexpr int red_ld = 20;                                   // fp32 words per pixel row of a partial-sum slice (16 + pad)
    // tile rows per reduction pass: the 4-byte element types (fp32, split precision) reduce in two passes of 4 rows -- half the slices,
    // 69 KB instead of 90 KB per workgroup, so TWO workgroups fit a CU's 160 KB (one wave per SIMD cannot hide its own latencies)
    static constexpr int red_rows = sizeof(T) == 4 ? 4 : kHeadRows;
    static constexpr int red_bytes = red_rows * 16 * red_ld * 4;        // one wave's slice: [red_rows * 16 px][20]
    static constexpr int bytes = patch_bytes + kHeadWaves * red_bytes;
};

// trunk values of one pixel held by a lane after GEMM1 -> B-operand chunk of GEMM2
template <typename T> struct TrunkPack;
template <> struct TrunkPack<bf16_t> {      // k-block = 32 trunk channels = D fragments (2kb, 2kb+1): 8 values per lane
    static constexpr int KBLK = 2;
    __device__ static __forceinline__ u32x4 make(const float (&t)[kHeadFN][4], int kb) {
        float v[8] = {t[2 * kb][0], t[2 * kb][1], t[2 * kb][2], t[2 * kb][3], t[2 * kb + 1][0], t[2 * kb + 1][1], t[2 * kb + 1][2], t[2 * kb + 1][3]};
        return ElemTraits<bf16_t>::pack(v);
    }
};
template <> struct TrunkPack<half_t> {
    static constexpr int KBLK = 2;
    __device__ static __forceinline__ u32x4 make(const float (&t)[kHeadFN][4], int kb) {
        float v[8] = {t[2 * kb][0], t[2 * kb][1], t[2 * kb][2], t[2 * kb][3], t[2 * kb + 1][0], t[2 * kb + 1][1], t[2 * kb + 1][2], t[2 * kb + 1][3]};
        return ElemTraits<half_t>::pack(v);
    }
};
template <> struct TrunkPack<float> {       // k-block = 16 trunk channels = D fragment kb: 4 values per lane
    static constexpr int KBLK = 4;
    __device__ static __forceinline__ u32x4 make(const float (&t)[kHeadFN][4], int kb) {
        return ElemTraits<float>::pack(t[kb]);
    }
};

template <> struct TrunkPack<f32s_t> {      // as float; the four values become one split-precision operand chunk
    static constexpr int KBLK = 4;
    __device__ static __forceinline__ u32x4 make(const float (&t)[kHeadFN][4], int kb) {
        return lds_operand<f32s_t>(ElemTraits<float>::pack(t[kb]));
    }
};

// w1p: fragment-major 3x3 weights  [branch][wn 4][step][j 4][lane 64][16 B]
// w2p: fragment-major 1x1 weights  [branch][wn 4][kblk][of 2][lane 64][16 B]  (K order matching TrunkPack)
// DBG (timing probes only, results are wrong; option "heads_dbg"): bit 0 = the K loop keeps the first step's weight fragments
// (no L2 -> register weight stream), bit 1 = it keeps the first pixel fragments (no LDS reads).
// split precision (T = f32s_t): the K loop runs over step PAIRS.  Both operands keep their hi halves in dwords 0-1 and their lo halves in
// dwords 2-3 of a chunk, so the hi (lo) halves of two consecutive steps form one 8-element MFMA operand [hi(step 2p) | hi(step 2p+1)]:
// the weights arrive re-packed that way (ops.pair_steps), the pixels by two 8-byte LDS reads.  Three products per pair -- hh.hh, hh.ll,
// ll.hh (lo.lo is below fp32 resolution) -- instead of the four of two single-step mma_chunk<f32s_t> calls: 25 % fewer MFMAs.
__device__ __forceinline__ u32x4 lds_read_pair(const char* p, int second) {
    const uint2 a = *reinterpret_cast<const uint2*>(p), b = *reinterpret_cast<const uint2*>(p + second);
    return u32x4{a.x, a.y, b.x, b.y};
}
__device__ __forceinline__ void mma_split3(const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);
}
// eight fp32 values -> their hi halves / lo halves as two MFMA operands
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hh, u32x4& ll) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x2 h = __builtin_convertvector((f32x2){v[2 * i], v[2 * i + 1]}, f16x2);
        const f16x2 l = __builtin_convertvector((f32x2){v[2 * i] - (float)h[0], v[2 * i + 1] - (float)h[1]}, f16x2);
        hh[i] = __builtin_bit_cast(uint32_t, h); ll[i] = __builtin_bit_cast(uint32_t, l);
    }
}

template <typename T, bool PL, int DBG = 0>
__global__ __launch_bounds__(kHeadWaves * 64, 2) void heads_fused_kernel(const T* __restrict__ x, const u32x4* __restrict__ w1p,
                                                                         const float* __restrict__ scale1, const float* __restrict__ shift1,
                                                                         const u32x4* __restrict__ w2p, const float* __restrict__ bias2,
                                                                         float* __restrict__ out, HeadGeom g, HeadTabs tabs) {
    constexpr int NT = kHeadWaves * 64, FM = kHeadRows, FN = kHeadFN;
    constexpr int ELEMS = ElemTraits<T>::ELEMS;
    constexpr int PS = HeadSmem<T, PL>::PS, PLANE = HeadSmem<T, PL>::PLANE;
    constexpr int KBLK = TrunkPack<T>::KBLK;
    constexpr int RLD = HeadSmem<T, PL>::red_ld;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    float* red = reinterpret_cast<float*>(smem + HeadSmem<T, PL>::patch_bytes);      // [4 waves][red_rows * 16 px][20]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xl = lane & 15, kq = lane >> 4;

    // Work = (tile, branch) units, tile-major.  Workgroup i of n takes the contiguous range [i*U/n, (i+1)*U/n): with one workgroup per
    // tile (n = tiles) that is the tile's nine branches; the persistent launch (g.persist: n = resident workgroups, 2 per CU)
    // hands every workgroup U/n +- 1 units, so the chip drains together instead of running a 3/4-full last round of whole tiles
    // (1920 tiles on 512 slots).  The patch is reloaded whenever the tile changes.
    const int wg = g.persist ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const long long units = (long long)g.tiles_x * g.tiles_y * g.B * g.nbranch;
    const int u0 = (int)(units * wg / gridDim.x), u1 = (int)(units * (wg + 1) / gridDim.x);
    if (u0 >= u1) return;
    const int steps = g.steps;                               // 9 * 64 / (4 * ELEMS): 18 (bf16) / 36 (f32)

    auto wsrc_of = [&](int br) { return w1p + ((size_t)(br * kHeadWaves + wn) * steps) * (FN * 64) + lane; };
    auto wfetch = [&](const u32x4* wsrc, int s, u32x4 (&wf)[FN]) {
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = wsrc[(size_t)(s * FN + j) * 64];
    };

    int cur_tile = -1, b = 0, x0 = 0, y0 = 0;
    for (int u = u0; u < u1; ++u) {
        const int tile_u = u / g.nbranch, br = u - tile_u * g.nbranch;
        if (tile_u != cur_tile) {
            // ---- halo patch (zero outside the image).  Every wave is past the previous unit's reduction barriers, i.e. done reading
            if (cur_tile >= 0) __syncthreads();
            cur_tile = tile_u;
            int ptid = tid;                                  // opaque here: hoisted out of the unit loop, the per-thread chunk offsets of
            asm volatile("" : "+v"(ptid));                   // the patch copy (11 chunks x address pairs) stay live through the K loop and spill
            int tile = tile_u;
            const int tx = tile % g.tiles_x; tile /= g.tiles_x;
            const int ty = tile % g.tiles_y; b = tile / g.tiles_y;
            x0 = tx * 16; y0 = ty * kHeadRows;
            constexpr int CPP = kHeadC * (int)sizeof(T) / 16;
            constexpr int nchunks = (kHeadRows + 2) * 18 * CPP;
            const T* xg = x + (size_t)b * g.H * g.W * kHeadC;
            constexpr int PU = 4;
            for (int base = 0; base < nchunks; base += NT * PU) {
                u32x4 pr[PU];
#pragma unroll
                for (int q = 0; q < PU; ++q) {
                    const int idx = base + q * NT + ptid;
                    const int pix = idx / CPP, ch = idx - pix * CPP;
                    const int py = pix / 18, px = pix - py * 18;
                    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                    u32x4 z = {0u, 0u, 0u, 0u};
                    if (idx < nchunks && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                        z = *reinterpret_cast<const u32x4*>(xg + ((size_t)iy * g.W + ix) * kHeadC + ch * ELEMS);
                    pr[q] = z;
                }
#pragma unroll
                for (int q = 0; q < PU; ++q) {
                    const int idx = base + q * NT + ptid;
                    if (idx < nchunks) *reinterpret_cast<u32x4*>(patch + (PL ? ((idx % CPP) & 3) * PLANE + (idx / CPP) * PS + ((idx % CPP) >> 2) * 16
                                                                           : (idx / CPP) * PS + (idx % CPP) * 16)) = lds_operand<T>(pr[q]);
                }
            }
            __syncthreads();
        }

        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- GEMM1 (transposed): D[j][i] rows = trunk channel 64wn+16j+.., cols = pixel (row i, x = lane&15)
        const u32x4* wsrc = wsrc_of(br);
        auto compute = [&](int s, const u32x4 (&wf)[FN]) {
            const int e = s * (4 * ELEMS) + kq * ELEMS;      // this lane's K chunk -> (tap, channel)
            const int tap = e >> 6, cl = e & 63;
            const int th = (tap * 21846) >> 16, tw = tap - th * 3;
            const char* ap = patch + (PL ? kq * PLANE + (th * 18 + xl + tw) * PS + (cl * (int)sizeof(T) / 64) * 16   // chunk kq + 4*(cl*sizeof/64)
                                         : (th * 18 + xl + tw) * PS + cl * (int)sizeof(T));
            // pixel fragments two rows ahead of the MFMAs that consume them: left to itself the scheduler (at 250+ VGPRs) sinks
            // each ds_read pair right in front of its 8 MFMAs and every pair then exposes a full LDS round trip
            u32x4 pf[2][2];
            pf[0][0] = *reinterpret_cast<const u32x4*>(ap);
            pf[0][1] = *reinterpret_cast<const u32x4*>(ap + 18 * PS);
#pragma unroll
            for (int i = 0; i < FM; i += 2) {
                const int cur = (DBG & 2) ? 0 : (i >> 1) & 1;
                if (i + 2 < FM && !(DBG & 2)) {
                    pf[cur ^ 1][0] = *reinterpret_cast<const u32x4*>(ap + (i + 2) * 18 * PS);
                    pf[cur ^ 1][1] = *reinterpret_cast<const u32x4*>(ap + (i + 3) * 18 * PS);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < FN; ++j) mma_chunk<T>(wf[j], pf[cur][0], acc[i][j]);
#pragma unroll
                for (int j = 0; j < FN; ++j) mma_chunk<T>(wf[j], pf[cur][1], acc[i + 1][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        constexpr bool SPLIT = std::is_same<T, f32s_t>::value;
        u32x4 wb[SPLIT ? 1 : 3][FN];
        const int cn = tabs.c_out[br];
        const bool two = cn > 16;                            // second 16-row output fragment needed?
        f32x4 s4[FN], h4[FN];                                // BN scale / shift of this wave's trunk channels
        u32x4 w2f[KBLK][2];
        auto load_scale = [&]() {
            const int ko = kq * 4;
#pragma unroll
            for (int j = 0; j < FN; ++j) s4[j] = *reinterpret_cast<const f32x4*>(scale1 + br * kHeadTrunk + wn * 64 + j * 16 + ko);
        };
        auto load_shift = [&]() {
            const int ko = kq * 4;
#pragma unroll
            for (int j = 0; j < FN; ++j) h4[j] = *reinterpret_cast<const f32x4*>(shift1 + br * kHeadTrunk + wn * 64 + j * 16 + ko);
        };
        // (split precision: w2f[2*kbp + hl][of] = the hi (hl 0) / lo (hl 1) operand of trunk-channel block pair kbp)
        auto load_w2 = [&](int of) {
            const u32x4* src = w2p + ((size_t)(br * kHeadWaves + wn) * KBLK) * (2 * 64) + lane;
#pragma unroll
            for (int kb = 0; kb < KBLK; ++kb) w2f[kb][of] = src[(size_t)(kb * 2 + of) * 64];
        };
        if constexpr (SPLIT) {
            // weights: [pair][hi | lo][j][lane] (ops.pair_steps); ring of two pairs, fetched one pair ahead
            u32x4 wh[2][FN], wl[2][FN];
            auto wfetch2 = [&](int p, u32x4 (&h)[FN], u32x4 (&l)[FN]) {
#pragma unroll
                for (int j = 0; j < FN; ++j) { h[j] = wsrc[(size_t)((p * 2 + 0) * FN + j) * 64]; l[j] = wsrc[(size_t)((p * 2 + 1) * FN + j) * 64]; }
            };
            auto compute2 = [&](int p, const u32x4 (&h)[FN], const u32x4 (&l)[FN]) {
                const int e = p * 32 + kq * 4;               // first step of the pair: (tap, channel); the second step is 16 channels on
                const int tap = e >> 6, cl = e & 63;
                const int th = (tap * 21846) >> 16, tw = tap - th * 3;
                const char* ap = patch + (th * 18 + xl + tw) * PS + cl * 4;
                u32x4 ph[2], pl[2];
                ph[0] = lds_read_pair(ap, 64); pl[0] = lds_read_pair(ap + 8, 64);
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int cur = i & 1;
                    if (i + 1 < FM) { ph[cur ^ 1] = lds_read_pair(ap + (i + 1) * 18 * PS, 64); pl[cur ^ 1] = lds_read_pair(ap + (i + 1) * 18 * PS + 8, 64); }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, h[j]), __builtin_bit_cast(f16x8, ph[cur]), acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, h[j]), __builtin_bit_cast(f16x8, pl[cur]), acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, l[j]), __builtin_bit_cast(f16x8, ph[cur]), acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            const int npairs = steps / 2;                    // 18
            wfetch2(0, wh[0], wl[0]);
            for (int p = 0; p < npairs; p += 2) {
                if (p + 1 < npairs) wfetch2(p + 1, wh[1], wl[1]);
                compute2(p, wh[0], wl[0]);
                if (p + 2 < npairs) wfetch2(p + 2, wh[0], wl[0]);
                if (p + 1 < npairs) compute2(p + 1, wh[1], wl[1]);
            }
        } else {
        wfetch(wsrc, 0, wb[0]);
        wfetch(wsrc, 1, wb[1]);
        if (DBG & 1) wfetch(wsrc, 2, wb[2]);
        for (int s = 0; s < steps; s += 3) {                 // steps is a multiple of 3 (18 / 36)
            if (!(DBG & 1)) wfetch(wsrc, s + 2, wb[2]);
            compute(s, wb[0]);
            if (!(DBG & 1) && s + 3 < steps) wfetch(wsrc, s + 3, wb[0]);
            compute(s + 1, wb[1]);
            if (!(DBG & 1) && s + 4 < steps) wfetch(wsrc, s + 4, wb[1]);
            compute(s + 2, wb[2]);
        }
        }
        // (fetching these between the last steps' MFMA blocks -- their ring slots are free by then -- was built and measured: the
        // extra live registers spill, the scratch traffic shares vmcnt with the loads, 516 -> 542 us.  Not kept.)
        load_scale();
        load_shift();
        load_w2(0);
        load_w2(1);

        // ---- BN + leaky in registers; GEMM2 straight from the accumulators
        const int co = tabs.ch_off[br];
        const float w2s = tabs.w2s[br];
        float* mine;
        // partial 1x1 outputs of this wave: po[i][of] = D2[o = 16*of + 4*kq + r][pixel (row i, x = xl)]
        f32x4 po[FM][2];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            float t[FN][4];
            // BN + LeakyReLU(0.01) on value PAIRS: v_pk_fma_f32, v_pk_mul_f32 and one v_max per value (max(v, 0.01 v) IS the leaky ReLU for
            // every finite v) -- two VALU instructions per value instead of four (fma, mul, compare, select): 128 values per unit and wave
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 v = f32x2{acc[i][j][r], acc[i][j][r + 1]} * f32x2{s4[j][r], s4[j][r + 1]} + f32x2{h4[j][r], h4[j][r + 1]};
                    const f32x2 u = v * f32x2{0.01f, 0.01f};
                    t[j][r] = fmaxf(v[0], u[0]); t[j][r + 1] = fmaxf(v[1], u[1]);
                }
            po[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; po[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) {
#pragma unroll
                for (int kbp = 0; kbp < KBLK / 2; ++kbp) {
                    const float v[8] = {t[2 * kbp][0], t[2 * kbp][1], t[2 * kbp][2], t[2 * kbp][3], t[2 * kbp + 1][0], t[2 * kbp + 1][1], t[2 * kbp + 1][2], t[2 * kbp + 1][3]};
                    u32x4 th, tl;
                    split8(v, th, tl);
                    mma_split3(w2f[2 * kbp][0], w2f[2 * kbp + 1][0], th, tl, po[i][0]);
                    if (two) mma_split3(w2f[2 * kbp][1], w2f[2 * kbp + 1][1], th, tl, po[i][1]);
                }
            } else {
#pragma unroll
            for (int kb = 0; kb < KBLK; ++kb) {
                const u32x4 tb = TrunkPack<T>::make(t, kb);
                mma_chunk<T>(w2f[kb][0], tb, po[i][0]);
                if (two) mma_chunk<T>(w2f[kb][1], tb, po[i][1]);
            }
            }
        }
        constexpr int RR = HeadSmem<T, PL>::red_rows, SLICE = RR * 16 * RLD;      // rows per reduction pass, floats per wave slice
        constexpr int PPW = RR * 16 / kHeadWaves, NIT = PPW * 4 / 64;             // pixels summed per wave per pass, items per lane
        mine = red + wn * SLICE;
        for (int of = 0; of < (two ? 2 : 1); ++of) {         // 16 output channels per pass
#pragma unroll
            for (int half = 0; half < FM / RR; ++half) {
                __syncthreads();                             // slices free (previous pass / branch fully summed)
#pragma unroll
                for (int i = 0; i < RR; ++i) *reinterpret_cast<f32x4*>(mine + (i * 16 + xl) * RLD + kq * 4) = po[half * RR + i][of];
                __syncthreads();
                // wave wn sums pixels [PPW*wn, PPW*wn + PPW) of the pass: lane -> (pixel, group of 4 outputs)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int item = it * 64 + lane;
                    const int px = wn * PPW + (item >> 2), og = item & 3;
                    f32x4 sum = *reinterpret_cast<const f32x4*>(red + px * RLD + og * 4);
#pragma unroll
                    for (int w = 1; w < kHeadWaves; ++w) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(red + w * SLICE + px * RLD + og * 4);
                        sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
                    }
                    const int tp = half * RR * 16 + px;      // pixel of the 8 x 16 tile
                    const int oy = y0 + (tp >> 4), ox = x0 + (tp & 15);
                    if (oy < g.H && ox < g.W) {
                        const size_t m = ((size_t)b * g.H + oy) * g.W + ox;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int o = of * 16 + og * 4 + r;
                            if (o < cn) {
                                const float res = sum[r] * w2s + bias2[br * 32 + o];
                                out[m * g.ld_out + co + o] = res;
                                if (br == 0 && g.planar && o < g.planar_c)
                                    g.planar[((size_t)b * g.planar_c + o) * g.H * g.W + (size_t)oy * g.W + ox] = res;
                            }
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The same kernel on v_mfma_f32_32x32x16 (r06; 16-bit element types).  Per wave the tile is still 64 trunk channels x 128 pixels of
// one branch: 2 row blocks of 32 channels x 4 column blocks of 32 pixels (two adjacent tile rows), 8 MFMAs of 32 cycles per 16 K (the 16x16x32 form
// issues 32 MFMAs of 16 cycles per 32 K: the same operand bytes per FLOP -- 4 weight fragments and 8 pixel fragments per 32 K either way, the tile
// decides that, not the instruction -- but half the instruction issues, and the microarchitecture guide measures the 32x32 shape at its exact 32-cycle
// cadence where the 16x16 one needs ~17).  D layout: lane (c = lane & 31 pixel, h = lane >> 5) holds rows 8 (r >> 2) + 4 h + (r & 3): sixteen trunk
// channels of one pixel per row block, whose registers 8t .. 8t + 7 ARE the B operand of GEMM2's K-step t (W2 is packed in that K order), so the
// trunk still never leaves the registers.  w1p32: [branch][wn 4][K-step 36][rb 2][lane 64][16 B], lane (row, h) = weights of channel 64 wn + 32 rb + row,
// k = 16 s + 8 h ..; w2p32: [branch][wn][rb 2][t 2][lane 64][16 B], lane (o, h) = W2[o][64 wn + 32 rb + 16 t + 8 (e >> 2) + 4 h + (e & 3)].
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <typename T> __device__ __forceinline__ f32x16 mma32(const u32x4& a, const u32x4& b, const f32x16& c);
template <> __device__ __forceinline__ f32x16 mma32<bf16_t>(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mma32<half_t>(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <typename T>
__global__ __launch_bounds__(kHeadWaves * 64, 2) void heads_fused32_kernel(const T* __restrict__ x, const u32x4* __restrict__ w1p, const float* __restrict__ scale1,
                                                                          const float* __restrict__ shift1, const u32x4* __restrict__ w2p,
                                                                          const float* __restrict__ bias2, float* __restrict__ out, HeadGeom g, HeadTabs tabs) {
    constexpr int NT = kHeadWaves * 64;
    constexpr int PS = HeadSmem<T, false>::PS, RLD = HeadSmem<T, false>::red_ld;
    constexpr int NB = 4, RB = 2, KS = 36;                  // pixel blocks (32 px), row blocks (32 channels) per wave, K-steps of 16
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;
    float* red = reinterpret_cast<float*>(smem + HeadSmem<T, false>::patch_bytes);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;

    const int wg = g.persist ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const long long units = (long long)g.tiles_x * g.tiles_y * g.B * g.nbranch;
    const int u0 = (int)(units * wg / gridDim.x), u1 = (int)(units * (wg + 1) / gridDim.x);
    if (u0 >= u1) return;

    int cur_tile = -1, b = 0, x0 = 0, y0 = 0;
    // lane part of a pixel-fragment address: pixel (row c >> 4 of the block's two, column c & 15), channel chunk h of the K-step
    const char* lb = patch + ((c >> 4) * 18 + (c & 15)) * PS + h * 16;
    for (int u = u0; u < u1; ++u) {
        const int tile_u = u / g.nbranch, br = u - tile_u * g.nbranch;
        if (tile_u != cur_tile) {
            if (cur_tile >= 0) __syncthreads();
            cur_tile = tile_u;
            int ptid = tid;
            asm volatile("" : "+v"(ptid));
            int tile = tile_u;
            const int tx = tile % g.tiles_x; tile /= g.tiles_x;
            const int ty = tile % g.tiles_y; b = tile / g.tiles_y;
            x0 = tx * 16; y0 = ty * kHeadRows;
            constexpr int CPP = kHeadC * (int)sizeof(T) / 16;
            constexpr int nchunks = (kHeadRows + 2) * 18 * CPP;
            const T* xg = x + (size_t)b * g.H * g.W * kHeadC;
            constexpr int PU = 4;
            for (int base = 0; base < nchunks; base += NT * PU) {
                u32x4 pr[PU];
#pragma unroll
                for (int q = 0; q < PU; ++q) {
                    const int idx = base + q * NT + ptid;
                    const int pix = idx / CPP, ch = idx - pix * CPP;
                    const int py = pix / 18, px = pix - py * 18;
                    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                    u32x4 z = {0u, 0u, 0u, 0u};
                    if (idx < nchunks && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                        z = *reinterpret_cast<const u32x4*>(xg + ((size_t)iy * g.W + ix) * kHeadC + ch * 8);
                    pr[q] = z;
                }
#pragma unroll
                for (int q = 0; q < PU; ++q) {
                    const int idx = base + q * NT + ptid;
                    if (idx < nchunks) *reinterpret_cast<u32x4*>(patch + (idx / CPP) * PS + (idx % CPP) * 16) = pr[q];
                }
            }
            __syncthreads();
        }

        f32x16 acc[NB][RB];
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // ---- GEMM1: D[rb][blk] rows = trunk channels 64 wn + 32 rb + .., cols = the block's 32 pixels.  Fully unrolled: every LDS offset is an immediate.
        const u32x4* wsrc = w1p + ((size_t)(br * kHeadWaves + wn) * KS) * (RB * 64) + lane;
        constexpr int WR = 6;                                 // weight ring: K-steps in flight (2 fragments each): 5 ahead x 256 MFMA cycles (36 % 6 == 0)
        u32x4 wb[WR][RB];
#pragma unroll
        for (int s = 0; s < WR - 1; ++s)
#pragma unroll
            for (int j = 0; j < RB; ++j) wb[s][j] = wsrc[(size_t)(s * RB + j) * 64];
        u32x4 pf[2][NB];
        auto pread = [&](int s, u32x4 (&dst)[NB]) {
            const int tap = s >> 2, th = tap / 3, tw = tap - th * 3;
#pragma unroll
            for (int i = 0; i < NB; ++i) dst[i] = *reinterpret_cast<const u32x4*>(lb + ((2 * i + th) * 18 + tw) * PS + 32 * (s & 3));
        };
        pread(0, pf[0]);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + WR - 1 < KS) {
#pragma unroll
                for (int j = 0; j < RB; ++j) wb[(s + WR - 1) % WR][j] = wsrc[(size_t)((s + WR - 1) * RB + j) * 64];
            }
            if (s + 1 < KS) pread(s + 1, pf[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < RB; ++j) acc[i][j] = mma32<T>(wb[s % WR][j], pf[s & 1][i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- BN + leaky in registers; GEMM2 straight from the accumulators
        const int cn = tabs.c_out[br];
        const bool two = cn > 16;
        u32x4 w2f[RB][2];
        {
            const u32x4* src = w2p + ((size_t)(br * kHeadWaves + wn) * (RB * 2)) * 64 + lane;
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) w2f[j][t] = src[(size_t)(j * 2 + t) * 64];
        }
        const int co = tabs.ch_off[br];
        const float w2s = tabs.w2s[br];
        f32x16 po[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) po[i][r] = 0.f;
        // row block by row block: the block's 16 scale / shift values per lane live only while its accumulators are consumed (all 32 at once, next to
        // 128 accumulator and 64 output registers, spilled)
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            f32x4 s4[4], h4[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int ch = br * kHeadTrunk + wn * 64 + j * 32 + gq * 8 + h * 4;
                s4[gq] = *reinterpret_cast<const f32x4*>(scale1 + ch);
                h4[gq] = *reinterpret_cast<const f32x4*>(shift1 + ch);
            }
            // (t outside i: consecutive MFMAs accumulate into four different po[] -- as `i, t` the two K-steps of a block were back-to-back dependent
            // 32x32x16 MFMAs, each waiting out the other's 16 passes)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    float v8[8];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const int r = 8 * t + e, gq = r >> 2, q = r & 3;
                        const f32x2 v = f32x2{acc[i][j][r], acc[i][j][r + 1]} * f32x2{s4[gq][q], s4[gq][q + 1]} + f32x2{h4[gq][q], h4[gq][q + 1]};
                        const f32x2 w_ = v * f32x2{0.01f, 0.01f};
                        v8[e] = fmaxf(v[0], w_[0]); v8[e + 1] = fmaxf(v[1], w_[1]);
                    }
                    po[i] = mma32<T>(w2f[j][t], ElemTraits<T>::pack(v8), po[i]);
                }
        }
        // ---- the four waves' partial 1x1 sums meet in LDS (16 outputs per pass); lane (c, h) holds outputs 8 (r >> 2) + 4 h + (r & 3) of pixel 32 blk + c
        constexpr int SLICE = kHeadRows * 16 * RLD;
        constexpr int PPW = kHeadRows * 16 / kHeadWaves, NIT = PPW * 4 / 64;
        float* mine = red + wn * SLICE;
        for (int of = 0; of < (two ? 2 : 1); ++of) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const f32x16& pv = po[i];
                    f32x4 v4;
                    if (of == 0) v4 = gg == 0 ? f32x4{pv[0], pv[1], pv[2], pv[3]} : f32x4{pv[4], pv[5], pv[6], pv[7]};
                    else v4 = gg == 0 ? f32x4{pv[8], pv[9], pv[10], pv[11]} : f32x4{pv[12], pv[13], pv[14], pv[15]};
                    *reinterpret_cast<f32x4*>(mine + (i * 32 + c) * RLD + gg * 8 + h * 4) = v4;
                }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int item = it * 64 + lane;
                const int px = wn * PPW + (item >> 2), og = item & 3;
                f32x4 sum = *reinterpret_cast<const f32x4*>(red + px * RLD + og * 4);
#pragma unroll
                for (int w = 1; w < kHeadWaves; ++w) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(red + w * SLICE + px * RLD + og * 4);
                    sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
                }
                const int oy = y0 + (px >> 4), ox = x0 + (px & 15);
                if (oy < g.H && ox < g.W) {
                    const size_t m = ((size_t)b * g.H + oy) * g.W + ox;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = of * 16 + og * 4 + r;
                        if (o < cn) {
                            const float res = sum[r] * w2s + bias2[br * 32 + o];
                            out[m * g.ld_out + co + o] = res;
                            if (br == 0 && g.planar && o < g.planar_c)
                                g.planar[((size_t)b * g.planar_c + o) * g.H * g.W + (size_t)oy * g.W + ox] = res;
                        }
                    }
                }
            }
        }
    }
}

template <typename T, bool PL, int DBG = 0> static int launch_heads(const mfx_heads_desc* d, hipStream_t st) {
    HeadGeom g;
    g.B = d->B; g.H = d->H; g.W = d->W; g.tiles_x = (d->W + 15) / 16; g.tiles_y = (d->H + kHeadRows - 1) / kHeadRows;
    g.nbranch = d->nbranch; g.ld_out = d->ld_out; g.planar = d->planar; g.planar_c = d->planar_c;
    g.steps = 9 * kHeadC / (4 * ElemTraits<T>::ELEMS);
    g.persist = 0;
    HeadTabs t;
    for (int i = 0; i < 16; ++i) { t.ch_off[i] = d->ch_off[i]; t.c_out[i] = d->c_out[i]; t.w2s[i] = d->w2_scale[i] != 0.f ? d->w2_scale[i] : 1.f; }
    const int tiles = g.tiles_x * g.tiles_y * d->B;
    int grid = tiles;
    if (g_opt_heads_persist) {                               // two workgroups fit a CU (LDS, 2 waves per SIMD): one unit range per slot
        static int slots = 0;
        if (!slots) {
            int dev = 0, cus = 0;
            MFX_HIP_CHECK(hipGetDevice(&dev));
            MFX_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            slots = cus * (2 * HeadSmem<T, PL>::bytes <= 160 * 1024 ? 2 : 1);
        }
        const int want = g_opt_heads_persist > 1 ? g_opt_heads_persist : slots;      // (> 1: that many workgroups -- tests)
        if (tiles > want) { grid = want; g.persist = 1; }
    }
    constexpr int smem = HeadSmem<T, PL>::bytes;
    if constexpr (!PL && DBG == 0 && (std::is_same<T, bf16_t>::value || std::is_same<T, half_t>::value)) {
        if (d->w1_32 && d->w2_32 && g_opt_heads_mfma32) {      // the 32x32x16 form (needs its own weight packs)
            auto k32 = heads_fused32_kernel<T>;
            static bool attr32 = false;
            if (!attr32 && smem > 64 * 1024) { MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k32), hipFuncAttributeMaxDynamicSharedMemorySize, smem)); attr32 = true; }
            hipLaunchKernelGGL(k32, dim3(grid), dim3(kHeadWaves * 64), smem, st, reinterpret_cast<const T*>(d->x),
                               reinterpret_cast<const u32x4*>(d->w1_32), d->scale1, d->shift1, reinterpret_cast<const u32x4*>(d->w2_32), d->bias2,
                               d->out, g, t);
            MFX_HIP_CHECK(hipGetLastError());
            return MFX_OK;
        }
    }
    auto k = heads_fused_kernel<T, PL, DBG>;
    static bool attr_set = false;
    if (!attr_set && smem > 64 * 1024) { MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem)); attr_set = true; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(kHeadWaves * 64), smem, st, reinterpret_cast<const T*>(d->x),
                       reinterpret_cast<const u32x4*>(d->w1), d->scale1, d->shift1, reinterpret_cast<const u32x4*>(d->w2), d->bias2,
                       d->out, g, t);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

}  // namespace mfx
using namespace mfx;

int g_opt_heads_planes = 0;    // option "heads_planes": 1 = four k-group planes (no LDS bank conflicts: 48 % -> 11 % of the LDS cycles, LDS-active
                               // cycles -42 %), 0 = one 144-byte record per pixel.  Same kernel time (555 vs 555 us, A/B in one run): not LDS-bound

extern "C" int mfx_heads_fused(const mfx_heads_desc* d, void* stream) {
    if (!d || !d->x || !d->w1 || !d->scale1 || !d->shift1 || !d->w2 || !d->bias2 || !d->out)
        return mfx_fail(MFX_ERR_ARG, "heads_fused: null pointer");
    if (d->nbranch < 1 || d->nbranch > 16) return mfx_fail(MFX_ERR_ARG, "heads_fused: 1..16 branches");
    if (d->K_pad != 9 * kHeadC) return mfx_fail(MFX_ERR_ARG, "heads_fused: K_pad must be 576 (fragment-major packing)");
    for (int i = 0; i < d->nbranch; ++i)
        if (d->c_out[i] < 1 || d->c_out[i] > 32 || d->ch_off[i] < 0 || d->ch_off[i] + d->c_out[i] > d->ld_out)
            return mfx_fail(MFX_ERR_ARG, "heads_fused: branch output channels out of range");
    if (d->B * d->H * d->W == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MFX_F32) return g_opt_heads_planes ? launch_heads<float, true>(d, st) : launch_heads<float, false>(d, st);
#ifdef MFX_PROBES      /* timing probes with WRONG results (tools/probes/heads_probe.py): only in a probe build (MFX_PROBES=1 python -m monoflex_amd.build) */
    if (d->dtype == MFX_BF16 && g_opt_heads_dbg == 1) return launch_heads<bf16_t, false, 1>(d, st);
    if (d->dtype == MFX_BF16 && g_opt_heads_dbg == 2) return launch_heads<bf16_t, false, 2>(d, st);
    if (d->dtype == MFX_BF16 && g_opt_heads_dbg == 3) return launch_heads<bf16_t, false, 3>(d, st);
#endif
    if (d->dtype == MFX_BF16) return g_opt_heads_planes ? launch_heads<bf16_t, true>(d, st) : launch_heads<bf16_t, false>(d, st);
    if (d->dtype == MFX_F16) return launch_heads<half_t, false>(d, st);
    if (d->dtype == MFX_F16X2) return launch_heads<f32s_t, false>(d, st);
    return mfx_fail(MFX_ERR_ARG, "heads_fused: bad dtype");
}

MFX_RANGE_FLAG_ACCESSOR(heads)      // split-precision range sentinel of this translation unit (common.h)
