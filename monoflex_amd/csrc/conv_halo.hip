// 3x3 / stride-1 / pad-1 convolution: LDS-staged input halo, wave-private output slices, weights
// streamed L2 -> registers.  No workgroup barrier inside the K loop.
//
// Why (measured on MI355X, profiles/r01_*): the generic implicit-GEMM kernel rebuilds its A tile from
// global memory every k-iteration (each input element fetched 9x), pushes both operands through
// ds_write_b128 (~79 B/clk/CU) and synchronises the workgroup every 16 MFMAs -> 10-20 % MFMA utilisation,
// insensitive to tile shape.  Here:
//   * a workgroup = WN waves owns an 8 x 16 block of output pixels of one image; its (8+2) x 18 input halo
//     patch (one channel group) is written to LDS once, all loads in flight together, zero-filled outside
//     the image; pixel stride C*sizeof(T)+16 bytes puts 16 consecutive pixels on 16 distinct 16-byte bank
//     slots, so fragment reads are conflict-free ds_read_b128;
//   * wave wn computes all 128 pixels x its FN*16 output channels (FM = 8 row fragments): every weight
//     fragment (16 n x 64 B of K, loaded straight from L1/L2 into VGPRs, prefetched one step ahead) feeds
//     8 MFMAs, every A fragment FN MFMAs; LDS carries only the A reads (<= 50 % of its bandwidth);
//   * waves never exchange data in the K loop -> no barriers; 5-6 waves per CU overlap each other's latency;
//   * epilogue: each wave transposes its accumulators through a private 16-pixel LDS buffer (no block
//     barrier) so stores and residual loads are 16 bytes per lane along channels.
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "igemm.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace mfx {

struct HaloGeom {
    int B, H, W, C, lgCG, CG, ngroups;     // CG = channels per patch pass (power of two), ngroups = C / CG
    int S, PW, PH, Ho, Wo;                 // stride (1 or 2); patch = PH x PW input pixels = (7 S + 3) x (15 S + 3); output map
    int tiles_x, tiles_y, tiles_n, K_pad;  // weight row length (elements); K index = tap*C + c
    int steps_per_group;                   // ceil(9*CG / (4*ELEMS)): 64-byte K steps per channel group
};

constexpr int kHaloRows = 8;               // output rows per workgroup (FM)

template <typename T, int WN, int FN, int WK = 1> struct HaloSmem {
    static constexpr int stage_ld = FN * 16 + 4;                              // fp32 words per staged pixel row
    static constexpr int stage_bytes = 16 * stage_ld * 4;                     // per wave
    static constexpr int reduce_bytes = WN * (WK - 1) * kHaloRows * FN * 64 * 16;          // split-K: partials of the rows a wave does not finish itself
    static __host__ __device__ constexpr int patch_stride(int CG) { return CG * (int)sizeof(T) + 16; }
    static __host__ __device__ constexpr int patch_bytes(int CG, int S = 1) { return ((kHaloRows - 1) * S + 3) * (15 * S + 3) * patch_stride(CG); }
    // the K-split reduction reuses the patch memory once the K loop is over
    static __host__ __device__ constexpr int main_bytes(int CG, int S = 1) { return patch_bytes(CG, S) > reduce_bytes ? patch_bytes(CG, S) : reduce_bytes; }
    static __host__ __device__ constexpr int total(int CG, int S = 1) { return main_bytes(CG, S) + WN * WK * stage_bytes; }
};

// WK > 1: WK waves share each output slice and split the K steps among themselves (step s belongs to wave s % WK); their
// partial accumulators meet in LDS after the K loop and each wave finishes FM / WK output rows.  For narrow outputs
// (the 27-channel DCN offset/mask conv: N = 32) a workgroup would otherwise be ONE wave walking the whole K alone.
// two output values as they will be stored (identity for fp32, one packed convert for bf16)
template <typename TO> __device__ __forceinline__ f32x2 halo_round2(float a, float b);
template <> __device__ __forceinline__ f32x2 halo_round2<float>(float a, float b) { return f32x2{a, b}; }
template <> __device__ __forceinline__ f32x2 halo_round2<f32s_t>(float a, float b) { return f32x2{a, b}; }
template <> __device__ __forceinline__ f32x2 halo_round2<bf16_t>(float a, float b) {
    const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, bf16x2));
    return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}

template <> __device__ __forceinline__ f32x2 halo_round2<half_t>(float a, float b) {
    const f16x2 h = __builtin_convertvector((f32x2){a, b}, f16x2);
    return f32x2{(float)h[0], (float)h[1]};
}

// PR (split precision only, channel groups of >= 32): the K loop walks step PAIRS.  Both operands keep their hi halves in dwords 0-1 and their
// lo halves in dwords 2-3 of a chunk, so the hi (lo) halves of two consecutive steps form ONE 8-element fp16 MFMA operand: the weights arrive
// re-packed that way (`wfm` = mfx_conv_desc.w_frag_pair, ops.pair_steps), the pixels by two 8-byte LDS reads 16 channels apart.  Three
// products per pair -- hi.hi, lo.hi, hi.lo (lo.lo is below fp32 resolution) -- instead of the four of two mma_chunk<f32s_t> calls.
template <typename T, typename TO, int WN, int FN, int WK = 1, bool ST = false, bool PR = false>
__global__ __launch_bounds__(WN * WK * 64, 2) void conv3x3_wave_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                                      const u32x4* __restrict__ wfm, HaloGeom g, EpiArgs ep) {
    constexpr int NT = WN * WK * 64, FM = kHaloRows;
    static_assert(FM % WK == 0, "rows must split evenly over the K-split waves");
    using SM = HaloSmem<T, WN, FN, WK>;
    constexpr int ELEMS = ElemTraits<T>::ELEMS;
    constexpr int BN = WN * FN * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wk = wave / WN;

    // tile decode: n-tile fastest, then x, y, image -> neighbours share halos in one XCD's L2
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % g.tiles_n; tile /= g.tiles_n;
    const int tx = tile % g.tiles_x; tile /= g.tiles_x;
    const int ty = tile % g.tiles_y; const int b = tile / g.tiles_y;
    const int x0 = tx * 16, y0 = ty * kHaloRows, n0 = tn * BN + wn * (FN * 16);

    const int PS = SM::patch_stride(g.CG);
    char* patch = smem;
    float* stage = reinterpret_cast<float*>(smem + SM::main_bytes(g.CG, g.S) + wave * SM::stage_bytes);
    const int PW = g.PW, rowb = g.S * g.PW * PS;             // patch width (pixels); bytes between the patch rows of consecutive output rows

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int CPP = g.CG * (int)sizeof(T) / 16;              // 16-byte chunks per patch pixel
    const int lgCPP = g.lgCG - (ELEMS == 8 ? 3 : 2);
    const int xl = lane & 15, kq = lane >> 4;
    const T* wrow = w + (size_t)(n0 + xl) * g.K_pad;         // this lane's weight row (fragment j adds 16 rows)
    const size_t wfrag = (size_t)16 * g.K_pad;
    // fragment-major weights: fragment (nf, step) is one contiguous KiB, lane-linear
    const int fsteps = g.K_pad / (4 * ELEMS);
    const u32x4* wfl = wfm ? wfm + (size_t)(n0 >> 4) * fsteps * 64 + lane : nullptr;

    // residual tile: fetched now, consumed in the epilogue (its latency hides behind the whole K loop)
    constexpr int OE_ = ElemTraits<TO>::ELEMS;
    constexpr int GPR_ = FN * 16 / OE_;
    constexpr int RITEMS = (16 * GPR_ + 63) / 64;            // epilogue items per lane per output row
    constexpr bool kResPrefetch = false;   // measured: the 32 extra VGPRs cost a wave of occupancy (3 -> 2 per SIMD), net loss
    u32x4 rpre[kResPrefetch ? FM : 1][RITEMS];
    if constexpr (kResPrefetch) {
        if (ep.res) {
            const T* res_ = reinterpret_cast<const T*>(ep.res);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int q = 0; q < RITEMS; ++q) {
                    const int it = q * 64 + lane;
                    const int px = it / GPR_, ng = it - px * GPR_;
                    const int oy = y0 + i, ox = x0 + px, gn = n0 + ng * OE_;
                    u32x4 z = {0u, 0u, 0u, 0u};
                    if (it < 16 * GPR_ && oy < g.Ho && ox < g.Wo && gn < ep.Cout)
                        z = *reinterpret_cast<const u32x4*>(res_ + (((size_t)b * g.Ho + oy) * g.Wo + ox) * ep.ldres + gn);
                    rpre[i][q] = z;
                }
        }
    }

    for (int grp = 0; grp < g.ngroups; ++grp) {
        if (grp > 0) __syncthreads();                         // every wave is done reading the previous patch
        // ---- halo patch: PH x PW input pixels (10 x 18 at stride 1, 17 x 33 at stride 2) x CG channels
        const T* xg = x + (size_t)b * g.H * g.W * g.C + grp * g.CG;
        const int nchunks = g.PH * PW * CPP;
        // batches of PU independent loads per lane (all in flight together), then the LDS writes
        constexpr int PU = FN >= 4 ? 4 : 8;
        for (int base = 0; base < nchunks; base += NT * PU) {
            u32x4 pr[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int idx = base + u * NT + tid;
                const int pix = idx >> lgCPP, ch = idx & (CPP - 1);
                const int py = pix / PW, px = pix - py * PW;
                const int iy = y0 * g.S - 1 + py, ix = x0 * g.S - 1 + px;
                // (r05: branch-free -- a clamped, always valid address and a select; as `if (inside) load` the compiler serialised some of the
                // batch's loads behind s_waitcnt vmcnt(0))
                const bool in = idx < nchunks && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                const int cy = min(max(iy, 0), g.H - 1), cx = min(max(ix, 0), g.W - 1);
                const u32x4 zz = *reinterpret_cast<const u32x4*>(xg + ((size_t)cy * g.W + cx) * g.C + ch * ELEMS);
                pr[u] = in ? zz : u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int idx = base + u * NT + tid;
                if (idx < nchunks) *reinterpret_cast<u32x4*>(patch + (idx >> lgCPP) * PS + (idx & (CPP - 1)) * 16) = lds_operand<T>(pr[u]);
            }
        }

        if constexpr (PR) {
            // pair p of this group = steps (2p, 2p+1): 32 consecutive channels of one tap (CG % 32 == 0: a pair never straddles taps, and
            // 9 * CG / 16 steps is even: no padding pairs)
            auto wfetch2 = [&](int p, u32x4 (&h)[FN], u32x4 (&l)[FN]) {
                const int e0 = p * 32;
                const int tap0 = e0 >> g.lgCG;
                const int st = (tap0 * g.C + grp * g.CG + (e0 & (g.CG - 1))) >> 4;          // global 16-element step index (even)
#pragma unroll
                for (int j = 0; j < FN; ++j) { h[j] = wfl[((size_t)j * fsteps + st) * 64]; l[j] = wfl[((size_t)j * fsteps + st + 1) * 64]; }
            };
            constexpr int RP = FN == 2 ? 3 : 2;
            u32x4 wh[RP][FN], wl[RP][FN];
            const int np = g.steps_per_group >> 1;
            const int nl = np > wk ? (np - wk + WK - 1) / WK : 0;
            auto P = [&](int j) { return wk + j * WK; };
#pragma unroll
            for (int u = 0; u < RP - 1; ++u)
                if (u < nl) wfetch2(P(u), wh[u], wl[u]);
            __syncthreads();                                      // patch visible to all waves
            auto compute2 = [&](int p, const u32x4 (&h)[FN], const u32x4 (&l)[FN]) {
                const int e = p * 32 + kq * 4;
                const int tap = e >> g.lgCG, cl = e & (g.CG - 1);
                const int th = (tap * 21846) >> 16, tw = tap - th * 3;
                const char* ap = patch + (th * PW + xl * g.S + tw) * PS + cl * 4;
                auto rd = [&](const char* q) {
                    const uint2 a = *reinterpret_cast<const uint2*>(q), b = *reinterpret_cast<const uint2*>(q + 64);
                    return u32x4{a.x, a.y, b.x, b.y};
                };
                u32x4 ph[2], pl[2];
                ph[0] = rd(ap); pl[0] = rd(ap + 8);
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int cur = i & 1;
                    if (i + 1 < FM) { ph[cur ^ 1] = rd(ap + (i + 1) * rowb); pl[cur ^ 1] = rd(ap + (i + 1) * rowb + 8); }
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ph[cur]), __builtin_bit_cast(f16x8, h[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, pl[cur]), __builtin_bit_cast(f16x8, h[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ph[cur]), __builtin_bit_cast(f16x8, l[j]), acc[i][j], 0, 0, 0);
                }
            };
            int s = 0;
            for (; s + RP <= nl; s += RP) {
#pragma unroll
                for (int u = 0; u < RP; ++u) {
                    if (s + u + RP - 1 < nl) wfetch2(P(s + u + RP - 1), wh[(u + RP - 1) % RP], wl[(u + RP - 1) % RP]);
                    compute2(P(s + u), wh[u], wl[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < RP - 1; ++u)
                if (s + u < nl) compute2(P(s + u), wh[u], wl[u]);
        } else {
        // lane's K chunk at step s: e = s*4*ELEMS + kq*ELEMS -> (tap, local channel)
        auto wfetch = [&](int s, u32x4 (&bf)[FN]) {
            const int e = s * (4 * ELEMS) + kq * ELEMS;
            const int tap = e >> g.lgCG, cl = e & (g.CG - 1);
            if (wfl) {
                // global K of this step's first chunk (kq = 0) -> 64-byte step index; wave-uniform
                const int e0 = s * (4 * ELEMS);
                const int tap0 = e0 >> g.lgCG;
                const int st = (tap0 * g.C + grp * g.CG + (e0 & (g.CG - 1))) / (4 * ELEMS);
                const bool ok = g.CG >= 4 * ELEMS ? tap0 < 9 : st < fsteps;     // K padding steps carry zeros / do not exist
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    u32x4 z = {0u, 0u, 0u, 0u};
                    if (ok) z = wfl[((size_t)j * fsteps + st) * 64];
                    bf[j] = z;
                }
                return;
            }
            const T* p = wrow + tap * g.C + grp * g.CG + cl;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                u32x4 z = {0u, 0u, 0u, 0u};
                if (tap < 9) z = *reinterpret_cast<const u32x4*>(p + j * wfrag);
                bf[j] = z;
            }
        };
        // weight fragments: ring of RING register buffers, fetched RING-1 steps ahead of their use.  A step is FM*FN MFMAs
        // = 107 ns (FN 2) .. 213 ns (FN 4) of matrix-pipe time per wave, an L2 round trip is ~400-500 ns: narrow waves need a
        // deeper ring than wide ones (option "halo_ring" to compare).  This wave's steps are S(j) = wk + j*WK, j < nl
        constexpr int RING = FN == 2 ? 5 : 3;               // FN 1 (the offset/mask convs, K-split over 4 waves) measured slower with 5
        u32x4 wb[RING][FN];
        const int ns = g.steps_per_group;
        const int nl = ns > wk ? (ns - wk + WK - 1) / WK : 0;
        auto S = [&](int j) { return wk + j * WK; };
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)
            if (u < nl) wfetch(S(u), wb[u]);
        __syncthreads();                                      // patch visible to all waves

        auto compute = [&](int s, const u32x4 (&bf)[FN]) {
            const int e = s * (4 * ELEMS) + kq * ELEMS;
            int tap = e >> g.lgCG;
            const int cl = e & (g.CG - 1);
            tap = tap > 8 ? 8 : tap;                          // K padding: weights are zero there, keep the address valid
            const int th = (tap * 21846) >> 16, tw = tap - th * 3;
            const char* ap = patch + (th * PW + xl * g.S + tw) * PS + cl * (int)sizeof(T);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const u32x4 af = *reinterpret_cast<const u32x4*>(ap + i * rowb);
#pragma unroll
                for (int j = 0; j < FN; ++j) mma_chunk<T>(af, bf[j], acc[i][j]);
            }
        };
        int s = 0;
        for (; s + RING <= nl; s += RING) {                   // unrolled by RING so the ring indices are static
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                if (s + u + RING - 1 < nl) wfetch(S(s + u + RING - 1), wb[(u + RING - 1) % RING]);
                compute(S(s + u), wb[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < RING - 1; ++u)                    // < RING steps left; their fragments are already in flight
            if (s + u < nl) compute(S(s + u), wb[u]);
        }
    }

    // ---- K-split: partial accumulators -> LDS (the patch is dead), wave wk sums and finishes rows wk*RW .. +RW
    constexpr int RW = FM / WK;
    if constexpr (WK > 1) {
        __syncthreads();                                      // every wave is done reading the patch
        // red[wn][dst][slot][r][j][lane]: what wave `src` accumulated for the RW rows finished by wave `dst`
        // (slot = src with dst skipped): a wave never writes the rows it finishes itself
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
                for (int q = 0; q < WK; ++q) if (q == wk) t = acc[q * RW + r][j];       // own partial (static register index)
#pragma unroll
                for (int slot = 0; slot < WK - 1; ++slot) t += red[((((wn * WK + wk) * (WK - 1) + slot) * RW + r) * FN + j) * 64 + lane];
                own[r][j] = t;
            }
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[r][j] = own[r][j];       // row wk*RW + r now lives in slot r
    }

    // ---- epilogue, wave-private: acc row-fragment i (16 pixels of output row y0+i) -> stage -> 16-byte stores
    constexpr int LDS_ = SM::stage_ld;
    constexpr int OE = ElemTraits<TO>::ELEMS;
    constexpr int GPR = FN * 16 / OE;                         // output chunks per pixel
    const T* res = reinterpret_cast<const T*>(ep.res);
    TO* y = reinterpret_cast<TO*>(ep.y);
    float sc[FN], sh[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        sc[j] = ep.scale ? ep.scale[n0 + j * 16 + xl] : 1.f;
        sh[j] = ep.shift ? ep.shift[n0 + j * 16 + xl] : 0.f;
    }
    // train-mode BN statistics of the output (ep.stats; the conv has no residual / activation then): every lane sums its column
    // n = j*16 + xl over its pixels, of the values AS STORED (rounded to the output type), then the four pixel groups of the
    // wave meet by shuffles and lanes 0..15 add into the layer's scratch -- the separate statistics pass over the map disappears
    // (two values per step in 2-wide vectors: packed convert + v_pk_add_f32 / v_pk_fma_f32 -- 128 values per lane on the widest
    // variant, and the epilogue's VALU work is not hidden behind MFMAs)
    f32x2 st_s[FN], st_q[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) { st_s[j] = f32x2{0.f, 0.f}; st_q[j] = f32x2{0.f, 0.f}; }
    constexpr bool st_on = ST;           // a SEPARATE instantiation: with the statistics code merely branched around, the widest variant
                                         // (<4,4>: 250+ VGPRs) lost a third of its speed even with ep.stats == nullptr (80 -> 130 us)
#pragma unroll
    for (int ii = 0; ii < RW; ++ii) {
        const int i = WK > 1 ? wk * RW + ii : ii;             // output row of accumulator slot ii
        // D layout: col n = lane&15, row (pixel x) = (lane>>4)*4 + r
        const bool row_in = y0 + i < g.Ho;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[ii][j][r] * sc[j] + sh[j];
                stage[((lane >> 4) * 4 + r) * LDS_ + j * 16 + xl] = v[r];
            }
            if (ST && row_in) {
                const int px = x0 + (lane >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    f32x2 vr = halo_round2<TO>(v[r], v[r + 1]);
                    if (px + r + 1 >= g.Wo) { if (px + r >= g.Wo) vr[0] = 0.f; vr[1] = 0.f; }      // ragged right edge only
                    st_s[j] += vr; st_q[j] += vr * vr;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                      // same wave: DS ops complete in order
        const int oy = y0 + i;
#pragma unroll
        for (int q = 0; q < RITEMS; ++q) {
            const int it = q * 64 + lane;
            if (it >= 16 * GPR) continue;
            const int px = it / GPR, ng = it - px * GPR;
            const int ox = x0 + px, gn = n0 + ng * OE;
            if (oy < g.Ho && ox < g.Wo && gn < ep.Cout) {
                const size_t gm = ((size_t)b * g.Ho + oy) * g.Wo + ox;
                float v[OE];
#pragma unroll
                for (int e = 0; e < OE; e += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(stage + px * LDS_ + ng * OE + e);
                    v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
                }
                if (res) {
                    const T* rp = res + gm * ep.ldres + gn;
                    if constexpr (kResPrefetch) {
                        float rv[OE];
                        ElemTraits<T>::unpack(rpre[ii][q], rv);
#pragma unroll
                        for (int e = 0; e < OE; ++e) v[e] += rv[e];
                    } else if constexpr (ElemTraits<T>::ELEMS == OE) {
                        float rv[OE];
                        ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(rp), rv);
#pragma unroll
                        for (int e = 0; e < OE; ++e) v[e] += rv[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < OE; ++e) v[e] += ElemTraits<T>::load(rp + e);
                    }
                }
apply_act_chunk<OE>(v, ep.act, gn);
                *reinterpret_cast<u32x4*>(y + gm * ep.ldy + gn) = ElemTraits<TO>::pack(v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (ST) {
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

static inline int cdivh(int a, int b) { return (a + b - 1) / b; }
static inline int ilog2h(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

int g_opt_halo = 1;          // 0 = generic kernel only, 1 = automatic, >= 2 = force variant (value - 1)
int g_opt_halo_cg = 0;       // max channels per patch pass (0 = default)
int g_opt_halo_s2 = 1;       // option "halo_s2": 0 = stride-2 3x3 convs stay on the generic implicit-GEMM kernel
int g_opt_halo_pair = 1;     // option "halo_pair": split precision walks K in step pairs where mfx_conv_desc.w_frag_pair is given (0 = two mma_chunk per pair)

template <typename T, typename TO, int WN, int FN, int WK = 1, bool ST = false>
static int launch_halo_st(const mfx_conv_desc* d, hipStream_t st);

int try_conv_cw(const mfx_conv_desc* d, int v, hipStream_t st);      // conv_cw.hip: 0 = ran, 1 = no instantiation, < 0 = error
int try_conv_cws(const mfx_conv_desc* d, int v, hipStream_t st);     // conv_cws.hip: the same for split precision

static bool g_halo_stats_ran = false;    // set by the launch that ran a statistics-accumulating instantiation (read by try_conv_halo)

// statistics-accumulating instantiations exist for output tiles up to 128 channels wide (wider ones are atomic-bound there,
// autograd.py keeps the separate pass for them)
template <typename T, typename TO, int WN, int FN, int WK = 1>
static int launch_halo(const mfx_conv_desc* d, hipStream_t st) {
    if constexpr (WN * FN * 16 <= 128 && std::is_same<T, TO>::value) {
        if (d->stats) { g_halo_stats_ran = true; return launch_halo_st<T, TO, WN, FN, WK, true>(d, st); }
    }
    return launch_halo_st<T, TO, WN, FN, WK, false>(d, st);
}

template <typename T, typename TO, int WN, int FN, int WK, bool ST>
static int launch_halo_st(const mfx_conv_desc* d, hipStream_t st) {
    using SM = HaloSmem<T, WN, FN, WK>;
    constexpr int ELEMS = ElemTraits<T>::ELEMS;
    constexpr int BN = WN * FN * 16;
    HaloGeom g;
    g.B = d->B; g.H = d->H; g.W = d->W; g.C = d->Ck;
    // 512-byte channel rows: 256 bf16 / 128 f32.  Split precision: 64 channels (a 49 KB patch instead of 95 KB: two workgroups per CU;
    // B = 8 step 6.58 -> 6.42 ms, `halo_cg` A/B on MI355X)
    int cg_max = g_opt_halo_cg > 0 ? g_opt_halo_cg : (std::is_same<T, f32s_t>::value ? 64 : 512 / (int)sizeof(T));
    g.S = d->stride; g.PW = 15 * g.S + 3; g.PH = (kHaloRows - 1) * g.S + 3; g.Ho = d->Ho; g.Wo = d->Wo;
    // stride 2: the 17 x 33 patch is 3.1x the stride-1 one -- 64-byte channel groups keep it at 45 KB (three workgroups per CU)
    if (g.S == 2 && g_opt_halo_cg <= 0) cg_max = 64 / (int)sizeof(T);
    if (cg_max < 2 * ELEMS) cg_max = 2 * ELEMS;
    g.CG = d->Ck < cg_max ? d->Ck : cg_max; g.lgCG = ilog2h(g.CG); g.ngroups = d->Ck / g.CG;
    g.tiles_x = cdivh(d->Wo, 16); g.tiles_y = cdivh(d->Ho, kHaloRows); g.tiles_n = d->Cout_pad / BN; g.K_pad = d->K_pad;
    g.steps_per_group = cdivh(9 * g.CG, 4 * ELEMS);
    EpiArgs ep;
    ep.scale = d->scale; ep.shift = d->shift; ep.res = d->res; ep.y = d->y; ep.ldy = d->ldy; ep.ldres = d->ldres;
    ep.Cout = d->Cout; ep.act = d->act; ep.K_pad = d->K_pad; ep.nk = 0; ep.tiles_n = g.tiles_n;
    ep.stats = ST ? d->stats : nullptr; ep.stats_ncopy = d->stats_ncopy > 0 ? d->stats_ncopy : 1;
    const int smem = SM::total(g.CG, g.S);
    const int tiles = g.tiles_n * g.tiles_x * g.tiles_y * d->B;
    if constexpr (std::is_same<T, f32s_t>::value && !ST) {
        if (g_opt_halo_pair && d->w_frag_pair && g.CG % 32 == 0) {       // split precision: the pair-walking K loop (3 products per pair)
            auto kp = conv3x3_wave_kernel<T, TO, WN, FN, WK, false, true>;
            static int attr_smem_p = 0;
            if (smem > 64 * 1024 && smem > attr_smem_p) {
                MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                attr_smem_p = smem;
            }
            hipLaunchKernelGGL(kp, dim3(tiles), dim3(WN * WK * 64), smem, st, reinterpret_cast<const T*>(d->x), reinterpret_cast<const T*>(d->w),
                               reinterpret_cast<const u32x4*>(d->w_frag_pair), g, ep);
            MFX_HIP_CHECK(hipGetLastError());
            return MFX_OK;
        }
    }
    auto k = conv3x3_wave_kernel<T, TO, WN, FN, WK, ST>;
    static int attr_smem = 0;
    if (smem > 64 * 1024 && smem > attr_smem) {
        MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_smem = smem;
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(WN * WK * 64), smem, st, reinterpret_cast<const T*>(d->x), reinterpret_cast<const T*>(d->w),
                       reinterpret_cast<const u32x4*>(d->w_frag), g, ep);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// variants (value of option "halo" minus 1):  1: BN16 (1 wave x FN1)  2: BN32 (1 x 2)  3: BN64 (1 x 4)  4: BN128 (2 x 4)
//                                              5: BN256 (4 x 4)        6: BN64 (2 x 2)  7: BN128 (4 x 2)
template <typename T, typename TO> static int halo_variant(int v, const mfx_conv_desc* d, hipStream_t st) {
    switch (v) {
        case 1: return launch_halo<T, TO, 1, 1>(d, st);
        case 2: return launch_halo<T, TO, 1, 2>(d, st);
        case 8: return launch_halo<T, TO, 1, 2, 4>(d, st);    // BN32, 4 waves split K
        case 9: return launch_halo<T, TO, 1, 2, 2>(d, st);    // BN32, 2 waves split K
        case 10: return launch_halo<T, TO, 1, 1, 4>(d, st);   // BN16, 4 waves split K
        default: break;
    }
    if constexpr (std::is_same<T, TO>::value) {
        switch (v) {
            case 3: return launch_halo<T, TO, 1, 4>(d, st);
            case 4: return launch_halo<T, TO, 2, 4>(d, st);
            case 5: return launch_halo<T, TO, 4, 4>(d, st);
            case 6: return launch_halo<T, TO, 2, 2>(d, st);
            case 7: return launch_halo<T, TO, 4, 2>(d, st);
            case 11: return launch_halo<T, TO, 4, 2, 2>(d, st);   // BN128, 2-way K split (8 waves)
            case 12: return launch_halo<T, TO, 2, 2, 2>(d, st);   // BN64, 2-way K split
            case 13: return launch_halo<T, TO, 2, 2, 4>(d, st);   // BN64, 4-way K split (8 waves)
            case 14: return launch_halo<T, TO, 2, 4, 2>(d, st);   // BN128 (2 x FN4), 2-way K split
            default: break;
        }
    }
    return mfx_fail(MFX_ERR_UNSUPPORTED, "conv halo: no such variant");
}

static int variant_bn(int v) { const int bn[15] = {0, 16, 32, 64, 128, 256, 64, 128, 32, 32, 16, 128, 64, 64, 128}; return (v >= 1 && v <= 14) ? bn[v] : 0; }

// returns 1 if the halo kernel handled the convolution, 0 if the caller should use the generic kernel, <0 on error
static int try_conv_halo_impl(const mfx_conv_desc* d, hipStream_t st);

int try_conv_halo(const mfx_conv_desc* d, hipStream_t st, int* stats_ran) {
    g_halo_stats_ran = false;
    const int rc = try_conv_halo_impl(d, st);
    if (stats_ran) *stats_ran = (rc == 1 && g_halo_stats_ran) ? 1 : 0;
    return rc;
}

static int try_conv_halo_impl(const mfx_conv_desc* d, hipStream_t st) {
    if (g_opt_halo == 0) return 0;
    if (d->kh != 3 || d->kw != 3 || (d->stride != 1 && d->stride != 2) || d->pad_h != 1 || d->pad_w != 1 || d->dil_w != 1) return 0;
    if (d->stride == 2 && !g_opt_halo_s2) return 0;
    if (d->rowmap || d->x_pixstride != d->Ck || d->Ho != (d->H - 1) / d->stride + 1 || d->Wo != (d->W - 1) / d->stride + 1 || d->M != d->B * d->Ho * d->Wo) return 0;
    const int elems = (d->dtype == MFX_F32 || d->dtype == MFX_F16X2) ? 4 : 8;
    if (d->Ck < 2 * elems || d->K_pad < 9 * d->Ck) return 0;
    const int N = d->Cout_pad;
    const int px_tiles = d->B * cdivh(d->Ho, kHaloRows) * cdivh(d->Wo, 16);
    // variant table from tools/conv_probe.py on MI355X (profiles/r01_*): two or four waves share a patch,
    // 32 output channels per wave; narrow outputs on small maps split N further to get more waves in flight
    // variant table from tools/conv_bench.py on MI355X, B = 8 (graph replay, us; profiles/r02_conv_variants.md):
    //   N = 32 (level1, the DCN offset/mask convs): 32ch@192x640 v2 44 / v10 76; 64ch@96x320 v9 22 / v10 27; 128ch@48x160 v8 11;
    //     256ch@24x80 v10 12; 512ch@12x40 v10 20 -- the fewer pixel tiles, the more the waves split K
    //   N = 64: 64ch@96x320 v6 32 / v12 36; 64ch@48x160 v12 10.5 / v6 14; 128ch@96x320 v12 58 / v6 66;
    //     256ch@96x320 (the heads' data gradient) v13 103 / v12 137
    //   N % 128: 64->256@96x320 (the heads' trunks in training) v5 81 / v7 99; 256->128@48x160 v11 47 / v7 61;
    //     128->128@48x160 v7 26; level4/5 (120-240 pixel tiles) v11
    int v;
    const int Ck = d->Ck;
    if (N == 16) v = 1;
    else if (N == 32) v = px_tiles >= 8000 ? 2 : (px_tiles >= 1500 ? 9 : (px_tiles >= 300 ? 8 : 10));
    else if (N == 64) v = Ck >= 256 ? 13 : (Ck >= 128 ? 12 : (px_tiles >= 1500 ? 6 : 12));
    else if (N % 128 == 0) {
        if (N % 256 == 0 && Ck <= 64 && px_tiles >= 1500) v = 5;
        else v = (Ck >= 256 || px_tiles * (N / 128) < 300) ? 11 : 7;
    } else v = 6;
    if (d->dtype == MFX_F16X2 && N == 32) {
        // split precision (three MFMAs per step pair, 49 KB fp32 patches): the 27-channel DCN offset/mask convs take the 4-way K split on
        // every map size -- B = 8 step 6.27 -> 6.20 ms, same box (bench.py --dtype fp16x2); the same move for N = 64 (v12) and N % 128 (v11)
        // measured +-0.01 ms and is not made
        v = 8;
    }
    if (g_opt_halo >= 2) {
        const int f = g_opt_halo - 1, bn = variant_bn(f);
        if (bn && N % bn == 0 && (bn >= 64) == (N >= 64)) v = f;
    }
    if ((d->dtype == MFX_BF16 || d->dtype == MFX_F16) && g_opt_halo_cg <= 0) {
        // compile-time-geometry form of the same decomposition (conv_cw.hip): bit-identical output
        const int r = try_conv_cw(d, v, st);
        if (r == 0 && d->stats) g_halo_stats_ran = true;
        if (r <= 0) return r == 0 ? 1 : r;
        static const bool trace = getenv("MFX_TRACE_CW") != nullptr;      // which layers stay on the run-time-geometry kernel
        if (trace) fprintf(stderr, "cw-fallback v=%d Ck=%d Cout=%d/%d s=%d HxW=%dx%d B=%d stats=%d out=%d res=%d act=%d\n", v, d->Ck, d->Cout, d->Cout_pad, d->stride,
                           d->H, d->W, d->B, d->stats ? 1 : 0, d->out_dtype, d->res ? 1 : 0, d->act);
    }
    if (d->dtype == MFX_F16X2 && g_opt_halo_cg <= 0 && g_opt_halo_pair) {
        // split precision, compile-time-geometry form of the pair-walking kernel (conv_cws.hip): bit-identical output
        const int r = try_conv_cws(d, v, st);
        if (r <= 0) return r == 0 ? 1 : r;
    }
    int rc;
    if (d->dtype == MFX_F32) rc = halo_variant<float, float>(v, d, st);
    else if (d->dtype == MFX_F16X2) rc = halo_variant<f32s_t, f32s_t>(v, d, st);
    else if (d->dtype == MFX_F16) rc = d->out_dtype == MFX_F16 ? halo_variant<half_t, half_t>(v, d, st) : halo_variant<half_t, float>(v, d, st);
    else if (d->out_dtype == MFX_BF16) rc = halo_variant<bf16_t, bf16_t>(v, d, st);
    else rc = halo_variant<bf16_t, float>(v, d, st);
    return rc == MFX_OK ? 1 : rc;
}

}  // namespace mfx

MFX_RANGE_FLAG_ACCESSOR(conv_halo)      // split-precision range sentinel of this translation unit (common.h)
