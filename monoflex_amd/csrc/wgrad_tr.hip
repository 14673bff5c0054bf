// Convolution weight gradient on the matrix cores, second generation (bf16 activations):
//     dW[o][k] = sum_m dy[m][o] * A[m][k],   A = implicit im2col of x (k = tap * Ck + c) or a dense [M][K] matrix
// (reference: torch autograd of nn.Conv2d, engine/trainer.py:116-117; DCN: dcn_v2_cuda.cu:292-318).
//
// The reduction runs over pixels, the SLOW axis of both NHWC operands, while an MFMA fragment wants 8 consecutive reduction
// elements per lane.  The first generation (train_kernels.hip, conv_wgrad_mfma_kernel) transposed on the way INTO LDS: sixteen
// 4-byte ds_write per thread and step for four MFMAs per wave -- 170 TFLOP/s over the step's 1.4 PFLOP of weight gradients,
// the largest line of the training profile (profiles/r02_b_train_step_kernel_stats.md).  Here the tiles go into LDS in their
// natural layout with 16-byte stores and are transposed on the way OUT by ds_read_b64_tr_b16 (gfx950): within a 16-lane group
// lanes 4j..4j+3 supply row j (4 x 8 bytes) and lane l receives column l of rows 0..3 (semantics probed on the device,
// tools/probes/tr_probe.hip).  LDS image per operand and step: 16-channel sub-tiles of 32 pixel rows x 32 bytes, rows stored
// at position p(r) = r with bits 2 and 3 swapped, so that the two 16-lane groups of a 32-lane LDS pass (pixel rows 8g..8g+3
// and 8g+8..8g+11) read eight consecutive 32-byte rows = all 64 banks once.
//
// Workgroup = WO x 3 waves; every wave owns a 64 (o) x 64 (k) block = 4 x 4 MFMA fragments: 16 MFMAs per 16 transposed reads
// and step.  BK = 192 divides K = 9 * Ck exactly for every 3x3 convolution with Ck a multiple of 64.  The pixel range is
// split into slabs over blockIdx.z; partial tiles go to the workspace with plain stores (wgrad_reduce_kernel sums them).
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"
#include "fill.h"
#include "wgrad.h"
#include <algorithm>

namespace mfx {

constexpr int TR_BK = 192, TR_STEP = 32;

__device__ __forceinline__ int tr_rowpos(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// four 16x(8 rows) fragments of one operand: sub-tiles at byte offsets 0 / 1056 / 2112 / 3168 from the two lane addresses
// (rows 8g..8g+3 and 8g+4..8g+7); ONE asm block so that the wait sits behind all eight reads and nothing that consumes the
// results can be scheduled in front of it
__device__ __forceinline__ void tr_read4(uint32_t alo, uint32_t ahi, u32x4 (&f)[4]) {
    uint64_t l0, h0, l1, h1, l2, h2, l3, h3;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\t"
        "ds_read_b64_tr_b16 %1, %9\n\t"
        "ds_read_b64_tr_b16 %2, %8 offset:1056\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:1056\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:2112\n\t"
        "ds_read_b64_tr_b16 %5, %9 offset:2112\n\t"
        "ds_read_b64_tr_b16 %6, %8 offset:3168\n\t"
        "ds_read_b64_tr_b16 %7, %9 offset:3168\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
        : "v"(alo), "v"(ahi)
        : "memory");
    f[0] = u32x4{(uint32_t)l0, (uint32_t)(l0 >> 32), (uint32_t)h0, (uint32_t)(h0 >> 32)};
    f[1] = u32x4{(uint32_t)l1, (uint32_t)(l1 >> 32), (uint32_t)h1, (uint32_t)(h1 >> 32)};
    f[2] = u32x4{(uint32_t)l2, (uint32_t)(l2 >> 32), (uint32_t)h2, (uint32_t)(h2 >> 32)};
    f[3] = u32x4{(uint32_t)l3, (uint32_t)(l3 >> 32), (uint32_t)h3, (uint32_t)(h3 >> 32)};
}

template <typename T, int WO>                                 // T = bf16_t or half_t (the staging and the transposed reads move 16-bit words)
__global__ __launch_bounds__(WO * 192) void conv_wgrad_tr_kernel(const T* __restrict__ x, const T* __restrict__ dy, WgradGeom g) {
    constexpr int NT = WO * 192, BO = WO * 64, BK = TR_BK;
    constexpr int OB = BO / 16, KB = BK / 16;                 // 16-channel sub-tiles per operand
    constexpr int SUB = TR_STEP * 32 + 32;                    // bytes per sub-tile: 32 rows x 32 B, + 32 so that the sub-tiles of one
                                                              // pixel (the 16-byte stores of 8 neighbouring lanes) start 8 banks apart
    constexpr int DY_BYTES = OB * SUB, A_BYTES = KB * SUB, STAGE = DY_BYTES + A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char lds[];        // [2][STAGE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave / 3, wk = wave - wo * 3;
    const int k0 = blockIdx.x * BK, o0 = blockIdx.y * BO;
    const int m_begin = blockIdx.z * g.m_per_block, m_end = min(m_begin + g.m_per_block, g.M);

    // ---- loader roles.  A tile: 32 pixel rows x 24 chunks of 8 channels; thread -> fixed chunk column, rows r0 + j * (NT / 24)
    constexpr int A_RPP = NT / 24, A_N = TR_STEP / A_RPP;     // rows per pass (8 or 16), passes (4 or 2)
    const int a_kc = tid % 24, a_r0 = tid / 24;
    const int a_kk = k0 + a_kc * 8;
    const bool a_ok = a_kk < g.K;
    const int a_tap = a_ok ? a_kk / g.Ck : 0, a_ch = a_kk - a_tap * g.Ck;
    const int a_th = a_tap / g.kw, a_tw = a_tap - a_th * g.kw;
    // dy tile: 32 rows x OB*2 chunks; item id = tid + j * NT, row = id / (2 * OB), chunk = id % (2 * OB)
    constexpr int D_CPR = 2 * OB, D_ITEMS = TR_STEP * D_CPR, D_N = (D_ITEMS + NT - 1) / NT;
    const int hw = g.Ho * g.Wo;

    struct Regs { u32x4 a[A_N]; u32x4 d[D_N]; };
    // (b, oh, ow) of this thread's A rows, advanced by 32 pixels per load: integer divisions per chunk made the first version
    // VALU-bound (12.5 VALU instructions per MFMA, PMC pass of round 2: SQ_INSTS_VALU / SQ_INSTS_MFMA)
    int pb[A_N], poh[A_N], pow_[A_N];
#pragma unroll
    for (int j = 0; j < A_N; ++j) {
        const int m = m_begin + a_r0 + j * A_RPP;
        pb[j] = m / hw; const int rem = m - pb[j] * hw; poh[j] = rem / g.Wo; pow_[j] = rem - poh[j] * g.Wo;
    }
    int m_next = m_begin;                                      // first pixel of the next tile to load (loads are issued in order)
    auto gload = [&](Regs& r, int m0) {                        // tiles of pixels m0 .. m0 + 31
#pragma unroll
        for (int j = 0; j < A_N; ++j) {
            r.a[j] = u32x4{0u, 0u, 0u, 0u};
            const int m = m0 + a_r0 + j * A_RPP;
            if (a_ok && m < m_end) {
                if (g.direct) r.a[j] = *reinterpret_cast<const u32x4*>(x + (size_t)m * g.x_pixstride + a_kk);
                else {
                    const int ih = poh[j] * g.stride - g.pad_h + a_th, iw = pow_[j] * g.stride - g.pad_w + a_tw * g.dil_w;
                    if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
                        r.a[j] = *reinterpret_cast<const u32x4*>(x + ((size_t)(pb[j] * g.H + ih) * g.W + iw) * g.x_pixstride + a_ch);
                }
            }
            pow_[j] += TR_STEP;                                // next tile: 32 pixels further (row-major over (b, oh, ow))
            while (pow_[j] >= g.Wo) { pow_[j] -= g.Wo; if (++poh[j] == g.Ho) { poh[j] = 0; ++pb[j]; } }
        }
        (void)m_next;
#pragma unroll
        for (int j = 0; j < D_N; ++j) {
            r.d[j] = u32x4{0u, 0u, 0u, 0u};
            const int id = tid + j * NT;
            const int row = id / D_CPR, ch = id - row * D_CPR;
            const int m = m0 + row;
            if ((D_ITEMS % NT == 0 || id < D_ITEMS) && m < m_end && o0 + ch * 8 < g.Cout)
                r.d[j] = *reinterpret_cast<const u32x4*>(dy + (size_t)m * g.ldy + o0 + ch * 8);
        }
    };
    auto lstore = [&](int buf, const Regs& r) {
        char* base = lds + buf * STAGE;
#pragma unroll
        for (int j = 0; j < A_N; ++j) {
            const int row = a_r0 + j * A_RPP;
            *reinterpret_cast<u32x4*>(base + DY_BYTES + (a_kc >> 1) * SUB + tr_rowpos(row) * 32 + (a_kc & 1) * 16) = r.a[j];
        }
#pragma unroll
        for (int j = 0; j < D_N; ++j) {
            const int id = tid + j * NT;
            const int row = id / D_CPR, ch = id - row * D_CPR;
            if (D_ITEMS % NT == 0 || id < D_ITEMS)
                *reinterpret_cast<u32x4*>(base + (ch >> 1) * SUB + tr_rowpos(row) * 32 + (ch & 1) * 16) = r.d[j];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed fragment reads: lane l of 16-lane group gq supplies 8 bytes of row 8*gq + (l16 >> 2) (+4 for the second
    // half), channels (l16 & 3) * 4 .. +3 of the sub-tile, and receives channel l16, rows 8*gq .. 8*gq+3 (+4)
    const int l16 = lane & 15, gq = lane >> 4;
    const int frag_lo = tr_rowpos(8 * gq + (l16 >> 2)) * 32 + (l16 & 3) * 8;
    const int frag_hi = tr_rowpos(8 * gq + 4 + (l16 >> 2)) * 32 + (l16 & 3) * 8;
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    static_assert(SUB == 1056, "tr_read4 immediates");
    auto compute = [&](int buf) {
        const uint32_t sb = lds_base + buf * STAGE;
        u32x4 af[4], bf[4];
        tr_read4(sb + wo * 4 * SUB + frag_lo, sb + wo * 4 * SUB + frag_hi, af);
        tr_read4(sb + DY_BYTES + wk * 4 * SUB + frag_lo, sb + DY_BYTES + wk * 4 * SUB + frag_hi, bf);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma_chunk<T>(af[i], bf[j], acc[i][j]);
    };

    // steps s = 0 .. ns-1; register set R[s & 1] holds the data of step s.  Global loads run TWO steps ahead of their LDS
    // store (one step of 16 MFMAs per wave is far shorter than an L2 / HBM round trip): iteration s issues the loads of step
    // s+2 into the set that step s has just vacated, computes step s, then stores step s+1 (loaded one iteration ago).
    const int ns = (m_end - m_begin + TR_STEP - 1) / TR_STEP;
    Regs R0, R1;
    gload(R0, m_begin);
    lstore(0, R0);
    if (ns > 1) gload(R1, m_begin + TR_STEP);
    __syncthreads();
    int s = 0;
    for (; s + 2 <= ns; s += 2) {                             // unrolled by two: static register sets
        if (s + 2 < ns) gload(R0, m_begin + (s + 2) * TR_STEP);
        compute(0);
        if (s + 1 < ns) lstore(1, R1);
        __syncthreads();
        if (s + 3 < ns) gload(R1, m_begin + (s + 3) * TR_STEP);
        compute(1);
        if (s + 2 < ns) lstore(0, R0);
        __syncthreads();
    }
    if (s < ns) compute(0);                                   // odd tail (its data was stored by the last iteration)

    // D: col (lane & 15) = k, row (lane >> 4) * 4 + r = o; partial tile -> workspace slab (plain stores)
    float* wsb = g.ws + (size_t)blockIdx.z * g.ws_slab;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + wo * 64 + i * 16 + (lane >> 4) * 4 + r, k = k0 + wk * 64 + j * 16 + (lane & 15);
                if (o < g.Cout && k < g.K) wsb[(size_t)o * g.ws_ld + k] = acc[i][j][r];
            }
}

}  // namespace mfx
using namespace mfx;

int g_opt_wgrad_tr = 1;          // option "wgrad_tr": 0 = first-generation kernel everywhere
int g_opt_wgrad_tr_blocks = 512;   // option "wgrad_tr_blocks": target workgroup count (tiles x pixel slabs)

// returns 1 if handled (partial tiles are in g.ws: the caller runs wgrad_reduce_kernel), 0 to fall through
template <typename T> static int conv_wgrad_tr_t(const void* x, const void* dy, WgradGeom& g, void* workspace, size_t workspace_bytes, int* nslab_out, hipStream_t st);
int try_conv_wgrad_tr(const void* x, const void* dy, WgradGeom& g, int dtype, void* workspace, size_t workspace_bytes, int* nslab_out, hipStream_t st) {
    if (dtype == MFX_F16) return conv_wgrad_tr_t<half_t>(x, dy, g, workspace, workspace_bytes, nslab_out, st);
    return conv_wgrad_tr_t<bf16_t>(x, dy, g, workspace, workspace_bytes, nslab_out, st);
}
template <typename T> static int conv_wgrad_tr_t(const void* x, const void* dy, WgradGeom& g, void* workspace, size_t workspace_bytes, int* nslab_out, hipStream_t st) {
    if (!g_opt_wgrad_tr || !workspace) return 0;
    // Cout = 64: the 64 x 192 workgroup tile (49 FLOP per L1 byte) measured slower than the first-generation kernel
    // (188 vs 137 us on 64->64 @ 96x320, B=8), so those layers fall through unless option wgrad_tr = 2 forces it
    if (g.Cout % 128 != 0 && g_opt_wgrad_tr != 2) return 0;
    if (g.K % TR_BK != 0 || g.Cout % 64 != 0 || g.Ck % 8 != 0 || g.x_pixstride % 8 != 0 || g.ldy % 8 != 0 || g.M < 4096) return 0;
    const int wo = g.Cout % 128 == 0 ? 2 : 1, bo = wo * 64;
    const int tiles = (g.K / TR_BK) * (g.Cout / bo);
    const int ws_ld = g.K;
    const long ws_slab = (long)g.Cout * ws_ld;
    int slabs = std::max(1, g_opt_wgrad_tr_blocks / tiles);
    slabs = (int)std::min<long>(slabs, (long)(workspace_bytes / sizeof(float)) / ws_slab);
    if (slabs < 1) return 0;
    g.m_per_block = std::max(512, (int)(((long)g.M / slabs + TR_STEP - 1) / TR_STEP * TR_STEP));
    const int nslab = (g.M + g.m_per_block - 1) / g.m_per_block;
    g.ws = reinterpret_cast<float*>(workspace); g.ws_ld = ws_ld; g.ws_slab = ws_slab;
    const dim3 grid(g.K / TR_BK, g.Cout / bo, nslab);
    if (wo == 2) {
        constexpr int smem = 2 * ((128 / 16) + (TR_BK / 16)) * (TR_STEP * 32 + 32);
        hipLaunchKernelGGL((conv_wgrad_tr_kernel<T, 2>), grid, dim3(384), smem, st, (const T*)x, (const T*)dy, g);
    } else {
        constexpr int smem = 2 * ((64 / 16) + (TR_BK / 16)) * (TR_STEP * 32 + 32);
        hipLaunchKernelGGL((conv_wgrad_tr_kernel<T, 1>), grid, dim3(192), smem, st, (const T*)x, (const T*)dy, g);
    }
    *nslab_out = nslab;
    return 1;
}

// ======================================================================================================================
// Third form, for 3x3 / stride 1 / pad 1 convolutions (the heads' nine 64->256 trunks, the DLA trunk, the DCN offset/mask
// convs): the implicit-im2col operand is NOT re-read from memory once per tap.  A workgroup stages the input PATCH of a
// 4 x 32 pixel tile (6 x 34 pixels x 64 channels, 26 KB) and the tile's dy rows (128 pixels x 64 output channels, 16 KB) in
// LDS once and derives all nine taps' operand fragments from the patch with shifted transposed reads -- 225 FLOP per byte
// staged instead of 49-77 -- while the whole (64 o) x (9 taps x 64 c) gradient block stays in the accumulators of its six
// waves across ALL tiles of the workgroup's pixel slab.
//
// Transposed reads (ds_read_b64_tr_b16) need no row permutation here: the MFMA k index is mapped to pixels as
// k-group g <-> pixels {4g..4g+3} and {16+4g..16+4g+3} of a 32-pixel row segment, for BOTH operands, so the two 16-lane
// groups of an LDS pass read eight consecutive 32-byte pixel rows (all 64 banks once) at any pixel shift.
// ======================================================================================================================
namespace mfx {

constexpr int WP_TH = 4, WP_TW = 32, WP_PW = WP_TW + 2, WP_PPIX = (WP_TH + 2) * WP_PW;      // patch 6 x 34 pixels
constexpr int WP_XSUB = WP_PPIX * 32 + 32, WP_DSUB = WP_TH * WP_TW * 32 + 32;              // sub-tile strides (+32: banks)


struct WpGeom { int B, H, W, Cin, Cout, ldy, tiles_x, tiles_y, ntiles, tiles_per_slab, K; float* ws; long ws_slab; };

// OS = 16-channel sub-tiles of dy handled by the workgroup (4: 64 output channels, 2: 32, 1: 16); XS = 16-channel sub-tiles of
// the input slice (4: a 64-channel slice; 2 / 1: the whole input of the 32- and 16-channel full-resolution layers); NW = waves,
// each owning JW = 9 * XS / NW of the (tap, 16-channel) operand sub-tiles
template <typename T, int OS, int NW, int XS = 4>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_patch_kernel(const T* __restrict__ x, const T* __restrict__ dy, WpGeom g) {
    constexpr int WP_NT = 64 * NW, JW = 9 * XS / NW;
    static_assert(9 * XS % NW == 0 && (JW == 6 || JW <= 3), "operand sub-tiles per wave");
    constexpr int XCH = WP_PPIX * 2 * XS, DCH = WP_TH * WP_TW * 2 * OS;      // 16-byte chunks per tile: x patch, dy
    constexpr int XN = (XCH + WP_NT - 1) / WP_NT, DN = (DCH + WP_NT - 1) / WP_NT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int STAGE = XS * WP_XSUB + OS * WP_DSUB;                       // x patch [XS sub][204 px][32 B] | dy [OS sub][128 px][32 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);               // k sub-tiles JW*wave .. JW*wave+JW-1 of 9*XS
    const int o0 = blockIdx.x * (16 * OS), cs = blockIdx.y;                  // first output channel, input slice of 16*XS channels
    const int t_begin = blockIdx.z * g.tiles_per_slab, t_end = min(t_begin + g.tiles_per_slab, g.ntiles);

    struct Regs { u32x4 xr[XN]; u32x4 dr[DN]; };
    auto gload = [&](Regs& r, int t) {
        const int tx = t % g.tiles_x; int q = t / g.tiles_x;
        const int ty = q % g.tiles_y, b = q / g.tiles_y;
        const int x0 = tx * WP_TW, y0 = ty * WP_TH;
        const T* xb = x + (size_t)b * g.H * g.W * g.Cin + cs * (16 * XS);
        const T* db = dy + (size_t)b * g.H * g.W * g.ldy + o0;
#pragma unroll
        for (int u = 0; u < XN; ++u) {
            const int id = tid + u * WP_NT;
            const int p = id / (2 * XS), c8 = id - p * (2 * XS);
            const int pr = p / WP_PW, pc = p - pr * WP_PW;
            const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
            u32x4 z = {0u, 0u, 0u, 0u};
            if ((XCH % WP_NT == 0 || id < XCH) && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                z = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * g.W + ix) * g.Cin + c8 * 8);
            r.xr[u] = z;
        }
#pragma unroll
        for (int u = 0; u < DN; ++u) {
            const int id = tid + u * WP_NT;
            const int px = id / (2 * OS), c8 = id - px * (2 * OS);
            const int oy = y0 + px / WP_TW, ox = x0 + (px % WP_TW);
            u32x4 z = {0u, 0u, 0u, 0u};
            if ((DCH % WP_NT == 0 || id < DCH) && oy < g.H && ox < g.W && o0 + c8 * 8 < g.Cout)
                z = *reinterpret_cast<const u32x4*>(db + ((size_t)oy * g.W + ox) * g.ldy + c8 * 8);
            r.dr[u] = z;
        }
    };
    auto lstore = [&](int buf, const Regs& r) {
        char* xs = lds + buf * STAGE;
        char* ds = xs + XS * WP_XSUB;
#pragma unroll
        for (int u = 0; u < XN; ++u) {
            const int id = tid + u * WP_NT;
            const int p = id / (2 * XS), c8 = id - p * (2 * XS);
            if (XCH % WP_NT == 0 || id < XCH)
                *reinterpret_cast<u32x4*>(xs + (c8 >> 1) * WP_XSUB + p * 32 + (c8 & 1) * 16) = r.xr[u];
        }
#pragma unroll
        for (int u = 0; u < DN; ++u) {
            const int id = tid + u * WP_NT;
            const int px = id / (2 * OS), c8 = id - px * (2 * OS);
            if (DCH % WP_NT == 0 || id < DCH)
                *reinterpret_cast<u32x4*>(ds + (c8 >> 1) * WP_DSUB + px * 32 + (c8 & 1) * 16) = r.dr[u];
        }
    };

    f32x4 acc[OS][JW];
#pragma unroll
    for (int i = 0; i < OS; ++i)
#pragma unroll
        for (int j = 0; j < JW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // lane part of every transposed read: pixel 4*gq + (l16 >> 2) of the 32-pixel segment, 8-byte quarter (l16 & 3)
    const int l16 = lane & 15, gq = lane >> 4;
    const uint32_t lane_off = (uint32_t)((4 * gq + (l16 >> 2)) * 32 + (l16 & 3) * 8);
    const uint32_t xs_a = (uint32_t)(uintptr_t)lds, ds_a = xs_a + XS * WP_XSUB;
    uint32_t xoff[JW];                                                       // this wave's (tap, channel sub-tile) operands
#pragma unroll
    for (int j = 0; j < JW; ++j) {
        const int ks = wave * JW + j, tap = ks / XS, sub = ks % XS, th = tap / 3, tw = tap - th * 3;
        xoff[j] = xs_a + (uint32_t)(sub * WP_XSUB + (th * WP_PW + tw) * 32) + lane_off;
    }
    auto compute = [&](int buf) {
        const uint32_t sb = (uint32_t)(buf * STAGE);
#pragma unroll
        for (int r = 0; r < WP_TH; ++r) {
            u32x4 df[OS], xf[JW];
            {
                const uint32_t da = ds_a + sb + (uint32_t)(r * WP_TW * 32) + lane_off;
                uint64_t l0, h0, l1, h1, l2 = 0, h2 = 0, l3 = 0, h3 = 0;
                if constexpr (OS == 4) {
                    asm volatile(
                        "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:512\n\t"
                        "ds_read_b64_tr_b16 %2, %8 offset:4128\n\tds_read_b64_tr_b16 %3, %8 offset:4640\n\t"
                        "ds_read_b64_tr_b16 %4, %8 offset:8256\n\tds_read_b64_tr_b16 %5, %8 offset:8768\n\t"
                        "ds_read_b64_tr_b16 %6, %8 offset:12384\n\tds_read_b64_tr_b16 %7, %8 offset:12896\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3) : "v"(da) : "memory");
                } else if constexpr (OS == 2) {
                    asm volatile(
                        "ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\t"
                        "ds_read_b64_tr_b16 %2, %4 offset:4128\n\tds_read_b64_tr_b16 %3, %4 offset:4640\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1) : "v"(da) : "memory");
                } else {
                    l1 = 0; h1 = 0;
                    asm volatile(
                        "ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(l0), "=&v"(h0) : "v"(da) : "memory");
                }
                df[0] = u32x4{(uint32_t)l0, (uint32_t)(l0 >> 32), (uint32_t)h0, (uint32_t)(h0 >> 32)};
                if constexpr (OS >= 2) df[1] = u32x4{(uint32_t)l1, (uint32_t)(l1 >> 32), (uint32_t)h1, (uint32_t)(h1 >> 32)};
                if constexpr (OS == 4) {
                    df[2] = u32x4{(uint32_t)l2, (uint32_t)(l2 >> 32), (uint32_t)h2, (uint32_t)(h2 >> 32)};
                    df[3] = u32x4{(uint32_t)l3, (uint32_t)(l3 >> 32), (uint32_t)h3, (uint32_t)(h3 >> 32)};
                }
            }
            {
                const uint32_t ro = sb + (uint32_t)(r * WP_PW * 32);
                if constexpr (JW == 6) {
                    const uint32_t a0 = xoff[0] + ro, a1 = xoff[1] + ro, a2 = xoff[2] + ro, a3 = xoff[3] + ro, a4 = xoff[4] + ro, a5 = xoff[5] + ro;
                    uint64_t l0, h0, l1, h1, l2, h2, l3, h3, l4, h4, l5, h5;
                    asm volatile(
                        "ds_read_b64_tr_b16 %0, %12\n\tds_read_b64_tr_b16 %1, %12 offset:512\n\t"
                        "ds_read_b64_tr_b16 %2, %13\n\tds_read_b64_tr_b16 %3, %13 offset:512\n\t"
                        "ds_read_b64_tr_b16 %4, %14\n\tds_read_b64_tr_b16 %5, %14 offset:512\n\t"
                        "ds_read_b64_tr_b16 %6, %15\n\tds_read_b64_tr_b16 %7, %15 offset:512\n\t"
                        "ds_read_b64_tr_b16 %8, %16\n\tds_read_b64_tr_b16 %9, %16 offset:512\n\t"
                        "ds_read_b64_tr_b16 %10, %17\n\tds_read_b64_tr_b16 %11, %17 offset:512\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3), "=&v"(l4), "=&v"(h4), "=&v"(l5), "=&v"(h5)
                        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5) : "memory");
                    xf[0] = u32x4{(uint32_t)l0, (uint32_t)(l0 >> 32), (uint32_t)h0, (uint32_t)(h0 >> 32)};
                    xf[1] = u32x4{(uint32_t)l1, (uint32_t)(l1 >> 32), (uint32_t)h1, (uint32_t)(h1 >> 32)};
                    xf[2] = u32x4{(uint32_t)l2, (uint32_t)(l2 >> 32), (uint32_t)h2, (uint32_t)(h2 >> 32)};
                    xf[3 % JW] = u32x4{(uint32_t)l3, (uint32_t)(l3 >> 32), (uint32_t)h3, (uint32_t)(h3 >> 32)};
                    xf[4 % JW] = u32x4{(uint32_t)l4, (uint32_t)(l4 >> 32), (uint32_t)h4, (uint32_t)(h4 >> 32)};
                    xf[5 % JW] = u32x4{(uint32_t)l5, (uint32_t)(l5 >> 32), (uint32_t)h5, (uint32_t)(h5 >> 32)};
                } else {
                    const uint32_t a0 = xoff[0] + ro, a1 = xoff[1 % JW] + ro, a2 = xoff[2 % JW] + ro;
                    uint64_t l0, h0, l1, h1, l2, h2;
                    asm volatile(
                        "ds_read_b64_tr_b16 %0, %6\n\tds_read_b64_tr_b16 %1, %6 offset:512\n\t"
                        "ds_read_b64_tr_b16 %2, %7\n\tds_read_b64_tr_b16 %3, %7 offset:512\n\t"
                        "ds_read_b64_tr_b16 %4, %8\n\tds_read_b64_tr_b16 %5, %8 offset:512\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2)
                        : "v"(a0), "v"(a1), "v"(a2) : "memory");
                    xf[0] = u32x4{(uint32_t)l0, (uint32_t)(l0 >> 32), (uint32_t)h0, (uint32_t)(h0 >> 32)};
                    xf[1 % JW] = u32x4{(uint32_t)l1, (uint32_t)(l1 >> 32), (uint32_t)h1, (uint32_t)(h1 >> 32)};
                    xf[2 % JW] = u32x4{(uint32_t)l2, (uint32_t)(l2 >> 32), (uint32_t)h2, (uint32_t)(h2 >> 32)};
                }
            }
#pragma unroll
            for (int i = 0; i < OS; ++i)
#pragma unroll
                for (int j = 0; j < JW; ++j) mma_chunk<T>(df[i], xf[j], acc[i][j]);
        }
    };

    // two LDS stages: the next tile is stored while the other stage is being read, one barrier per tile
    Regs R;
    if (t_begin < t_end) {
        gload(R, t_begin);
        lstore(0, R);
        __syncthreads();
        for (int t = t_begin; t < t_end; ++t) {
            const bool more = t + 1 < t_end;
            const int cur = (t - t_begin) & 1;
            if (more) gload(R, t + 1);                        // next tile's global loads fly during this tile's MFMAs
            compute(cur);
            if (more) lstore(cur ^ 1, R);
            __syncthreads();
        }
    }
    // partial gradient block -> workspace slab [Cout][K]: row o, column k = tap * Cin + cs * 64 + sub * 16 + (lane & 15)
    float* wsb = g.ws + (size_t)blockIdx.z * g.ws_slab;
#pragma unroll
    for (int j = 0; j < JW; ++j) {
        const int ks = wave * JW + j, tap = ks / XS, sub = ks % XS;
        const int k = tap * g.Cin + cs * (16 * XS) + sub * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < OS; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + i * 16 + (lane >> 4) * 4 + r;
                if (o < g.Cout) wsb[(size_t)o * g.K + k] = acc[i][j][r];
            }
    }
}

}  // namespace mfx
using namespace mfx;

int g_opt_wgrad_patch = 1;          // option "wgrad_patch": 0 = off
int g_opt_wgrad_patch_blocks = 256; // option "wgrad_patch_blocks": target workgroup count
int g_opt_wgrad_patch_waves = 12;   // option "wgrad_patch_waves": 6 (64 x 96 block per wave) or 12 (64 x 48)

// 3x3 / s1 / p1, Cin % 64 == 0: returns 1 if launched (partials in g.ws, *nslab slabs), 0 to fall through
template <typename T> static int conv_wgrad_patch_t(const void* x, const void* dy, WgradGeom& g, void* workspace, size_t workspace_bytes, int* nslab_out, hipStream_t st);
int try_conv_wgrad_patch(const void* x, const void* dy, WgradGeom& g, int dtype, void* workspace, size_t workspace_bytes, int* nslab_out, hipStream_t st) {
    if (dtype == MFX_F16) return conv_wgrad_patch_t<half_t>(x, dy, g, workspace, workspace_bytes, nslab_out, st);
    return conv_wgrad_patch_t<bf16_t>(x, dy, g, workspace, workspace_bytes, nslab_out, st);
}
template <typename T> static int conv_wgrad_patch_t(const void* x, const void* dy, WgradGeom& g, void* workspace, size_t workspace_bytes, int* nslab_out, hipStream_t st) {
    if (!g_opt_wgrad_patch || !workspace || g.direct) return 0;
    if (g.kh != 3 || g.kw != 3 || g.stride != 1 || g.pad_h != 1 || g.pad_w != 1 || g.dil_w != 1 || g.Ho != g.H || g.Wo != g.W) return 0;
    if ((g.Ck % 64 != 0 && g.Ck != 32 && g.Ck != 16) || g.x_pixstride != g.Ck || g.ldy % 8 != 0 || g.Cout % 8 != 0 || g.M < 2048) return 0;
    if (g.Ck < 64 && g.Cout > 32) return 0;
    WpGeom w;
    w.B = g.B; w.H = g.H; w.W = g.W; w.Cin = g.Ck; w.Cout = g.Cout; w.ldy = g.ldy; w.K = g.K;
    w.tiles_x = (g.W + WP_TW - 1) / WP_TW; w.tiles_y = (g.H + WP_TH - 1) / WP_TH; w.ntiles = w.tiles_x * w.tiles_y * g.B;
    const int xs = g.Ck >= 64 ? 4 : g.Ck / 16;
    const int os = g.Cout > 32 ? 4 : (g.Cout > 16 || xs == 4 ? 2 : 1);
    const int otiles = (g.Cout + 16 * os - 1) / (16 * os), slices = g.Ck / (16 * xs);
    const long ws_slab = (long)g.Cout * g.K;
    int nslab = std::max(1, (xs < 4 ? 8 : 1) * g_opt_wgrad_patch_blocks / (otiles * slices));    // small-channel form: ~7 of its workgroups fit a CU
    nslab = (int)std::min<long>(nslab, (long)(workspace_bytes / sizeof(float)) / ws_slab);
    nslab = std::min(nslab, std::max(1, w.ntiles / 4));       // at least four tiles per slab: the 147 KB epilogue must amortise
    if (nslab < 1) return 0;
    w.tiles_per_slab = (w.ntiles + nslab - 1) / nslab;
    nslab = (w.ntiles + w.tiles_per_slab - 1) / w.tiles_per_slab;
    w.ws = reinterpret_cast<float*>(workspace); w.ws_slab = ws_slab;
    g.ws = w.ws; g.ws_ld = g.K; g.ws_slab = ws_slab;
    const dim3 grid(otiles, slices, nslab);
    const int nw = g_opt_wgrad_patch_waves == 12 ? 12 : 6;
    if (xs < 4) {
        // the 16- and 32-channel full-resolution layers (level0 / level1 of the DLA base): one slice holds the whole input;
        // memory-bound (10 KB staged per 128 pixels for 36-72 MFMAs), so many slabs and small workgroups
#define WP_LAUNCH_SMALL(OS_, NW_, XS_)                                                                                              \
        do {                                                                                                                        \
            constexpr int smem = 2 * (XS_ * WP_XSUB + OS_ * WP_DSUB);                                                               \
            hipLaunchKernelGGL((conv_wgrad_patch_kernel<T, OS_, NW_, XS_>), grid, dim3(64 * NW_), smem, st, (const T*)x, (const T*)dy, w); \
        } while (0)
        if (xs == 1 && os == 1) WP_LAUNCH_SMALL(1, 3, 1);
        else if (xs == 1) WP_LAUNCH_SMALL(2, 3, 1);
        else if (os == 1) WP_LAUNCH_SMALL(1, 6, 2);
        else WP_LAUNCH_SMALL(2, 6, 2);
#undef WP_LAUNCH_SMALL
        MFX_HIP_CHECK(hipGetLastError());
        *nslab_out = nslab;
        return 1;
    }
    if (os == 4) {
        constexpr int smem = 2 * (4 * WP_XSUB + 4 * WP_DSUB);
        if (nw == 12) {
            auto k = conv_wgrad_patch_kernel<T, 4, 12>;
            static bool a12 = false; if (!a12) { MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem)); a12 = true; }
            hipLaunchKernelGGL(k, grid, dim3(768), smem, st, (const T*)x, (const T*)dy, w);
        } else {
            auto k = conv_wgrad_patch_kernel<T, 4, 6>;
            static bool a6 = false; if (!a6) { MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem)); a6 = true; }
            hipLaunchKernelGGL(k, grid, dim3(384), smem, st, (const T*)x, (const T*)dy, w);
        }
    } else {
        constexpr int smem = 2 * (4 * WP_XSUB + 2 * WP_DSUB);
        auto k = conv_wgrad_patch_kernel<T, 2, 6>;
        static bool a2 = false; if (!a2) { MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem)); a2 = true; }
        hipLaunchKernelGGL(k, grid, dim3(384), smem, st, (const T*)x, (const T*)dy, w);
    }
    MFX_HIP_CHECK(hipGetLastError());
    *nslab_out = nslab;
    return 1;
}

// ======================================================================================================================
// Stem (7x7 / stride 1 / pad 3, 3 -> 16 channels at full resolution; dla_dcn.py:268-272): weight gradient in the forward
// stem's operand layout -- the zero-padded NHWC4 image, super-taps of two 4-channel pixels: dW[o][th][dx*4 + c] with
// dx = 0..7 (dx = 7 and c = 3 are padding columns the caller drops).
// For a fixed kernel row th the implicit-im2col operand is Toeplitz: column (dx, c) of output pixel x is element
// 4*(x + dx) + c of the FLAT image row, i.e. the 32 columns of pixel x are the 64 bytes starting at byte 8*x.  A transposed
// read takes four 8-byte pieces per pixel row at ANY 8-byte aligned address, so the operand fragments come straight from the
// flat rows in LDS (320 bytes per row for 32 output pixels) -- no im2col, and the 126 MB of dy + 31 MB of image are read
// from memory once (the generic kernel re-read the image 28 times through a K = 224, Cout = 16 tile: 774 us at B = 8).
// ======================================================================================================================
namespace mfx {

constexpr int SW_TH = 8, SW_TW = 32, SW_ROWS = SW_TH + 6, SW_ROWB = (SW_TW + 8) * 8;     // 14 image rows x 320 bytes
constexpr int SW_XBYTES = SW_ROWS * SW_ROWB, SW_DBYTES = SW_TH * SW_TW * 32, SW_STAGE = SW_XBYTES + SW_DBYTES;
constexpr int SW_K = 224;

struct SwGeom { int B, H, W, Hp, Wp, tiles_x, tiles_y, ntiles, tiles_per_block; float* ws; };

template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const T* __restrict__ xp, const T* __restrict__ dy, SwGeom g) {
    __shared__ __attribute__((aligned(16))) char lds[2 * SW_STAGE];
    static_assert(2 * SW_STAGE >= 14 * 64 * 16, "the cross-wave reduction reuses the staging buffers");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t_begin = blockIdx.x * g.tiles_per_block, t_end = min(t_begin + g.tiles_per_block, g.ntiles);
    constexpr int XCH = SW_ROWS * (SW_ROWB / 16);                            // 280 image chunks (and 512 dy chunks) of 16 bytes
    static_assert(SW_TH * SW_TW * 2 == 512, "two dy chunks per thread");
    struct Regs { u32x4 xr[2]; u32x4 dr[2]; };
    auto gload = [&](Regs& r, int t) {
        const int tx = t % g.tiles_x; int q = t / g.tiles_x;
        const int ty = q % g.tiles_y, b = q / g.tiles_y;
        const int x0 = tx * SW_TW, y0 = ty * SW_TH;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int id = tid + u * 256;
            const int pr = id / (SW_ROWB / 16), c = id - pr * (SW_ROWB / 16);
            u32x4 z = {0u, 0u, 0u, 0u};
            if (id < XCH && y0 + pr < g.Hp && x0 + 2 * c + 1 < g.Wp)
                z = *reinterpret_cast<const u32x4*>(xp + (((size_t)b * g.Hp + y0 + pr) * g.Wp + x0 + 2 * c) * 4);
            r.xr[u] = z;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int id = tid + u * 256;
            const int px = id >> 1, c8 = id & 1;
            const int oy = y0 + px / SW_TW, ox = x0 + (px % SW_TW);
            u32x4 z = {0u, 0u, 0u, 0u};
            if (oy < g.H && ox < g.W) z = *reinterpret_cast<const u32x4*>(dy + (((size_t)b * g.H + oy) * g.W + ox) * 16 + c8 * 8);
            r.dr[u] = z;
        }
    };
    auto lstore = [&](int buf, const Regs& r) {
        char* xs = lds + buf * SW_STAGE;
        char* ds = xs + SW_XBYTES;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int id = tid + u * 256;
            if (id < XCH) *reinterpret_cast<u32x4*>(xs + id * 16) = r.xr[u];
            *reinterpret_cast<u32x4*>(ds + id * 16) = r.dr[u];
        }
    };
    f32x4 acc[7][2];
#pragma unroll
    for (int i = 0; i < 7; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int l16 = lane & 15, gq = lane >> 4;
    const uint32_t base = (uint32_t)(uintptr_t)lds;
    const uint32_t d_off = (uint32_t)((4 * gq + (l16 >> 2)) * 32 + (l16 & 3) * 8);       // dy: 32-byte pixel rows
    const uint32_t x_off = (uint32_t)((4 * gq + (l16 >> 2)) * 8 + (l16 & 3) * 8);        // image: the Toeplitz rows start 8 bytes apart
    auto tr2 = [](uint32_t a, uint32_t second) {
        uint64_t l, h;
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(l), "=&v"(h) : "v"(a), "v"(a + second) : "memory");
        return u32x4{(uint32_t)l, (uint32_t)(l >> 32), (uint32_t)h, (uint32_t)(h >> 32)};
    };
    auto compute = [&](int buf) {
        const uint32_t xs = base + (uint32_t)(buf * SW_STAGE), ds = xs + SW_XBYTES;
#pragma unroll
        for (int rr = 0; rr < SW_TH / 4; ++rr) {
            const int r = wave + 4 * rr;                                                  // this wave's output rows of the tile
            const u32x4 df = tr2(ds + (uint32_t)(r * SW_TW * 32) + d_off, 16 * 32);
#pragma unroll
            for (int th = 0; th < 7; ++th) {
                const uint32_t xa = xs + (uint32_t)((r + th) * SW_ROWB) + x_off;
                const u32x4 x0 = tr2(xa, 16 * 8), x1 = tr2(xa + 32, 16 * 8);
                mma_chunk<T>(df, x0, acc[th][0]);
                mma_chunk<T>(df, x1, acc[th][1]);
            }
        }
    };
    Regs R;
    if (t_begin < t_end) {
        gload(R, t_begin);
        lstore(0, R);
        __syncthreads();
        for (int t = t_begin; t < t_end; ++t) {
            const bool more = t + 1 < t_end;
            const int cur = (t - t_begin) & 1;
            if (more) gload(R, t + 1);
            compute(cur);
            if (more) lstore(cur ^ 1, R);
            __syncthreads();
        }
    }
    // the four waves' partial blocks -> one [16][224] slab per workgroup (summed through LDS, one wave after the other)
    float* red = reinterpret_cast<float*>(lds);                                           // [14][64 lanes][4]
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 7; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    f32x4* p = reinterpret_cast<f32x4*>(red + ((i * 2 + c) * 64 + lane) * 4);
                    f32x4 v = acc[i][c];
                    if (w) { const f32x4 q = *p; v[0] += q[0]; v[1] += q[1]; v[2] += q[2]; v[3] += q[3]; }
                    *p = v;
                }
        }
        __syncthreads();
    }
    float* slab = g.ws + (size_t)blockIdx.x * (16 * SW_K);
    for (int e = tid; e < 14 * 64 * 4; e += 256) {
        const int reg = e & 3, ln = (e >> 2) & 63, blk = e >> 8;                          // blk = th * 2 + column tile
        const int o = (ln >> 4) * 4 + reg, col = (blk >> 1) * 32 + (blk & 1) * 16 + (ln & 15);
        slab[o * SW_K + col] = red[e];
    }
}

__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ ws, int nslab, int n, float* __restrict__ out) {
    __shared__ float part[4][64];                                     // 64 outputs per workgroup, a quarter of the slabs per wave
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6, i = blockIdx.x * 64 + col;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        int z = q;
        for (; z + 12 < nslab; z += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += ws[(size_t)(z + 4 * u) * n + i];
        }
        for (; z < nslab; z += 4) a[0] += ws[(size_t)z * n + i];
    }
    part[q][col] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (q == 0 && i < n) out[i] = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
}

}  // namespace mfx

int g_opt_stem_wgrad_blocks = 512;      // option "stem_wgrad_blocks"

extern "C" int mfx_stem_wgrad_bf16(const void* xp, const void* dy, float* dw, int B, int H, int W, int Hp, int Wp, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    return mfx_stem_wgrad_16(xp, dy, dw, B, H, W, Hp, Wp, MFX_BF16, workspace, workspace_bytes, stream);
}

extern "C" int mfx_stem_wgrad_16(const void* xp, const void* dy, float* dw, int B, int H, int W, int Hp, int Wp, int dtype, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (dtype != MFX_BF16 && dtype != MFX_F16) return mfx_fail(MFX_ERR_ARG, "stem_wgrad: 16-bit activations (bf16 / fp16) only");
    if (!xp || !dy || !dw || !workspace) return mfx_fail(MFX_ERR_ARG, "stem_wgrad: null pointer");
    if (Hp < H + 6 || Wp < W + 8 || (Wp & 1)) return mfx_fail(MFX_ERR_ARG, "stem_wgrad: the padded image must be (H+6) x (W+8), even width");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SwGeom g;
    g.B = B; g.H = H; g.W = W; g.Hp = Hp; g.Wp = Wp;
    g.tiles_x = (W + SW_TW - 1) / SW_TW; g.tiles_y = (H + SW_TH - 1) / SW_TH; g.ntiles = g.tiles_x * g.tiles_y * B;
    if (g.ntiles == 0) { MFX_HIP_CHECK(mfx::zero_async(dw, sizeof(float) * 16 * SW_K, st)); return MFX_OK; }
    int nb = std::min(g.ntiles, std::max(1, g_opt_stem_wgrad_blocks));
    nb = (int)std::min<size_t>((size_t)nb, workspace_bytes / (sizeof(float) * 16 * SW_K));
    if (nb < 1) return mfx_fail(MFX_ERR_WORKSPACE, "stem_wgrad: workspace too small");
    g.tiles_per_block = (g.ntiles + nb - 1) / nb;
    nb = (g.ntiles + g.tiles_per_block - 1) / g.tiles_per_block;
    g.ws = reinterpret_cast<float*>(workspace);
    if (dtype == MFX_BF16) hipLaunchKernelGGL(stem_wgrad_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)xp, (const bf16_t*)dy, g);
    else hipLaunchKernelGGL(stem_wgrad_kernel<half_t>, dim3(nb), dim3(256), 0, st, (const half_t*)xp, (const half_t*)dy, g);
    hipLaunchKernelGGL(slab_sum_kernel, dim3((16 * SW_K + 63) / 64), dim3(256), 0, st, (const float*)g.ws, nb, 16 * SW_K, dw);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}
