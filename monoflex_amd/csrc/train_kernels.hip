// Training-path kernels (NHWC): weight/bias gradients, train-mode BatchNorm (+activation, +residual) forward and
// backward, max-pool / depthwise-deconv backward, zero-insertion for stride-2 data gradients.
// Data gradients of convolutions reuse the forward implicit-GEMM kernels (flipped / transposed weights packed by the
// host), so no separate dgrad kernel exists.  All reductions accumulate in fp32.
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"
#include "fill.h"
#include "wgrad.h"
#include <algorithm>

namespace mfx {

static inline int cdivt(long a, long b) { return (int)((a + b - 1) / b); }
// elementwise BN passes: at most ~8 workgroups per CU, each streaming many chunks (amortises the per-workgroup table)
#define BN_APPLY_GRID(total) dim3((unsigned)(cdivt((total), 256) < 2048 ? cdivt((total), 256) : 2048))
#define WR_GRID(total) dim3((unsigned)(cdivt((total), 64) < 8192 ? cdivt((total), 64) : 8192))
#define TR_GRID(total) dim3((unsigned)(cdivt((total), 256) < 16384 ? cdivt((total), 256) : 16384))

template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void load4<half_t>(const half_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    const f16x2 a = __builtin_bit_cast(f16x2, t.x), b = __builtin_bit_cast(f16x2, t.y);
    v[0] = (float)a[0]; v[1] = (float)a[1]; v[2] = (float)b[0]; v[3] = (float)b[1];
}

// ------------------------------------------------------------------------------------------------
// conv weight gradient: dW[o][tap][c] = sum_m dy[m][o] * x[pixel(m,tap)][c]       (fp32 [Cout][taps][Ck])
// Block = 64 k x 64 o tile over a slab of pixels; 256 threads x 4x4 register blocks; fp32 atomics on the small result.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wgrad_add(float* dw, const WgradGeom& g, int o, int k, float v) {
    if (g.ws) { g.ws[(size_t)blockIdx.z * g.ws_slab + (size_t)o * g.ws_ld + k] = v; return; }
    if (!g.oihw) { unsafeAtomicAdd(dw + (size_t)o * g.K + k, v); return; }
    const int tap = k / g.Ck, c = k - tap * g.Ck;
    if (o < g.Cout_out && c < g.Cin_out) unsafeAtomicAdd(dw + ((size_t)o * g.Cin_out + c) * (g.kh * g.kw) + tap, v);
}

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, WgradGeom g, float* __restrict__ dw) {
    constexpr int MCH = 16;
    __shared__ float cs[MCH][64 + 4];
    __shared__ float gs[MCH][64 + 4];
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * 64, o0 = blockIdx.y * 64;
    const int m_begin = blockIdx.z * g.m_per_block, m_end = min(m_begin + g.m_per_block, g.M);
    const int tk = (tid & 15) * 4, to = (tid >> 4) * 4;
    const int sr = tid >> 4, sc = (tid & 15) * 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int k = k0 + sc;
    const bool k_ok = k < g.K;
    const int tap = k_ok ? k / g.Ck : 0, c = k - tap * g.Ck;          // Ck % 4 == 0: the 4 k's share a tap
    const int th = tap / g.kw, tw = tap - th * g.kw;
    const bool o_ok = o0 + sc < g.Cout;
    const int hw = g.Ho * g.Wo;
    for (int mb = m_begin; mb < m_end; mb += MCH) {
        const int m = mb + sr;
        float cv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < m_end) {
            if (o_ok) load4<T>(dy + (size_t)m * g.ldy + o0 + sc, gv);
            if (k_ok && g.direct) load4<T>(x + (size_t)m * g.x_pixstride + k, cv);
            else if (k_ok) {
                const int b = m / hw, rem = m - b * hw, oh = rem / g.Wo, ow = rem - oh * g.Wo;
                const int ih = oh * g.stride - g.pad_h + th, iw = ow * g.stride - g.pad_w + tw * g.dil_w;
                if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
                    load4<T>(x + ((size_t)(b * g.H + ih) * g.W + iw) * g.x_pixstride + c, cv);
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) { cs[sr][sc + e] = cv[e]; gs[sr][sc + e] = gv[e]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MCH; ++r) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&cs[r][tk]);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(&gs[r][to]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * bb[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + tk + i < g.K && o0 + to + j < g.Cout) wgrad_add(dw, g, o0 + to + j, k0 + tk + i, acc[i][j]);
}

// ------------------------------------------------------------------------------------------------
// conv weight gradient on the matrix cores (bf16 activations): dW[o][k] = sum_m dy[m][o] * A[m][k], A = implicit im2col.
// The reduction runs over pixels, which is the SLOW axis of both NHWC operands, while an MFMA fragment wants 8
// consecutive reduction elements per lane.  Each k-step therefore stages 32 pixels of both operands in LDS TRANSPOSED
// ([channel][pixel]): a thread loads the same 8-channel chunk of two adjacent pixels (2 x 16 B, coalesced along channels)
// and writes eight 4-byte (pixel, pixel+1) pairs; fragments are then plain ds_read_b128.  Workgroup = 4 waves (2 x 2),
// tile 64 o x 64 k, slab of m_per_block pixels, fp32 atomics on the small result.  Needs Ck % 8 == 0.
// ------------------------------------------------------------------------------------------------
constexpr int kWgRow = 80;                                   // LDS row: 32 pixels x 2 B + 16 B pad

// BT = 64: tile 64 o x 64 k, threads 0..127 stage dy and 128..255 the im2col operand, 4 MFMAs per wave and step.
// BT = 128: tile 128 x 128, every thread stages one chunk pair of each operand, 16 MFMAs per wave and step: with 64-wide tiles
// the head-trunk weight gradients alone re-read ~20 GB of operand tiles from L2 per step; 128-wide tiles halve that.
// Global loads run TWO steps ahead of their LDS store (two register sets): one step of MFMAs is far shorter than an L2 round
// trip and the large tiles leave only a few waves per SIMD to hide it.
template <typename T, int BT>                                 // T = bf16_t or half_t: the staging moves 16-bit words, only the MFMA knows the format
__global__ __launch_bounds__(256) void conv_wgrad_mfma_kernel(const T* __restrict__ x, const T* __restrict__ dy, WgradGeom g,
                                                              float* __restrict__ dw) {
    constexpr int NCH = BT / 8, FR = BT / 32;                 // 16-byte chunks per tile row; 16-row fragments per wave and operand
    __shared__ __attribute__((aligned(16))) char lds[2][2][BT * kWgRow];      // [buffer][0 = dy^T, 1 = A^T][BT rows]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo = wave >> 1, wk = wave & 1;
    const int k0 = blockIdx.x * BT, o0 = blockIdx.y * BT;
    const int m_begin = blockIdx.z * g.m_per_block, m_end = min(m_begin + g.m_per_block, g.M);

    // loader roles; item = (pixel pair pp, 8-channel chunk c8)
    const bool do_dy = BT == 128 || tid < 128, do_a = BT == 128 || tid >= 128;
    const int lt = BT == 128 ? tid : (tid & 127), pp = lt / NCH, c8 = (lt % NCH) * 8;
    const int kk = k0 + c8;                                   // first of this thread's 8 consecutive k (one tap: Ck % 8 == 0)
    const bool a_ok = do_a && kk < g.K, d_ok = do_dy && o0 + c8 < g.Cout;
    const int tap = a_ok ? kk / g.Ck : 0, ch = kk - tap * g.Ck;
    const int th = tap / g.kw, tw = tap - th * g.kw;
    const int hw = g.Ho * g.Wo;
    // (b, oh, ow) of the pixel pair this thread loads next, advanced by 32 pixels per load
    int m = m_begin + 2 * pp;
    int pb = m / hw, rem = m - pb * hw, poh = rem / g.Wo, pow_ = rem - poh * g.Wo;
    auto advance = [&](int& b_, int& oh_, int& ow_, int n) {
        ow_ += n;
        while (ow_ >= g.Wo) { ow_ -= g.Wo; if (++oh_ == g.Ho) { oh_ = 0; ++b_; } }
    };
    struct Regs { u32x4 d0, d1, a0, a1; };
    auto gload = [&](Regs& r) {                               // chunks of pixels m and m+1 (zero when masked), then m += 32
        r.d0 = u32x4{0u, 0u, 0u, 0u}; r.d1 = r.d0; r.a0 = r.d0; r.a1 = r.d0;
        if (d_ok) {
            if (m < m_end) r.d0 = *reinterpret_cast<const u32x4*>(dy + (size_t)m * g.ldy + o0 + c8);
            if (m + 1 < m_end) r.d1 = *reinterpret_cast<const u32x4*>(dy + (size_t)(m + 1) * g.ldy + o0 + c8);
        }
        if (a_ok && g.direct) {
            if (m < m_end) r.a0 = *reinterpret_cast<const u32x4*>(x + (size_t)m * g.x_pixstride + kk);
            if (m + 1 < m_end) r.a1 = *reinterpret_cast<const u32x4*>(x + (size_t)(m + 1) * g.x_pixstride + kk);
        } else if (a_ok) {
            int b2 = pb, oh2 = poh, ow2 = pow_;
            if (m < m_end) {
                const int ih = poh * g.stride - g.pad_h + th, iw = pow_ * g.stride - g.pad_w + tw * g.dil_w;
                if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
                    r.a0 = *reinterpret_cast<const u32x4*>(x + ((size_t)(pb * g.H + ih) * g.W + iw) * g.x_pixstride + ch);
            }
            if (m + 1 < m_end) {
                advance(b2, oh2, ow2, 1);
                const int ih = oh2 * g.stride - g.pad_h + th, iw = ow2 * g.stride - g.pad_w + tw * g.dil_w;
                if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
                    r.a1 = *reinterpret_cast<const u32x4*>(x + ((size_t)(b2 * g.H + ih) * g.W + iw) * g.x_pixstride + ch);
            }
        }
        m += 32; advance(pb, poh, pow_, 32);
    };
    auto tstore = [&](char* base, const u32x4& r0, const u32x4& r1) {       // transposed: row = channel, (pixel, pixel+1) pairs
        const uint32_t a[4] = {r0.x, r0.y, r0.z, r0.w}, b[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<uint32_t*>(base + (2 * q) * kWgRow) = (a[q] & 0xffffu) | (b[q] << 16);
            *reinterpret_cast<uint32_t*>(base + (2 * q + 1) * kWgRow) = (a[q] >> 16) | (b[q] & 0xffff0000u);
        }
    };
    auto lstore = [&](int buf, const Regs& r) {
        if (do_dy) tstore(lds[buf][0] + c8 * kWgRow + pp * 4, r.d0, r.d1);
        if (do_a) tstore(lds[buf][1] + c8 * kWgRow + pp * 4, r.a0, r.a1);
    };

    f32x4 acc[FR][FR];
#pragma unroll
    for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int j = 0; j < FR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frag = (lane & 15) * kWgRow + (lane >> 4) * 16;
    auto compute = [&](int buf) {
        u32x4 af[FR], bf[FR];
#pragma unroll
        for (int i = 0; i < FR; ++i) af[i] = *reinterpret_cast<const u32x4*>(lds[buf][0] + (wo * (BT / 2) + i * 16) * kWgRow + frag);
#pragma unroll
        for (int j = 0; j < FR; ++j) bf[j] = *reinterpret_cast<const u32x4*>(lds[buf][1] + (wk * (BT / 2) + j * 16) * kWgRow + frag);
#pragma unroll
        for (int i = 0; i < FR; ++i)
#pragma unroll
            for (int j = 0; j < FR; ++j) mma_chunk<T>(af[i], bf[j], acc[i][j]);
    };

    // steps s = 0 .. ns-1; registers R[s & 1] hold the data of step s once loaded (two steps ahead of its LDS store)
    const int ns = (m_end - m_begin + 31) / 32;
    Regs R0, R1;
    gload(R0);                                               // step 0
    lstore(0, R0);
    if (ns > 1) gload(R1);                                   // step 1
    if (ns > 2) gload(R0);                                   // step 2
    __syncthreads();
    int s = 0;
    for (; s + 2 <= ns; s += 2) {                            // unrolled by two: static register sets
        compute(0);                                          // step s   (buffer 0)
        if (s + 1 < ns) lstore(1, R1);                       // step s+1 -> buffer 1
        if (s + 3 < ns) gload(R1);                           // step s+3
        __syncthreads();
        if (s + 1 < ns) compute(1);                          // step s+1 (buffer 1)
        if (s + 2 < ns) lstore(0, R0);                       // step s+2 -> buffer 0
        if (s + 4 < ns) gload(R0);                           // step s+4
        __syncthreads();
    }
    if (s < ns) compute(0);                                  // odd tail
    // D: col (lane&15) = k, row (lane>>4)*4 + r = o
#pragma unroll
    for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int j = 0; j < FR; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + wo * (BT / 2) + i * 16 + (lane >> 4) * 4 + r, k = k0 + wk * (BT / 2) + j * 16 + (lane & 15);
                if (o < g.Cout && k < g.K) wgrad_add(dw, g, o, k, acc[i][j][r]);
            }
}

// sums the per-slab partial tiles of conv_wgrad_mfma_kernel and writes the gradient in its final layout
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int slabs, long ws_slab, int ws_ld, WgradGeom g, float* __restrict__ dw) {
    // a workgroup owns 64 consecutive output elements; its four waves each sum a quarter of the slabs (four independent
    // accumulators per thread), then the quarters meet in LDS.  (One thread per element walking all slabs left the pass
    // latency-bound and the chip under-filled: 144 workgroups for a 64x576 gradient, 10 us per layer, 84 layers per step.)
    __shared__ float part[4][64];
    const long total = (long)g.Cout * g.K;
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (long i0 = (long)blockIdx.x * 64; i0 < total; i0 += (long)gridDim.x * 64) {
        const long i = i0 + col;
        float s = 0.f;
        int o = 0, k = 0;
        if (i < total) {
            o = (int)(i / g.K); k = (int)(i - (long)o * g.K);
            const float* p = ws + (size_t)o * ws_ld + k;
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            int z = q;
            for (; z + 12 < slabs; z += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] += p[(size_t)(z + 4 * u) * ws_slab];
            }
            for (; z < slabs; z += 4) a[0] += p[(size_t)z * ws_slab];
            s = (a[0] + a[1]) + (a[2] + a[3]);
        }
        part[q][col] = s;
        __syncthreads();
        if (q == 0 && i < total) {
            s = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
            if (!g.oihw) dw[(size_t)o * g.K + k] = s;
            else {
                const int tap = k / g.Ck, c = k - tap * g.Ck;
                if (o < g.Cout_out && c < g.Cin_out) dw[((size_t)o * g.Cin_out + c) * (g.kh * g.kw) + tap] = s;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing for the training step: fp32 OIHW parameter -> K-contiguous [rows_pad][K_pad] (+ fragment-major copy) of the
// compute dtype in one launch.  mode 0: forward operand, row = o, k = tap*Cin + c.  mode 1: data-gradient operand (the
// flipped, in/out-swapped kernel): row = c, k = tap'*ck + o with tap' the 180-degree rotated tap, ck = channels of dy.
// (As torch ops this was ~8 small launches per conv in forward and ~12 in backward, 60 convs per step.)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int kh, int kw, int mode, T* __restrict__ packed,
                                        T* __restrict__ frag, int rows_pad, int K_pad, int ck) {
    constexpr int E = ElemTraits<T>::ELEMS;
    const long total = (long)rows_pad * K_pad;
    const int taps = kh * kw;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / K_pad), k = (int)(i - (long)n * K_pad);
        const int tap = k / ck, cc = k - tap * ck;
        float v = 0.f;
        if (tap < taps) {
            if (mode == 0) { if (n < Cout && cc < Cin) v = w[((size_t)n * Cin + cc) * taps + tap]; }
            else if (n < Cin && cc < Cout) v = w[((size_t)cc * Cin + n) * taps + (taps - 1 - tap)];
        }
        ElemTraits<T>::store(packed + i, v);
        if (frag) {
            const size_t f = (((size_t)(n >> 4) * (K_pad / (4 * E)) + k / (4 * E)) * 4 + (k % (4 * E)) / E) * (16 * E) + (n & 15) * E + k % E;
            ElemTraits<T>::store(frag + f, v);
        }
    }
}

// All conv operands of a training step in ONE launch.  The operands are cut into chunks of PACK_CHUNK elements; workgroup b
// packs chunk b, which belongs to descriptor d with prefix[d] <= b < prefix[d+1] (prefix counts CHUNKS; wave-uniform binary
// search over ~150 descriptors, descriptor fetched once per workgroup); same element rule as pack_conv_weight_kernel.
// (One launch per operand was 148 launches of ~5 us per step; a per-element search made the single launch 330 us.)
//
// The chunk's source values go through LDS: an operand row is a PERMUTATION of parameter memory -- forward rows interleave
// (channel, tap) -> (tap, channel), data-gradient rows gather one input channel's taps from every output channel, 9 floats out of
// every Cin * 9 -- so element-wise gathers touched one cache line per lane (mode 1) or a ninth of each line per pass (mode 0), and
// every line was fetched once per tap: 308 us per step for 84 MB of parameters.  Here the workgroup first copies the parameter
// range its rows need into LDS with consecutive lanes on consecutive addresses (whole rows for mode 0; for mode 1 the run of
// (rows x taps) floats of every output channel), then every lane forms 8 consecutive operand elements from LDS and stores them as
// 16-byte pieces (packed copy and fragment-major copy alike).  Rows that do not fit the staging buffer take the element-wise path.
constexpr int PACK_CHUNK = 2048;
constexpr int PACK_STAGE = 10240;                              // floats of LDS staging (two rows of a 512-channel 3x3 operand + slack)
template <typename T> __device__ __forceinline__ void pack_store8(T* p, const float (&v)[8]) { *reinterpret_cast<u32x4*>(p) = ElemTraits<T>::pack(v); }
template <> __device__ __forceinline__ void pack_store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weight_batched_kernel(const mfx_pack_desc* __restrict__ descs, const long long* __restrict__ prefix, int n) {
    constexpr int E = ElemTraits<T>::ELEMS;
    __shared__ float stage[PACK_STAGE];
    __shared__ int below[4];
    const long chunk = blockIdx.x;
    // descriptor of this chunk = (number of prefix entries <= chunk) - 1: every lane tests one entry and the four waves' ballots are
    // counted -- ONE round trip to the table instead of the eight dependent ones of a binary search, which were most of a workgroup's
    // life (20 k workgroups of ~10 us on 1024 resident slots)
    int cnt = 0;
    for (int base = 0; base < n; base += 256) {
        const int idx = base + (int)threadIdx.x;
        cnt += __popcll(__ballot(idx < n && (long)prefix[idx] <= chunk));
    }
    if ((threadIdx.x & 63) == 0) below[threadIdx.x >> 6] = cnt;
    __syncthreads();
    const int lo = below[0] + below[1] + below[2] + below[3] - 1;
    const mfx_pack_desc d = descs[lo];
    const long total = (long)d.rows_pad * d.K_pad, j0 = (chunk - (long)prefix[lo]) * PACK_CHUNK;
    const long j1 = j0 + PACK_CHUNK < total ? j0 + PACK_CHUNK : total;
    const int taps = d.kh * d.kw;
    T* packed = reinterpret_cast<T*>(d.packed);
    T* frag = reinterpret_cast<T*>(d.frag);
    // operand rows of this chunk and the parameter rows behind them (mode 0: output channels, mode 1: input channels)
    const int r0 = (int)(j0 / d.K_pad), r1 = (int)((j1 - 1) / d.K_pad);
    const int src_rows = d.mode == 0 ? d.Cout : d.Cin;
    const int rv = (r1 < src_rows ? r1 : src_rows - 1) - r0 + 1;          // valid rows (<= 0: the chunk is all padding)
    const int run = rv > 0 ? rv * taps : 0;                                // mode 1: floats per output channel
    const long need = rv <= 0 ? 0 : (d.mode == 0 ? (long)rv * d.Cin * taps : (long)d.Cout * run);
    const bool staged = need <= PACK_STAGE && (d.K_pad & 7) == 0;
    if (staged && need > 0) {
        // (eight loads in flight per lane before the first LDS write: one load per iteration left every workgroup waiting on ~16
        // dependent round trips)
        const float* src = d.w + (size_t)r0 * (d.mode == 0 ? d.Cin * taps : taps);
        for (int base = threadIdx.x; base < (int)need; base += 256 * 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256;
                t[u] = 0.f;
                if (i < (int)need) {
                    if (d.mode == 0) t[u] = src[i];
                    else { const int o = i / run, rem = i - o * run; t[u] = src[(size_t)o * d.Cin * taps + rem]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256;
                if (i < (int)need) stage[i] = t[u];
            }
        }
    }
    __syncthreads();
    if (staged) {
        const long j = j0 + (long)threadIdx.x * 8;                             // 8 consecutive elements: one row, one tap (ck % E == 0, K_pad % 8 == 0)
        if (j >= j1) return;
        const int nrow = (int)(j / d.K_pad), k = (int)(j - (long)nrow * d.K_pad);
        float v[8];
#pragma unroll
        for (int h = 0; h < 8; h += E) {                                       // (fp32 operands: ck is a multiple of 4 only, so per 4-element half)
            const int kk = k + h, tap = kk / d.ck, cc = kk - tap * d.ck;
#pragma unroll
            for (int e = 0; e < E && e < 8; ++e) {
                float x = 0.f;
                if (tap < taps && nrow - r0 < rv) {
                    if (d.mode == 0) { if (cc + e < d.Cin) x = stage[((nrow - r0) * d.Cin + cc + e) * taps + tap]; }
                    else if (cc + e < d.Cout) x = stage[(cc + e) * run + (nrow - r0) * taps + (taps - 1 - tap)];
                }
                v[h + e] = x;
            }
        }
        pack_store8<T>(packed + j, v);
        if (frag) {
#pragma unroll
            for (int h = 0; h < 8; h += E) {
                const int kk = k + h;
                const size_t f = (((size_t)(nrow >> 4) * (d.K_pad / (4 * E)) + kk / (4 * E)) * 4 + (kk % (4 * E)) / E) * (16 * E) + (nrow & 15) * E;
                if (E == 8) pack_store8<T>(frag + f, v);
                else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(frag) + f) = f32x4{v[h], v[h + 1], v[h + 2], v[h + 3]};
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < PACK_CHUNK / 256; ++u) {
        const long j = j0 + u * 256 + threadIdx.x;
        if (j >= total) break;
        const int nrow = (int)(j / d.K_pad), k = (int)(j - (long)nrow * d.K_pad);
        const int tap = k / d.ck, cc = k - tap * d.ck;
        float v = 0.f;
        if (tap < taps) {
            if (d.mode == 0) { if (nrow < d.Cout && cc < d.Cin) v = d.w[((size_t)nrow * d.Cin + cc) * taps + tap]; }
            else if (nrow < d.Cin && cc < d.Cout) v = d.w[((size_t)cc * d.Cin + nrow) * taps + (taps - 1 - tap)];
        }
        ElemTraits<T>::store(packed + j, v);
        if (frag) {
            const size_t f = (((size_t)(nrow >> 4) * (d.K_pad / (4 * E)) + k / (4 * E)) * 4 + (k % (4 * E)) / E) * (16 * E) + (nrow & 15) * E + k % E;
            ElemTraits<T>::store(frag + f, v);
        }
    }
}

// column sums: out[c] += sum_m x[m][c]      (bias gradient)
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, int M, int C, int ld, int rows_per_block, float* __restrict__ out) {
    const int c = blockIdx.y * 64 + threadIdx.x;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, M);
    float s = 0.f;
    if (c < C) for (int r = r0 + threadIdx.y; r < r1; r += blockDim.y) s += ElemTraits<T>::load(x + (size_t)r * ld + c);
    __shared__ float red[4][64];
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && c < C) unsafeAtomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// Train-mode BatchNorm over an [M][C] NHWC tensor
// thread -> fixed 16-byte channel chunk column (cc = tid % CPR), rows strided: per-thread register accumulation, then
// a short LDS + global atomic reduction per block.
// ------------------------------------------------------------------------------------------------
// Block reduction of per-thread column partials: thread (roff, cc) holds NV vectors of E floats for the 16-byte channel chunk
// cc; partials with equal cc are summed over roff through LDS with plain stores (LDS float atomics retire about one lane per
// clock on gfx950 -- sixteen of them per thread made these kernels atomic-bound), then ONE global atomic per column and block.
template <int E, int NV>
__device__ __forceinline__ void block_reduce_columns(const float (&v)[NV][E], float* sred, int C, int CPR, int tid, float* const (&out)[NV]) {
    const int cc = tid % CPR, roff = tid / CPR, rstep = 256 / CPR;
    const int ld = NV * C;
    if (roff < rstep) {
#pragma unroll
        for (int n = 0; n < NV; ++n)
#pragma unroll
            for (int e = 0; e < E; e += 4)
                *reinterpret_cast<f32x4*>(sred + roff * ld + n * C + cc * E + e) = f32x4{v[n][e], v[n][e + 1], v[n][e + 2], v[n][e + 3]};
    }
    __syncthreads();
    for (int i = tid; i < ld; i += 256) {
        float t = 0.f;
        for (int r = 0; r < rstep; ++r) t += sred[r * ld + i];
        unsafeAtomicAdd(out[i / C] + (i % C), t);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, long M, int C, int rows_per_block,
                                                       float* __restrict__ sum, float* __restrict__ sumsq, int ncopy) {
    if (ncopy > 1) { const int k = blockIdx.x % ncopy; sum += (size_t)k * 2 * C; sumsq += (size_t)k * 2 * C; }      // spread the atomics over ncopy [sum|sumsq] pairs
    constexpr int E = ElemTraits<T>::ELEMS;
    extern __shared__ float sred[];                          // [256 / CPR][2 * C]
    const int CPR = C / E, tid = threadIdx.x;
    const int cc = tid % CPR, rstep = 256 / CPR, roff = tid / CPR;
    float acc[2][E];
#pragma unroll
    for (int e = 0; e < E; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(r0 + (long)rows_per_block, M);
    if (roff < rstep) {
        const T* px = x + cc * E;
        long r = r0 + roff;
        for (; r + 3L * rstep < r1; r += 4L * rstep) {       // four independent 16-byte loads in flight per thread
            u32x4 c4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c4[u] = *reinterpret_cast<const u32x4*>(px + (r + (long)u * rstep) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[E];
                ElemTraits<T>::unpack(c4[u], v);
#pragma unroll
                for (int e = 0; e < E; ++e) { acc[0][e] += v[e]; acc[1][e] += v[e] * v[e]; }
            }
        }
        for (; r < r1; r += rstep) {
            float v[E];
            ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(px + r * C), v);
#pragma unroll
            for (int e = 0; e < E; ++e) { acc[0][e] += v[e]; acc[1][e] += v[e] * v[e]; }
        }
    }
    float* const outs[2] = {sum, sumsq};
    block_reduce_columns<E, 2>(acc, sred, C, CPR, tid, outs);
}

// column sums with the bn_stats thread mapping (16-byte chunks along channels, rows strided): used for bias gradients when
// the row is a power-of-two number of chunks (every padded output map is)
template <typename T>
__global__ __launch_bounds__(256) void colsum_chunk_kernel(const T* __restrict__ x, long M, int C, int ld, int rows_per_block, float* __restrict__ out) {
    constexpr int E = ElemTraits<T>::ELEMS;
    extern __shared__ float sred[];                          // [256 / CPR][C]
    const int CPR = C / E, tid = threadIdx.x;
    const int cc = tid % CPR, rstep = 256 / CPR, roff = tid / CPR;
    float acc[1][E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[0][e] = 0.f;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(r0 + (long)rows_per_block, M);
    if (roff < rstep) {
        const T* px = x + cc * E;
        long r = r0 + roff;
        for (; r + 3L * rstep < r1; r += 4L * rstep) {
            u32x4 c4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c4[u] = *reinterpret_cast<const u32x4*>(px + (r + (long)u * rstep) * ld);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[E];
                ElemTraits<T>::unpack(c4[u], v);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[0][e] += v[e];
            }
        }
        for (; r < r1; r += rstep) {
            float v[E];
            ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(px + r * ld), v);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[0][e] += v[e];
        }
    }
    float* const outs[1] = {out};
    block_reduce_columns<E, 1>(acc, sred, C, CPR, tid, outs);
}

// per-channel epilogue of the statistics pass: mean / rstd / folded scale+shift and the running-statistics update in ONE
// launch (as separate tensor ops this was ~10 tiny launches per BN layer, 66 layers per step)
__global__ void bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps,
                                   float inv_count, float unbias, float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float m = sum[c] * inv_count;
    const float v = fmaxf(sumsq[c] * inv_count - m * m, 0.f);
    const float r = rsqrtf(v + eps);
    const float sc = gamma[c] * r;
    mean[c] = m; rstd[c] = r; scale[c] = sc; shift[c] = beta[c] - m * sc;
    if (running_mean) {
        running_mean[c] = running_mean[c] * (1.f - momentum) + m * momentum;
        running_var[c] = running_var[c] * (1.f - momentum) + v * unbias * momentum;
    }
}

// y = act(x*scale[c] + shift[c] (+ res)); per-channel parameters are fetched as 16-byte vectors (8 scalar loads per
// parameter and chunk made these streaming kernels instruction-bound at ~1.2 TB/s)
template <int E> __device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[E]) {
#pragma unroll
    for (int e = 0; e < E; e += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + e);
        v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
    }
}

template <typename T>
__global__ void bn_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                  const T* __restrict__ res, T* __restrict__ y, long total_chunks, int C, int act) {
    constexpr int E = ElemTraits<T>::ELEMS;
    const int CPR = C / E;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total_chunks; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % CPR) * E;
        float v[E], sc[E], sh[E];
        ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(x + i * E), v);
        load_vec<E>(scale + c0, sc); load_vec<E>(shift + c0, sh);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = v[e] * sc[e] + sh[e];
        if (res) {
            float r[E];
            ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(res + i * E), r);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] += r[e];
        }
        apply_act_chunk<E>(v, act, 0);
        *reinterpret_cast<u32x4*>(y + i * E) = ElemTraits<T>::pack(v);
    }
}

__device__ __forceinline__ float act_grad(float a, int act) {    // derivative of the activation, from its OUTPUT a
    if (act == ACT_RELU) return a > 0.f ? 1.f : 0.f;
    if (act == ACT_LEAKY) return a > 0.f ? 1.f : 0.01f;
    return 1.f;
}

// backward reductions: g = da * act'(a);  sg[c] = sum g,  sgx[c] = sum g * xhat,  xhat = (x - mean[c]) * rstd[c]
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ a, const T* __restrict__ da,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            long M, int C, int rows_per_block, int act,
                                                            float* __restrict__ sg, float* __restrict__ sgx, int ncopy,
                                                            const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr) {
    if (ncopy > 1) { const int k = blockIdx.x % ncopy; sg += (size_t)k * 2 * C; sgx += (size_t)k * 2 * C; }
    constexpr int E = ElemTraits<T>::ELEMS;
    extern __shared__ float sred[];                          // [256 / CPR][2 * C]
    const int CPR = C / E, tid = threadIdx.x;
    const int cc = tid % CPR, rstep = 256 / CPR, roff = tid / CPR;
    // a == nullptr with an activation: no residual entered the activation, so its sign is the sign of x*scale + shift, recomputed
    // with the forward's own expression instead of reading the stored output (one of the three streams of this pass)
    const bool recompute = act != ACT_NONE && a == nullptr;
    float acc[2][E], mu[E], rs[E], sc[E], sh[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        acc[0][e] = 0.f; acc[1][e] = 0.f; mu[e] = mean[cc * E + e]; rs[e] = rstd[cc * E + e];
        sc[e] = recompute ? gamma[cc * E + e] * rs[e] : 0.f;
        sh[e] = recompute ? beta[cc * E + e] - mu[e] * sc[e] : 0.f;
    }
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(r0 + (long)rows_per_block, M);
    auto fold = [&](const u32x4& cx, const u32x4& cd, const u32x4& ca) {
        float xv[E], av[E], dv[E];
        ElemTraits<T>::unpack(cx, xv); ElemTraits<T>::unpack(cd, dv);
        if (act != ACT_NONE && !recompute) ElemTraits<T>::unpack(ca, av);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (recompute) av[e] = xv[e] * sc[e] + sh[e];
            const float gq = act != ACT_NONE ? dv[e] * act_grad(av[e], act) : dv[e];
            acc[0][e] += gq; acc[1][e] += gq * (xv[e] - mu[e]) * rs[e];
        }
    };
    if (roff < rstep) {
        long r = r0 + roff;
        for (; r + rstep < r1; r += 2L * rstep) {            // two rows = six 16-byte loads in flight per thread
            const size_t o0 = (size_t)r * C + cc * E, o1 = (size_t)(r + rstep) * C + cc * E;
            const u32x4 x0 = *reinterpret_cast<const u32x4*>(x + o0), x1 = *reinterpret_cast<const u32x4*>(x + o1);
            const u32x4 d0 = *reinterpret_cast<const u32x4*>(da + o0), d1 = *reinterpret_cast<const u32x4*>(da + o1);
            u32x4 a0 = x0, a1 = x1;
            if (act != ACT_NONE && !recompute) { a0 = *reinterpret_cast<const u32x4*>(a + o0); a1 = *reinterpret_cast<const u32x4*>(a + o1); }
            fold(x0, d0, a0); fold(x1, d1, a1);
        }
        for (; r < r1; r += rstep) {
            const size_t o = (size_t)r * C + cc * E;
            const u32x4 x0 = *reinterpret_cast<const u32x4*>(x + o), d0 = *reinterpret_cast<const u32x4*>(da + o);
            u32x4 a0 = x0;
            if (act != ACT_NONE && !recompute) a0 = *reinterpret_cast<const u32x4*>(a + o);
            fold(x0, d0, a0);
        }
    }
    float* const outs[2] = {sg, sgx};
    block_reduce_columns<E, 2>(acc, sred, C, CPR, tid, outs);
}

// dx = gamma*rstd * (g - sg/M - xhat*sgx/M) = A[c]*g + B[c]*x + D[c];  dres = g (optional).  Every workgroup first builds the
// [A | B | D] table (3*C floats) in LDS, then streams: three 16-byte table reads per chunk instead of 40 scalar loads.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ a, const T* __restrict__ da,
                                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sg, const float* __restrict__ sgx, float invM,
                                    T* __restrict__ dx, T* __restrict__ dres, long total_chunks, int C, int act) {
    constexpr int E = ElemTraits<T>::ELEMS;
    extern __shared__ float coef[];                          // [3][C]
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float ca = gamma[c] * rstd[c], cb = -ca * rstd[c] * sgx[c] * invM;
        coef[c] = ca; coef[C + c] = cb; coef[2 * C + c] = -cb * mean[c] - ca * sg[c] * invM;
    }
    __syncthreads();
    const int CPR = C / E;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total_chunks; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % CPR) * E;
        float xv[E], av[E], dv[E], gq[E], ov[E], ca[E], cb[E], cd[E];
        ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(x + i * E), xv);
        ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(da + i * E), dv);
        if (act != ACT_NONE) ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(a + i * E), av);
        load_vec<E>(coef + c0, ca); load_vec<E>(coef + C + c0, cb); load_vec<E>(coef + 2 * C + c0, cd);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            gq[e] = act != ACT_NONE ? dv[e] * act_grad(av[e], act) : dv[e];
            ov[e] = ca[e] * gq[e] + cb[e] * xv[e] + cd[e];
        }
        *reinterpret_cast<u32x4*>(dx + i * E) = ElemTraits<T>::pack(ov);
        if (dres) *reinterpret_cast<u32x4*>(dres + i * E) = ElemTraits<T>::pack(gq);
    }
}

// ---- two-launch forms (no zero-fill, finalize or copy launches) ----------------------------------
// The column reductions above add into one of `ncopy` [2C] pairs of a persistent, ZERO scratch (ncopy * 2C = 1024 floats:
// 64 cache lines take the atomics instead of 4).  The streaming kernel that consumes the sums first folds the copies (every
// workgroup, 4 loads per thread), and the LAST workgroup to have done so zeroes the scratch again for the next call -- the
// 66 BN layers x (zero-fill + finalize + 2 gradient copies + counter add) launches of a step disappear.
constexpr int BN_SCRATCH_COLS = 1024;                            // floats per direction: ncopy * 2C

__device__ __forceinline__ void bn_fold_copies(const float* __restrict__ sums, int C, int ncopy, float* colsum, float* part, int tid) {
    const int cols = 2 * C;
    if (cols <= 256) {
        const int groups = 256 / cols, j = tid % cols, g0 = tid / cols;
        float a = 0.f;
        for (int k = g0; k < ncopy; k += groups) a += sums[(size_t)k * cols + j];
        part[tid] = a;
        __syncthreads();
        if (tid < cols) {
            float t = 0.f;
            for (int g = 0; g < groups; ++g) t += part[g * cols + tid];
            colsum[tid] = t;
        }
    } else {
        for (int j = tid; j < cols; j += 256) {
            float t = 0.f;
            for (int k = 0; k < ncopy; ++k) t += sums[(size_t)k * cols + j];
            colsum[j] = t;
        }
    }
    __syncthreads();
}

// Every workgroup takes a ticket once it has read the sums (right after its fold) and looks at it only after streaming.  One
// counter for ~2000 workgroups drains at ~8 ns per atomic (16 us, longer than the small layers' streaming), so the tickets are
// two-level: 64 first-level counters on separate cache lines (workgroup index mod 64), the last arrival of each takes a
// second-level ticket, and the holder of the last of those clears the sums.  Counters reset themselves.
constexpr int BN_TICKET_GROUPS = 64, BN_TICKET_STRIDE = 32;      // counters: [64][32 words] first level, then one second-level word
__device__ __forceinline__ unsigned bn_take_ticket(unsigned* cnt) { return atomicAdd(cnt + (blockIdx.x % BN_TICKET_GROUPS) * BN_TICKET_STRIDE, 1u); }
__device__ __forceinline__ void bn_release_scratch(float* sums, unsigned* cnt, unsigned ticket, int tid, int* s_last) {
    if (tid == 0) {
        int last = 0;
        const unsigned grp = blockIdx.x % BN_TICKET_GROUPS;
        const unsigned members = (gridDim.x - grp + BN_TICKET_GROUPS - 1) / BN_TICKET_GROUPS;
        if (ticket == members - 1) {
            cnt[grp * BN_TICKET_STRIDE] = 0u;
            const unsigned groups = gridDim.x < (unsigned)BN_TICKET_GROUPS ? gridDim.x : (unsigned)BN_TICKET_GROUPS;
            if (atomicAdd(cnt + BN_TICKET_GROUPS * BN_TICKET_STRIDE, 1u) == groups - 1) { cnt[BN_TICKET_GROUPS * BN_TICKET_STRIDE] = 0u; last = 1; }
        }
        *s_last = last;
    }
    __syncthreads();
    if (*s_last)
        for (int i = tid; i < BN_SCRATCH_COLS; i += 256) sums[i] = 0.f;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_fused_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
        const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean, float* running_var, long long* nbt,
        float momentum, float eps, float inv_count, float unbias, float* sums, unsigned* counter, int ncopy,
        float* __restrict__ mean_out, float* __restrict__ rstd_out, long total_chunks, int C, int act) {
    constexpr int E = ElemTraits<T>::ELEMS;
    __shared__ float colsum[BN_SCRATCH_COLS], tab[BN_SCRATCH_COLS], part[256];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    bn_fold_copies(sums, C, ncopy, colsum, part, tid);
    for (int c = tid; c < C; c += 256) {
        const float m = colsum[c] * inv_count;
        const float v = fmaxf(colsum[C + c] * inv_count - m * m, 0.f);
        const float r = rsqrtf(v + eps), sc = gamma[c] * r;
        tab[c] = sc; tab[C + c] = beta[c] - m * sc;
        if (blockIdx.x == 0) {
            mean_out[c] = m; rstd_out[c] = r;
            if (running_mean) {
                running_mean[c] = running_mean[c] * (1.f - momentum) + m * momentum;
                running_var[c] = running_var[c] * (1.f - momentum) + v * unbias * momentum;
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0 && nbt) *nbt += 1;
    __syncthreads();
    const unsigned ticket = tid == 0 ? bn_take_ticket(counter) : 0u;
    const int CPR = C / E;
    const long stride = (long)gridDim.x * blockDim.x;
    auto finish = [&](long i, const u32x4& cx, const u32x4& cr) {
        const int c0 = (int)(i % CPR) * E;
        float v[E], sc[E], sh[E];
        ElemTraits<T>::unpack(cx, v);
        load_vec<E>(tab + c0, sc); load_vec<E>(tab + C + c0, sh);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = v[e] * sc[e] + sh[e];
        if (res) {
            float r[E];
            ElemTraits<T>::unpack(cr, r);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] += r[e];
        }
        apply_act_chunk<E>(v, act, 0);
        *reinterpret_cast<u32x4*>(y + i * E) = ElemTraits<T>::pack(v);
    };
    long i = blockIdx.x * (long)blockDim.x + tid;
    for (; i + stride < total_chunks; i += 2 * stride) {         // two chunks (up to four 16-byte loads) in flight per thread
        const long j = i + stride;
        const u32x4 x0 = *reinterpret_cast<const u32x4*>(x + i * E), x1 = *reinterpret_cast<const u32x4*>(x + j * E);
        u32x4 r0 = x0, r1 = x1;
        if (res) { r0 = *reinterpret_cast<const u32x4*>(res + i * E); r1 = *reinterpret_cast<const u32x4*>(res + j * E); }
        finish(i, x0, r0); finish(j, x1, r1);
    }
    for (; i < total_chunks; i += stride) {
        const u32x4 x0 = *reinterpret_cast<const u32x4*>(x + i * E);
        u32x4 r0 = x0;
        if (res) r0 = *reinterpret_cast<const u32x4*>(res + i * E);
        finish(i, x0, r0);
    }
    bn_release_scratch(sums, counter, ticket, tid, &s_last);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(const T* __restrict__ x, const T* __restrict__ a, const T* __restrict__ da,
        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma, float* sums, unsigned* counter,
        int ncopy, float invM, T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ dgamma, float* __restrict__ dbeta,
        long total_chunks, int C, int act, const float* __restrict__ beta) {
    constexpr int E = ElemTraits<T>::ELEMS;
    __shared__ float colsum[BN_SCRATCH_COLS], coef[5 * (BN_SCRATCH_COLS / 2)], part[256];      // [A | B | D | scale | shift]
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const bool recompute = act != ACT_NONE && a == nullptr;          // see bn_bwd_reduce_kernel
    bn_fold_copies(sums, C, ncopy, colsum, part, tid);
    for (int c = tid; c < C; c += 256) {
        const float sg = colsum[c], sgx = colsum[C + c];
        const float ca = gamma[c] * rstd[c], cb = -ca * rstd[c] * sgx * invM;
        coef[c] = ca; coef[C + c] = cb; coef[2 * C + c] = -cb * mean[c] - ca * sg * invM;
        if (recompute) { coef[3 * C + c] = ca; coef[4 * C + c] = beta[c] - mean[c] * ca; }
        if (blockIdx.x == 0) { dgamma[c] = sgx; dbeta[c] = sg; }
    }
    __syncthreads();
    const unsigned ticket = tid == 0 ? bn_take_ticket(counter) : 0u;
    const int CPR = C / E;
    const long stride = (long)gridDim.x * blockDim.x;
    auto finish = [&](long i, const u32x4& cx, const u32x4& cdv, const u32x4& cav) {
        const int c0 = (int)(i % CPR) * E;
        float xv[E], av[E], dv[E], gq[E], ov[E], ca[E], cb[E], cd[E];
        ElemTraits<T>::unpack(cx, xv);
        ElemTraits<T>::unpack(cdv, dv);
        if (recompute) {
            float sc[E], sh[E];
            load_vec<E>(coef + 3 * C + c0, sc); load_vec<E>(coef + 4 * C + c0, sh);
#pragma unroll
            for (int e = 0; e < E; ++e) av[e] = xv[e] * sc[e] + sh[e];
        } else if (act != ACT_NONE) ElemTraits<T>::unpack(cav, av);
        load_vec<E>(coef + c0, ca); load_vec<E>(coef + C + c0, cb); load_vec<E>(coef + 2 * C + c0, cd);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            gq[e] = act != ACT_NONE ? dv[e] * act_grad(av[e], act) : dv[e];
            ov[e] = ca[e] * gq[e] + cb[e] * xv[e] + cd[e];
        }
        *reinterpret_cast<u32x4*>(dx + i * E) = ElemTraits<T>::pack(ov);
        if (dres) *reinterpret_cast<u32x4*>(dres + i * E) = ElemTraits<T>::pack(gq);
    };
    long i = blockIdx.x * (long)blockDim.x + tid;
    for (; i + stride < total_chunks; i += 2 * stride) {         // two chunks = six 16-byte loads in flight per thread
        const long j = i + stride;
        const u32x4 x0 = *reinterpret_cast<const u32x4*>(x + i * E), x1 = *reinterpret_cast<const u32x4*>(x + j * E);
        const u32x4 d0 = *reinterpret_cast<const u32x4*>(da + i * E), d1 = *reinterpret_cast<const u32x4*>(da + j * E);
        u32x4 a0 = x0, a1 = x1;
        if (act != ACT_NONE && !recompute) { a0 = *reinterpret_cast<const u32x4*>(a + i * E); a1 = *reinterpret_cast<const u32x4*>(a + j * E); }
        finish(i, x0, d0, a0); finish(j, x1, d1, a1);
    }
    for (; i < total_chunks; i += stride) {
        const u32x4 x0 = *reinterpret_cast<const u32x4*>(x + i * E), d0 = *reinterpret_cast<const u32x4*>(da + i * E);
        u32x4 a0 = x0;
        if (act != ACT_NONE && !recompute) a0 = *reinterpret_cast<const u32x4*>(a + i * E);
        finish(i, x0, d0, a0);
    }
    bn_release_scratch(sums, counter, ticket, tid, &s_last);
}

// ------------------------------------------------------------------------------------------------
// max-pool 2x2 backward: gradient goes to the first maximum of each window (torch semantics)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C) {
    constexpr int E = ElemTraits<T>::ELEMS;
    const int Ho = H / 2, Wo = W / 2, CG = C / E;
    const long total = (long)B * Ho * Wo * CG;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        long p = i / CG;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const size_t base = ((size_t)(b * H + oh * 2) * W + ow * 2) * C + cg * E;
        const size_t offs[4] = {base, base + C, base + (size_t)W * C, base + (size_t)W * C + C};
        float v[4][E], g[E], o[4][E];
#pragma unroll
        for (int q = 0; q < 4; ++q) ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(x + offs[q]), v[q]);
        ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(dy + ((size_t)(b * Ho + oh) * Wo + ow) * C + cg * E), g);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            int best = 0; float bv = v[0][e];
#pragma unroll
            for (int q = 1; q < 4; ++q) if (v[q][e] > bv) { bv = v[q][e]; best = q; }
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q][e] = q == best ? g[e] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(dx + offs[q]) = ElemTraits<T>::pack(o[q]);
    }
}

// ------------------------------------------------------------------------------------------------
// depthwise ConvTranspose2d(k=2f, s=f, p=f/2) backward:  dx[b,ih,iw,c] = sum_{kh,kw} dy[b, ih*f-p+kh, iw*f-p+kw, c] * w[kh*k+kw][c]
//                                                         dw[kh*k+kw][c] = sum_{b,ih,iw} x[b,ih,iw,c] * dy[b, ih*f-p+kh, iw*f-p+kw, c]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void upsample_bwd_dx_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int B, int H, int W, int C, int f) {
    constexpr int E = ElemTraits<T>::ELEMS;
    const int Ho = H * f, Wo = W * f, CG = C / E, p_ = f / 2, k = 2 * f;
    const long total = (long)B * H * W * CG;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        long p = i / CG;
        const int iw = (int)(p % W); p /= W;
        const int ih = (int)(p % H);
        const int b = (int)(p / H);
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        for (int kh = 0; kh < k; ++kh) {
            const int oh = ih * f - p_ + kh;
            if (oh < 0 || oh >= Ho) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int ow = iw * f - p_ + kw;
                if (ow < 0 || ow >= Wo) continue;
                float g[E];
                ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(dy + ((size_t)(b * Ho + oh) * Wo + ow) * C + cg * E), g);
                const float* wp = w + (size_t)(kh * k + kw) * C + cg * E;
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] += g[e] * wp[e];
            }
        }
        *reinterpret_cast<u32x4*>(dx + i * E) = ElemTraits<T>::pack(acc);
    }
}

// depthwise-deconv weight gradient: dw[tap][c] = sum over input pixels of x[b,ih,iw,c] * dy[b, ih*f - f/2 + kh, iw*f - f/2 + kw, c].
// A workgroup of 1024 threads walks `rows_per_block` input rows (b, ih); thread -> (item = (tap, 16-byte channel chunk),
// column group): the S = 1024 / items groups take interleaved columns, four 16-byte load pairs in flight each, and meet in
// LDS before ONE atomic per (tap, channel) and workgroup.  (With one thread per item and 256-thread workgroups the f = 2
// layers ran 128 threads per workgroup, 768 waves on the whole chip, each walking a full row: 85-110 us per call.)
template <typename T>
__global__ __launch_bounds__(1024) void upsample_bwd_dw_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw,
                                                               int B, int H, int W, int C, int f, int rows_per_block, float* __restrict__ part) {
    // `part` != nullptr: the workgroup's sums go to part[blockIdx.x][tap][c] with plain stores (every element written) and
    // upsample_dw_sum_kernel adds the workgroups in order -- 384 workgroups x 1024 atomics on the same 64 cache lines serialised in L2
    // (66 us for a 7 us streaming job) and needed a zero-fill; the partial form is also bit-reproducible.
    if (part) dw = part + (size_t)blockIdx.x * (size_t)(4 * f * f) * C;
    constexpr int E = ElemTraits<T>::ELEMS;
    extern __shared__ float red[];                                    // [S][items * E] when S > 1
    const int k = 2 * f, p_ = f / 2, Ho = H * f, Wo = W * f, taps = k * k, CG = C / E, items = taps * CG;
    const int S = items >= 1024 ? 1 : 1024 / items;                   // items is a power of two
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, B * H);
    for (int it0 = 0; it0 < items; it0 += 1024) {
        const int item = it0 + (S > 1 ? (int)threadIdx.x % items : (int)threadIdx.x), grp = S > 1 ? (int)threadIdx.x / items : 0;
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        if (item < items && grp < S) {
            const int cg = item % CG, tap = item / CG, kh = tap / k, kw = tap - kh * k;
            // columns iw with 0 <= iw*f - p_ + kw < Wo
            const int iw_lo = max(0, (p_ - kw + f - 1) / f), iw_hi = min(W - 1, (Wo - 1 + p_ - kw) / f);
            for (int row = r0; row < r1; ++row) {
                const int b = row / H, ih = row - b * H, oh = ih * f - p_ + kh;
                if (oh < 0 || oh >= Ho) continue;
                const T* xr = x + (size_t)row * W * C + cg * E;
                const T* dr = dy + ((size_t)(b * Ho + oh) * Wo + (kw - p_)) * C + cg * E;      // + iw*f*C per column
                int iw = iw_lo + grp;
                for (; iw + 3 * S <= iw_hi; iw += 4 * S) {
                    u32x4 xv[4], dv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        xv[u] = *reinterpret_cast<const u32x4*>(xr + (size_t)(iw + u * S) * C);
                        dv[u] = *reinterpret_cast<const u32x4*>(dr + (size_t)(iw + u * S) * f * C);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float a[E], d[E];
                        ElemTraits<T>::unpack(xv[u], a); ElemTraits<T>::unpack(dv[u], d);
#pragma unroll
                        for (int e = 0; e < E; ++e) acc[e] += a[e] * d[e];
                    }
                }
                for (; iw <= iw_hi; iw += S) {
                    float a[E], d[E];
                    ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(xr + (size_t)iw * C), a);
                    ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(dr + (size_t)iw * f * C), d);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] += a[e] * d[e];
                }
            }
        }
        if (S > 1) {
            if (grp < S) {
#pragma unroll
                for (int e = 0; e < E; ++e) red[(size_t)grp * items * E + (size_t)e * items + item] = acc[e];
            }
            __syncthreads();
            for (int i = threadIdx.x; i < items * E; i += 1024) {
                float t = 0.f;
                for (int q = 0; q < S; ++q) t += red[(size_t)q * items * E + i];
                const int e = i / items, im = i - e * items, cg = im % CG, tap = im / CG;
                if (part) dw[(size_t)tap * C + cg * E + e] = t;
                else unsafeAtomicAdd(dw + (size_t)tap * C + cg * E + e, t);
            }
            __syncthreads();
        } else if (item < items) {
            const int cg = item % CG, tap = item / CG;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (part) dw[(size_t)tap * C + cg * E + e] = acc[e];
                else unsafeAtomicAdd(dw + (size_t)tap * C + cg * E + e, acc[e]);
            }
        }
    }
}

// sum of the workgroups' partial blocks, fixed order: 16 columns x 16 row groups per workgroup (every thread 1/16 of the rows with
// independent loads in flight, then an LDS fold) -- one thread per column walking all 256 rows took 15 us of pure load latency
// `C_tr` > 0: the sums are written in the PARAMETER's layout (C, 1, k, k) = [c][tap] instead of [tap][c]
__global__ __launch_bounds__(256) void upsample_dw_sum_kernel(const float* __restrict__ part, int nblk, int n, float* __restrict__ dw, int C_tr) {
    __shared__ float red[16][17];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + col;
    float a = 0.f;
    if (i < n)
        for (int b = grp; b < nblk; b += 16) a += part[(size_t)b * n + i];
    red[grp][col] = a;
    __syncthreads();
    if (grp == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][col];
        if (C_tr > 0) { const int tap = i / C_tr, c = i - tap * C_tr; dw[(size_t)c * (n / C_tr) + tap] = t; }
        else dw[i] = t;
    }
}

// zero insertion for the data gradient of a stride-2 conv: up[b, 2*oh, 2*ow, :] = dy[b, oh, ow, :], zeros elsewhere
template <typename T>
__global__ void zero_insert2_kernel(const T* __restrict__ dy, T* __restrict__ up, int B, int Ho, int Wo, int C, int H, int W) {
    constexpr int E = ElemTraits<T>::ELEMS;
    const int CG = C / E;
    const long total = (long)B * H * W * CG;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        long p = i / CG;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int b = (int)(p / H);
        u32x4 z = {0u, 0u, 0u, 0u};
        if (!(h & 1) && !(w & 1) && (h >> 1) < Ho && (w >> 1) < Wo)
            z = *reinterpret_cast<const u32x4*>(dy + ((size_t)(b * Ho + (h >> 1)) * Wo + (w >> 1)) * C + cg * E);
        *reinterpret_cast<u32x4*>(up + i * E) = z;
    }
}

}  // namespace mfx
using namespace mfx;

int g_opt_wgrad_blocks = 600;   // option "wgrad_blocks": target workgroup count of the MFMA weight-gradient kernel (measured, B=8 step:
                                // 64 -> 146 ms, 150 -> 90, 300 -> 78, 600 -> 73, 2048 -> 75, 8192 -> 81: the tile atomics of every slab cost
                                // more than the extra workgroups hide)
int g_opt_wgrad_ws = 1;        // option "wgrad_ws": 0 = always accumulate the tiles with atomics
int g_opt_wgrad_ws_blocks = 1200;  // option "wgrad_ws_blocks": target workgroup count when partial tiles go to the workspace (step: 1200 -> 58.0 ms, 2400 -> 58.3, 4800 -> 58.6; atomics: 59.6)
int g_opt_wgrad_min_m = 128;   // option "wgrad_min_m": fewest pixels of a slab of the MFMA weight-gradient kernel.  r06: 1024 left the 1x1 / Root layers of the 24x80 and 12x40 maps with 8-16 slabs (120-240 workgroups of 32 iterations each: 31 us per layer for 4 GFLOP); same-box step 17.50 (1024) / 17.31 (512) / 17.27 (256) / 17.23-17.31 (128) / 17.26 (64) ms
int g_opt_wgrad_mfma = 1;     // option "wgrad_mfma": 0 = VALU kernel for bf16 too, 1 = 64x64 MFMA tiles, 3 = 128x128 where they fit

// CALL_16 is written once for both 16-bit activation types: T16 = bf16_t or half_t
#define DISPATCH_T(dtype, CALL_F32, CALL_16) do { if ((dtype) == MFX_F32) { CALL_F32; } else if ((dtype) == MFX_BF16) { using T16 = bf16_t; CALL_16; } \
    else if ((dtype) == MFX_F16) { using T16 = half_t; CALL_16; } else return mfx_fail(MFX_ERR_ARG, "bad dtype"); } while (0)

static int conv_wgrad_impl(const void* x, const void* dy, float* dw, int B, int H, int W, int x_pixstride, int Ck,
                           int kh, int kw, int stride, int pad_h, int pad_w, int Ho, int Wo, int Cout, int ldy,
                           int dtype, int oihw, int Cin_out, int Cout_out, void* stream, int dil_w = 1,
                           void* workspace = nullptr, size_t workspace_bytes = 0, int direct = 0) {
    if (!x || !dy || !dw) return mfx_fail(MFX_ERR_ARG, "conv_wgrad: null pointer");
    if (Ck % 4 != 0 || Cout % 4 != 0) return mfx_fail(MFX_ERR_ARG, "conv_wgrad: Ck and Cout must be multiples of 4");
    WgradGeom g;
    g.B = B; g.H = H; g.W = W; g.Ho = Ho; g.Wo = Wo; g.x_pixstride = x_pixstride; g.Ck = Ck; g.kh = kh; g.kw = kw; g.stride = stride;
    g.pad_h = pad_h; g.pad_w = pad_w; g.dil_w = dil_w; g.M = B * Ho * Wo; g.K = kh * kw * Ck; g.Cout = Cout; g.ldy = ldy; g.m_per_block = 2048;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    g.oihw = oihw; g.Cin_out = Cin_out; g.Cout_out = Cout_out; g.ws = nullptr; g.ws_ld = 0; g.ws_slab = 0; g.direct = direct;
    const size_t dw_bytes = (oihw ? (size_t)Cout_out * Cin_out * kh * kw : (size_t)Cout * g.K) * sizeof(float);
    if (g.M == 0) { MFX_HIP_CHECK(mfx::zero_async(dw, dw_bytes, st)); return MFX_OK; }
    // (x chunks only need 4-byte alignment: the stem reads 8-element super-taps at a pixel stride of 4 elements)
    const bool is16 = dtype == MFX_BF16 || dtype == MFX_F16;
    if (is16 && g_opt_wgrad_mfma) {
        int nslab_tr = 0;
        int rc_tr = try_conv_wgrad_patch(x, dy, g, dtype, workspace, workspace_bytes, &nslab_tr, st);
        if (rc_tr != 1) rc_tr = try_conv_wgrad_tr(x, dy, g, dtype, workspace, workspace_bytes, &nslab_tr, st);
        if (rc_tr == 1) {
            const long total = (long)Cout * g.K;
            hipLaunchKernelGGL(wgrad_reduce_kernel, WR_GRID(total), dim3(256), 0, st, g.ws, nslab_tr, g.ws_slab, g.ws_ld, g, dw);
            MFX_HIP_CHECK(hipGetLastError());
            return MFX_OK;
        }
        g.ws = nullptr; g.ws_ld = 0; g.ws_slab = 0; g.m_per_block = 2048;
    }
    // (the workspace paths end in wgrad_reduce_kernel, which writes every element of dw; only the atomic paths need zeros)
    if (is16 && Ck % 8 == 0 && x_pixstride % 2 == 0 && ldy % 8 == 0 && g_opt_wgrad_mfma) {
        const int bt = (Cout >= 128 && g.K >= 128 && g_opt_wgrad_mfma == 3) ? 128 : 64;     // 128-wide tiles measured slower (83 vs 78 ms)
        const int tiles = cdivt(g.K, bt) * cdivt(Cout, bt);
        // pixel slabs.  With a workspace every slab writes its partial tile with plain stores and wgrad_reduce_kernel sums them,
        // so the slab count only has to fill the chip; without one the tile is accumulated with fp32 atomics, which bounds it
        // (measured optimum ~600 workgroups: the atomics of every extra slab cost more than the extra workgroups hide).
        const int ws_ld = cdivt(g.K, bt) * bt;
        const long ws_slab = (long)cdivt(Cout, bt) * bt * ws_ld;
        const bool use_ws = workspace && g_opt_wgrad_ws;
        int slabs = std::max(1, (use_ws ? g_opt_wgrad_ws_blocks : g_opt_wgrad_blocks) / tiles);
        if (use_ws) slabs = (int)std::min<long>(slabs, (long)(workspace_bytes / sizeof(float)) / ws_slab);
        if (use_ws && slabs < 1) slabs = 1;
        g.m_per_block = std::max(std::max(32, g_opt_wgrad_min_m / 32 * 32), (int)(((long)g.M / slabs + 31) / 32 * 32));
        const int nslab = cdivt(g.M, g.m_per_block);
        bool ws_ok = use_ws && (size_t)nslab * ws_slab * sizeof(float) <= workspace_bytes;
        if (!ws_ok && g_opt_det) { g.m_per_block = (g.M + 31) / 32 * 32; }     // atomics: a single slab per tile adds into zeros exactly once
        if (ws_ok) { g.ws = reinterpret_cast<float*>(workspace); g.ws_ld = ws_ld; g.ws_slab = ws_slab; }
        else MFX_HIP_CHECK(mfx::zero_async(dw, dw_bytes, st));
        dim3 grid(cdivt(g.K, bt), cdivt(Cout, bt), cdivt(g.M, g.m_per_block));
        if (dtype == MFX_BF16) {
            if (bt == 128) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<bf16_t, 128>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, g, dw);
            else hipLaunchKernelGGL((conv_wgrad_mfma_kernel<bf16_t, 64>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, g, dw);
        } else {
            if (bt == 128) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<half_t, 128>), grid, dim3(256), 0, st, (const half_t*)x, (const half_t*)dy, g, dw);
            else hipLaunchKernelGGL((conv_wgrad_mfma_kernel<half_t, 64>), grid, dim3(256), 0, st, (const half_t*)x, (const half_t*)dy, g, dw);
        }
        if (ws_ok) {
            const long total = (long)Cout * g.K;
            hipLaunchKernelGGL(wgrad_reduce_kernel, WR_GRID(total), dim3(256), 0, st, g.ws, nslab, ws_slab, ws_ld, g, dw);
        }
        MFX_HIP_CHECK(hipGetLastError());
        return MFX_OK;
    }
    MFX_HIP_CHECK(mfx::zero_async(dw, dw_bytes, st));
    if (g_opt_det) g.m_per_block = g.M;                          // one slab: every element of dw receives exactly one add
    dim3 grid(cdivt(g.K, 64), cdivt(Cout, 64), cdivt(g.M, g.m_per_block));
    DISPATCH_T(dtype, hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)dy, g, dw),
                      hipLaunchKernelGGL(conv_wgrad_kernel<T16>, grid, dim3(256), 0, st, (const T16*)x, (const T16*)dy, g, dw));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// library-internal (dcn_bwd_tile.hip): sum `nslab` partial gradient blocks ws[slab][Cout][K] (k = tap*Ck + c) into dW (Cout, Ck, kh, kw)
int mfx_internal_wgrad_slab_sum(const float* ws, int nslab, int Cout, int Ck, int kh, int kw, float* dw_oihw, void* stream) {
    WgradGeom g = {};
    g.Cout = Cout; g.Ck = Ck; g.kh = kh; g.kw = kw; g.K = kh * kw * Ck; g.oihw = 1; g.Cin_out = Ck; g.Cout_out = Cout;
    const long total = (long)Cout * g.K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, WR_GRID(total), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ws, nslab, total, g.K, g, dw_oihw);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// library-internal (dcn_bwd_tile.hip): the DCN weight gradient is a weight gradient over the dense columns matrix
int mfx_internal_conv_wgrad(const void* x, const void* dy, float* dw, int B, int H, int W, int x_pixstride, int Ck,
                            int kh, int kw, int stride, int pad_h, int pad_w, int Ho, int Wo, int Cout, int ldy,
                            int dtype, int oihw, int Cin_out, int Cout_out, void* stream, int dil_w,
                            void* workspace, size_t workspace_bytes, int direct) {
    return conv_wgrad_impl(x, dy, dw, B, H, W, x_pixstride, Ck, kh, kw, stride, pad_h, pad_w, Ho, Wo, Cout, ldy, dtype, oihw, Cin_out, Cout_out,
                           stream, dil_w, workspace, workspace_bytes, direct);
}

extern "C" int mfx_conv_wgrad_nhwc(const void* x, const void* dy, float* dw, int B, int H, int W, int x_pixstride, int Ck,
                                   int kh, int kw, int stride, int pad_h, int pad_w, int Ho, int Wo, int Cout, int ldy,
                                   int dtype, void* stream) {
    return conv_wgrad_impl(x, dy, dw, B, H, W, x_pixstride, Ck, kh, kw, stride, pad_h, pad_w, Ho, Wo, Cout, ldy, dtype, 0, 0, 0, stream);
}

extern "C" int mfx_conv_wgrad_nhwc_dil(const void* x, const void* dy, float* dw, int B, int H, int W, int x_pixstride, int Ck,
                                       int kh, int kw, int stride, int pad_h, int pad_w, int dil_w, int Ho, int Wo, int Cout, int ldy,
                                       int dtype, void* stream) {
    if (dil_w < 1) return mfx_fail(MFX_ERR_ARG, "conv_wgrad: dil_w must be >= 1");
    return conv_wgrad_impl(x, dy, dw, B, H, W, x_pixstride, Ck, kh, kw, stride, pad_h, pad_w, Ho, Wo, Cout, ldy, dtype, 0, 0, 0, stream, dil_w);
}

extern "C" int mfx_conv_wgrad_oihw(const void* x, const void* dy, float* dw, int B, int H, int W, int x_pixstride, int Ck,
                                   int kh, int kw, int stride, int pad_h, int pad_w, int Ho, int Wo, int Cout, int ldy,
                                   int Cout_real, int Cin_real, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (Cout_real < 1 || Cout_real > Cout || Cin_real < 1 || Cin_real > Ck) return mfx_fail(MFX_ERR_ARG, "conv_wgrad_oihw: bad real channel counts");
    return conv_wgrad_impl(x, dy, dw, B, H, W, x_pixstride, Ck, kh, kw, stride, pad_h, pad_w, Ho, Wo, Cout, ldy, dtype, 1, Cin_real, Cout_real, stream, 1,
                           workspace, workspace_bytes);
}

extern "C" int mfx_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int kh, int kw, int mode, void* packed, void* frag,
                                    int rows_pad, int K_pad, int ck, int dtype, void* stream) {
    if (!w_oihw || !packed) return mfx_fail(MFX_ERR_ARG, "pack_conv_weight: null pointer");
    const int E = dtype == MFX_F32 ? 4 : 8;
    if (mode < 0 || mode > 1 || ck < 1 || K_pad < kh * kw * ck || K_pad % (4 * E) != 0 || rows_pad % 16 != 0 ||
        rows_pad < (mode == 0 ? Cout : Cin) || ck < (mode == 0 ? Cin : Cout))
        return mfx_fail(MFX_ERR_ARG, "pack_conv_weight: bad geometry");
    const long total = (long)rows_pad * K_pad;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_conv_weight_kernel<float>, TR_GRID(total), dim3(256), 0, st, w_oihw, Cout, Cin, kh, kw, mode, (float*)packed, (float*)frag, rows_pad, K_pad, ck),
                      hipLaunchKernelGGL(pack_conv_weight_kernel<T16>, TR_GRID(total), dim3(256), 0, st, w_oihw, Cout, Cin, kh, kw, mode, (T16*)packed, (T16*)frag, rows_pad, K_pad, ck));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_pack_chunk_elems(void) { return PACK_CHUNK; }

extern "C" int mfx_pack_conv_weights_batched(const mfx_pack_desc* descs_dev, const long long* prefix_dev, int n, long long total_chunks,
                                             int dtype, void* stream) {
    if (n <= 0 || total_chunks <= 0) return MFX_OK;
    if (!descs_dev || !prefix_dev) return mfx_fail(MFX_ERR_ARG, "pack_conv_weights_batched: null pointer");
    if (total_chunks >= (1LL << 31)) return mfx_fail(MFX_ERR_ARG, "pack_conv_weights_batched: too many chunks");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_conv_weight_batched_kernel<float>, dim3((unsigned)total_chunks), dim3(256), 0, st, descs_dev, prefix_dev, n),
                      hipLaunchKernelGGL(pack_conv_weight_batched_kernel<T16>, dim3((unsigned)total_chunks), dim3(256), 0, st, descs_dev, prefix_dev, n));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

static int bn_rows_per_block(long M, int C, int dtype, int owners);

// library-internal (dcn_bwd_tile.hip): column sums ADDED into `out`, which an earlier kernel of the caller has zeroed
int mfx_internal_colsum_add(const void* x, float* out, long M, int C, int ld, int dtype, void* stream);

extern "C" int mfx_colsum(const void* x, float* out, long M, int C, int ld, int dtype, void* stream) {
    if (!x || !out) return mfx_fail(MFX_ERR_ARG, "colsum: null pointer");
    MFX_HIP_CHECK(mfx::zero_async(out, (size_t)C * sizeof(float), reinterpret_cast<hipStream_t>(stream)));
    return mfx_internal_colsum_add(x, out, M, C, ld, dtype, stream);
}

extern "C" int mfx_colsum_add(const void* x, float* out, long M, int C, int ld, int dtype, void* stream) {
    if (!x || !out) return mfx_fail(MFX_ERR_ARG, "colsum_add: null pointer");
    return mfx_internal_colsum_add(x, out, M, C, ld, dtype, stream);
}

int mfx_internal_colsum_add(const void* x, float* out, long M, int C, int ld, int dtype, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (M == 0) return MFX_OK;
    {
        const int E = dtype == MFX_F32 ? 4 : 8;
        if (C % E == 0 && ld % E == 0 && C / E <= 256 && 256 % (C / E) == 0) {
            const int rows2 = bn_rows_per_block(M, C, dtype, 1);
            const size_t smem = (size_t)(256 / (C / E)) * C * sizeof(float);
            DISPATCH_T(dtype, hipLaunchKernelGGL(colsum_chunk_kernel<float>, dim3(cdivt(M, rows2)), dim3(256), smem, st, (const float*)x, M, C, ld, rows2, out),
                              hipLaunchKernelGGL(colsum_chunk_kernel<T16>, dim3(cdivt(M, rows2)), dim3(256), smem, st, (const T16*)x, M, C, ld, rows2, out));
            MFX_HIP_CHECK(hipGetLastError());
            return MFX_OK;
        }
    }
    const int rows = g_opt_det ? (int)M : (M >= (1 << 18) ? 1024 : 128);             // >= ~2 workgroups per CU also on the small head maps
    dim3 grid(cdivt(M, rows), cdivt(C, 64)), block(64, 4);
    DISPATCH_T(dtype, hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, st, (const float*)x, (int)M, C, ld, rows, out),
                      hipLaunchKernelGGL(colsum_kernel<T16>, grid, block, 0, st, (const T16*)x, (int)M, C, ld, rows, out));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

int g_opt_bn_apply_blocks = 1024;   // option "bn_apply_blocks": workgroup cap of the two-launch forms' streaming kernels
int g_opt_bn_blocks = 768;     // option "bn_blocks": target workgroup count of the column reductions (BN statistics / backward sums / bias sums)

// rows per workgroup of the column reductions: every workgroup ends with one global atomic per column, all workgroups on the
// same 2C addresses (~12 ns each when they collide), so the count is bounded (~3 per CU, 4+ rows in flight per thread hide the
// latency instead of more workgroups); a multiple of the rows one pass of the 256 threads covers
// `owners`: the number of separate destination copies the launch adds into (workgroup b -> copy b % owners).  Deterministic mode
// launches at most one workgroup per copy, so every copy has a single writer and its adds happen in program order.
static int bn_rows_per_block(long M, int C, int dtype, int owners) {
    const int E = dtype == MFX_F32 ? 4 : 8, rstep = std::max(1, 256 / (C / E));
    const int target = g_opt_det ? std::max(1, owners) : (g_opt_bn_blocks > 0 ? g_opt_bn_blocks : 768);
    if (g_opt_det) {
        long rows = (M + target - 1) / target;
        rows = (rows + rstep - 1) / rstep * rstep;
        return (int)std::max<long>(rows, rstep);
    }
    long rows = (M + target - 1) / target;
    rows = std::max<long>(rows, 8L * rstep);
    rows = (rows + rstep - 1) / rstep * rstep;
    return (int)std::min<long>(rows, 1 << 20);
}

static int bn_check(int C, int dtype) {
    const int E = dtype == MFX_F32 ? 4 : 8;
    if (C % E != 0 || C / E > 256 || (256 % (C / E)) != 0) return mfx_fail(MFX_ERR_ARG, "bn: C must be a power-of-two multiple of one 16-byte chunk (<= 256 chunks)");
    return MFX_OK;
}

extern "C" int mfx_bn_stats(const void* x, float* sum, float* sumsq, long M, int C, int dtype, void* stream) {
    if (!x || !sum || !sumsq) return mfx_fail(MFX_ERR_ARG, "bn_stats: null pointer");
    int rc = bn_check(C, dtype); if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (sumsq == sum + C) { MFX_HIP_CHECK(mfx::zero_async(sum, (size_t)2 * C * sizeof(float), st)); }      // one fill for the usual packed pair
    else {
        MFX_HIP_CHECK(mfx::zero_async(sum, (size_t)C * sizeof(float), st));
        MFX_HIP_CHECK(mfx::zero_async(sumsq, (size_t)C * sizeof(float), st));
    }
    if (M == 0) return MFX_OK;
    const int rows = bn_rows_per_block(M, C, dtype, 1);
    const size_t smem = (size_t)(256 / (C / (dtype == MFX_F32 ? 4 : 8))) * 2 * C * sizeof(float);
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(cdivt(M, rows)), dim3(256), smem, st, (const float*)x, M, C, rows, sum, sumsq, 1),
                      hipLaunchKernelGGL(bn_stats_kernel<T16>, dim3(cdivt(M, rows)), dim3(256), smem, st, (const T16*)x, M, C, rows, sum, sumsq, 1));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_bn_finalize(const float* sum, const float* sumsq, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, long count, float* mean, float* rstd, float* scale,
                               float* shift, int C, void* stream) {
    if (!sum || !sumsq || !gamma || !beta || !mean || !rstd || !scale || !shift) return mfx_fail(MFX_ERR_ARG, "bn_finalize: null pointer");
    if ((running_mean == nullptr) != (running_var == nullptr)) return mfx_fail(MFX_ERR_ARG, "bn_finalize: running_mean/var must come together");
    if (C <= 0 || count <= 0) return MFX_OK;
    const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdivt(C, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), sum, sumsq, gamma, beta,
                       running_mean, running_var, momentum, eps, (float)(1.0 / (double)count), unbias, mean, rstd, scale, shift, C);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_bn_act_fwd(const void* x, const float* scale, const float* shift, const void* res, void* y,
                              long M, int C, int act, int dtype, void* stream) {
    if (!x || !scale || !shift || !y) return mfx_fail(MFX_ERR_ARG, "bn_act_fwd: null pointer");
    int rc = bn_check(C, dtype); if (rc) return rc;
    if (M == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long chunks = M * (C / (dtype == MFX_F32 ? 4 : 8));
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_act_fwd_kernel<float>, TR_GRID(chunks), dim3(256), 0, st, (const float*)x, scale, shift, (const float*)res, (float*)y, chunks, C, act),
                      hipLaunchKernelGGL(bn_act_fwd_kernel<T16>, TR_GRID(chunks), dim3(256), 0, st, (const T16*)x, scale, shift, (const T16*)res, (T16*)y, chunks, C, act));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_bn_bwd_reduce(const void* x, const void* a, const void* da, const float* mean, const float* rstd,
                                 float* sg, float* sgx, long M, int C, int act, int dtype, void* stream) {
    if (!x || !da || !mean || !rstd || !sg || !sgx || (act != MFX_ACT_NONE && !a)) return mfx_fail(MFX_ERR_ARG, "bn_bwd_reduce: null pointer");
    int rc = bn_check(C, dtype); if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (sgx == sg + C) { MFX_HIP_CHECK(mfx::zero_async(sg, (size_t)2 * C * sizeof(float), st)); }
    else {
        MFX_HIP_CHECK(mfx::zero_async(sg, (size_t)C * sizeof(float), st));
        MFX_HIP_CHECK(mfx::zero_async(sgx, (size_t)C * sizeof(float), st));
    }
    if (M == 0) return MFX_OK;
    const int rows = bn_rows_per_block(M, C, dtype, 1);
    const size_t smem = (size_t)(256 / (C / (dtype == MFX_F32 ? 4 : 8))) * 2 * C * sizeof(float);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(cdivt(M, rows)), dim3(256), smem, st, (const float*)x, (const float*)a, (const float*)da, mean, rstd, M, C, rows, act, sg, sgx, 1),
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<T16>, dim3(cdivt(M, rows)), dim3(256), smem, st, (const T16*)x, (const T16*)a, (const T16*)da, mean, rstd, M, C, rows, act, sg, sgx, 1));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_bn_bwd_apply(const void* x, const void* a, const void* da, const float* mean, const float* rstd, const float* gamma,
                                const float* sg, const float* sgx, void* dx, void* dres, long M, long M_total, int C, int act,
                                int dtype, void* stream) {
    if (!x || !da || !mean || !rstd || !gamma || !sg || !sgx || !dx || (act != MFX_ACT_NONE && !a)) return mfx_fail(MFX_ERR_ARG, "bn_bwd_apply: null pointer");
    int rc = bn_check(C, dtype); if (rc) return rc;
    if (M == 0) return MFX_OK;
    if (M_total < M) return mfx_fail(MFX_ERR_ARG, "bn_bwd_apply: M_total < M");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long chunks = M * (C / (dtype == MFX_F32 ? 4 : 8));
    const float invM = 1.f / (float)M_total;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, BN_APPLY_GRID(chunks), dim3(256), (size_t)3 * C * sizeof(float), st, (const float*)x, (const float*)a, (const float*)da, mean, rstd, gamma, sg, sgx, invM, (float*)dx, (float*)dres, chunks, C, act),
        hipLaunchKernelGGL(bn_bwd_apply_kernel<T16>, BN_APPLY_GRID(chunks), dim3(256), (size_t)3 * C * sizeof(float), st, (const T16*)x, (const T16*)a, (const T16*)da, mean, rstd, gamma, sg, sgx, invM, (T16*)dx, (T16*)dres, chunks, C, act));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// scratch of the two-launch forms: [0, 1024) forward sums, [1024, 2048) backward sums, then the two ticket blocks; ZERO on entry,
// zero again on exit.  One per BN layer, persistent.
#define BN_FUSED_GRID(total) dim3((unsigned)std::min<long>(cdivt((total), 256), std::max(64, g_opt_bn_apply_blocks)))
static int bn_ncopy(int C) { return std::max(1, BN_SCRATCH_COLS / (2 * C)); }

constexpr int BN_TICKET_WORDS = BN_TICKET_GROUPS * BN_TICKET_STRIDE + BN_TICKET_STRIDE;
extern "C" int mfx_bn_ncopy(int C) { return C > 0 && 2 * C <= BN_SCRATCH_COLS ? bn_ncopy(C) : 0; }

// grid-barrier words of the one-pass kernels (below): 32 first-level counters on separate lines, then [top, stuck flag, ...], then 32 go words
constexpr int BN_BAR_GROUPS = 32, BN_BAR_STRIDE = 32;
constexpr int BN_BAR_WORDS = (2 * BN_BAR_GROUPS + 1) * BN_BAR_STRIDE;
// layout (floats / words): [fwd sums 1024][bwd sums 1024][fwd tickets][bwd tickets][barrier]
extern "C" size_t mfx_bn_scratch_bytes(void) { return (size_t)(2 * BN_SCRATCH_COLS + 2 * BN_TICKET_WORDS + BN_BAR_WORDS) * sizeof(float); }


// ------------------------------------------------------------------------------------------------
// ONE-pass train-mode BN (r06): statistics (or the backward sums) and the element-wise pass in ONE launch, the map held in REGISTERS
// between the two.  The two-launch forms read every operand twice; on the 12x40 .. 96x320 maps of DLA levels 2-5 and the DCN modules each
// of the two launches is latency-bound (15 us for 2-8 MB), so half of a step's 2.9 ms of BN passes is launch + first-touch latency.
// Here every thread loads its NCH 16-byte chunks of each operand at once (all loads in flight), folds them into the column sums
// (LDS, then one global atomic per column and workgroup into the layer's self-clearing scratch, as the two-launch forms do), meets
// the other workgroups at a GRID BARRIER, reads the totals and finishes from its registers.  The grid never exceeds the co-resident
// capacity of the device (occupancy query x CUs, checked by the launcher), so the barrier cannot starve; its counters are two-level
// (32 first-level lines), reset by the last workgroup to leave, and the spin is bounded (a flag word is set instead of hanging).
// Only ONE such kernel may be in flight on the device: they are launched on the step's main stream only.
// ------------------------------------------------------------------------------------------------
// No cache maintenance is needed around this barrier -- and none is done: agent-scope release / acquire fences write back and invalidate the
// whole L2 of an XCD, and 1900 waves doing that cost 25-30 us per launch (first form of this kernel, r06 call 48: three times SLOWER than the two
// launches).  Everything the workgroups exchange (the column sums, the counters) is only ever touched with agent-scope ATOMICS, which are performed
// at the device's coherence point past every cache; what the barrier must guarantee is order, and s_waitcnt vmcnt(0) gives it: a thread's adds into
// the sums are acknowledged before its workgroup arrives.
__device__ unsigned g_bn_onepass_stuck = 0u;      // raised when a barrier's bounded spin ran out (mfx_bn_onepass_stuck reads it)

// Arrival is two-level (32 first-level lines); the very last arrival raises one GO word per group (32 lanes, 32 lines) and every workgroup polls
// its own group's word: 16 pollers per line, none of them on a line that still takes arrivals.
__device__ __forceinline__ void bn_grid_barrier(unsigned* bar, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this thread's adds into the sums have been performed
    __syncthreads();
    if (tid < 64) {
        const unsigned G = gridDim.x, grp = blockIdx.x % BN_BAR_GROUPS;
        const unsigned members = (G - grp + BN_BAR_GROUPS - 1) / BN_BAR_GROUPS;
        const unsigned groups = G < (unsigned)BN_BAR_GROUPS ? G : (unsigned)BN_BAR_GROUPS;
        unsigned* top = bar + BN_BAR_GROUPS * BN_BAR_STRIDE;
        unsigned* go = bar + (BN_BAR_GROUPS + 1) * BN_BAR_STRIDE;
        int last = 0;
        if (tid == 0 && __hip_atomic_fetch_add(bar + grp * BN_BAR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1)
            last = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
        last = __shfl(last, 0);
        if (last && (unsigned)tid < groups) __hip_atomic_store(go + tid * BN_BAR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0 && !last) {
            unsigned spins = 0;
            while (__hip_atomic_load(go + grp * BN_BAR_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 21)) { atomicOr(top + 1, 1u); atomicOr(&g_bn_onepass_stuck, 1u); break; }   // never expected in one process: the grid is co-resident by construction
            }
        }
    }
    __syncthreads();
}

// The chunks stay PACKED in registers across the barrier: without this the compiler keeps the unpacked floats of the first phase alive (twice the
// registers for 16-bit maps) and spills.
__device__ __forceinline__ void bn_keep_packed(u32x4& c) { asm volatile("" : "+v"(c)); }

// the totals, read past every non-coherent cache (the adds were agent-scope atomics of other XCDs' workgroups in THIS launch)
__device__ __forceinline__ void bn_fold_copies_coherent(const float* sums, int C, int ncopy, float* colsum, int tid) {
    const int cols = 2 * C;
    for (int j = tid; j < cols; j += 256) {
        float t = 0.f;
        for (int k = 0; k < ncopy; ++k) t += __hip_atomic_load(sums + (size_t)k * cols + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        colsum[j] = t;
    }
    __syncthreads();
}

// bn_release_scratch + the barrier words (every workgroup is past the barrier once it holds a leave ticket)
__device__ __forceinline__ void bn_release_scratch_bar(float* sums, unsigned* cnt, unsigned* bar, unsigned ticket, int tid, int* s_last) {
    bn_release_scratch(sums, cnt, ticket, tid, s_last);
    if (*s_last && tid < 2 * BN_BAR_GROUPS + 1) bar[tid * BN_BAR_STRIDE] = 0u;   // 32 first-level counters, `top` (the stuck flag beside it stays), 32 go words
}

template <typename T, int NCH, bool HAS_RES>
__global__ __launch_bounds__(256, 2) void bn_fwd_onepass_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
        const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean, float* running_var, long long* nbt,
        float momentum, float eps, float inv_count, float unbias, float* sums, unsigned* counter, unsigned* bar, int ncopy,
        float* __restrict__ mean_out, float* __restrict__ rstd_out, long total_chunks, int C, int act) {
    constexpr int E = ElemTraits<T>::ELEMS;
    extern __shared__ float sred[];                          // [256 / CPR][2 * C]
    __shared__ float colsum[BN_SCRATCH_COLS];
    __shared__ int s_last;
    const int tid = threadIdx.x, CPR = C / E, cc = tid % CPR;
    const uint32_t nthr = gridDim.x * 256u, i0 = blockIdx.x * 256u + tid, total = (uint32_t)total_chunks;      // 256 and the grid stride are multiples of CPR: one channel chunk per thread;
    const char* xb = reinterpret_cast<const char*>(x);        // 32-bit byte offsets from uniform bases (the launcher bounds the map at 32 MB): no 64-bit address per slot and operand
    u32x4 xr[NCH], rr[HAS_RES ? NCH : 1];
#pragma unroll
    for (int s = 0; s < NCH; ++s) {
        const uint32_t i = i0 + (uint32_t)s * nthr;
        xr[s] = u32x4{0u, 0u, 0u, 0u};
        if (i < total) xr[s] = *reinterpret_cast<const u32x4*>(xb + i * 16u);
    }
    if constexpr (HAS_RES) {
#pragma unroll
        for (int s = 0; s < NCH; ++s) {
            const uint32_t i = i0 + (uint32_t)s * nthr;
            rr[s] = u32x4{0u, 0u, 0u, 0u};
            if (i < total) rr[s] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(res) + i * 16u);
        }
    }
    float acc[2][E];
#pragma unroll
    for (int e = 0; e < E; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
#pragma unroll
    for (int s = 0; s < NCH; ++s) {                          // (a missing chunk is zeros: it adds nothing)
        float v[E];
        ElemTraits<T>::unpack(xr[s], v);
#pragma unroll
        for (int e = 0; e < E; ++e) { acc[0][e] += v[e]; acc[1][e] += v[e] * v[e]; }
    }
    {
        float* sp = sums + (size_t)(blockIdx.x % ncopy) * 2 * C;
        float* const outs[2] = {sp, sp + C};
        block_reduce_columns<E, 2>(acc, sred, C, CPR, tid, outs);
    }
    bn_grid_barrier(bar, tid);
#pragma unroll
    for (int s = 0; s < NCH; ++s) { bn_keep_packed(xr[s]); if constexpr (HAS_RES) bn_keep_packed(rr[s]); }
    bn_fold_copies_coherent(sums, C, ncopy, colsum, tid);
    const unsigned ticket = tid == 0 ? bn_take_ticket(counter) : 0u;
    float sc[E], sh[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int c = cc * E + e;
        const float m = colsum[c] * inv_count;
        const float v = fmaxf(colsum[C + c] * inv_count - m * m, 0.f);
        const float r = rsqrtf(v + eps);
        sc[e] = gamma[c] * r; sh[e] = beta[c] - m * sc[e];
    }
    if (blockIdx.x == 0) {
        for (int c = tid; c < C; c += 256) {
            const float m = colsum[c] * inv_count;
            const float v = fmaxf(colsum[C + c] * inv_count - m * m, 0.f);
            mean_out[c] = m; rstd_out[c] = rsqrtf(v + eps);
            if (running_mean) {
                running_mean[c] = running_mean[c] * (1.f - momentum) + m * momentum;
                running_var[c] = running_var[c] * (1.f - momentum) + v * unbias * momentum;
            }
        }
        if (tid == 0 && nbt) *nbt += 1;
    }
#pragma unroll
    for (int s = 0; s < NCH; ++s) {
        const uint32_t i = i0 + (uint32_t)s * nthr;
        if (i < total) {
            float v[E];
            ElemTraits<T>::unpack(xr[s], v);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = v[e] * sc[e] + sh[e];
            if constexpr (HAS_RES) {
                float r[E];
                ElemTraits<T>::unpack(rr[s], r);
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] += r[e];
            }
            apply_act_chunk<E>(v, act, 0);
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(y) + i * 16u) = ElemTraits<T>::pack(v);
        }
    }
    bn_release_scratch_bar(sums, counter, bar, ticket, tid, &s_last);
}

template <typename T, int NCH, bool HAS_A>
__global__ __launch_bounds__(256, 2) void bn_bwd_onepass_kernel(const T* __restrict__ x, const T* __restrict__ a, const T* __restrict__ da,
        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
        float* sums, unsigned* counter, unsigned* bar, int ncopy, float invM, T* __restrict__ dx, T* __restrict__ dres,
        float* __restrict__ dgamma, float* __restrict__ dbeta, long total_chunks, int C, int act) {
    constexpr int E = ElemTraits<T>::ELEMS;
    extern __shared__ float sred[];                          // [256 / CPR][2 * C]
    __shared__ float colsum[BN_SCRATCH_COLS];
    __shared__ int s_last;
    const int tid = threadIdx.x, CPR = C / E, cc = tid % CPR;
    const uint32_t nthr = gridDim.x * 256u, i0 = blockIdx.x * 256u + tid, total = (uint32_t)total_chunks;
    const char* xb = reinterpret_cast<const char*>(x);
    const bool recompute = act != ACT_NONE && !HAS_A;              // see bn_bwd_reduce_kernel
    u32x4 xr[NCH], dr[NCH], ar[HAS_A ? NCH : 1];
#pragma unroll
    for (int s = 0; s < NCH; ++s) {
        const uint32_t i = i0 + (uint32_t)s * nthr;
        xr[s] = u32x4{0u, 0u, 0u, 0u}; dr[s] = u32x4{0u, 0u, 0u, 0u};
        if (i < total) { xr[s] = *reinterpret_cast<const u32x4*>(xb + i * 16u); dr[s] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(da) + i * 16u); }
        if constexpr (HAS_A) {
            ar[s] = u32x4{0u, 0u, 0u, 0u};
            if (i < total) ar[s] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a) + i * 16u);
        }
    }
    float mu[E], rs[E], sc[E], sh[E], acc[2][E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int c = cc * E + e;
        acc[0][e] = 0.f; acc[1][e] = 0.f; mu[e] = mean[c]; rs[e] = rstd[c];
        sc[e] = recompute ? gamma[c] * rs[e] : 0.f;
        sh[e] = recompute ? beta[c] - mu[e] * sc[e] : 0.f;
    }
    // derivative of the activation from its output: 1 above the threshold, `slope` below (none: threshold -inf; relu: 0 / 0; leaky: 0 / 0.01)
    const float thr = act == ACT_NONE ? -__builtin_inff() : 0.f, slope = act == ACT_LEAKY ? 0.01f : (act == ACT_RELU ? 0.f : 1.f);
    auto grad = [&](int s, float (&xv)[E], float (&gq)[E]) {      // g = da * act'(a), the two-launch forms' expression
        float dv[E], av[E];
        ElemTraits<T>::unpack(xr[s], xv); ElemTraits<T>::unpack(dr[s], dv);
        if constexpr (HAS_A) ElemTraits<T>::unpack(ar[s], av);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (!HAS_A) av[e] = xv[e] * sc[e] + sh[e];
            gq[e] = dv[e] * (av[e] > thr ? 1.f : slope);      // act_grad() without a branch on `act` (the compiler made three copies of the kernel body)
        }
    };
#pragma unroll
    for (int s = 0; s < NCH; ++s) {                          // (a missing chunk has da = 0: it adds nothing)
        float xv[E], gq[E];
        grad(s, xv, gq);
#pragma unroll
        for (int e = 0; e < E; ++e) { acc[0][e] += gq[e]; acc[1][e] += gq[e] * (xv[e] - mu[e]) * rs[e]; }
        if (NCH >= 16) __builtin_amdgcn_sched_barrier(0);     // one slot's floats at a time: the packed chunks are what must stay in registers
    }
    {
        float* sp = sums + (size_t)(blockIdx.x % ncopy) * 2 * C;
        float* const outs[2] = {sp, sp + C};
        block_reduce_columns<E, 2>(acc, sred, C, CPR, tid, outs);
    }
    bn_grid_barrier(bar, tid);
#pragma unroll
    for (int s = 0; s < NCH; ++s) { bn_keep_packed(xr[s]); bn_keep_packed(dr[s]); if constexpr (HAS_A) bn_keep_packed(ar[s]); }
    bn_fold_copies_coherent(sums, C, ncopy, colsum, tid);
    const unsigned ticket = tid == 0 ? bn_take_ticket(counter) : 0u;
    float ca[E], cb[E], cd[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int c = cc * E + e;
        const float sg = colsum[c], sgx = colsum[C + c];
        ca[e] = gamma[c] * rs[e]; cb[e] = -ca[e] * rs[e] * sgx * invM; cd[e] = -cb[e] * mu[e] - ca[e] * sg * invM;
    }
    if (blockIdx.x == 0)
        for (int c = tid; c < C; c += 256) { dgamma[c] = colsum[C + c]; dbeta[c] = colsum[c]; }
#pragma unroll
    for (int s = 0; s < NCH; ++s) {
        const uint32_t i = i0 + (uint32_t)s * nthr;
        if (i < total) {
            float xv[E], gq[E], ov[E];
            grad(s, xv, gq);
#pragma unroll
            for (int e = 0; e < E; ++e) ov[e] = ca[e] * gq[e] + cb[e] * xv[e] + cd[e];
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(dx) + i * 16u) = ElemTraits<T>::pack(ov);
            if (dres) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(dres) + i * 16u) = ElemTraits<T>::pack(gq);
        }
        if (NCH >= 16) __builtin_amdgcn_sched_barrier(0);
    }
    bn_release_scratch_bar(sums, counter, bar, ticket, tid, &s_last);
}

int g_opt_bn_onepass = 3;          // option "bn_onepass": bit 0 = backward, bit 1 = forward in one launch where the map fits (0 = the two-launch forms everywhere)
int g_opt_bn_onepass_min_chunks = 200000;       // option "bn_onepass_min_chunks": smaller maps keep the two launches (backward)
int g_opt_bn_onepass_fwd_min_chunks = 900000;   // option "bn_onepass_fwd_min_chunks": the same for the forward
int g_opt_bn_onepass_grid = 0;     // option "bn_onepass_grid": workgroup cap (0 = by map size, see bn_onepass_plan)

// co-resident workgroups of the one-pass kernels: 2 per CU by their launch bounds; asked of the runtime once per kernel
template <typename K> static int bn_onepass_capacity(K kernel, size_t smem) {
    int dev = 0, per_cu = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, smem) != hipSuccess) return 0;
    return std::min(per_cu, 2) * cus;
}

// slots per thread (2 / 4 / 8 / 16) and grid for `chunks` 16-byte chunks; 0 = the map does not fit the registers of one resident grid
static int bn_onepass_plan(long chunks, int cap, int* grid, int max_nch, long min_chunks) {
    // measured per shape (tools/bn_bench.py, B = 8, bf16; r06 calls 49 / 50, us one-pass vs two launches): the barrier costs about what a launch
    // boundary costs, so one launch wins by the re-read it saves and by nothing else.  Backward: 64 ch @ 96x320 29.6 vs 37.1 (512 workgroups),
    // 128 @ 48x160 20.4 vs 28.8 (256), 256 @ 24x80 16.4 vs 21.8 (256), 512 @ 12x40 16.5 vs 20.8 (128).  Forward: 19.2 vs 21.6, 12.9 vs 15.0, then
    // a loss (14.6 vs 12.4 at 256 channels): the forward keeps its two launches below ~0.9 M chunks.  Fewer, fatter workgroups on the smaller maps:
    // every workgroup adds 2C columns into the sums and takes part in the barrier.
    if (chunks < min_chunks) return 0;
    const int want = g_opt_bn_onepass_grid > 0 ? g_opt_bn_onepass_grid : (chunks > (1 << 20) ? 512 : (chunks > 300000 ? 256 : 128));
    cap = std::min(cap, want);
    if (cap < 32 || chunks * 16 >= (1ll << 31)) return 0;
    for (int nch = 2; nch <= max_nch; nch *= 2)
        if ((long)nch * 256 * cap >= chunks) { *grid = (int)((chunks + 256L * nch - 1) / (256L * nch)); return nch; }
    return 0;
}

#define BN_ONEPASS_NCH(NCH_, ...) switch (NCH_) { case 2: { constexpr int N_ = 2; __VA_ARGS__; } break; case 4: { constexpr int N_ = 4; __VA_ARGS__; } break; \
                                                  case 8: { constexpr int N_ = 8; __VA_ARGS__; } break; default: { constexpr int N_ = 16; __VA_ARGS__; } break; }

template <typename T>
static int bn_fwd_onepass(const void* x, const void* res, void* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          long long* nbt, float momentum, float eps, long M, int C, int act, float* scratch, float* mean, float* rstd, hipStream_t st) {
    constexpr int E = ElemTraits<T>::ELEMS;
    const size_t smem = (size_t)(256 / (C / E)) * 2 * C * sizeof(float);
    static int cap_res = -1, cap_nores = -1;
    int& cap = res ? cap_res : cap_nores;
    if (cap < 0) cap = res ? bn_onepass_capacity(bn_fwd_onepass_kernel<T, 16, true>, smem) : bn_onepass_capacity(bn_fwd_onepass_kernel<T, 16, false>, smem);
    const long chunks = M * (C / E);
    int grid = 0;
    const int nch = bn_onepass_plan(chunks, cap, &grid, 16, std::min(g_opt_bn_onepass_min_chunks, g_opt_bn_onepass_fwd_min_chunks) == 0 ? 0 : g_opt_bn_onepass_fwd_min_chunks);
    if (!nch) return 1;
    const int ncopy = bn_ncopy(C);
    const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
    unsigned* counter = reinterpret_cast<unsigned*>(scratch + 2 * BN_SCRATCH_COLS);
    unsigned* bar = counter + 2 * BN_TICKET_WORDS;
    if (res) {
        BN_ONEPASS_NCH(nch, hipLaunchKernelGGL((bn_fwd_onepass_kernel<T, N_, true>), dim3(grid), dim3(256), smem, st, (const T*)x, (const T*)res, (T*)y, gamma, beta,
                                  running_mean, running_var, nbt, momentum, eps, (float)(1.0 / (double)M), unbias, scratch, counter, bar, ncopy, mean, rstd, chunks, C, act))
    } else {
        BN_ONEPASS_NCH(nch, hipLaunchKernelGGL((bn_fwd_onepass_kernel<T, N_, false>), dim3(grid), dim3(256), smem, st, (const T*)x, (const T*)res, (T*)y, gamma, beta,
                                  running_mean, running_var, nbt, momentum, eps, (float)(1.0 / (double)M), unbias, scratch, counter, bar, ncopy, mean, rstd, chunks, C, act))
    }
    return 0;
}

template <typename T>
static int bn_bwd_onepass(const void* x, const void* a, const void* da, const float* mean, const float* rstd, const float* gamma, const float* beta,
                          void* dx, void* dres, float* dgamma, float* dbeta, long M, int C, int act, float* scratch, hipStream_t st) {
    constexpr int E = ElemTraits<T>::ELEMS;
    const size_t smem = (size_t)(256 / (C / E)) * 2 * C * sizeof(float);
    static int cap_a = -1, cap_noa = -1;
    int& cap = a ? cap_a : cap_noa;
    if (cap < 0) cap = a ? bn_onepass_capacity(bn_bwd_onepass_kernel<T, 16, true>, smem) : bn_onepass_capacity(bn_bwd_onepass_kernel<T, 16, false>, smem);
    const long chunks = M * (C / E);
    int grid = 0;
    const int nch = bn_onepass_plan(chunks, cap, &grid, a ? 8 : 16, g_opt_bn_onepass_min_chunks);      // three operands x 16 slots do not fit 256 registers
    if (!nch) return 1;
    const int ncopy = bn_ncopy(C);
    float* sums = scratch + BN_SCRATCH_COLS;
    unsigned* counter = reinterpret_cast<unsigned*>(scratch + 2 * BN_SCRATCH_COLS) + BN_TICKET_WORDS;
    unsigned* bar = reinterpret_cast<unsigned*>(scratch + 2 * BN_SCRATCH_COLS) + 2 * BN_TICKET_WORDS;
    if (a) {
        BN_ONEPASS_NCH(nch, hipLaunchKernelGGL((bn_bwd_onepass_kernel<T, N_, true>), dim3(grid), dim3(256), smem, st, (const T*)x, (const T*)a, (const T*)da, mean, rstd, gamma, beta,
                                sums, counter, bar, ncopy, 1.f / (float)M, (T*)dx, (T*)dres, dgamma, dbeta, chunks, C, act))
    } else {
        BN_ONEPASS_NCH(nch, hipLaunchKernelGGL((bn_bwd_onepass_kernel<T, N_, false>), dim3(grid), dim3(256), smem, st, (const T*)x, (const T*)a, (const T*)da, mean, rstd, gamma, beta,
                                sums, counter, bar, ncopy, 1.f / (float)M, (T*)dx, (T*)dres, dgamma, dbeta, chunks, C, act))
    }
    return 0;
}

// 1 if a one-pass BN launch gave up waiting at its grid barrier since the last reset (its outputs are then wrong): another process's kernels
// held the CUs its remaining workgroups needed (two processes training on ONE device), or two such launches were in flight.  Synchronises.
extern "C" int mfx_bn_onepass_stuck(int reset) {
    unsigned v = 0u;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_bn_onepass_stuck), sizeof(v)) != hipSuccess) return mfx_fail(MFX_ERR_LAUNCH, "bn_onepass_stuck: copy from the device failed");
    if (reset && v) { const unsigned z = 0u; if (hipMemcpyToSymbol(HIP_SYMBOL(g_bn_onepass_stuck), &z, sizeof(z)) != hipSuccess) return mfx_fail(MFX_ERR_LAUNCH, "bn_onepass_stuck: reset failed"); }
    return v ? 1 : 0;
}

extern "C" int mfx_bn_train_fwd(const void* x, const void* res, void* y, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, long long* num_batches_tracked, float momentum, float eps, long M, int C, int act,
                                int dtype, float* scratch, float* mean, float* rstd, int stats_done, void* stream) {
    if (!x || !y || !gamma || !beta || !scratch || !mean || !rstd) return mfx_fail(MFX_ERR_ARG, "bn_train_fwd: null pointer");
    if ((running_mean == nullptr) != (running_var == nullptr)) return mfx_fail(MFX_ERR_ARG, "bn_train_fwd: running_mean/var must come together");
    int rc = bn_check(C, dtype); if (rc) return rc;
    if (2 * C > BN_SCRATCH_COLS) return mfx_fail(MFX_ERR_ARG, "bn_train_fwd: C > 512");
    if (M <= 0) return mfx_fail(MFX_ERR_ARG, "bn_train_fwd: empty batch");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int E = dtype == MFX_F32 ? 4 : 8, ncopy = bn_ncopy(C);
    const int rows = bn_rows_per_block(M, C, dtype, ncopy);
    const size_t smem = (size_t)(256 / (C / E)) * 2 * C * sizeof(float);
    if (!stats_done && (g_opt_bn_onepass & 2) && !g_opt_det) {          // statistics + element-wise pass in one launch where the map fits the registers of one grid
        int r = 1;
        DISPATCH_T(dtype, r = bn_fwd_onepass<float>(x, res, y, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, M, C, act, scratch, mean, rstd, st),
                          r = bn_fwd_onepass<T16>(x, res, y, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, M, C, act, scratch, mean, rstd, st));
        if (r == 0) { MFX_HIP_CHECK(hipGetLastError()); return MFX_OK; }
    }
    if (!stats_done)
        DISPATCH_T(dtype, hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(cdivt(M, rows)), dim3(256), smem, st, (const float*)x, M, C, rows, scratch, scratch + C, ncopy),
                          hipLaunchKernelGGL(bn_stats_kernel<T16>, dim3(cdivt(M, rows)), dim3(256), smem, st, (const T16*)x, M, C, rows, scratch, scratch + C, ncopy));
    MFX_HIP_CHECK(hipGetLastError());
    const long chunks = M * (C / E);
    const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
    unsigned* counter = reinterpret_cast<unsigned*>(scratch + 2 * BN_SCRATCH_COLS);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(bn_act_fwd_fused_kernel<float>, BN_FUSED_GRID(chunks), dim3(256), 0, st, (const float*)x, (const float*)res, (float*)y, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, (float)(1.0 / (double)M), unbias, scratch, counter, ncopy, mean, rstd, chunks, C, act),
        hipLaunchKernelGGL(bn_act_fwd_fused_kernel<T16>, BN_FUSED_GRID(chunks), dim3(256), 0, st, (const T16*)x, (const T16*)res, (T16*)y, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, (float)(1.0 / (double)M), unbias, scratch, counter, ncopy, mean, rstd, chunks, C, act));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// statistics only (the sparse regression heads apply the BN themselves, at the object pixels): fold the copies, finalize, clear
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(float* sums, int ncopy, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* running_mean, float* running_var, long long* nbt, float momentum, float eps,
                                                                float inv_count, float unbias, float* __restrict__ mean_out, float* __restrict__ rstd_out, int C) {
    __shared__ float colsum[BN_SCRATCH_COLS], part[256];
    const int tid = threadIdx.x;
    bn_fold_copies(sums, C, ncopy, colsum, part, tid);
    for (int c = tid; c < C; c += 256) {
        const float m = colsum[c] * inv_count;
        const float v = fmaxf(colsum[C + c] * inv_count - m * m, 0.f);
        mean_out[c] = m; rstd_out[c] = rsqrtf(v + eps);
        if (running_mean) {
            running_mean[c] = running_mean[c] * (1.f - momentum) + m * momentum;
            running_var[c] = running_var[c] * (1.f - momentum) + v * unbias * momentum;
        }
    }
    if (tid == 0 && nbt) *nbt += 1;
    __syncthreads();
    for (int i = tid; i < BN_SCRATCH_COLS; i += 256) sums[i] = 0.f;
}

extern "C" int mfx_bn_train_stats(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  long long* num_batches_tracked, float momentum, float eps, long M, int C, int dtype, float* scratch,
                                  float* mean, float* rstd, int stats_done, void* stream) {
    if (!x || !gamma || !beta || !scratch || !mean || !rstd) return mfx_fail(MFX_ERR_ARG, "bn_train_stats: null pointer");
    if ((running_mean == nullptr) != (running_var == nullptr)) return mfx_fail(MFX_ERR_ARG, "bn_train_stats: running_mean/var must come together");
    int rc = bn_check(C, dtype); if (rc) return rc;
    if (2 * C > BN_SCRATCH_COLS || M <= 0) return mfx_fail(MFX_ERR_ARG, "bn_train_stats: C > 512 or empty batch");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int E = dtype == MFX_F32 ? 4 : 8, ncopy = bn_ncopy(C);
    const int rows = bn_rows_per_block(M, C, dtype, ncopy);
    const size_t smem = (size_t)(256 / (C / E)) * 2 * C * sizeof(float);
    if (!stats_done)
        DISPATCH_T(dtype, hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(cdivt(M, rows)), dim3(256), smem, st, (const float*)x, M, C, rows, scratch, scratch + C, ncopy),
                          hipLaunchKernelGGL(bn_stats_kernel<T16>, dim3(cdivt(M, rows)), dim3(256), smem, st, (const T16*)x, M, C, rows, scratch, scratch + C, ncopy));
    const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(1), dim3(256), 0, st, scratch, ncopy, gamma, beta, running_mean, running_var, num_batches_tracked,
                       momentum, eps, (float)(1.0 / (double)M), unbias, mean, rstd, C);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_bn_train_bwd(const void* x, const void* a, const void* da, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, long M, int C, int act, int dtype,
                                float* scratch, void* stream) {
    if (!x || !da || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !scratch || (act != MFX_ACT_NONE && !a && !beta))
        return mfx_fail(MFX_ERR_ARG, "bn_train_bwd: null pointer");
    if (act != MFX_ACT_NONE && !a && dres) return mfx_fail(MFX_ERR_ARG, "bn_train_bwd: a residual entered the activation: pass the forward output `a`");
    int rc = bn_check(C, dtype); if (rc) return rc;
    if (2 * C > BN_SCRATCH_COLS) return mfx_fail(MFX_ERR_ARG, "bn_train_bwd: C > 512");
    if (M <= 0) return mfx_fail(MFX_ERR_ARG, "bn_train_bwd: empty batch");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int E = dtype == MFX_F32 ? 4 : 8, ncopy = bn_ncopy(C);
    const int rows = bn_rows_per_block(M, C, dtype, ncopy);
    const size_t smem = (size_t)(256 / (C / E)) * 2 * C * sizeof(float);
    if ((g_opt_bn_onepass & 1) && !g_opt_det) {
        int r = 1;
        DISPATCH_T(dtype, r = bn_bwd_onepass<float>(x, a, da, mean, rstd, gamma, beta, dx, dres, dgamma, dbeta, M, C, act, scratch, st),
                          r = bn_bwd_onepass<T16>(x, a, da, mean, rstd, gamma, beta, dx, dres, dgamma, dbeta, M, C, act, scratch, st));
        if (r == 0) { MFX_HIP_CHECK(hipGetLastError()); return MFX_OK; }
    }
    float* sums = scratch + BN_SCRATCH_COLS;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(cdivt(M, rows)), dim3(256), smem, st, (const float*)x, (const float*)a, (const float*)da, mean, rstd, M, C, rows, act, sums, sums + C, ncopy, gamma, beta),
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<T16>, dim3(cdivt(M, rows)), dim3(256), smem, st, (const T16*)x, (const T16*)a, (const T16*)da, mean, rstd, M, C, rows, act, sums, sums + C, ncopy, gamma, beta));
    MFX_HIP_CHECK(hipGetLastError());
    const long chunks = M * (C / E);
    unsigned* counter = reinterpret_cast<unsigned*>(scratch + 2 * BN_SCRATCH_COLS) + BN_TICKET_WORDS;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(bn_bwd_apply_fused_kernel<float>, BN_FUSED_GRID(chunks), dim3(256), 0, st, (const float*)x, (const float*)a, (const float*)da, mean, rstd, gamma, sums, counter, ncopy, 1.f / (float)M, (float*)dx, (float*)dres, dgamma, dbeta, chunks, C, act, beta),
        hipLaunchKernelGGL(bn_bwd_apply_fused_kernel<T16>, BN_FUSED_GRID(chunks), dim3(256), 0, st, (const T16*)x, (const T16*)a, (const T16*)da, mean, rstd, gamma, sums, counter, ncopy, 1.f / (float)M, (T16*)dx, (T16*)dres, dgamma, dbeta, chunks, C, act, beta));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_bn_act_bwd(const void* x, const void* a, const void* da, const float* mean, const float* rstd, const float* gamma,
                              float* sg, float* sgx, void* dx, void* dres, long M, int C, int act, int dtype, void* stream) {
    int rc = mfx_bn_bwd_reduce(x, a, da, mean, rstd, sg, sgx, M, C, act, dtype, stream);
    if (rc) return rc;
    return mfx_bn_bwd_apply(x, a, da, mean, rstd, gamma, sg, sgx, dx, dres, M, M, C, act, dtype, stream);
}

extern "C" int mfx_maxpool2x2_bwd_nhwc(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int dtype, void* stream) {
    if (!x || !dy || !dx) return mfx_fail(MFX_ERR_ARG, "maxpool_bwd: null pointer");
    const int E = dtype == MFX_F32 ? 4 : 8;
    if (C % E != 0 || (H & 1) || (W & 1)) return mfx_fail(MFX_ERR_ARG, "maxpool_bwd: C must be a multiple of 16 bytes, H/W even");
    const long total = (long)B * (H / 2) * (W / 2) * (C / E);
    if (total == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<float>, TR_GRID(total), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, B, H, W, C),
                      hipLaunchKernelGGL(maxpool_bwd_kernel<T16>, TR_GRID(total), dim3(256), 0, st, (const T16*)x, (const T16*)dy, (T16*)dx, B, H, W, C));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

static int upsample_dw_rows_per_block(int nrows) { return (nrows + 255) / 256; }      // <= 256 workgroups: one resident round, 256 partial blocks to sum

extern "C" size_t mfx_upsample_bwd_workspace_bytes(int B, int H, int C, int f) {
    const int nrows = B * H, ppb = upsample_dw_rows_per_block(nrows);
    return (size_t)((nrows + ppb - 1) / ppb) * 4 * f * f * C * sizeof(float);
}

static int upsample_bwd_impl(const void* x, const float* w, const void* dy, void* dx, float* dw,
                             int B, int H, int W, int C, int f, int dtype, void* workspace, size_t workspace_bytes, void* stream, int dw_oihw) {
    if (!x || !w || !dy || !dx || !dw) return mfx_fail(MFX_ERR_ARG, "upsample_bwd: null pointer");
    const int E = dtype == MFX_F32 ? 4 : 8;
    if (C % E != 0 || f < 1) return mfx_fail(MFX_ERR_ARG, "upsample_bwd: bad C or f");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long total = (long)B * H * W * (C / E);
    const int nrows = B * H;
    float* part = nullptr;
    int ppb;
    if (workspace && workspace_bytes >= mfx_upsample_bwd_workspace_bytes(B, H, C, f) && total > 0) {
        part = reinterpret_cast<float*>(workspace);            // partial sums per workgroup, summed in order: no atomics, no zero-fill
        ppb = upsample_dw_rows_per_block(nrows);
    } else {
        if (dw_oihw) return mfx_fail(MFX_ERR_WORKSPACE, "upsample_bwd (parameter-layout gradient): needs the workspace of mfx_upsample_bwd_workspace_bytes");
        MFX_HIP_CHECK(mfx::zero_async(dw, (size_t)4 * f * f * C * sizeof(float), st));
        ppb = g_opt_det ? nrows : (nrows >= 1024 ? 2 : 1);     // input rows per block: >= ~512 blocks (deterministic: one workgroup)
    }
    if (total == 0) return MFX_OK;
    const int dw_items = 4 * f * f * (C / E);
    if (dw_items & (dw_items - 1)) return mfx_fail(MFX_ERR_ARG, "upsample_bwd: taps x channel chunks must be a power of two");
    const size_t dw_smem = dw_items >= 1024 ? 0 : (size_t)1024 * E * sizeof(float);          // [S][items * E], S * items = 1024
    DISPATCH_T(dtype,
        { hipLaunchKernelGGL(upsample_bwd_dx_kernel<float>, TR_GRID(total), dim3(256), 0, st, (const float*)dy, w, (float*)dx, B, H, W, C, f);
          hipLaunchKernelGGL(upsample_bwd_dw_kernel<float>, dim3(cdivt(nrows, ppb)), dim3(1024), dw_smem, st, (const float*)x, (const float*)dy, dw, B, H, W, C, f, ppb, part); },
        { hipLaunchKernelGGL(upsample_bwd_dx_kernel<T16>, TR_GRID(total), dim3(256), 0, st, (const T16*)dy, w, (T16*)dx, B, H, W, C, f);
          hipLaunchKernelGGL(upsample_bwd_dw_kernel<T16>, dim3(cdivt(nrows, ppb)), dim3(1024), dw_smem, st, (const T16*)x, (const T16*)dy, dw, B, H, W, C, f, ppb, part); });
    if (part) {
        const int n = 4 * f * f * C;
        hipLaunchKernelGGL(upsample_dw_sum_kernel, dim3(cdivt(n, 16)), dim3(256), 0, st, part, cdivt(nrows, ppb), n, dw, dw_oihw ? C : 0);
    }
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_upsample_bwd_nhwc(const void* x, const float* w, const void* dy, void* dx, float* dw,
                                     int B, int H, int W, int C, int f, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return upsample_bwd_impl(x, w, dy, dx, dw, B, H, W, C, f, dtype, workspace, workspace_bytes, stream, 0);
}

extern "C" int mfx_upsample_bwd_nhwc_oihw(const void* x, const float* w, const void* dy, void* dx, float* dw,
                                          int B, int H, int W, int C, int f, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return upsample_bwd_impl(x, w, dy, dx, dw, B, H, W, C, f, dtype, workspace, workspace_bytes, stream, 1);
}

extern "C" int mfx_zero_insert2_nhwc(const void* dy, void* up, int B, int Ho, int Wo, int C, int H, int W, int dtype, void* stream) {
    if (!dy || !up) return mfx_fail(MFX_ERR_ARG, "zero_insert2: null pointer");
    const int E = dtype == MFX_F32 ? 4 : 8;
    if (C % E != 0) return mfx_fail(MFX_ERR_ARG, "zero_insert2: C must be a multiple of 16 bytes");
    const long total = (long)B * H * W * (C / E);
    if (total == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DISPATCH_T(dtype, hipLaunchKernelGGL(zero_insert2_kernel<float>, TR_GRID(total), dim3(256), 0, st, (const float*)dy, (float*)up, B, Ho, Wo, C, H, W),
                      hipLaunchKernelGGL(zero_insert2_kernel<T16>, TR_GRID(total), dim3(256), 0, st, (const T16*)dy, (T16*)up, B, Ho, Wo, C, H, W));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}
