// Regression branches of the TRAINING step through the input's patch Gram matrix, as HIP kernels (round 5).
//
// Reference: model/head/detector_predictor.py:125-165 (conv3x3 64 -> 256, no bias -> InPlaceABN(batch statistics, leaky 0.01) -> 1x1 heads per
// branch), read by the loss at the object centres only (model/layers/utils.py:120-145).  The algebra is monoflex_amd/gram_heads.py's
// (DESIGN section 4.6): with x_p(px) the 3x3 x 64-channel patch of the shared feature map,
//     m = sum_px x_p,  G = sum_px x_p x_p^T      ->  sum_px y = W m,  sum_px y^2 = diag(W G W^T)        (batch statistics of every branch, no dense conv)
//     y(r) = W x_p(r)                             at the object rows (and, for the 3d_offset branch, at the edge-sequence pixels)
// and the backward pass sends d loss / d G through ONE 5x5 convolution of x.  Round 4 ran the 576-wide algebra and every row-sized tensor
// as torch ops differentiated by torch.autograd.grad -- 321 small `at::native` launches and 16 Tensile GEMMs per step, 2.1 ms of the 19.4 ms
// step.  Here the same arithmetic is ~20 kernels of this file plus the library's own weight-gradient / conv launches, with the
// backward written out by hand:
//     z = y sc + sh,  act = leaky(z),  out = act W2^T + b2;     sc = gamma rstd,  sh = beta - mean sc,  rstd = (var + eps)^-1/2,
//     mean = s1 / M,  var = s2 / M - mean^2,                      s1 = W m,  s2_c = w_c^T G w_c
//     d sc_tot = d sc - mean d sh;  d gamma = d sc_tot rstd;  d beta = d sh;  d var = -1/2 d sc_tot gamma rstd^3 (0 where the clamp at 0 is active)
//     d mean = -sc d sh - 2 mean d var;  d s1 = d mean / M;  d s2 = d var / M
//     d W = d y^T A  +  d s1 (x) m  +  2 diag(d s2) W G;      d m = W^T d s1;      d G = W^T diag(d s2) W
//     d A_frame = -2 A_frame d G - d m  (G = autocorrelation part - A_frame^T A_frame,  m = 9 x S0 - colsum(A_frame))
//     d x = conv5x5(x; K[a][b][d] = d R5[a][b][d] + d R5[b][a][-d]) + d S0 + scatter(d A rows)
// Precision: every GEMM whose operands are exact in the 16-bit activation type (gathered patches, the rounded trunk weights) runs on the bf16 /
// fp16 MFMA with fp32 accumulation; the row gradients d y are split into hi + lo halves (two MFMAs); W G is an f32 MFMA; gradients of sums
// over every pixel (1e-9 .. 1e-6) are normalised by their maximum before they become 16-bit operands (fp16 has no such exponents).
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"
#include "fill.h"
#include <algorithm>

namespace mfx {

constexpr int GK = 576, GC = 64, GT = 256;     // patch length (9 taps x 64 channels), input channels, trunk channels per branch
constexpr int G_ROW = MFX_OBJ_ROW;

typedef mfx_gram_desc GD;

template <typename T> __device__ __forceinline__ float g_ld(const T* p) { return ElemTraits<T>::load(p); }
template <typename T> __device__ __forceinline__ void g_st(T* p, float v) { ElemTraits<T>::store(p, v); }

// (image, centre row, centre column) of gathered row r: frame rows | object rows | edge rows
__device__ __forceinline__ void gram_coords(const GD& d, int r, int& b, int& cy, int& cx) {
    const int nf = 2 * (d.W + 2) + 2 * d.H;
    if (r < d.F) {
        b = r / nf;
        int j = r - b * nf;
        if (j < d.W + 2) { cy = -1; cx = j - 1; }
        else if (j < 2 * (d.W + 2)) { cy = d.H; cx = j - (d.W + 2) - 1; }
        else { j -= 2 * (d.W + 2); if (j < d.H) { cy = j; cx = -1; } else { cy = j - d.H; cx = d.W; } }
    } else if (r < d.F + d.N) {
        const float* t = d.rows + (size_t)(r - d.F) * G_ROW;
        b = (int)fminf(fmaxf(t[57], 0.f), (float)(d.B - 1));
        cy = (int)fminf(fmaxf(t[3], 0.f), (float)(d.H - 1));
        cx = (int)fminf(fmaxf(t[2], 0.f), (float)(d.W - 1));
    } else {
        const long long p = d.extra_rows[r - d.F - d.N];
        const long long hw = (long long)d.H * d.W;
        b = (int)(p / hw);
        const int q = (int)(p - (long long)b * hw);
        cy = q / d.W; cx = q - cy * d.W;
    }
}

// A[r][tap * 64 + c] = x[b][cy + ty][cx + tx][c] (zero outside the image), exact copies in the activation type
template <typename T>
__global__ __launch_bounds__(256) void gram_gather_kernel(GD d) {
    const T* x = reinterpret_cast<const T*>(d.x);
    T* A = reinterpret_cast<T*>(d.A);
    const long total = (long)(d.F + d.N + d.Ne) * 72;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / 72), q = (int)(i - (long)r * 72), t = q >> 3, cq = q & 7;
        int b, cy, cx;
        gram_coords(d, r, b, cy, cx);
        const int y = cy + t / 3 - 1, xx = cx + t % 3 - 1;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (y >= 0 && y < d.H && xx >= 0 && xx < d.W) v = *reinterpret_cast<const u32x4*>(x + ((size_t)(b * d.H + y) * d.W + xx) * GC + cq * 8);
        *reinterpret_cast<u32x4*>(A + (size_t)r * GK + t * GC + cq * 8) = v;
    }
}

// the branches' packed trunk weights ([256][576] each, k = tap * 64 + c) as ONE matrix Wkc [CH][576] and its transpose WkT [576][CH]
template <typename T>
__global__ __launch_bounds__(256) void gram_wk_kernel(GD d) {
    const int CH = d.nbranch * GT;
    T* Wkc = reinterpret_cast<T*>(d.Wkc);
    T* WkT = reinterpret_cast<T*>(d.WkT);
    const long total = (long)CH * GK;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i / GK), k = (int)(i - (long)c * GK);
        const T v = reinterpret_cast<const T*>(d.wk[c >> 8])[(size_t)(c & 255) * GK + k];
        Wkc[i] = v;
        WkT[(size_t)k * CH + c] = v;
    }
}

// G[(t1,a)][(t2,b)] = R5[a][b][t2 - t1] - (A_f^T A_f)[(t1,a)][(t2,b)];   m[(t,b)] = S0[b] - colsum(A_f)[(t,b)]
__global__ __launch_bounds__(256) void gram_build_kernel(GD d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= GK * GK) return;
    const int j = i / GK, k = i - j * GK;
    const int t1 = j >> 6, a = j & 63, t2 = k >> 6, b = k & 63;
    const int dy = t2 / 3 - t1 / 3, dx = t2 % 3 - t1 % 3;
    // R5h[a][b][kh][kw], kh = dy + 2 in 0..2 (displacement rows -2, -1, 0): R[a][b][d] = R[b][a][-d] supplies dy = 1, 2
    const float r = dy <= 0 ? d.R5[((a * GC + b) * 3 + dy + 2) * 5 + dx + 2] : d.R5[((b * GC + a) * 3 + 2 - dy) * 5 + 2 - dx];
    d.G[i] = r - d.P[i];
    if (i < GK) d.m[i] = d.S0[i & 63] - d.csA[i];
}

template <typename T> __device__ __forceinline__ f32x4 g_ld4(const T* p);
template <> __device__ __forceinline__ f32x4 g_ld4<bf16_t>(const bf16_t* p) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)};
}
template <> __device__ __forceinline__ f32x4 g_ld4<half_t>(const half_t* p) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    const uint32_t d0 = t.x, d1 = t.y;
    const f16x2 h0 = __builtin_bit_cast(f16x2, d0), h1 = __builtin_bit_cast(f16x2, d1);
    return f32x4{(float)h0[0], (float)h0[1], (float)h1[0], (float)h1[1]};
}

// Tm = Wk G (f32 MFMA: G carries sums over every pixel -- full fp32 operands), s1 = Wk m, s2_c = sum_k Tm[c][k] Wk[c][k].
// Workgroup = 16 trunk channels x 576 columns; wave w takes columns [144 w, 144 w + 144) = nine 16-column accumulator blocks.
template <typename T>
__global__ __launch_bounds__(256) void gram_stats_kernel(GD d) {
    __shared__ float red[4][16];
    const int CH = d.nbranch * GT;
    const T* Wkc = reinterpret_cast<const T*>(d.Wkc);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane & 15, kq = lane >> 4;
    const int c0 = blockIdx.x * 16, n0 = wave * 144;
    f32x4 acc[9];
#pragma unroll
    for (int f = 0; f < 9; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float p1 = 0.f;
    const T* wrow = Wkc + (size_t)(c0 + row) * GK + kq * 4;
    for (int s = 0; s < GK / 16; ++s) {
        const f32x4 a4 = g_ld4<T>(wrow + s * 16);
        const u32x4 a = {__float_as_uint(a4[0]), __float_as_uint(a4[1]), __float_as_uint(a4[2]), __float_as_uint(a4[3])};
        if (wave == 0) {
            const f32x4 mm = *reinterpret_cast<const f32x4*>(d.m + s * 16 + kq * 4);
            p1 += a4[0] * mm[0] + a4[1] * mm[1] + a4[2] * mm[2] + a4[3] * mm[3];
        }
#pragma unroll
        for (int f = 0; f < 9; ++f) {
            const u32x4 bq = *reinterpret_cast<const u32x4*>(d.G + (size_t)(n0 + f * 16 + row) * GK + s * 16 + kq * 4);     // G is symmetric
            mma_chunk<float>(a, bq, acc[f]);
        }
    }
    if (wave == 0) {
        p1 += __shfl_xor(p1, 16); p1 += __shfl_xor(p1, 32);
        if (lane < 16) d.sums[c0 + lane] = p1;
    }
    float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = c0 + 4 * kq + r, col = n0 + f * 16 + row;
            d.Tm[(size_t)ch * GK + col] = acc[f][r];
            q[r] += acc[f][r] * g_ld<T>(Wkc + (size_t)ch * GK + col);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) q[r] += __shfl_xor(q[r], o);
        if (row == 0) red[wave][4 * kq + r] = q[r];
    }
    __syncthreads();
    if (tid < 16) d.sums[CH + c0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// stat[0..4][CH] = mean, var (unclamped), rstd, sc = gamma rstd, sh = beta - mean sc;  running statistics (momentum update, unbiased variance)
__global__ __launch_bounds__(256) void gram_finalize_kernel(GD d) {
    const int CH = d.nbranch * GT;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= CH) return;
    const int b = c >> 8, o = c & 255;
    const double Mt = (double)d.Mt;
    const double mean = (double)d.sums[c] / Mt;
    const double varr = (double)d.sums[CH + c] / Mt - mean * mean;
    const float var = fmaxf((float)varr, 0.f);
    const float rstd = rsqrtf(var + d.eps[b]);
    const float sc = rstd * d.gamma[b][o];
    d.stat[c] = (float)mean; d.stat[CH + c] = (float)varr; d.stat[2 * CH + c] = rstd; d.stat[3 * CH + c] = sc;
    d.stat[4 * CH + c] = d.beta[b][o] - (float)mean * sc;
    if (d.run_mean[b]) {
        const float unb = var * (d.Mt / fmaxf(d.Mt - 1.f, 1.f));
        d.run_mean[b][o] += d.momentum * ((float)mean - d.run_mean[b][o]);
        d.run_var[b][o] += d.momentum * (unb - d.run_var[b][o]);
        if (o == 0 && d.nbt[b]) d.nbt[b][0] += 1;
    }
}

// out[r][n] = sum_k (A[r][k] (+ Alo[r][k])) Bm[n][k]   -- both operands K-contiguous, so a 16-byte chunk IS an MFMA fragment: no LDS.
// Workgroup = 16 rows x (FN * 64) columns, wave w = FN 16-column blocks.  MODE 0: out (fp32) = alpha * sum.
// MODE 1 (trunk rows): Y (fp32) = sum, act = leaky(Y sc + sh) as fp32 (objects) or in the activation type (edge rows).
struct RowGemm { const void* A; const void* Alo; const void* Bm; float* out; void* act; const float* sc; const float* sh; const float* alpha;
                 int R, K, lda, ldb, ldo, ncols, act16; float alpha_mul; };

template <typename T, int FN, int MODE>
__global__ __launch_bounds__(256) void gram_rowgemm_kernel(RowGemm g) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane & 15, kq = lane >> 4;
    const int r0 = blockIdx.x * 16, n0 = blockIdx.y * (FN * 64) + wave * (FN * 16);
    const T* A = reinterpret_cast<const T*>(g.A);
    const T* Alo = reinterpret_cast<const T*>(g.Alo);
    const T* Bm = reinterpret_cast<const T*>(g.Bm);
    const int ra = min(r0 + row, g.R - 1);
    f32x4 acc[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const T* ap = A + (size_t)ra * g.lda + kq * 8;
    const T* lp = Alo ? Alo + (size_t)ra * g.lda + kq * 8 : nullptr;
    const T* bp = Bm + (size_t)(n0 + row) * g.ldb + kq * 8;
    // fragments of step s + 1 are in flight while step s multiplies (both operands come straight from L2: no LDS, no barrier)
    const int ns = g.K / 32;
    u32x4 a = *reinterpret_cast<const u32x4*>(ap), al = {0u, 0u, 0u, 0u}, bq[FN];
    if (lp) al = *reinterpret_cast<const u32x4*>(lp);
#pragma unroll
    for (int j = 0; j < FN; ++j) bq[j] = *reinterpret_cast<const u32x4*>(bp + (size_t)j * 16 * g.ldb);
    for (int s = 0; s < ns; ++s) {
        u32x4 an = a, aln = al, bn[FN];
        const int s1 = s + 1 < ns ? s + 1 : s;
        an = *reinterpret_cast<const u32x4*>(ap + s1 * 32);
        if (lp) aln = *reinterpret_cast<const u32x4*>(lp + s1 * 32);
#pragma unroll
        for (int j = 0; j < FN; ++j) bn[j] = *reinterpret_cast<const u32x4*>(bp + (size_t)j * 16 * g.ldb + s1 * 32);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            mma_chunk<T>(a, bq[j], acc[j]);
            if (lp) mma_chunk<T>(al, bq[j], acc[j]);
        }
        a = an; al = aln;
#pragma unroll
        for (int j = 0; j < FN; ++j) bq[j] = bn[j];
    }
    const float alpha = MODE == 0 ? g.alpha_mul * (g.alpha ? g.alpha[0] : 1.f) : 1.f;
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = r0 + 4 * kq + r, n = n0 + j * 16 + row;
            if (rr >= g.R || n >= g.ncols) continue;
            const float v = acc[j][r];
            if (MODE == 0) g.out[(size_t)rr * g.ldo + n] = v * alpha;
            else {
                g.out[(size_t)rr * g.ldo + n] = v;
                const float z = v * g.sc[n] + g.sh[n];
                const float a_ = z > 0.f ? z : 0.01f * z;
                if (g.act16) g_st<T>(reinterpret_cast<T*>(g.act) + (size_t)rr * g.ldo + n, a_);
                else reinterpret_cast<float*>(g.act)[(size_t)rr * g.ldo + n] = a_;
            }
        }
}

// out[n][off_b + k] = b2[k] + sum_c W2_b[k][c] act[n][256 b + c]   (zero rows for the empty slots of the object table)
__global__ __launch_bounds__(256) void gram_out_kernel(GD d) {
    __shared__ float part[4][32];
    const int CH = d.nbranch * GT;
    const int n = blockIdx.x, b = blockIdx.y, c = threadIdx.x, lane = c & 63, wave = c >> 6;
    float* o = d.out + (size_t)n * d.ld_out + d.off[b];
    const int kb = d.k[b];
    if (d.rows[(size_t)n * G_ROW] <= 0.f) { if (c < kb) o[c] = 0.f; return; }
    const float a = d.act[(size_t)n * CH + b * GT + c];
    for (int k = 0; k < kb; ++k) {
        float v = d.w2[b][k * GT + c] * a;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
        if (lane == 0) part[wave][k] = v;
    }
    __syncthreads();
    if (c < kb) o[c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]) + (d.b2[b] ? d.b2[b][c] : 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------------------------

template <typename T> __device__ __forceinline__ void g_split(float v, T* hi, T* lo, size_t i) {
    g_st<T>(hi + i, v);
    g_st<T>(lo + i, v - g_ld<T>(hi + i));
}

// object rows (blockIdx.x < nchunk_obj) and edge rows of branch `extra_branch`: thread = trunk channel.
//   d act -> d z -> d y = d z sc (hi + lo halves);  d sc += d z y,  d sh += d z;  d W2 += d out (x) act,  d b2 += d out
template <typename T>
__global__ __launch_bounds__(256) void gram_rows_bwd_kernel(GD d, int rpb, int nchunk_obj, int rpb_e) {
    const int CH = d.nbranch * GT;
    const int c = threadIdx.x, b = blockIdx.y, ch = b * GT + c;
    const float sc = d.stat[3 * CH + ch], sh = d.stat[4 * CH + ch];
    float dsc = 0.f, dsh = 0.f;
    if ((int)blockIdx.x < nchunk_obj) {
        const int n0 = blockIdx.x * rpb, n1 = min(n0 + rpb, d.N);
        const int kb = d.k[b];
        float w[32], dw[32];
        for (int k = 0; k < 32; ++k) { w[k] = k < kb ? d.w2[b][k * GT + c] : 0.f; dw[k] = 0.f; }
        float db = 0.f;
        T* yh = reinterpret_cast<T*>(d.dYh); T* yl = reinterpret_cast<T*>(d.dYl);
        for (int n = n0; n < n1; ++n) {
            float gy = 0.f;
            if (d.rows[(size_t)n * G_ROW] > 0.f) {
                const float y = d.Y[(size_t)n * CH + ch];
                const float z = y * sc + sh, a = z > 0.f ? z : 0.01f * z;
                const float* dd = d.dout + (size_t)n * d.ld_out + d.off[b];
                float da = 0.f;
#pragma unroll 4
                for (int k = 0; k < kb; ++k) { const float dk = dd[k]; da += dk * w[k]; dw[k] += dk * a; }
                const float gq = da * (z > 0.f ? 1.f : 0.01f);
                dsh += gq; dsc += gq * y;
                if (c < kb) db += dd[c];
                gy = gq * sc;
            }
            g_split<T>(gy, yh, yl, (size_t)n * CH + ch);
        }
        for (int k = 0; k < kb; ++k) unsafeAtomicAdd(d.dw2[b] + k * GT + c, dw[k]);
        if (c < kb && d.db2[b]) unsafeAtomicAdd(d.db2[b] + c, db);
    } else {
        if (b != d.extra_branch || d.Ne == 0) return;
        const int e0 = ((int)blockIdx.x - nchunk_obj) * rpb_e, e1 = min(e0 + rpb_e, d.Ne);
        const T* de = reinterpret_cast<const T*>(d.dact_e);
        T* yh = reinterpret_cast<T*>(d.dYeh); T* yl = reinterpret_cast<T*>(d.dYel);
        for (int e = e0; e < e1; ++e) {
            const float y = d.Ye[(size_t)e * GT + c];
            const float z = y * sc + sh;
            const float gq = (de ? g_ld<T>(de + (size_t)e * GT + c) : 0.f) * (z > 0.f ? 1.f : 0.01f);
            dsh += gq; dsc += gq * y;
            g_split<T>(gq * sc, yh, yl, (size_t)e * GT + c);
        }
    }
    unsafeAtomicAdd(d.dsum + ch, dsc);
    unsafeAtomicAdd(d.dsum + CH + ch, dsh);
}

// per channel: d gamma, d beta, d s1, d s2 from (d sc, d sh)
__global__ __launch_bounds__(256) void gram_stats_bwd_kernel(GD d) {
    const int CH = d.nbranch * GT;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= CH) return;
    const int b = c >> 8, o = c & 255;
    const float mean = d.stat[c], varr = d.stat[CH + c], rstd = d.stat[2 * CH + c], sc = d.stat[3 * CH + c];
    const float dsc = d.dsum[c], dsh = d.dsum[CH + c];
    const float gam = d.gamma[b][o];
    const float dsc_tot = dsc - mean * dsh;
    if (d.dgamma[b]) d.dgamma[b][o] = dsc_tot * rstd;
    if (d.dbeta[b]) d.dbeta[b][o] = dsh;
    const float dvar = varr >= 0.f ? -0.5f * dsc_tot * gam * rstd * rstd * rstd : 0.f;
    const float dmean = -sc * dsh - 2.f * mean * dvar;
    d.ds[c] = dmean / d.Mt;
    d.ds[CH + c] = dvar / d.Mt;
}

// scal[slot] = max |v[i]|: the bit pattern of a non-negative float orders like an unsigned integer, so the workgroups meet in one atomicMax
// (exact and order-independent); the slot is zero before (phase 3 clears the arena that holds `scal`)
__global__ __launch_bounds__(256) void gram_absmax_kernel(const float* v, long n, float* scal, int slot) {
    __shared__ float red[4];
    float m = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(v[i]));
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned int*>(scal) + slot, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// Dw16[c][j] = (d s2[c] / max |d s2|) Wk[c][j] in the activation type;   d m[j] = sum_c Wk[c][j] d s1[c]
template <typename T>
__global__ __launch_bounds__(256) void gram_dw16_kernel(GD d) {
    const int CH = d.nbranch * GT;
    const T* Wkc = reinterpret_cast<const T*>(d.Wkc);
    T* D = reinterpret_cast<T*>(d.Dw16);
    const float smax = d.scal[0], inv = smax > 0.f ? 1.f / smax : 0.f;
    const long total = (long)CH * GK;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i / GK);
        g_st<T>(D + i, d.ds[CH + c] * inv * g_ld<T>(Wkc + i));
    }
}

// d m[j] = sum_c Wk[c][j] d s1[c]: workgroup = 64 columns, the channels split over its four waves
template <typename T>
__global__ __launch_bounds__(256) void gram_dm_kernel(GD d) {
    __shared__ float red[4][64];
    const int CH = d.nbranch * GT;
    const T* Wkc = reinterpret_cast<const T*>(d.Wkc);
    const int jj = threadIdx.x & 63, part = threadIdx.x >> 6, j = blockIdx.x * 64 + jj;
    float s = 0.f;
    for (int c = part; c < CH; c += 4) s += g_ld<T>(Wkc + (size_t)c * GK + j) * d.ds[c];
    red[part][jj] = s;
    __syncthreads();
    if (part == 0) d.dm[j] = (red[0][jj] + red[1][jj]) + (red[2][jj] + red[3][jj]);
}

// Kxs[a][b][kh][kw] = dRs[a][b][d] + dRs[b][a][-d],  dRs[a][b][d] = sum over tap pairs (t1, t2) with t2 - t1 = d of dGs[(t1,a)][(t2,b)]
// (dGs = d G / max |d s2|)
__global__ __launch_bounds__(256) void gram_kx_kernel(GD d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= GC * GC * 25) return;
    const int kw = i % 5, kh = (i / 5) % 5, b = (i / 25) % GC, a = i / (25 * GC);
    const int dy = kh - 2, dx = kw - 2;
    float s = 0.f;
    for (int t1 = 0; t1 < 9; ++t1) {
        const int y1 = t1 / 3, x1 = t1 % 3;
        // + d: t2 = t1 + d;  - d: t2 = t1 - d with (a, b) exchanged
        const int y2 = y1 + dy, x2 = x1 + dx;
        if (y2 >= 0 && y2 < 3 && x2 >= 0 && x2 < 3) s += d.dGs[(size_t)(t1 * GC + a) * GK + (y2 * 3 + x2) * GC + b];
        const int y3 = y1 - dy, x3 = x1 - dx;
        if (y3 >= 0 && y3 < 3 && x3 >= 0 && x3 < 3) s += d.dGs[(size_t)(t1 * GC + b) * GK + (y3 * 3 + x3) * GC + a];
    }
    d.Kx[i] = s;
}

// Kx /= max |Kx|;  conv epilogue scale = max |Kx| max |d s2| (per channel, all equal), shift = d S0[a] = sum_t d m[t * 64 + a];
// Gn16 = dGs / max |dGs| in the activation type
template <typename T>
__global__ __launch_bounds__(256) void gram_kxfin_kernel(GD d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float smax = d.scal[0], kmax = d.scal[1], gmax = d.scal[2];
    if (i < GC * GC * 25) d.Kx[i] = kmax > 0.f ? d.Kx[i] / kmax : 0.f;
    if (i < GK * GK) g_st<T>(reinterpret_cast<T*>(d.Gn16) + i, gmax > 0.f ? d.dGs[i] / gmax : 0.f);
    if (i < GC) {
        d.cscale[i] = kmax * smax;
        float s = 0.f;
        for (int t = 0; t < 9; ++t) s += d.dm[t * GC + i];
        d.cshift[i] = s;
        if (i == 0) d.scal[3] = -2.f * smax * gmax;             // d A_frame = scal[3] * (A_frame Gn16) - d m
    }
}

// border pixels: d x[q] += sum over the <= `width` (frame row, tap) entries that touch q of d A_frame[row][tap] (unique pixels, fixed order)
template <typename T>
__global__ __launch_bounds__(256) void gram_ring_kernel(GD d) {
    const long total = (long)d.nring * GC;
    T* dx = reinterpret_cast<T*>(d.dx);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i >> 6), c = (int)(i & 63);
        float s = 0.f;
        for (int w = 0; w < d.ring_width; ++w) {
            const long long e = d.ring_inv[(size_t)q * d.ring_width + w];
            if (e >= (long long)d.F * 9) continue;
            const int r = (int)(e / 9), t = (int)(e - (long long)r * 9);
            s += d.dAf[(size_t)r * GK + t * GC + c] - d.dm[t * GC + c];
        }
        T* p = dx + (size_t)d.ring_idx[q] * GC + c;
        g_st<T>(p, g_ld<T>(p) + s);
    }
}

__device__ __forceinline__ void g_atomic_add8(bf16_t* p, const float (&v)[8]) {
    typedef short s2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const uint32_t t = ElemTraits<bf16_t>::pack2(v[k], v[k + 1]);
        __builtin_amdgcn_global_atomic_fadd_v2bf16((s2_t __attribute__((address_space(1)))*)(p + k), __builtin_bit_cast(s2_t, t));
    }
}
__device__ __forceinline__ void g_atomic_add8(half_t* p, const float (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const f16x2 t = __builtin_convertvector((f32x2){v[k], v[k + 1]}, f16x2);
        __builtin_amdgcn_global_atomic_fadd_v2f16((f16x2 __attribute__((address_space(1)))*)(p + k), t);
    }
}

// object and edge rows: d x[centre + tap] += d A[row][tap] (inside the image).  Rows can share pixels: packed 16-bit atomics
template <typename T>
__global__ __launch_bounds__(256) void gram_scatter_kernel(GD d) {
    const long total = (long)(d.N + d.Ne) * 72;
    T* dx = reinterpret_cast<T*>(d.dx);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int rr = (int)(i / 72), q = (int)(i - (long)rr * 72), t = q >> 3, cq = q & 7;
        if (rr < d.N && d.rows[(size_t)rr * G_ROW] <= 0.f) continue;
        int b, cy, cx;
        gram_coords(d, d.F + rr, b, cy, cx);
        const int y = cy + t / 3 - 1, xx = cx + t % 3 - 1;
        if (y < 0 || y >= d.H || xx < 0 || xx >= d.W) continue;
        const float* src = d.dArows + (size_t)rr * GK + t * GC + cq * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[e];
        g_atomic_add8(dx + ((size_t)(b * d.H + y) * d.W + xx) * GC + cq * 8, v);
    }
}

// d W_b (O, I, 3, 3) = d y^T A (object rows [+ edge rows of branch e]) + d s1 (x) m + 2 d s2 (Wk G)
__global__ __launch_bounds__(256) void gram_dwk_kernel(GD d) {
    const int CH = d.nbranch * GT;
    const long total = (long)CH * GK;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i / GK), k = (int)(i - (long)c * GK);
        const int b = c >> 8, o = c & 255;
        if (!d.dwt[b]) continue;
        float v = d.dwo[i] + d.ds[c] * d.m[k] + 2.f * d.ds[CH + c] * d.Tm[i];
        if (b == d.extra_branch && d.Ne > 0) v += d.dwe[(size_t)o * GK + k];
        const int t = k >> 6, ci = k & 63;
        d.dwt[b][((size_t)o * GC + ci) * 9 + t] = v;
    }
}

static inline unsigned g_grid(long total) { return (unsigned)std::min<long>((total + 255) / 256, 4096); }

template <typename T, int FN, int MODE> static void launch_rowgemm(const RowGemm& g, hipStream_t st) {
    hipLaunchKernelGGL((gram_rowgemm_kernel<T, FN, MODE>), dim3((g.R + 15) / 16, (g.ncols + FN * 64 - 1) / (FN * 64)), dim3(256), 0, st, g);
}

template <typename T> static int gram_phase(const GD* dp, int phase, hipStream_t st) {
    const GD& d = *dp;
    const int CH = d.nbranch * GT;
    const int R = d.F + d.N + d.Ne;
    switch (phase) {
        case 0:                                                   // gathered patches + the contiguous weight matrices
            hipLaunchKernelGGL(gram_gather_kernel<T>, dim3(g_grid((long)R * 72)), dim3(256), 0, st, d);
            hipLaunchKernelGGL(gram_wk_kernel<T>, dim3(g_grid((long)CH * GK)), dim3(256), 0, st, d);
            break;
        case 1:                                                   // G, m, Tm = Wk G, sums (caller: all-reduce of `sums` for SyncBN)
            hipLaunchKernelGGL(gram_build_kernel, dim3((GK * GK + 255) / 256), dim3(256), 0, st, d);
            hipLaunchKernelGGL(gram_stats_kernel<T>, dim3(CH / 16), dim3(256), 0, st, d);
            break;
        case 2: {                                                 // statistics, trunk rows + activation, 1x1 heads
            hipLaunchKernelGGL(gram_finalize_kernel, dim3((CH + 255) / 256), dim3(256), 0, st, d);
            RowGemm g = {};
            g.A = reinterpret_cast<const T*>(d.A) + (size_t)d.F * GK; g.Bm = d.Wkc; g.out = d.Y; g.act = d.act; g.sc = d.stat + 3 * CH; g.sh = d.stat + 4 * CH;
            g.R = d.N; g.K = GK; g.lda = GK; g.ldb = GK; g.ldo = CH; g.ncols = CH; g.act16 = 0;
            if (d.N > 0) launch_rowgemm<T, 4, 1>(g, st);
            if (d.Ne > 0 && d.extra_branch >= 0) {
                const int e = d.extra_branch;
                g.A = reinterpret_cast<const T*>(d.A) + (size_t)(d.F + d.N) * GK; g.Bm = reinterpret_cast<const T*>(d.Wkc) + (size_t)e * GT * GK;
                g.out = d.Ye; g.act = d.act_e; g.sc = d.stat + 3 * CH + e * GT; g.sh = d.stat + 4 * CH + e * GT;
                g.R = d.Ne; g.ldo = GT; g.ncols = GT; g.act16 = 1;
                launch_rowgemm<T, 4, 1>(g, st);
            }
            if (d.N > 0) hipLaunchKernelGGL(gram_out_kernel, dim3(d.N, d.nbranch), dim3(256), 0, st, d);
            break;
        }
        case 3: {                                                 // row gradients, statistics gradients (caller: all-reduce of `ds` for SyncBN)
            // ONE fill: the caller carved dsum, scal and every branch's d W2 / d b2 out of one arena (d.dsum = its start, scal_bytes its length)
            MFX_HIP_CHECK(mfx::zero_async(d.dsum, (size_t)d.arena_bytes, st));
            const int rpb = g_opt_det ? std::max(1, d.N) : 16, nco = d.N > 0 ? (d.N + rpb - 1) / rpb : 0;
            const int rpe = g_opt_det ? std::max(1, d.Ne) : 64, nce = (d.Ne > 0 && d.extra_branch >= 0) ? (d.Ne + rpe - 1) / rpe : 0;
            if (nco + nce > 0) hipLaunchKernelGGL(gram_rows_bwd_kernel<T>, dim3(nco + nce, d.nbranch), dim3(256), 0, st, d, rpb, nco, rpe);
            hipLaunchKernelGGL(gram_stats_bwd_kernel, dim3((CH + 255) / 256), dim3(256), 0, st, d);
            break;
        }
        case 4:                                                   // max |d s2|, its 16-bit operand, d m
            hipLaunchKernelGGL(gram_absmax_kernel, dim3(8), dim3(256), 0, st, (const float*)(d.ds + CH), (long)CH, d.scal, 0);
            hipLaunchKernelGGL(gram_dw16_kernel<T>, dim3(g_grid((long)CH * GK)), dim3(256), 0, st, d);
            hipLaunchKernelGGL(gram_dm_kernel<T>, dim3(GK / 64), dim3(256), 0, st, d);
            break;
        case 5: {                                                 // (after dGs = Dw16^T Wk) 5x5 kernel, normalised d G, d A rows
            hipLaunchKernelGGL(gram_kx_kernel, dim3((GC * GC * 25 + 255) / 256), dim3(256), 0, st, d);
            hipLaunchKernelGGL(gram_absmax_kernel, dim3(100), dim3(256), 0, st, (const float*)d.Kx, (long)GC * GC * 25, d.scal, 1);
            hipLaunchKernelGGL(gram_absmax_kernel, dim3(324), dim3(256), 0, st, (const float*)d.dGs, (long)GK * GK, d.scal, 2);
            hipLaunchKernelGGL(gram_kxfin_kernel<T>, dim3((GK * GK + 255) / 256), dim3(256), 0, st, d);
            RowGemm g = {};
            g.A = d.A; g.Bm = d.Gn16; g.out = d.dAf; g.alpha = d.scal + 3; g.alpha_mul = 1.f;       // d A_frame + d m = scal[3] * A_frame Gn16
            g.R = d.F; g.K = GK; g.lda = GK; g.ldb = GK; g.ldo = GK; g.ncols = GK;
            if (d.F > 0) launch_rowgemm<T, 3, 0>(g, st);
            g = RowGemm{};
            g.A = d.dYh; g.Alo = d.dYl; g.Bm = d.WkT; g.out = d.dArows; g.alpha_mul = 1.f;
            g.R = d.N; g.K = CH; g.lda = CH; g.ldb = CH; g.ldo = GK; g.ncols = GK;
            if (d.N > 0) launch_rowgemm<T, 3, 0>(g, st);
            if (d.Ne > 0 && d.extra_branch >= 0) {
                g.A = d.dYeh; g.Alo = d.dYel; g.Bm = reinterpret_cast<const T*>(d.WkT) + (size_t)d.extra_branch * GT; g.out = d.dArows + (size_t)d.N * GK;
                g.R = d.Ne; g.K = GT; g.lda = GT;
                launch_rowgemm<T, 3, 0>(g, st);
            }
            break;
        }
        case 6:                                                   // (after d x = conv5x5) border pixels, object / edge scatter
            if (d.nring > 0) hipLaunchKernelGGL(gram_ring_kernel<T>, dim3(g_grid((long)d.nring * GC)), dim3(256), 0, st, d);
            if (d.N + d.Ne > 0) hipLaunchKernelGGL(gram_scatter_kernel<T>, dim3(g_grid((long)(d.N + d.Ne) * 72)), dim3(256), 0, st, d);
            break;
        case 7:                                                   // (after dwo = dYh^T A_o, dwe = dYeh^T A_e) trunk weight gradients
            hipLaunchKernelGGL(gram_dwk_kernel, dim3(g_grid((long)CH * GK)), dim3(256), 0, st, d);
            break;
        default: return mfx_fail(MFX_ERR_ARG, "gram_heads: bad phase");
    }
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

}  // namespace mfx
using namespace mfx;

extern "C" int mfx_gram_heads(const mfx_gram_desc* d, int phase, void* stream) {
    if (!d || !d->x || !d->rows) return mfx_fail(MFX_ERR_ARG, "gram_heads: null pointer");
    if (d->nbranch < 1 || d->nbranch > MFX_HEAD_MAX_BRANCH || d->C != GC) return mfx_fail(MFX_ERR_ARG, "gram_heads: 1..8 branches on a 64-channel map");
    if (d->Ne > 0 && !d->extra_rows) return mfx_fail(MFX_ERR_ARG, "gram_heads: edge rows without their pixel indices");
    for (int b = 0; b < d->nbranch; ++b)
        if (d->k[b] < 1 || d->k[b] > 32 || d->off[b] < 0 || d->off[b] + d->k[b] > d->ld_out) return mfx_fail(MFX_ERR_ARG, "gram_heads: branch outputs out of range");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MFX_BF16) return gram_phase<bf16_t>(d, phase, st);
    if (d->dtype == MFX_F16) return gram_phase<half_t>(d, phase, st);
    return mfx_fail(MFX_ERR_UNSUPPORTED, "gram_heads: 16-bit activations only (fp32 runs the torch form, monoflex_amd/gram_heads.py)");
}
