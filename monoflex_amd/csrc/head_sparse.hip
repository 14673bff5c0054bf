// Regression heads of the TRAINING step, evaluated where the loss reads them.
//
// Reference: model/head/detector_predictor.py:125-169 runs conv3x3 -> InPlaceABN(leaky 0.01) -> conv1x1 densely for every
// regression branch, and model/head/detector_loss.py:116-180 (select_point_of_interest, layers/utils.py:120-145) then gathers the
// 50 regression channels at the <= B*MAX_OBJECTS object centres only.  The dense part that cannot be avoided is the trunk
// convolution and its batch statistics; everything after them is needed at the object pixels alone:
//
//   forward   out[n][k] = b2[k] + sum_c W2[k][c] * leaky(y[p_n][c] * scale[c] + shift[c])            (one workgroup per row)
//   backward  g[n][c]   = leaky'(.) * sum_k dout[n][k] * W2[k][c]   -> Sg, Sgx (BN sums), dW2, db2   (row-chunk workgroups)
//             dx[p][c]  = B[c] * y[p][c] + D[c]  for EVERY pixel (the batch-statistics terms of the BN backward are dense),
//                         + A[c] * sum_{m: p_m = p} g[m][c] at the object pixels                      (dense pass + fix-up)
//
// so a branch's backward is ONE dense pass (read y, write dx) instead of the 1x1 data gradient, the 1x1 weight gradient, the
// bias sum, the BN reduction and the BN apply over its 126 MB activation, and its forward needs neither the activation map
// nor the dense 1x1 conv.  Branches whose activation is read elsewhere (the class head: dense focal loss; the 3d_offset head:
// edge fusion) keep the dense path.
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"
#include "fill.h"

namespace mfx {

constexpr int HS_C = 256, HS_ROW = MFX_OBJ_ROW;
enum { HS_VALID = 0, HS_CX = 2, HS_CY = 3, HS_B = 57 };          // object_loss_math.h R_VALID / R_CX / R_CY / R_B

struct HsBranch { const void* y; const float* mean; const float* rstd; const float* gamma; const float* beta; const float* w2; const float* b2;
                  float* sums; float* dw2; float* db2; void* dx; int k, off; };
struct HsGeom { int nbranch, N, B, H, W, ld_out; HsBranch br[MFX_HEAD_MAX_BRANCH]; };

__device__ __forceinline__ long hs_pixel(const float* t, int B, int H, int W) {
    const int b = min(max((int)t[HS_B], 0), B - 1), cx = min(max((int)t[HS_CX], 0), W - 1), cy = min(max((int)t[HS_CY], 0), H - 1);
    return ((long)b * H + cy) * W + cx;
}
__device__ __forceinline__ float hs_leaky(float z) { return z > 0.f ? z : 0.01f * z; }

// one workgroup (256 threads = 256 trunk channels) per (object row, branch)
template <typename T>
__global__ __launch_bounds__(256) void head_sparse_fwd_kernel(HsGeom g, const float* __restrict__ rows, float* __restrict__ out) {
    __shared__ float part[4][32];
    const int n = blockIdx.x, c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const HsBranch& b = g.br[blockIdx.y];
    const float* t = rows + (size_t)n * HS_ROW;
    float* o = out + (size_t)n * g.ld_out + b.off;
    if (t[HS_VALID] == 0.f) { if (c < b.k) o[c] = 0.f; return; }
    const long p = hs_pixel(t, g.B, g.H, g.W);
    const float sc = b.gamma[c] * b.rstd[c], sh = b.beta[c] - b.mean[c] * sc;
    const float a = hs_leaky(ElemTraits<T>::load(reinterpret_cast<const T*>(b.y) + p * HS_C + c) * sc + sh);
    for (int k = 0; k < b.k; ++k) {
        float v = b.w2[k * HS_C + c] * a;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) part[wave][k] = v;
    }
    __syncthreads();
    if (c < b.k) o[c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]) + (b.b2 ? b.b2[c] : 0.f);
}

// gradient rows: workgroup (row chunk, branch), thread = channel; Sg / Sgx / dW2 accumulate in registers over the chunk's rows
template <typename T>
__global__ __launch_bounds__(256) void head_sparse_bwd_rows_kernel(HsGeom g, const float* __restrict__ rows, const float* __restrict__ dout,
                                                                   float* __restrict__ grows, int rows_per_block) {
    const int c = threadIdx.x;
    const HsBranch& b = g.br[blockIdx.y];
    const int n0 = blockIdx.x * rows_per_block, n1 = min(n0 + rows_per_block, g.N);
    const float mu = b.mean[c], rs = b.rstd[c], sc = b.gamma[c] * rs, sh = b.beta[c] - mu * sc;
    float w[32], dw[32];
    for (int k = 0; k < 32; ++k) { w[k] = k < b.k ? b.w2[k * HS_C + c] : 0.f; dw[k] = 0.f; }
    float sg = 0.f, sgx = 0.f, db = 0.f;
    float* gb = grows + (size_t)blockIdx.y * g.N * HS_C;
    for (int n = n0; n < n1; ++n) {
        const float* t = rows + (size_t)n * HS_ROW;
        float gq = 0.f;
        if (t[HS_VALID] != 0.f) {
            const long p = hs_pixel(t, g.B, g.H, g.W);
            const float x = ElemTraits<T>::load(reinterpret_cast<const T*>(b.y) + p * HS_C + c);
            const float z = x * sc + sh, a = hs_leaky(z);
            const float* d = dout + (size_t)n * g.ld_out + b.off;
            float da = 0.f;
#pragma unroll 4
            for (int k = 0; k < b.k; ++k) { const float dk = d[k]; da += dk * w[k]; dw[k] += dk * a; }
            gq = da * (z > 0.f ? 1.f : 0.01f);
            sg += gq; sgx += gq * (x - mu) * rs;
            if (c < b.k) db += d[c];
        }
        gb[(size_t)n * HS_C + c] = gq;
    }
    unsafeAtomicAdd(b.sums + c, sg);
    unsafeAtomicAdd(b.sums + HS_C + c, sgx);
    for (int k = 0; k < b.k; ++k) unsafeAtomicAdd(b.dw2 + k * HS_C + c, dw[k]);
    if (c < b.k && b.db2) unsafeAtomicAdd(b.db2 + c, db);
}

// coefficients of dx = A*g + B*x + D from the branch's sums (bn_bwd_apply_kernel's table)
__device__ __forceinline__ void hs_coef(const HsBranch& b, int c, float invM, float& ca, float& cb, float& cd) {
    const float sg = b.sums[c], sgx = b.sums[HS_C + c];
    ca = b.gamma[c] * b.rstd[c]; cb = -ca * b.rstd[c] * sgx * invM; cd = -cb * b.mean[c] - ca * sg * invM;
}

// the dense pass: dx = B[c]*y + D[c] over the whole map of every branch (grid.y = branch)
template <typename T>
__global__ __launch_bounds__(256) void head_sparse_apply_kernel(HsGeom g, long chunks, float invM) {
    constexpr int E = ElemTraits<T>::ELEMS;
    __shared__ float tab[2][HS_C];
    const HsBranch& b = g.br[blockIdx.y];
    { float ca, cb, cd; hs_coef(b, threadIdx.x, invM, ca, cb, cd); tab[0][threadIdx.x] = cb; tab[1][threadIdx.x] = cd; }
    __syncthreads();
    const T* x = reinterpret_cast<const T*>(b.y);
    T* dx = reinterpret_cast<T*>(b.dx);
    constexpr int CPR = HS_C / E;
    const long stride = (long)gridDim.x * blockDim.x;
    auto finish = [&](long i, const u32x4& cx) {
        const int c0 = (int)(i % CPR) * E;
        float v[E];
        ElemTraits<T>::unpack(cx, v);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = tab[0][c0 + e] * v[e] + tab[1][c0 + e];
        *reinterpret_cast<u32x4*>(dx + i * E) = ElemTraits<T>::pack(v);
    };
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < chunks; i += 4 * stride) {
        u32x4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const u32x4*>(x + (i + u * stride) * E);
#pragma unroll
        for (int u = 0; u < 4; ++u) finish(i + u * stride, q[u]);
    }
    for (; i < chunks; i += stride) finish(i, *reinterpret_cast<const u32x4*>(x + i * E));
}

// object pixels: dx = A * (sum of the gradient rows that share the pixel) + B*y + D, rewritten whole (no atomics; rows that share
// a pixel write the same value)
template <typename T>
__global__ __launch_bounds__(256) void head_sparse_fix_kernel(HsGeom g, const float* __restrict__ rows, const float* __restrict__ grows, float invM) {
    const int n = blockIdx.x, c = threadIdx.x;
    const HsBranch& b = g.br[blockIdx.y];
    const float* t = rows + (size_t)n * HS_ROW;
    if (t[HS_VALID] == 0.f) return;
    const long p = hs_pixel(t, g.B, g.H, g.W);
    const float* gb = grows + (size_t)blockIdx.y * g.N * HS_C;
    float gs = 0.f;
    for (int m = 0; m < g.N; ++m) {
        const float* tm = rows + (size_t)m * HS_ROW;
        if (tm[HS_VALID] != 0.f && hs_pixel(tm, g.B, g.H, g.W) == p) gs += gb[(size_t)m * HS_C + c];
    }
    float ca, cb, cd;
    hs_coef(b, c, invM, ca, cb, cd);
    const float x = ElemTraits<T>::load(reinterpret_cast<const T*>(b.y) + p * HS_C + c);
    ElemTraits<T>::store(reinterpret_cast<T*>(b.dx) + p * HS_C + c, ca * gs + cb * x + cd);
}

static int hs_geom(const mfx_head_sparse_desc* d, HsGeom& g, bool backward) {
    if (!d || !d->rows) return mfx_fail(MFX_ERR_ARG, "head_sparse: null pointer");
    if (d->nbranch < 1 || d->nbranch > MFX_HEAD_MAX_BRANCH || d->C != HS_C) return mfx_fail(MFX_ERR_ARG, "head_sparse: 1..8 branches of 256 trunk channels");
    if (d->N < 0 || d->B < 1 || d->H < 1 || d->W < 1) return mfx_fail(MFX_ERR_ARG, "head_sparse: bad sizes");
    g.nbranch = d->nbranch; g.N = d->N; g.B = d->B; g.H = d->H; g.W = d->W; g.ld_out = d->ld_out;
    for (int i = 0; i < d->nbranch; ++i) {
        if (!d->y[i] || !d->mean[i] || !d->rstd[i] || !d->gamma[i] || !d->beta[i] || !d->w2[i]) return mfx_fail(MFX_ERR_ARG, "head_sparse: null branch pointer");
        if (d->k[i] < 1 || d->k[i] > 32 || d->out_off[i] < 0 || d->out_off[i] + d->k[i] > d->ld_out) return mfx_fail(MFX_ERR_ARG, "head_sparse: branch outputs out of range");
        if (backward && (!d->sums[i] || !d->dw2[i] || !d->dx[i])) return mfx_fail(MFX_ERR_ARG, "head_sparse_bwd: null output pointer");
        HsBranch& b = g.br[i];
        b.y = d->y[i]; b.mean = d->mean[i]; b.rstd = d->rstd[i]; b.gamma = d->gamma[i]; b.beta = d->beta[i]; b.w2 = d->w2[i]; b.b2 = d->b2[i];
        b.sums = d->sums[i]; b.dw2 = d->dw2[i]; b.db2 = d->db2[i]; b.dx = d->dx[i]; b.k = d->k[i]; b.off = d->out_off[i];
    }
    return MFX_OK;
}

}  // namespace mfx
using namespace mfx;

extern "C" int mfx_head_sparse_fwd(const mfx_head_sparse_desc* d, void* stream) {
    HsGeom g;
    int rc = hs_geom(d, g, false); if (rc) return rc;
    if (!d->out) return mfx_fail(MFX_ERR_ARG, "head_sparse_fwd: null output");
    if (d->N == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MFX_F32) hipLaunchKernelGGL(head_sparse_fwd_kernel<float>, dim3(d->N, d->nbranch), dim3(256), 0, st, g, d->rows, d->out);
    else if (d->dtype == MFX_BF16) hipLaunchKernelGGL(head_sparse_fwd_kernel<bf16_t>, dim3(d->N, d->nbranch), dim3(256), 0, st, g, d->rows, d->out);
    else if (d->dtype == MFX_F16) hipLaunchKernelGGL(head_sparse_fwd_kernel<half_t>, dim3(d->N, d->nbranch), dim3(256), 0, st, g, d->rows, d->out);
    else return mfx_fail(MFX_ERR_ARG, "head_sparse_fwd: bad dtype");
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_head_sparse_bwd(const mfx_head_sparse_desc* d, void* stream) {
    HsGeom g;
    int rc = hs_geom(d, g, true); if (rc) return rc;
    if (!d->dout || !d->g || !d->arena || d->arena_bytes == 0) return mfx_fail(MFX_ERR_ARG, "head_sparse_bwd: null pointer");
    if (d->dtype != MFX_F32 && d->dtype != MFX_BF16 && d->dtype != MFX_F16) return mfx_fail(MFX_ERR_ARG, "head_sparse_bwd: bad dtype");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MFX_HIP_CHECK(mfx::zero_async(d->arena, d->arena_bytes, st));        // sums, dW2, db2 of every branch: carved from one arena by the caller
    const long M = (long)d->B * d->H * d->W;
    const float invM = 1.f / (float)M;
    const int E = d->dtype == MFX_F32 ? 4 : 8;
    const long chunks = M * (HS_C / E);
    const int rpb = g_opt_det ? std::max(1, d->N) : 16, nchunk = (d->N + rpb - 1) / rpb;     // deterministic: one row chunk per branch (single writer)
    const unsigned ablocks = (unsigned)std::min<long>((chunks + 255) / 256, 1024);
#define HS_LAUNCH(T)                                                                                                                              \
    do {                                                                                                                                          \
        if (d->N > 0) hipLaunchKernelGGL(head_sparse_bwd_rows_kernel<T>, dim3(nchunk, d->nbranch), dim3(256), 0, st, g, d->rows, d->dout, d->g, rpb); \
        hipLaunchKernelGGL(head_sparse_apply_kernel<T>, dim3(ablocks, d->nbranch), dim3(256), 0, st, g, chunks, invM);                            \
        if (d->N > 0) hipLaunchKernelGGL(head_sparse_fix_kernel<T>, dim3(d->N, d->nbranch), dim3(256), 0, st, g, d->rows, (const float*)d->g, invM); \
    } while (0)
    if (d->dtype == MFX_F32) HS_LAUNCH(float); else if (d->dtype == MFX_BF16) HS_LAUNCH(bf16_t); else HS_LAUNCH(half_t);
#undef HS_LAUNCH
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}
