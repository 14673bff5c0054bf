// Fused loss kernels of the training step (reference model/head/detector_loss.py:267-300, model/layers/focal_loss.py:29-55).
//
// Heat-map term (penalty-reduced focal loss of CenterNet) over the (B, classes, H, W) class map: the reference runs
//   sigmoid -> clamp(1e-4, 1-1e-4) -> eq/lt/ge masks -> pow -> log -> mul ... -> two sums -> backward of all of it,
// ~35 elementwise passes over 737 k elements per step.  Here ONE pass reads the NHWC logits and the NCHW target map,
// accumulates [loss_sum, num_pos] and writes d(loss_sum)/d(logit) for the backward pass (which is then a single scale).
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"
#include "fill.h"

namespace mfx {

__global__ __launch_bounds__(256) void focal_loss_kernel(const float* __restrict__ logits, const float* __restrict__ heat,
                                                        int B, int HW, int ncls, float alpha, float beta,
                                                        float* __restrict__ sums, float* __restrict__ dz) {
    const long total = (long)B * HW * ncls;
    float ls = 0.f, np = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i indexes the NHWC logits: (b, p, c); the target is NCHW: (b, c, p)
        const int c = (int)(i % ncls);
        const long bp = i / ncls;
        const int p = (int)(bp % HW), b = (int)(bp / HW);
        const float z = logits[i], t = heat[((size_t)b * ncls + c) * HW + p];
        const float s = 1.f / (1.f + expf(-z));
        const bool clamped = s < 1e-4f || s > 1.f - 1e-4f;           // sigmoid_hm: clamp(min=1e-4, max=1-1e-4) (layers/utils.py:39-42)
        const float pr = fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
        const float dpdz = clamped ? 0.f : pr * (1.f - pr);
        float l = 0.f, dldp = 0.f;
        if (t == 1.f) {                                              // positive: log(p) * (1-p)^alpha
            const float q = 1.f - pr, qa = powf(q, alpha);
            l = logf(pr) * qa;
            dldp = qa / pr - alpha * powf(q, alpha - 1.f) * logf(pr);
            np += 1.f;
        } else if (t < 1.f && t >= 0.f) {                            // negative: log(1-p) * p^alpha * (1-t)^beta
            const float nw = powf(1.f - t, beta), pa = powf(pr, alpha), q = 1.f - pr;
            l = logf(q) * pa * nw;
            dldp = nw * (alpha * powf(pr, alpha - 1.f) * logf(q) - pa / q);
        }
        ls -= l;                                                     // loss_sum = -sum(pos) - sum(neg)
        dz[i] = -dldp * dpdz;
    }
    // block reduction -> two global atomics per block
    __shared__ float red[2][4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ls += __shfl_xor(ls, off); np += __shfl_xor(np, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ls; red[1][threadIdx.x >> 6] = np; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(sums, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        unsafeAtomicAdd(sums + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

}  // namespace mfx
using namespace mfx;

extern "C" int mfx_focal_loss(const float* logits_nhwc, const float* heat_nchw, int B, int H, int W, int ncls, float alpha, float beta,
                              float* sums2, float* dlogits_nhwc, void* stream) {
    if (!logits_nhwc || !heat_nchw || !sums2 || !dlogits_nhwc) return mfx_fail(MFX_ERR_ARG, "focal_loss: null pointer");
    if (ncls < 1 || B < 0 || H < 0 || W < 0) return mfx_fail(MFX_ERR_ARG, "focal_loss: bad sizes");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MFX_HIP_CHECK(mfx::zero_async(sums2, 8, st));
    const long total = (long)B * H * W * ncls;
    if (total == 0) return MFX_OK;
    unsigned blocks = (unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    if (g_opt_det) blocks = 1;                                  // one workgroup: the two sums have a single writer
    hipLaunchKernelGGL(focal_loss_kernel, dim3(blocks), dim3(256), 0, st, logits_nhwc, heat_nchw, B, H * W, ncls, alpha, beta, sums2, dlogits_nhwc);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// ------------------------------------------------------------------------------------------------
// Per-object regression terms: one wavefront per object row (object_loss_math.h)
// ------------------------------------------------------------------------------------------------
#include "object_loss_math.h"

namespace mfx {
using namespace oloss;

struct PixelReader {                                        // channel ch of the object's pixel; lane `seed` differentiates w.r.t. its own channel
    const float* p; int seed;
    __device__ Dual operator()(int ch) const { return Dual{p[ch], ch == seed ? 1.f : 0.f}; }
};

// dense map: the row's (image, centre) pixel.  B == 0: `base` is already the gathered table, one row of `ld` floats per object row n
__device__ __forceinline__ const float* object_pixel(const float* base, const float* t, int n, int B, int H, int W, int ld, int ch_off) {
    if (B == 0) return base + (size_t)n * ld + ch_off;
    const int b = min(max((int)t[R_B], 0), B - 1), cx = min(max((int)t[R_CX], 0), W - 1), cy = min(max((int)t[R_CY], 0), H - 1);
    return base + ((size_t)(b * H + cy) * W + cx) * ld + ch_off;
}

__global__ __launch_bounds__(64) void object_loss_kernel(const float* __restrict__ reg, int B, int H, int W, int ld, int ch_off,
                                                         const float* __restrict__ rows, int N, mfx_object_loss_cfg c,
                                                         float* __restrict__ vals, float* __restrict__ G) {
    const int lane = threadIdx.x;
    float cn[NNORM];
#pragma unroll
    for (int i = 0; i < NNORM; ++i) cn[i] = 0.f;
    for (int r = lane; r < N; r += 64) {                    // the batch-wide selection counts: every wave recounts them (N <= a few hundred rows)
        float q[NNORM];
        row_counts(rows + (size_t)r * ROW, q);
#pragma unroll
        for (int i = 0; i < NNORM; ++i) cn[i] += q[i];
    }
#pragma unroll
    for (int i = 0; i < NNORM; ++i)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cn[i] += __shfl_xor(cn[i], off);
    // (deterministic mode launches ONE wave that walks the objects in order: its adds into `vals` are then sequential)
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        const float* t = rows + (size_t)n * ROW;
        Dual out[NVAL];
        const PixelReader X{object_pixel(reg, t, n, B, H, W, ld, ch_off), lane};
        object_terms(X, t, c, cn, out);
#pragma unroll
        for (int k = 0; k < NTERM; ++k) G[((size_t)n * NTERM + k) * 64 + lane] = out[k].d;
        if (lane == 0 && t[R_VALID] != 0.f)
            for (int k = 0; k < NVAL; ++k) unsafeAtomicAdd(vals + k, out[k].v);
    }
}

__global__ __launch_bounds__(64) void object_loss_bwd_kernel(const float* __restrict__ G, const float* __restrict__ gout,
                                                             const float* __restrict__ rows, int N, int B, int H, int W,
                                                             float* __restrict__ dreg, int ld, int ch_off) {
    const int lane = threadIdx.x;
    for (int n = blockIdx.x; n < N; n += gridDim.x) {       // (deterministic mode: one wave, objects that share a pixel add in order)
        const float* t = rows + (size_t)n * ROW;
        if (t[R_VALID] == 0.f || lane >= 50) continue;
        float g = 0.f;
#pragma unroll
        for (int k = 0; k < NTERM; ++k) g += gout[k] * G[((size_t)n * NTERM + k) * 64 + lane];
        float* p = const_cast<float*>(object_pixel(dreg, t, n, B, H, W, ld, ch_off));
        unsafeAtomicAdd(p + lane, g);
    }
}

}  // namespace mfx

extern "C" int mfx_object_loss(const float* reg_nhwc, int B, int H, int W, int ld, int ch_off, const float* rows, int N,
                               const mfx_object_loss_cfg* cfg, float* vals, float* G, void* stream) {
    if (!reg_nhwc || !rows || !cfg || !vals || !G) return mfx_fail(MFX_ERR_ARG, "object_loss: null pointer");
    if (B < 0 || (B > 0 && (H < 1 || W < 1)) || N < 0 || ch_off < 0 || ch_off + 50 > ld) return mfx_fail(MFX_ERR_ARG, "object_loss: bad sizes");
    for (int i = 0; i < 9; ++i)
        if (cfg->ch[i] < 0 || cfg->ch[i] >= 50) return mfx_fail(MFX_ERR_ARG, "object_loss: channel offset outside the 50 regression channels");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MFX_HIP_CHECK(mfx::zero_async(vals, sizeof(float) * MFX_OBJ_VALUES, st));
    if (N == 0) return MFX_OK;
    hipLaunchKernelGGL(object_loss_kernel, dim3(g_opt_det ? 1 : N), dim3(64), 0, st, reg_nhwc, B, H, W, ld, ch_off, rows, N, *cfg, vals, G);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_object_loss_backward(const float* G, const float* gout_terms, const float* rows, int N, int B, int H, int W,
                                        float* dreg_nhwc, int ld, int ch_off, void* stream) {
    if (!G || !gout_terms || !rows || !dreg_nhwc) return mfx_fail(MFX_ERR_ARG, "object_loss_backward: null pointer");
    if (B < 0 || (B > 0 && (H < 1 || W < 1)) || N < 0 || ch_off < 0 || ch_off + 50 > ld) return mfx_fail(MFX_ERR_ARG, "object_loss_backward: bad sizes");
    if (N == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(object_loss_bwd_kernel, dim3(g_opt_det ? 1 : N), dim3(64), 0, st, G, gout_terms, rows, N, B, H, W, dreg_nhwc, ld, ch_off);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}
