// AdamW over ALL parameter tensors of the model in one launch (r06).  The reference's optimizer is torch.optim.AdamW with one group per parameter
// (solver/__init__.py:10-60: 280 groups); this build's host side merges them into two groups (weights / biases, monoflex_amd/solver.py) and ran torch's
// fused multi-tensor kernel: 25 launches and 345 us of a 18 ms training step for 560 MB of traffic (20 M parameters x (read p, g, m, v + write p, m, v)).
// Here every workgroup owns one 4096-element chunk of one tensor (a pointer table on the device says which: the chunk -> tensor search is one ballot per
// 256 table entries, as in pack_conv_weight_batched_kernel), streams it with 16-byte accesses and applies
//     p -= lr wd p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (decoupled weight decay, bias correction in double: the arithmetic of torch's `_fused_adamw_`, capturable form: t and lr are read from the device).
// The per-tensor step counters (torch keeps one per parameter) are advanced by a one-workgroup launch in front, so every chunk reads a settled value.
// `found_inf` (the fp16 loss scaler's flag): when set, nothing moves -- parameters, moments and counters stay as they are.
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "common.h"

namespace mfx {

constexpr int AW_CHUNK = 4096;

__global__ __launch_bounds__(256) void adamw_bump_steps_kernel(const mfx_adamw_desc* __restrict__ descs, int n, const float* __restrict__ found_inf) {
    if (found_inf && *found_inf != 0.f) return;
    for (int i = threadIdx.x; i < n; i += blockDim.x) *descs[i].step += 1.f;
}

__global__ __launch_bounds__(256) void adamw_multi_kernel(const mfx_adamw_desc* __restrict__ descs, const long long* __restrict__ prefix, int n,
                                                         const mfx_adamw_group* __restrict__ groups, const float* __restrict__ found_inf) {
    if (found_inf && *found_inf != 0.f) return;
    __shared__ int below[4];
    const long chunk = blockIdx.x;
    int cnt = 0;
    for (int base = 0; base < n; base += 256) {
        const int idx = base + (int)threadIdx.x;
        cnt += __popcll(__ballot(idx < n && (long)prefix[idx] <= chunk));
    }
    if ((threadIdx.x & 63) == 0) below[threadIdx.x >> 6] = cnt;
    __syncthreads();
    const int ti = below[0] + below[1] + below[2] + below[3] - 1;
    const mfx_adamw_desc d = descs[ti];
    const mfx_adamw_group gr = groups[d.group];
    const long j0 = (chunk - (long)prefix[ti]) * AW_CHUNK;
    const long j1 = j0 + AW_CHUNK < d.numel ? j0 + AW_CHUNK : d.numel;
    // scalars of the step (uniform per tensor)
    const double lr = (double)*gr.lr, b1 = (double)gr.beta1, b2 = (double)gr.beta2;
    const double t = (double)*d.step;                                         // already advanced (adamw_bump_steps_kernel)
    const float decay = (float)(lr * (double)gr.weight_decay);
    const float step_size = (float)(lr / (1.0 - pow(b1, t)));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(b2, t));
    const float w1 = (float)(1.0 - b1), fb1 = gr.beta1, fb2 = gr.beta2, w2 = (float)(1.0 - b2), eps = gr.eps;
    float* p = reinterpret_cast<float*>(d.p);
    const float* g = reinterpret_cast<const float*>(d.g);
    float* m = reinterpret_cast<float*>(d.m);
    float* v = reinterpret_cast<float*>(d.v);
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        pp -= decay * pp;
        mm = fb1 * mm + w1 * gg;
        vv = fb2 * vv + w2 * gg * gg;
        pp -= step_size * mm / (sqrtf(vv) / bc2_sqrt + eps);
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    long j = j0 + (long)threadIdx.x * 4;
    if (vec) {
        for (; j + 4 <= j1; j += 1024) {
            f32x4 pv = *reinterpret_cast<const f32x4*>(p + j), mv = *reinterpret_cast<const f32x4*>(m + j), vv = *reinterpret_cast<const f32x4*>(v + j);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pv[e], me = mv[e], ve = vv[e];
                upd(pe, gv[e], me, ve);
                pv[e] = pe; mv[e] = me; vv[e] = ve;
            }
            *reinterpret_cast<f32x4*>(p + j) = pv; *reinterpret_cast<f32x4*>(m + j) = mv; *reinterpret_cast<f32x4*>(v + j) = vv;
        }
    }
    // the chunk's ragged end (and every element of a tensor whose storage is not 16-byte aligned)
    for (; j < j1; j += 1024)
        for (long e = j; e < j + 4 && e < j1; ++e) upd(p[e], g[e], m[e], v[e]);
}

}  // namespace mfx
using namespace mfx;

extern "C" int mfx_adamw_chunk_elems(void) { return AW_CHUNK; }

extern "C" int mfx_adamw_multi(const mfx_adamw_desc* descs_dev, const long long* prefix_dev, int n, long long total_chunks,
                               const mfx_adamw_group* groups_dev, const float* found_inf, void* stream) {
    if (n <= 0 || total_chunks <= 0) return MFX_OK;
    if (!descs_dev || !prefix_dev || !groups_dev) return mfx_fail(MFX_ERR_ARG, "adamw_multi: null pointer");
    if (total_chunks >= (1LL << 31)) return mfx_fail(MFX_ERR_ARG, "adamw_multi: too many chunks");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adamw_bump_steps_kernel, dim3(1), dim3(256), 0, st, descs_dev, n, found_inf);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, st, descs_dev, prefix_dev, n, groups_dev, found_inf);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}
