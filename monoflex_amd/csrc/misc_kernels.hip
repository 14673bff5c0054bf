// Memory-bound helper kernels (NHWC): 2x2 max-pool, depthwise transposed-conv upsample + skip add,
// NCHW<->NHWC layout transforms, stem input packing, edge-fusion scatter-add.
// All of them move 16-byte chunks per lane along the channel axis (coalesced per wavefront).
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"

namespace mfx {

static inline int cdiv_i(long a, long b) { return (int)((a + b - 1) / b); }

// ---- 2x2 stride-2 max pool -------------------------------------------------------------------
template <typename T>
__global__ void maxpool2x2_kernel(const T* x, T* y, int B, int H, int W, int C) {
    constexpr int E = ElemTraits<T>::ELEMS;
    const int Ho = H / 2, Wo = W / 2, CG = C / E;
    const long total = (long)B * Ho * Wo * CG;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        long p = i / CG;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const T* src = x + ((size_t)(b * H + oh * 2) * W + ow * 2) * C + cg * E;
        float a[E], t[E];
        ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(src), a);
        ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(src + C), t);
#pragma unroll
        for (int e = 0; e < E; ++e) a[e] = fmaxf(a[e], t[e]);
        ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(src + (size_t)W * C), t);
#pragma unroll
        for (int e = 0; e < E; ++e) a[e] = fmaxf(a[e], t[e]);
        ElemTraits<T>::unpack(*reinterpret_cast<const u32x4*>(src + (size_t)W * C + C), t);
#pragma unroll
        for (int e = 0; e < E; ++e) a[e] = fmaxf(a[e], t[e]);
        *reinterpret_cast<u32x4*>(y + ((size_t)(b * Ho + oh) * Wo + ow) * C + cg * E) = ElemTraits<T>::pack(a);
    }
}

// ---- depthwise ConvTranspose2d(k=2f, s=f, p=f/2) + skip ------------------------------------------
// out pixel oh gets contributions from kh in {t%f, t%f+f} with t = oh+p, ih = (t-kh)/f.
// F > 0: the stride as a compile-time constant (2 and 4 are the only ones the network has: every % and / below folds into shifts);
// the per-channel tap weights are fetched as 16-byte vectors (eight scalar loads per tap and thread had made the kernel
// instruction-bound: 19 us for 70 MB at 64 channels, 96x320 out)
template <typename T, int F = 0>
__global__ void upsample_add_kernel(const T* __restrict__ x, const float* __restrict__ w, const T* __restrict__ skip, T* __restrict__ y,
                                    int B, int H, int W, int C, int f_rt) {
    constexpr int E = ElemTraits<T>::ELEMS;
    const int f = F > 0 ? F : f_rt;
    const int Ho = H * f, Wo = W * f, CG = C / E, p_ = f / 2, k = 2 * f;
    const long total = (long)B * Ho * Wo * CG;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        long p = i / CG;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        const int th = oh + p_, tw = ow + p_;
        // r06: branch-free.  The four taps' input chunks, their eight weight vectors and the skip chunk are thirteen independent loads issued together
        // (clamped, always valid addresses; a tap outside the map contributes an exact zero: its input chunk is replaced by zeros) -- as `if (outside)
        // continue` around each tap every load sat behind its own branch and was waited for alone (r05 PMC: these launches issued 7-17 % of their cycles).
        // Same sums in the same (ascending input index) order: fp32 maps come out bit-identical; 16-bit maps differ from the branchy form in ~1e-5 of their
        // elements by one ulp (the compiler contracts multiply + add differently in straight-line code).  8 launches of the step: 132 -> 123 us (tools/upsample_bench.py).
        const size_t o = ((size_t)(b * Ho + oh) * Wo + ow) * C + cg * E;
        u32x4 sv = {0u, 0u, 0u, 0u};
        if (skip) sv = *reinterpret_cast<const u32x4*>(skip + o);
        u32x4 xv[2][2]; f32x4 wv[2][2][E / 4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int kh = th % f + a * f, dh = th - kh, ih = dh / f;
            const bool vh = dh >= 0 && ih < H;
            const int ihc = min(max(ih, 0), H - 1);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int kw = tw % f + c2 * f, dw = tw - kw, iw = dw / f;
                const bool vw = dw >= 0 && iw < W;
                const int iwc = min(max(iw, 0), W - 1);
                const u32x4 v = *reinterpret_cast<const u32x4*>(x + ((size_t)(b * H + ihc) * W + iwc) * C + cg * E);
                xv[a][c2] = (vh && vw) ? v : u32x4{0u, 0u, 0u, 0u};
                const float* wp = w + (size_t)(kh * k + kw) * C + cg * E;
#pragma unroll
                for (int e = 0; e < E; e += 4) wv[a][c2][e / 4] = *reinterpret_cast<const f32x4*>(wp + e);
            }
        }
        // ascending (ih, iw) order = ascending input index, the order a direct scatter would add in
#pragma unroll
        for (int a = 1; a >= 0; --a)
#pragma unroll
            for (int c2 = 1; c2 >= 0; --c2) {
                float v[E];
                ElemTraits<T>::unpack(xv[a][c2], v);
#pragma unroll
                for (int e = 0; e < E; e += 4) {
                    const f32x4 w4 = wv[a][c2][e / 4];
                    acc[e] += v[e] * w4[0]; acc[e + 1] += v[e + 1] * w4[1]; acc[e + 2] += v[e + 2] * w4[2]; acc[e + 3] += v[e + 3] * w4[3];
                }
            }
        if (skip) {
            float s[E];
            ElemTraits<T>::unpack(sv, s);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += s[e];
        }
        *reinterpret_cast<u32x4*>(y + o) = ElemTraits<T>::pack(acc);
    }
}

// ---- NCHW fp32 <-> NHWC T, 32x32 tile transpose through LDS --------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, T* y, int C, int HW, int ldy) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int c = c0 + ty + j, p = p0 + tx;
        tile[ty + j][tx] = (c < C && p < HW) ? x[((size_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int p = p0 + ty + j, c = c0 + tx;
        if (p < HW && c < ldy) ElemTraits<T>::store(y + ((size_t)b * HW + p) * ldy + c, c < C ? tile[tx][ty + j] : 0.f);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* x, float* y, int C, int HW, int ldx) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int p = p0 + ty + j, c = c0 + tx;
        tile[ty + j][tx] = (p < HW && c < C) ? ElemTraits<T>::load(x + ((size_t)b * HW + p) * ldx + c) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int c = c0 + ty + j, p = p0 + tx;
        if (c < C && p < HW) y[((size_t)b * C + c) * HW + p] = tile[tx][ty + j];
    }
}

// ---- stem input: NCHW fp32 (B,3,H,W) -> zero-padded NHWC4 ----------------------------------------
template <typename T>
__global__ void pack_image_kernel(const float* x, T* y, int B, int H, int W, int ph, int pwl, int pwr) {
    const int Hp = H + 2 * ph, Wp = W + pwl + pwr;
    const long total = (long)B * Hp * Wp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int wp = (int)(i % Wp);
        long p = i / Wp;
        const int hp = (int)(p % Hp);
        const int b = (int)(p / Hp);
        const int h = hp - ph, w = wp - pwl;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (h >= 0 && h < H && w >= 0 && w < W) {
            const size_t o = ((size_t)b * 3 * H + h) * W + w;
            v[0] = x[o]; v[1] = x[o + (size_t)H * W]; v[2] = x[o + 2 * (size_t)H * W];
        }
        T* d = y + (size_t)i * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) ElemTraits<T>::store(d + e, v[e]);
    }
}

// ---- edge fusion tail: add the fused edge outputs back at the border pixels ----------------------
__global__ void edge_scatter_add_kernel(float* out, int ld_out, int ch_off, int C, const float* v, int ldv,
                                        const int* edge_xy, const int* edge_len, int B, int L, int H, int W, float* planar) {
    const int total = B * L * C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i % C, j = (i / C) % L, b = i / (C * L);
        if (j >= edge_len[b]) continue;
        const int x = edge_xy[(b * L + j) * 2], y = edge_xy[(b * L + j) * 2 + 1];
        // indices of the first edge_len points are unique -> plain read-modify-write
        const float add = v[(size_t)(b * L + j) * ldv + c];
        out[((size_t)(b * H + y) * W + x) * ld_out + ch_off + c] += add;
        if (planar) planar[((size_t)b * C + c) * H * W + y * W + x] += add;
    }
}

}  // namespace mfx
using namespace mfx;

#define MFX_GRID(total, threads) dim3((unsigned)(cdiv_i((total), (threads)) < 16384 ? cdiv_i((total), (threads)) : 16384))

extern "C" int mfx_maxpool2x2_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
    if (!x || !y) return mfx_fail(MFX_ERR_ARG, "maxpool: null pointer");
    const int E = dtype == MFX_F32 ? 4 : 8;
    if (C % E != 0 || (H & 1) || (W & 1)) return mfx_fail(MFX_ERR_ARG, "maxpool: C must be a multiple of 16 bytes, H/W even");
    const long total = (long)B * (H / 2) * (W / 2) * (C / E);
    if (total == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MFX_F32) hipLaunchKernelGGL(maxpool2x2_kernel<float>, MFX_GRID(total, 256), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C);
    else if (dtype == MFX_F16) hipLaunchKernelGGL(maxpool2x2_kernel<half_t>, MFX_GRID(total, 256), dim3(256), 0, st, (const half_t*)x, (half_t*)y, B, H, W, C);
    else hipLaunchKernelGGL(maxpool2x2_kernel<bf16_t>, MFX_GRID(total, 256), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_upsample_add_nhwc(const void* x, const float* w, const void* skip, void* y,
                                     int B, int H, int W, int C, int f, int dtype, void* stream) {
    if (!x || !w || !y) return mfx_fail(MFX_ERR_ARG, "upsample: null pointer");
    const int E = dtype == MFX_F32 ? 4 : 8;
    if (C % E != 0 || f < 1) return mfx_fail(MFX_ERR_ARG, "upsample: bad C or f");
    const long total = (long)B * H * f * W * f * (C / E);
    if (total == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define MFX_UP_LAUNCH(T, F_)                                                                                                                   \
    hipLaunchKernelGGL((upsample_add_kernel<T, F_>), MFX_GRID(total, 256), dim3(256), 0, st, (const T*)x, w, (const T*)skip, (T*)y, B, H, W, C, f)
#define MFX_UP_DISPATCH(T)                                                                                                                     \
    do { if (f == 2) MFX_UP_LAUNCH(T, 2); else if (f == 4) MFX_UP_LAUNCH(T, 4); else MFX_UP_LAUNCH(T, 0); } while (0)
    if (dtype == MFX_F32) MFX_UP_DISPATCH(float);
    else if (dtype == MFX_F16) MFX_UP_DISPATCH(half_t);
    else MFX_UP_DISPATCH(bf16_t);
#undef MFX_UP_DISPATCH
#undef MFX_UP_LAUNCH
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_nchw_to_nhwc(const float* x, void* y, int B, int C, int H, int W, int ldy, int dtype, void* stream) {
    if (!x || !y || ldy < C) return mfx_fail(MFX_ERR_ARG, "nchw_to_nhwc: bad arguments");
    if (B * C * H * W == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(cdiv_i((long)H * W, 32), cdiv_i(ldy, 32), B), block(32, 8);
    if (dtype == MFX_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, block, 0, st, x, (float*)y, C, H * W, ldy);
    else if (dtype == MFX_F16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<half_t>, grid, block, 0, st, x, (half_t*)y, C, H * W, ldy);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, block, 0, st, x, (bf16_t*)y, C, H * W, ldy);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_nhwc_to_nchw(const void* x, float* y, int B, int C, int H, int W, int ldx, int dtype, void* stream) {
    if (!x || !y || ldx < C) return mfx_fail(MFX_ERR_ARG, "nhwc_to_nchw: bad arguments");
    if (B * C * H * W == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(cdiv_i((long)H * W, 32), cdiv_i(C, 32), B), block(32, 8);
    if (dtype == MFX_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, block, 0, st, (const float*)x, y, C, H * W, ldx);
    else if (dtype == MFX_F16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<half_t>, grid, block, 0, st, (const half_t*)x, y, C, H * W, ldx);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, y, C, H * W, ldx);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_pack_image_nhwc4(const float* x, void* y, int B, int H, int W, int pad_h, int pad_w_left,
                                    int pad_w_right, int dtype, void* stream) {
    if (!x || !y) return mfx_fail(MFX_ERR_ARG, "pack_image: null pointer");
    const long total = (long)B * (H + 2 * pad_h) * (W + pad_w_left + pad_w_right);
    if (total == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MFX_F32) hipLaunchKernelGGL(pack_image_kernel<float>, MFX_GRID(total, 256), dim3(256), 0, st, x, (float*)y, B, H, W, pad_h, pad_w_left, pad_w_right);
    else if (dtype == MFX_F16) hipLaunchKernelGGL(pack_image_kernel<half_t>, MFX_GRID(total, 256), dim3(256), 0, st, x, (half_t*)y, B, H, W, pad_h, pad_w_left, pad_w_right);
    else hipLaunchKernelGGL(pack_image_kernel<bf16_t>, MFX_GRID(total, 256), dim3(256), 0, st, x, (bf16_t*)y, B, H, W, pad_h, pad_w_left, pad_w_right);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_edge_scatter_add(float* out, int ld_out, int ch_off, int C, const float* v, int ldv,
                                    const int32_t* edge_xy, const int32_t* edge_len, int B, int L, int H, int W,
                                    float* planar, void* stream) {
    if (!out || !v || !edge_xy || !edge_len) return mfx_fail(MFX_ERR_ARG, "edge_scatter_add: null pointer");
    const int total = B * L * C;
    if (total == 0) return MFX_OK;
    hipLaunchKernelGGL(edge_scatter_add_kernel, MFX_GRID(total, 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       out, ld_out, ch_off, C, v, ldv, edge_xy, edge_len, B, L, H, W, planar);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}
