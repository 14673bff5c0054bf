// Backward of the modulated deformable convolution at the reference `_ext.dcn_v2_backward` boundary
// (/root/reference/model/backbone/DCNv2/src/dcn_v2.h:48-92; math of src/cuda/dcn_v2_cuda.cu:206-335 and
// src/cuda/dcn_v2_im2col_cuda.cu:197-327), restructured for NHWC / gfx950:
//   * no per-image host loop: the whole batch is one set of launches;
//   * d(columns) = grad_out x W is one MFMA GEMM (the shared implicit-GEMM kernel as a 1x1 conv);
//   * grad_offset / grad_mask: one wavefront per (pixel, tap), lanes over channels, wave reduction
//     (the reference loops over channels inside one thread);
//   * grad_input: the same wavefront scatters to the four corners; lanes hit consecutive channels,
//     so the float atomics are coalesced 256-byte bursts;
//   * grad_weight: re-samples the columns tile into LDS (never to HBM) and contracts it with grad_out
//     over a slab of pixels per workgroup (split over pixels, atomics on the small weight gradient).
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"
#include "fill.h"
#include <type_traits>

namespace mfx {

static inline int next_pow2_(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int cdv(long a, long b) { return (int)((a + b - 1) / b); }

template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    return f32x4{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)};
}

struct BwdGeom { int B, H, W, Cp, lgC, Ho, Wo, kh, kw, kk, stride, pad, dil, M, K, Kp, Coutp, stride_w, pad_w, dil_w; };      // stride / pad / dil: rows; *_w: columns

struct TapGeo { int off[4]; float w[4]; float lh, lw, hh, hw; bool inside; bool cv[4]; };

// geometry of one (pixel, tap) sample -- same rules as the forward sampler (dcn_v2_im2col_cuda.cu:25-54,178-189)
__device__ __forceinline__ TapGeo tap_geometry(const BwdGeom& g, const float* om_row, int b, int oh, int ow, int tap) {
    TapGeo t;
    const int th = tap / g.kw, tw = tap - th * g.kw;
    const float h = (float)(oh * g.stride - g.pad + th * g.dil) + om_row[2 * tap];
    const float w = (float)(ow * g.stride_w - g.pad_w + tw * g.dil_w) + om_row[2 * tap + 1];
    t.inside = h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
    const float hf = floorf(h), wf = floorf(w);
    const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
    t.lh = h - hf; t.lw = w - wf; t.hh = 1.f - t.lh; t.hw = 1.f - t.lw;
    const bool t0 = t.inside && h0 >= 0, t1 = t.inside && h1 <= g.H - 1, l0 = w0 >= 0, l1 = w1 <= g.W - 1;
    const int ch0 = min(max(h0, 0), g.H - 1), ch1 = min(max(h1, 0), g.H - 1);
    const int cw0 = min(max(w0, 0), g.W - 1), cw1 = min(max(w1, 0), g.W - 1);
    const int pix0 = b * g.H * g.W;
    t.off[0] = pix0 + ch0 * g.W + cw0; t.cv[0] = t0 && l0; t.w[0] = t.hh * t.hw;
    t.off[1] = pix0 + ch0 * g.W + cw1; t.cv[1] = t0 && l1; t.w[1] = t.hh * t.lw;
    t.off[2] = pix0 + ch1 * g.W + cw0; t.cv[2] = t1 && l0; t.w[2] = t.lh * t.hw;
    t.off[3] = pix0 + ch1 * g.W + cw1; t.cv[3] = t1 && l1; t.w[3] = t.lh * t.lw;
    return t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// grad wrt offsets, mask and input from d(columns).  One wavefront per (pixel m, tap).
template <typename T>
__global__ __launch_bounds__(256) void dcn_bwd_sample_kernel(const T* x, const float* om, const T* gcol, BwdGeom g,
                                                             float* gx, float* gom) {
    const int lane = threadIdx.x & 63;
    const long wave_id = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long pair = wave_id; pair < (long)g.M * g.kk; pair += nwaves) {
        const int m = (int)(pair / g.kk), tap = (int)(pair - (long)m * g.kk);
        const int hw = g.Ho * g.Wo, b = m / hw, rem = m - b * hw, oh = rem / g.Wo, ow = rem - oh * g.Wo;
        const float* om_row = om + (size_t)m * 32;
        const float mask = om_row[18 + tap];
        const TapGeo t = tap_geometry(g, om_row, b, oh, ow, tap);
        float gm = 0.f, gh = 0.f, gw = 0.f;
        if (t.inside) {
            for (int c = lane; c < g.Cp; c += 64) {
                const float gc = ElemTraits<T>::load(gcol + (size_t)m * g.Kp + tap * g.Cp + c);
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = t.cv[q] ? ElemTraits<T>::load(x + (size_t)t.off[q] * g.Cp + c) : 0.f;
                gm += gc * (t.w[0] * v[0] + t.w[1] * v[1] + t.w[2] * v[2] + t.w[3] * v[3]);
                // dmcn_get_coordinate_weight (dcn_v2_im2col_cuda.cu:82-122)
                gh += (-t.hw * v[0] - t.lw * v[1] + t.hw * v[2] + t.lw * v[3]) * gc * mask;
                gw += (-t.hh * v[0] + t.hh * v[1] - t.lh * v[2] + t.lh * v[3]) * gc * mask;
                const float top = gc * mask;                  // col2im (dcn_v2_im2col_cuda.cu:197-254)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (t.cv[q]) unsafeAtomicAdd(gx + (size_t)t.off[q] * g.Cp + c, t.w[q] * top);
            }
        }
        gm = wave_sum(gm); gh = wave_sum(gh); gw = wave_sum(gw);
        if (lane == 0) {
            float* o = gom + (size_t)m * 32;
            o[2 * tap] = gh; o[2 * tap + 1] = gw; o[18 + tap] = gm;
        }
    }
}

int g_opt_dcn_wgrad_m = 512;    // option "dcn_wgrad_m": pixels per workgroup slab of the DCN weight-gradient kernel (step: 512 -> 73.1 ms, 2048 -> 73.8, 8192 -> 83.6)

// grad_weight[o][k] += sum over a slab of pixels of go[m][o] * col[m][k]; col re-sampled into LDS.
// Block = 64 k x 64 o output tile, 256 threads each owning a 4x4 register block.
constexpr int WG_MCH = 16;     // pixels staged per step
template <typename T>
__global__ __launch_bounds__(256) void dcn_bwd_wgrad_kernel(const T* x, const float* om, const T* go, BwdGeom g,
                                                            int m_per_block, float* gw) {
    __shared__ float cs[WG_MCH][64 + 4];     // col chunk  [m][k]
    __shared__ float gs[WG_MCH][64 + 4];     // grad chunk [m][o]
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * 64, o0 = blockIdx.y * 64;
    const int m_begin = blockIdx.z * m_per_block, m_end = min(m_begin + m_per_block, g.M);
    const int tk = (tid & 15) * 4, to = (tid >> 4) * 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    // staging role: thread -> (row r = tid/16, 4 consecutive k / o at (tid%16)*4)
    const int sr = tid >> 4, sc = (tid & 15) * 4;
    const int hw = g.Ho * g.Wo;
    for (int mb = m_begin; mb < m_end; mb += WG_MCH) {
        const int m = mb + sr;
        float cv4[4] = {0.f, 0.f, 0.f, 0.f}, gv4[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < m_end) {
            const f32x4 gg = ld4<T>(go + (size_t)m * g.Coutp + o0 + sc);
            gv4[0] = gg[0]; gv4[1] = gg[1]; gv4[2] = gg[2]; gv4[3] = gg[3];
            const int k = k0 + sc;
            if (k < g.K) {                                   // Cp >= 16 -> the 4 k's share one tap
                const int tap = k >> g.lgC, c = k & (g.Cp - 1);
                const int b = m / hw, rem = m - b * hw, oh = rem / g.Wo, ow = rem - oh * g.Wo;
                const float* om_row = om + (size_t)m * 32;
                const TapGeo t = tap_geometry(g, om_row, b, oh, ow, tap);
                const float mask = om_row[18 + tap];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (t.cv[q]) {
                        const f32x4 v = ld4<T>(x + (size_t)t.off[q] * g.Cp + c);
                        const float wq = t.w[q] * mask;
                        cv4[0] += wq * v[0]; cv4[1] += wq * v[1]; cv4[2] += wq * v[2]; cv4[3] += wq * v[3];
                    }
            }
        }
        __syncthreads();                                      // previous step's readers are done
#pragma unroll
        for (int e = 0; e < 4; ++e) { cs[sr][sc + e] = cv4[e]; gs[sr][sc + e] = gv4[e]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < WG_MCH; ++r) {
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
            if (k0 + tk + i < g.K) unsafeAtomicAdd(gw + (size_t)(o0 + to + j) * g.K + k0 + tk + i, acc[i][j]);
}

// grad_bias[o] = sum_m go[m][o]
template <typename T>
__global__ void dcn_bwd_bias_kernel(const T* go, int M, int Coutp, int rows_per_block, float* gb) {
    const int o = blockIdx.y * 64 + threadIdx.x;
    const int r0 = blockIdx.x * rows_per_block;
    float s = 0.f;
    for (int r = r0 + threadIdx.y; r < min(r0 + rows_per_block, M); r += blockDim.y) s += ElemTraits<T>::load(go + (size_t)r * Coutp + o);
    __shared__ float red[4][64];
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0) unsafeAtomicAdd(gb + o, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// packing / unpacking between the reference layouts and the kernels' layouts
__global__ void bwd_pack_offmask(const float* offset, const float* mask, float* om, int B, int HW, int kk) {
    const long total = (long)B * HW * 32;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i & 31);
        const long m = i >> 5;
        const int b = (int)(m / HW), p = (int)(m - (long)b * HW);
        float v = 0.f;
        if (ch < 18) { if (ch < 2 * kk) v = offset[((size_t)b * 2 * kk + ch) * HW + p]; }
        else if (ch < 27) { if (ch - 18 < kk) v = mask[((size_t)b * kk + (ch - 18)) * HW + p]; }
        om[i] = v;
    }
}
__global__ void bwd_unpack_offmask(const float* gom, float* goff, float* gmask, int B, int HW, int kk) {
    const long total = (long)B * HW * 3 * kk;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const int ch = (int)((i / HW) % (3 * kk));
        const int b = (int)(i / ((long)HW * 3 * kk));
        const size_t m = (size_t)b * HW + p;
        if (ch < 2 * kk) goff[((size_t)b * 2 * kk + ch) * HW + p] = gom[m * 32 + ch];
        else gmask[((size_t)b * kk + (ch - 2 * kk)) * HW + p] = gom[m * 32 + 18 + (ch - 2 * kk)];
    }
}
// weight (Cout,C,kk) -> transposed pack wT[Kp][Coutp]: wT[tap*Cp+c][o]
template <typename T>
__global__ void bwd_pack_weight_t(const float* w, T* wT, int Cout, int C, int kk, int Cp, int Kp, int Coutp) {
    const long total = (long)Kp * Coutp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i % Coutp);
        const int k = (int)(i / Coutp);
        const int tap = k / Cp, c = k - tap * Cp;
        ElemTraits<T>::store(wT + i, (o < Cout && c < C && tap < kk) ? w[((size_t)o * C + c) * kk + tap] : 0.f);
    }
}
// packed grad [Coutp][K] (k = tap*Cp+c) -> (Cout,C,kk)
__global__ void bwd_unpack_weight(const float* gwp, float* gw, int Cout, int C, int kk, int Cp, int K) {
    const long total = (long)Cout * C * kk;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % kk);
        const int c = (int)((i / kk) % C);
        const int o = (int)(i / ((long)kk * C));
        gw[i] = gwp[(size_t)o * K + tap * Cp + c];
    }
}

}  // namespace mfx
using namespace mfx;

#define BWD_GRID(total) dim3((unsigned)(cdv((total), 256) < 8192 ? cdv((total), 256) : 8192))

struct BwdLayout { size_t x, om, wT, go, gcol, gx, gom, gwp, gb, total; };
static BwdLayout bwd_layout(const BwdGeom& g) {
    BwdLayout L; size_t o = 0;
    L.x = o;    o += al256((size_t)g.B * g.H * g.W * g.Cp * 4);
    L.om = o;   o += al256((size_t)g.M * 32 * 4);
    L.wT = o;   o += al256((size_t)g.Kp * g.Coutp * 4);
    L.go = o;   o += al256((size_t)g.M * g.Coutp * 4);
    L.gcol = o; o += al256((size_t)g.M * g.Kp * 4);
    L.gx = o;   o += al256((size_t)g.B * g.H * g.W * g.Cp * 4);
    L.gom = o;  o += al256((size_t)g.M * 32 * 4);
    L.gwp = o;  o += al256((size_t)g.Coutp * g.K * 4);
    L.gb = o;   o += al256((size_t)g.Coutp * 4);
    L.total = o;
    return L;
}
static BwdGeom bwd_geom(int B, int C, int H, int W, int Cout, int kh, int kw, int s, int p, int d, int sw = -1, int pw = -1, int dw = -1) {
    BwdGeom g;
    g.B = B; g.H = H; g.W = W; g.Cp = next_pow2_(C < 16 ? 16 : C); g.lgC = 0; while ((1 << g.lgC) < g.Cp) ++g.lgC;
    g.kh = kh; g.kw = kw; g.kk = kh * kw; g.stride = s; g.pad = p; g.dil = d;
    g.stride_w = sw < 0 ? s : sw; g.pad_w = pw < 0 ? p : pw; g.dil_w = dw < 0 ? d : dw;
    g.Ho = (H + 2 * p - (d * (kh - 1) + 1)) / s + 1; g.Wo = (W + 2 * g.pad_w - (g.dil_w * (kw - 1) + 1)) / g.stride_w + 1;
    g.M = B * g.Ho * g.Wo; g.K = g.kk * g.Cp; g.Kp = ((g.K + 63) / 64) * 64;
    g.Coutp = next_pow2_(Cout < 64 ? 64 : Cout);
    return g;
}

// gradients of the NHWC fp32 problem: gcol scratch [M][Kp]; gx [B*H*W][Cp], gwp [Coutp][K], gb [Coutp] must be zeroed
template <typename T>
static int dcn_bwd_core(const T* x, const float* om, const T* wT, const T* go, T* gcol, float* gx, float* gom,
                        float* gwp, float* gb, const BwdGeom& g, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // d(columns)[m][k] = sum_o go[m][o] * W[o][k]   (dcn_v2_cuda.cu:273) as a 1x1 implicit GEMM
    mfx_conv_desc cd = {};
    cd.x = go; cd.w = wT; cd.w_frag = nullptr; cd.scale = nullptr; cd.shift = nullptr; cd.res = nullptr; cd.y = gcol; cd.rowmap = nullptr;
    cd.B = 1; cd.H = 1; cd.W = g.M; cd.x_pixstride = g.Coutp; cd.Ck = g.Coutp; cd.kh = 1; cd.kw = 1; cd.stride = 1;
    cd.pad_h = 0; cd.pad_w = 0; cd.dil_w = 1; cd.Ho = 1; cd.Wo = g.M; cd.M = g.M; cd.Cout = g.Kp; cd.Cout_pad = g.Kp;
    cd.K_pad = g.Coutp; cd.ldy = g.Kp; cd.ldres = 0; cd.act = MFX_ACT_NONE; cd.dtype = std::is_same<T, float>::value ? MFX_F32 : MFX_BF16; cd.out_dtype = cd.dtype;
    cd.workspace = nullptr; cd.workspace_bytes = 0;
    int rc = mfx_conv2d_nhwc(&cd, stream);
    if (rc) return rc;
    {   // grad_offset, grad_mask, grad_input
        const long pairs = (long)g.M * g.kk;
        const int blocks = (int)((pairs + 3) / 4 < 65536 ? (pairs + 3) / 4 : 65536);
        hipLaunchKernelGGL(dcn_bwd_sample_kernel<T>, dim3(blocks), dim3(256), 0, st, x, om, gcol, g, gx, gom);
    }
    {   // grad_weight
        const int m_per_block = g_opt_dcn_wgrad_m;
        dim3 grid(g.Kp / 64, g.Coutp / 64, cdv(g.M, m_per_block));
        hipLaunchKernelGGL(dcn_bwd_wgrad_kernel<T>, grid, dim3(256), 0, st, x, om, go, g, m_per_block, gwp);
    }
    {   // grad_bias
        const int rows = g.M >= (1 << 18) ? 256 : 128;          // ~1000+ workgroups (240 left this pass latency-bound)
        hipLaunchKernelGGL(dcn_bwd_bias_kernel<T>, dim3(cdv(g.M, rows), g.Coutp / 64), dim3(64, 4), 0, st, go, g.M, g.Coutp, rows, gb);
    }
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

// ---- fast route of the `_ext` backward (r06): 3x3 / stride 1 / pad 1 / dilation 1 with power-of-two channel counts >= 64 -- the model's own DCN geometry -- runs the
// tile-owned second-generation kernels (dcn_bwd_tile.hip: d(columns) GEMM on the matrix cores, grad_input gathered per tile in LDS instead of scattered with
// global atomics, weight gradient as an MFMA GEMM) behind the same NCHW boundary; d_raw's mask channels are asked for as the gradient of the mask ITSELF
// (BtGeom.raw_mask): `_ext` takes the mask as an input and knows nothing of the sigmoid in front of it (src/dcn_v2.h:48-59).
int g_opt_ext_bwd_fast = 1;  // option "ext_bwd_fast": 0 = the first-generation scatter backward for every geometry
extern "C" size_t mfx_dcn_backward_v2_workspace_bytes(int B, int C, int H, int W, int Cout, int dtype);
int mfx_internal_dcn_backward_v2_f32_rawmask(const float* x, const float* offmask, const float* weight_oihw, const float* dy, float* dx, float* d_raw,
                                             float* dweight, float* dbias, int B, int C, int H, int W, int Cout, void* workspace, size_t workspace_bytes, void* stream);
static bool ext_bwd_fast_ok(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw) {
    return g_opt_ext_bwd_fast && !g_opt_det && kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && dh == 1 && dw == 1 && C >= 64 && (C & (C - 1)) == 0 &&
           Cout >= 64 && (Cout & (Cout - 1)) == 0 && H < 4096 && W < 4096 && (long)B * H * W < (1L << 31) / 32;
}
struct FastLayout { size_t x, om, go, gx, gom, v2, total; };
static FastLayout fast_layout(int B, int C, int H, int W, int Cout) {
    FastLayout L; size_t o = 0;
    const size_t M = (size_t)B * H * W;
    L.x = o;   o += al256(M * C * 4);
    L.om = o;  o += al256(M * 32 * 4);
    L.go = o;  o += al256(M * Cout * 4);
    L.gx = o;  o += al256(M * C * 4);
    L.gom = o; o += al256(M * 32 * 4);
    L.v2 = o;  o += al256(mfx_dcn_backward_v2_workspace_bytes(B, C, H, W, Cout, MFX_F32));
    L.total = o;
    return L;
}

extern "C" size_t mfx_dcn_v2_backward_workspace_bytes_(int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw) {
    const size_t slow = bwd_layout(bwd_geom(B, C, H, W, Cout, kh, kw, sh, ph, dh, sw, pw, dw)).total;
    // the fast route may be taken by a channel SLICE of the layer (deformable groups: C / dg channels per call of backward_one): size for the largest
    // power of two <= C, which bounds every slice that qualifies (fast_layout grows with C)
    int Cp2 = 64;
    while (Cp2 * 2 <= C) Cp2 *= 2;
    if (C < 64 || !ext_bwd_fast_ok(B, Cp2, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw)) return slow;
    const size_t fast = fast_layout(B, Cp2, H, W, Cout).total;
    return fast > slow ? fast : slow;
}

int mfx_internal_ext_slice(const float* src, float* dst, int B, int Cs, int cs0, int Cd, int cd0, int Cg, int HW, int accumulate, void* stream);   // dcn_ext.hip

// one deformable group: contiguous NCHW fp32 operands and gradients
static int backward_one(const float* input, const float* weight, const float* offset, const float* mask, const float* grad_output,
                        float* grad_input, float* grad_offset, float* grad_mask, float* grad_weight, float* grad_bias,
                        int B, int C, int Cout, const BwdGeom& g, char* ws, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (ext_bwd_fast_ok(B, C, g.H, g.W, Cout, g.kh, g.kw, g.stride, g.stride_w, g.pad, g.pad_w, g.dil, g.dil_w)) {
        const FastLayout F = fast_layout(B, C, g.H, g.W, Cout);
        float* x = (float*)(ws + F.x); float* om = (float*)(ws + F.om); float* go = (float*)(ws + F.go); float* gx = (float*)(ws + F.gx); float* gom = (float*)(ws + F.gom);
        const int HW = g.H * g.W;
        int rc = mfx_nchw_to_nhwc(input, x, B, C, g.H, g.W, C, MFX_F32, stream);
        if (rc) return rc;
        rc = mfx_nchw_to_nhwc(grad_output, go, B, Cout, g.H, g.W, Cout, MFX_F32, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(bwd_pack_offmask, BWD_GRID((long)g.M * 32), dim3(256), 0, st, offset, mask, om, B, HW, g.kk);
        MFX_HIP_CHECK(hipGetLastError());
        // grad_weight arrives as (Cout, C, 3, 3) and grad_bias as (Cout): the caller's own layouts -- written in place
        rc = mfx_internal_dcn_backward_v2_f32_rawmask(x, om, weight, go, gx, gom, grad_weight, grad_bias, B, C, g.H, g.W, Cout, ws + F.v2, F.total - F.v2, stream);
        if (rc) return rc;
        rc = mfx_nhwc_to_nchw(gx, grad_input, B, C, g.H, g.W, C, MFX_F32, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(bwd_unpack_offmask, BWD_GRID((long)g.M * 3 * g.kk), dim3(256), 0, st, gom, grad_offset, grad_mask, B, HW, g.kk);
        MFX_HIP_CHECK(hipGetLastError());
        return MFX_OK;
    }
    const BwdLayout L = bwd_layout(g);
    float* x = (float*)(ws + L.x); float* om = (float*)(ws + L.om); float* wT = (float*)(ws + L.wT);
    float* go = (float*)(ws + L.go); float* gcol = (float*)(ws + L.gcol); float* gx = (float*)(ws + L.gx);
    float* gom = (float*)(ws + L.gom); float* gwp = (float*)(ws + L.gwp); float* gb = (float*)(ws + L.gb);
    const int HWo = g.Ho * g.Wo, H = g.H, W = g.W;

    int rc = mfx_nchw_to_nhwc(input, x, B, C, H, W, g.Cp, MFX_F32, stream);
    if (rc) return rc;
    rc = mfx_nchw_to_nhwc(grad_output, go, B, Cout, g.Ho, g.Wo, g.Coutp, MFX_F32, stream);   // .contiguous() is the caller's job (App. C item 18)
    if (rc) return rc;
    hipLaunchKernelGGL(bwd_pack_offmask, BWD_GRID((long)g.M * 32), dim3(256), 0, st, offset, mask, om, B, HWo, g.kk);
    hipLaunchKernelGGL(bwd_pack_weight_t<float>, BWD_GRID((long)g.Kp * g.Coutp), dim3(256), 0, st, weight, wT, Cout, C, g.kk, g.Cp, g.Kp, g.Coutp);
    MFX_HIP_CHECK(mfx::zero_async(gx, (size_t)B * H * W * g.Cp * 4, st));
    MFX_HIP_CHECK(mfx::zero_async(gwp, (size_t)g.Coutp * g.K * 4, st));
    MFX_HIP_CHECK(mfx::zero_async(gb, (size_t)g.Coutp * 4, st));
    MFX_HIP_CHECK(hipGetLastError());

    rc = dcn_bwd_core<float>(x, om, wT, go, gcol, gx, gom, gwp, gb, g, stream);
    if (rc) return rc;

    rc = mfx_nhwc_to_nchw(gx, grad_input, B, C, H, W, g.Cp, MFX_F32, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(bwd_unpack_offmask, BWD_GRID((long)g.M * 3 * g.kk), dim3(256), 0, st, gom, grad_offset, grad_mask, B, HWo, g.kk);
    hipLaunchKernelGGL(bwd_unpack_weight, BWD_GRID((long)Cout * C * g.kk), dim3(256), 0, st, gwp, grad_weight, Cout, C, g.kk, g.Cp, g.K);
    if (grad_bias) MFX_HIP_CHECK(mfx::copy_async(grad_bias, gb, (size_t)Cout * 4, st));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" size_t mfx_dcn_v2_workspace_bytes_g(int B, int C, int H, int W, int Cout, int kh, int kw,
                                               int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int deformable_group, int backward);

extern "C" int mfx_dcn_v2_backward(const float* input, const float* weight, const float* bias,
                                   const float* offset, const float* mask, const float* grad_output,
                                   float* grad_input, float* grad_offset, float* grad_mask,
                                   float* grad_weight, float* grad_bias,
                                   int B, int C, int H, int W, int Cout, int kh, int kw,
                                   int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                   int deformable_group, void* workspace, size_t workspace_bytes, void* stream) {
    (void)bias;
    if (!input || !weight || !offset || !mask || !grad_output || !grad_input || !grad_offset || !grad_mask || !grad_weight || !grad_bias)
        return mfx_fail(MFX_ERR_ARG, "dcn_v2_backward: null pointer");
    const int dg = deformable_group;
    if (dg < 1 || C % dg != 0) return mfx_fail(MFX_ERR_ARG, "dcn_v2_backward: deformable_group must divide the input channels");
    if (stride_h < 1 || stride_w < 1 || dil_h < 1 || dil_w < 1 || pad_h < 0 || pad_w < 0) return mfx_fail(MFX_ERR_ARG, "dcn_v2_backward: bad stride / padding / dilation");
    if (kh * kw > 9) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_v2_backward: at most 9 taps");
    const int Cg = C / dg, kk = kh * kw;
    const BwdGeom g = bwd_geom(B, Cg, H, W, Cout, kh, kw, stride_h, pad_h, dil_h, stride_w, pad_w, dil_w);
    if (g.Ho <= 0 || g.Wo <= 0) return mfx_fail(MFX_ERR_ARG, "dcn_v2_backward: empty output");
    if (!workspace || workspace_bytes < mfx_dcn_v2_workspace_bytes_g(B, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, dg, 1))
        return mfx_fail(MFX_ERR_WORKSPACE, "dcn_v2_backward: workspace too small (mfx_dcn_v2_workspace_bytes_g)");
    if (g.M == 0) return MFX_OK;
    char* ws = reinterpret_cast<char*>(workspace);
    if (dg == 1) return backward_one(input, weight, offset, mask, grad_output, grad_input, grad_offset, grad_mask, grad_weight, grad_bias, B, C, Cout, g, ws, stream);
    // deformable groups: group g's channel slice of the input / weight and its offset / mask channels give that slice's gradients (the output
    // gradient is shared); grad_bias is the same sum for every group
    const int HW = H * W, HWo = g.Ho * g.Wo;
    auto a256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
    char* t = ws + mfx_dcn_v2_backward_workspace_bytes_(B, C, H, W, Cout, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w);
    float* xg = (float*)t; t += a256((size_t)B * C * HW * 4);
    float* wg = (float*)t; t += a256((size_t)Cout * C * kk * 4);
    float* og = (float*)t; t += a256((size_t)B * 2 * kk * HWo * 4);
    float* mg = (float*)t; t += a256((size_t)B * kk * HWo * 4);
    float* gxg = (float*)t; t += a256((size_t)B * C * HW * 4);
    float* gwg = (float*)t; t += a256((size_t)Cout * C * kk * 4);
    float* gog = (float*)t; t += a256((size_t)B * 2 * kk * HWo * 4);
    float* gmg = (float*)t;
    int rc;
    for (int q = 0; q < dg; ++q) {
        if ((rc = mfx_internal_ext_slice(input, xg, B, C, q * Cg, Cg, 0, Cg, HW, 0, stream))) return rc;
        if ((rc = mfx_internal_ext_slice(weight, wg, Cout, C, q * Cg, Cg, 0, Cg, kk, 0, stream))) return rc;
        if ((rc = mfx_internal_ext_slice(offset, og, B, dg * 2 * kk, q * 2 * kk, 2 * kk, 0, 2 * kk, HWo, 0, stream))) return rc;
        if ((rc = mfx_internal_ext_slice(mask, mg, B, dg * kk, q * kk, kk, 0, kk, HWo, 0, stream))) return rc;
        if ((rc = backward_one(xg, wg, og, mg, grad_output, gxg, gog, gmg, gwg, q == 0 ? grad_bias : nullptr, B, Cg, Cout, g, ws, stream))) return rc;
        if ((rc = mfx_internal_ext_slice(gxg, grad_input, B, Cg, 0, C, q * Cg, Cg, HW, 0, stream))) return rc;
        if ((rc = mfx_internal_ext_slice(gwg, grad_weight, Cout, Cg, 0, C, q * Cg, Cg, kk, 0, stream))) return rc;
        if ((rc = mfx_internal_ext_slice(gog, grad_offset, B, 2 * kk, 0, dg * 2 * kk, q * 2 * kk, 2 * kk, HWo, 0, stream))) return rc;
        if ((rc = mfx_internal_ext_slice(gmg, grad_mask, B, kk, 0, dg * kk, q * kk, kk, HWo, 0, stream))) return rc;
    }
    return MFX_OK;
}

// Training entry (NHWC fp32 tensors straight from the model's activations; no layout transforms):
//   x [B][H][W][C], offmask [M][32] (mask already sigmoided), weight (Cout,C,kh,kw) fp32, dy [M][Cout]
//   -> dx [B][H][W][C], d_offmask [M][32] (grad wrt offsets and wrt the POST-sigmoid mask), dweight (Cout,C,kh,kw), dbias [Cout]
// C and Cout must be powers of two (C >= 16, Cout >= 64).  workspace: mfx_dcn_backward_nhwc_workspace_bytes().
extern "C" size_t mfx_dcn_backward_nhwc_workspace_bytes(int B, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil) {
    const BwdGeom g = bwd_geom(B, C, H, W, Cout, kh, kw, stride, pad, dil);
    return al256((size_t)g.Kp * g.Coutp * 4) + al256((size_t)g.M * g.Kp * 4) + al256((size_t)g.Coutp * g.K * 4) + al256((size_t)g.Coutp * 4);
}

template <typename T>
static int dcn_backward_nhwc_impl(const T* x, const float* offmask, const float* weight, const T* dy,
                                  float* dx, float* d_offmask, float* dweight, float* dbias,
                                  int B, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !offmask || !weight || !dy || !dx || !d_offmask || !dweight || !dbias) return mfx_fail(MFX_ERR_ARG, "dcn_backward_nhwc: null pointer");
    if (kh * kw > 9) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_backward_nhwc: at most 9 taps");
    const BwdGeom g = bwd_geom(B, C, H, W, Cout, kh, kw, stride, pad, dil);
    if (g.Cp != C || g.Coutp != Cout) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_backward_nhwc: C (>=16) and Cout (>=64) must be powers of two");
    if (workspace_bytes < mfx_dcn_backward_nhwc_workspace_bytes(B, C, H, W, Cout, kh, kw, stride, pad, dil) || !workspace)
        return mfx_fail(MFX_ERR_WORKSPACE, "dcn_backward_nhwc: workspace too small");
    if (g.M == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(workspace);                     // regions sized for fp32 (the bf16 build uses half of two of them)
    T* wT = (T*)ws; ws += al256((size_t)g.Kp * g.Coutp * 4);
    T* gcol = (T*)ws; ws += al256((size_t)g.M * g.Kp * 4);
    float* gwp = (float*)ws; ws += al256((size_t)g.Coutp * g.K * 4);
    float* gb = (float*)ws;
    hipLaunchKernelGGL(bwd_pack_weight_t<T>, BWD_GRID((long)g.Kp * g.Coutp), dim3(256), 0, st, weight, wT, Cout, C, g.kk, g.Cp, g.Kp, g.Coutp);
    MFX_HIP_CHECK(mfx::zero_async(dx, (size_t)B * H * W * C * 4, st));
    MFX_HIP_CHECK(mfx::zero_async(d_offmask, (size_t)g.M * 32 * 4, st));
    MFX_HIP_CHECK(mfx::zero_async(gwp, (size_t)g.Coutp * g.K * 4, st));
    MFX_HIP_CHECK(mfx::zero_async(gb, (size_t)g.Coutp * 4, st));
    int rc = dcn_bwd_core<T>(x, offmask, wT, dy, gcol, dx, d_offmask, gwp, gb, g, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(bwd_unpack_weight, BWD_GRID((long)Cout * C * g.kk), dim3(256), 0, st, gwp, dweight, Cout, C, g.kk, g.Cp, g.K);
    MFX_HIP_CHECK(mfx::copy_async(dbias, gb, (size_t)Cout * 4, st));
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_dcn_backward_nhwc(const float* x, const float* offmask, const float* weight, const float* dy,
                                     float* dx, float* d_offmask, float* dweight, float* dbias,
                                     int B, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    return dcn_backward_nhwc_impl<float>(x, offmask, weight, dy, dx, d_offmask, dweight, dbias, B, C, H, W, Cout, kh, kw, stride, pad, dil,
                                         workspace, workspace_bytes, stream);
}

// bf16 activations: x and dy bf16; d(columns) is produced by the bf16 MFMA GEMM and kept in bf16; every gradient output
// (dx included: it is accumulated with fp32 atomics) stays fp32
extern "C" int mfx_dcn_backward_nhwc_bf16(const void* x, const float* offmask, const float* weight, const void* dy,
                                          float* dx, float* d_offmask, float* dweight, float* dbias,
                                          int B, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    if (C < 64) return mfx_fail(MFX_ERR_UNSUPPORTED, "dcn_backward_nhwc_bf16: C >= 64");
    return dcn_backward_nhwc_impl<bf16_t>((const bf16_t*)x, offmask, weight, (const bf16_t*)dy, dx, d_offmask, dweight, dbias,
                                          B, C, H, W, Cout, kh, kw, stride, pad, dil, workspace, workspace_bytes, stream);
}
