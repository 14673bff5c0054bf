// Dense GEMM with a short K and a wide N, activation-stationary (r06): the projection half of the project-then-sample DCN (csrc/dcn_ps.hip),
//     y[m][n] = sum_k x[m][k] * w[n][k],      x: [M][K] 16-bit rows (K = 64 / 128 / 256 / 512 input channels), w: [N][K] K-contiguous (N = 9 Cout = 576 .. 2304).
// Why a kernel of its own: with K this short an output element costs K multiply-adds and 2 bytes of store -- the layer's floor is WRITING the map
// (70.8 MB at 5.5 TB/s = 13 us for 128 -> 64 @ 48 x 160, B = 8: tools/probes/membw_probe.py), and the tiled implicit-GEMM kernel (conv_kernels.hip) spends
// 36 us on it: two k-iterations per 64 x 64 tile, i.e. a prologue (operand tiles -> LDS, barrier) and an epilogue (LDS staging) per 128 MFMAs and nothing
// to overlap them with (the vendor's GEMM: 33 us, profiles/r06_dcn_ps.md).  Here a workgroup keeps its 128 (64 x FM) pixels' operand rows in REGISTERS for
// its whole life (K / 32 fragments per 16 pixels) and walks the output channels in chunks of BN: the chunk's weight rows go global -> registers (during the
// previous chunk's MFMAs) -> LDS (double-buffered: ONE barrier per chunk), every wave multiplies them against its own pixels and stores its results
// straight from the accumulators.  The GEMM runs transposed (weights = MFMA A operand) with the weight rows of a fragment PAIR permuted so that a lane
// ends up with 8 consecutive output channels of one pixel: one 16-byte store per pair, 64 contiguous bytes per pixel and pair, no LDS staging.
#include "../../include/monoflex_hip.h"
#include "err.h"
#include "common.h"

namespace mfx {

// LDS row r = 16 j + rho of a chunk holds channel chan(r): fragment j, row rho -> 32 (j >> 1) + 8 (rho >> 2) + 4 (j & 1) + (rho & 3)
__device__ __forceinline__ int as_chan(int r) { const int j = r >> 4, rho = r & 15; return 32 * (j >> 1) + 8 * (rho >> 2) + 4 * (j & 1) + (rho & 3); }

template <typename T, int K, int FM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_as_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, int M, int ldx, int ldy,
                                                        int nchunks, int nsplit) {
    constexpr int KS = K / 32, NJ = BN / 16, RS = K * 2 + 16;            // k-steps; fragments per chunk; LDS row stride (bytes)
    constexpr int CPR = K / 8, NLD = BN * CPR / 256;                     // 16-byte chunks per weight row; loads per thread and chunk
    static_assert(BN % 32 == 0 && BN * CPR % 256 == 0, "chunk geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];           // two chunk buffers of BN * RS bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xl = lane & 15, kq = lane >> 4;
    const int mt = blockIdx.x / nsplit, sp = blockIdx.x - mt * nsplit;
    const int c0 = (int)((long)nchunks * sp / nsplit), c1 = (int)((long)nchunks * (sp + 1) / nsplit);
    if (c0 >= c1) return;
    const int m0 = mt * (64 * FM) + wv * (16 * FM);

    // ---- this wave's pixel operand rows: FM fragments x KS k-steps, resident for every chunk
    u32x4 xb[FM][KS];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = min(m0 + 16 * i + xl, M - 1);
        const T* p = x + (size_t)m * ldx + kq * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) xb[i][s] = *reinterpret_cast<const u32x4*>(p + 32 * s);
    }
    // ---- weight chunk: thread t copies NLD 16-byte pieces; piece (row r, column c) comes from channel chan(r) of the chunk
    u32x4 wr[NLD];
    auto wload = [&](int c) {
        const T* wc = w + (size_t)c * BN * K;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = u * 256 + tid, r = idx / CPR, col = idx - r * CPR;
            wr[u] = *reinterpret_cast<const u32x4*>(wc + (size_t)as_chan(r) * K + col * 8);
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = u * 256 + tid, r = idx / CPR, col = idx - r * CPR;
            *reinterpret_cast<u32x4*>(smem + buf * (BN * RS) + r * RS + col * 16) = wr[u];
        }
    };
    wload(c0);
    wstore(0);
    __syncthreads();

    const char* abase = smem + xl * RS + kq * 16;
    for (int c = c0; c < c1; ++c) {
        const int buf = (c - c0) & 1;
        if (c + 1 < c1) wload(c + 1);
        f32x4 acc[NJ][FM];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < FM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* ab = abase + buf * (BN * RS);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 af[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) af[j] = *reinterpret_cast<const u32x4*>(ab + j * 16 * RS + s * 64);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < FM; ++i) mma_chunk<T>(af[j], xb[i][s], acc[j][i]);
        }
        // ---- stores: pair (2p, 2p + 1) of fragments = channels 32 p + 8 kq .. + 7 of pixel xl: 16 bytes per lane
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + 16 * i + xl;
            if (m < M) {
                T* yr = y + (size_t)m * ldy + (size_t)c * BN + kq * 8;
#pragma unroll
                for (int p = 0; p < NJ / 2; ++p) {
                    const float v[8] = {acc[2 * p][i][0], acc[2 * p][i][1], acc[2 * p][i][2], acc[2 * p][i][3],
                                        acc[2 * p + 1][i][0], acc[2 * p + 1][i][1], acc[2 * p + 1][i][2], acc[2 * p + 1][i][3]};
                    *reinterpret_cast<u32x4*>(yr + 32 * p) = ElemTraits<T>::pack(v);
                }
            }
        }
        if (c + 1 < c1) {
            wstore(buf ^ 1);                                  // (that buffer was last read in iteration c - 1: every wave is past the barrier that ended it)
            __syncthreads();
        }
    }
}

template <typename T, int K, int FM, int BN>
static int launch_gemm_as(const void* x, const void* w, void* y, int M, int N, int ldx, int ldy, hipStream_t st) {
    constexpr int smem = 2 * BN * (K * 2 + 16);
    static bool attr_done = false;
    if (!attr_done && smem > 64 * 1024) {
        MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_as_kernel<T, K, FM, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_done = true;
    }
    const int mtiles = (M + 64 * FM - 1) / (64 * FM), nchunks = N / BN;
    // enough workgroups for two per CU: split the channel chunks of a pixel tile over `nsplit` workgroups where the map is small
    int nsplit = 1;
    while (mtiles * nsplit < 512 && nsplit * 2 <= nchunks) nsplit *= 2;
    hipLaunchKernelGGL((gemm_as_kernel<T, K, FM, BN>), dim3(mtiles * nsplit), dim3(256), smem, st, reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(w),
                       reinterpret_cast<T*>(y), M, ldx, ldy, nchunks, nsplit);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

template <typename T> static int dispatch_gemm_as(const void* x, const void* w, void* y, int M, int K, int N, int ldx, int ldy, hipStream_t st) {
    if (K == 64) return launch_gemm_as<T, 64, 2, 64>(x, w, y, M, N, ldx, ldy, st);          // (r06: d(columns) = dy . W^T of the 64-output DCN layers' backward)
    if (K == 128) return launch_gemm_as<T, 128, 2, 64>(x, w, y, M, N, ldx, ldy, st);
    if (K == 256) return launch_gemm_as<T, 256, 2, 64>(x, w, y, M, N, ldx, ldy, st);
    if (K == 512) return launch_gemm_as<T, 512, 2, 32>(x, w, y, M, N, ldx, ldy, st);
    return mfx_fail(MFX_ERR_UNSUPPORTED, "project: K must be 64, 128, 256 or 512");
}

}  // namespace mfx
using namespace mfx;

extern "C" int mfx_project_nhwc(const void* x, const void* w, void* y, int M, int K, int N, int ldx, int ldy, int dtype, void* stream) {
    if (!x || !w || !y) return mfx_fail(MFX_ERR_ARG, "project: null pointer");
    if (M < 0 || N < 64 || N % 64 != 0 || ldx < K || ldy < N || ldx % 8 != 0 || ldy % 8 != 0) return mfx_fail(MFX_ERR_ARG, "project: bad shape (N a multiple of 64, 16-byte rows)");
    if (M == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MFX_BF16) return dispatch_gemm_as<bf16_t>(x, w, y, M, K, N, ldx, ldy, st);
    if (dtype == MFX_F16) return dispatch_gemm_as<half_t>(x, w, y, M, K, N, ldx, ldy, st);
    return mfx_fail(MFX_ERR_UNSUPPORTED, "project: 16-bit maps only (MFX_BF16 / MFX_F16)");
}
