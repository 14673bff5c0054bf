// Probes behind the split-precision ("f16x2": fp16 hi + fp16 lo per element) MFMA mode (gfx950):
//  1. v_permlane16_swap_b32 with both operands the same register: which 16-lane rows end up where?
//  2. v_mfma_f32_16x16x32_f16 with SUBNORMAL fp16 inputs: flushed or honoured?
//  3. accuracy of  acc += a*[bh bh bh bh] ; acc += a*[bl bl bl bl]  with a = [ah al ah al] chunks against an fp64 dot product,
//     next to the plain fp16 MFMA and the fp32 MFMA of the same data.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/split_probe.hip -o build_variants/split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ void swap_probe(uint32_t* out) {
    const uint32_t v = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(v, v + 100, false, false);
    out[128 + threadIdx.x] = q[0];
    out[192 + threadIdx.x] = q[1];
}

// one 16x16x32 MFMA: A[m][k] = av for k == 0 else 0, B[n][k] = bv for k == 0 else 0 -> D[m][n] = av*bv
__global__ void denorm_probe(const uint16_t* ab, float* out) {
    const int lane = threadIdx.x;
    f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((lane >> 4) == 0) { a[0] = __builtin_bit_cast(_Float16, ab[0]); b[0] = __builtin_bit_cast(_Float16, ab[1]); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

// D[16][16] = A[16][K] * B[16][K]^T three ways.  Asplit/Bsplit: per row, per 8 elements: 8 hi halves then 8 lo halves (32 B).
__global__ void split_gemm(const uint32_t* Asplit, const uint32_t* Bsplit, const float* A, const float* B, int K, float* Dsplit, float* Df16,
                           float* Df32) {
    const int lane = threadIdx.x, row = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acch = {0.f, 0.f, 0.f, 0.f}, accf = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {          // one step = 16 elements = 64 bytes of the split row
        const u32x4 a = *reinterpret_cast<const u32x4*>(Asplit + (size_t)row * K + k0 + kq * 4);       // kq 0: hi g0, 1: lo g0, 2: hi g1, 3: lo g1
        const u32x4 b = *reinterpret_cast<const u32x4*>(Bsplit + (size_t)row * K + k0 + kq * 4);
        u32x4 b1, b2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            auto r = __builtin_amdgcn_permlane16_swap(b[i], b[i], false, false);
            b1[i] = r[0]; b2[i] = r[1];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b1), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b2), acc, 0, 0, 0);
        // plain fp16: only the hi chunks (lanes with odd kq contribute zero)
        u32x4 ah = a, bh = b;
        if (kq & 1) { ah = u32x4{0, 0, 0, 0}; bh = u32x4{0, 0, 0, 0}; }
        acch = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), acch, 0, 0, 0);
        const f32x4 af = *reinterpret_cast<const f32x4*>(A + (size_t)row * K + k0 + kq * 4);
        const f32x4 bf = *reinterpret_cast<const f32x4*>(B + (size_t)row * K + k0 + kq * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) accf = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[i], accf, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) {
        Dsplit[(kq * 4 + r) * 16 + row] = acc[r];
        Df16[(kq * 4 + r) * 16 + row] = acch[r];
        Df32[(kq * 4 + r) * 16 + row] = accf[r];
    }
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

static void split_rows(const std::vector<float>& M, int rows, int K, std::vector<uint32_t>& out) {
    out.assign((size_t)rows * K, 0);
    uint16_t* o = reinterpret_cast<uint16_t*>(out.data());
    for (int r = 0; r < rows; ++r)
        for (int g = 0; g < K / 8; ++g)
            for (int e = 0; e < 8; ++e) {
                const float v = M[(size_t)r * K + g * 8 + e];
                const uint16_t hi = f2h(v);
                const uint16_t lo = f2h(v - h2f(hi));
                o[((size_t)r * K + g * 8) * 2 + e] = hi;
                o[((size_t)r * K + g * 8) * 2 + 8 + e] = lo;
            }
}

int main() {
    uint32_t* d_out; hipMalloc(&d_out, 256 * 4);
    hipLaunchKernelGGL(swap_probe, dim3(1), dim3(64), 0, 0, d_out);
    uint32_t h[256]; hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    printf("permlane16_swap(v, v), v = lane id: result rows (lane 0 / 16 / 32 / 48 of each output)\n");
    printf("  r[0]: %u %u %u %u   r[1]: %u %u %u %u\n", h[0], h[16], h[32], h[48], h[64], h[80], h[96], h[112]);
    printf("permlane32_swap(v, v+100): r[0]: %u %u %u %u   r[1]: %u %u %u %u\n", h[128], h[144], h[160], h[176], h[192], h[208], h[224], h[240]);
    bool ok16 = h[0] == 0 && h[16] == 0 && h[32] == 32 && h[48] == 32 && h[64] == 16 && h[80] == 16 && h[96] == 48 && h[112] == 48;
    printf("  permlane16_swap(b, b) = ([r0 r0 r2 r2], [r1 r1 r3 r3]): %s\n", ok16 ? "YES" : "NO");

    uint16_t* d_ab; float* d_f; hipMalloc(&d_ab, 4); hipMalloc(&d_f, 4);
    const float tests[][2] = {{ldexpf(1.f, -20), 1024.f}, {ldexpf(1.f, -24), 16384.f}, {ldexpf(3.f, -24), 1.f}, {ldexpf(1.f, -14), 1.f}, {ldexpf(1.f, -20), ldexpf(1.f, -20)}};
    for (auto& t : tests) {
        uint16_t ab[2] = {f2h(t[0]), f2h(t[1])};
        hipMemcpy(d_ab, ab, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(denorm_probe, dim3(1), dim3(64), 0, 0, d_ab, d_f);
        float r; hipMemcpy(&r, d_f, 4, hipMemcpyDeviceToHost);
        printf("mfma f16 subnormal probe: a = %g (0x%04x) b = %g -> %g (exact %g) %s\n", h2f(ab[0]), ab[0], h2f(ab[1]), r, (double)h2f(ab[0]) * h2f(ab[1]),
               r == h2f(ab[0]) * h2f(ab[1]) ? "HONOURED" : "FLUSHED/DIFFERENT");
    }

    for (int K : {576, 4608}) {
        for (int mode = 0; mode < 2; ++mode) {      // 0: activations ~ |N(0,1)| x weights ~ U(-b, b), b = 1/sqrt(K);  1: both N(0,1)
            std::mt19937 rng(7 + K + mode);
            std::normal_distribution<float> nd(0.f, 1.f);
            std::uniform_real_distribution<float> ud(-1.f, 1.f);
            std::vector<float> A(16 * K), B(16 * K);
            for (auto& v : A) v = mode == 0 ? fabsf(nd(rng)) : nd(rng);
            for (auto& v : B) v = mode == 0 ? ud(rng) / sqrtf((float)K) : nd(rng);
            std::vector<uint32_t> As, Bs; split_rows(A, 16, K, As); split_rows(B, 16, K, Bs);
            uint32_t *dAs, *dBs; float *dA, *dB, *dD;
            hipMalloc(&dAs, As.size() * 4); hipMalloc(&dBs, Bs.size() * 4); hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 3 * 256 * 4);
            hipMemcpy(dAs, As.data(), As.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dBs, Bs.data(), Bs.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(split_gemm, dim3(1), dim3(64), 0, 0, dAs, dBs, dA, dB, K, dD, dD + 256, dD + 512);
            std::vector<float> D(768); hipMemcpy(D.data(), dD, 768 * 4, hipMemcpyDeviceToHost);
            double es = 0, eh = 0, ef = 0, mag = 0;
            for (int m = 0; m < 16; ++m)
                for (int n = 0; n < 16; ++n) {
                    double ref = 0;
                    for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * B[(size_t)n * K + k];
                    es = fmax(es, fabs(D[m * 16 + n] - ref)); eh = fmax(eh, fabs(D[256 + m * 16 + n] - ref)); ef = fmax(ef, fabs(D[512 + m * 16 + n] - ref));
                    mag = fmax(mag, fabs(ref));
                }
            printf("K %4d mode %d: max|D| %.3f   max err  split %.3e   fp16 %.3e   fp32-mfma %.3e\n", K, mode, mag, es, eh, ef);
            hipFree(dAs); hipFree(dBs); hipFree(dA); hipFree(dB); hipFree(dD);
        }
    }
    return 0;
}
