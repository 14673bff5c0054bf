// Probe: __builtin_amdgcn_global_load_lds on gfx950 -- where does lane l's 16 bytes land?  (expected: wave-uniform LDS base + 16 * l)
// hipcc --offload-arch=gfx950 -O2 tools/probes/glds_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gl_void;

__global__ void probe(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, const uint32_t* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2048 / 4 * 2; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    // every lane fetches a PERMUTED 16-byte chunk: lane l reads chunk (l ^ 5); odd lanes of wave 1 read the zero page
    const uint32_t* g = src + (size_t)(wave * 64 + (lane ^ 5)) * 4;
    if (wave == 1 && (lane & 1)) g = zeros;
    char* base = smem + wave * 1024;                                   // wave-uniform
    __builtin_amdgcn_global_load_lds((gl_void*)g, (lds_void*)base, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048 / 4; i += blockDim.x) out[i] = reinterpret_cast<uint32_t*>(smem)[i];
}

int main() {
    std::vector<uint32_t> h(128 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i;
    uint32_t *d, *o, *z;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 2048); hipMalloc(&z, 64);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemset(z, 0, 64);
    hipLaunchKernelGGL(probe, dim3(1), dim3(128), 4096, 0, d, o, z);
    std::vector<uint32_t> r(512);
    if (hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    int bad = 0;
    for (int w = 0; w < 2; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                uint32_t want = (w == 1 && (l & 1)) ? 0u : (uint32_t)((w * 64 + (l ^ 5)) * 4 + e);
                if (r[(w * 64 + l) * 4 + e] != want) { if (bad < 8) printf("wave %d lane %d elem %d: got %u want %u\n", w, l, e, r[(w * 64 + l) * 4 + e], want); ++bad; }
            }
    printf("glds probe: %s (%d mismatches) -- lane l's 16 bytes land at base + 16*l\n", bad ? "MISMATCH" : "ok", bad);
    return bad != 0;
}
