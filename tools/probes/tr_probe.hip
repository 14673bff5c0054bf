// Probe of ds_read_b64_tr_b16 (gfx950): which 16-bit LDS elements does lane l receive when every lane supplies its own
// 8-byte-aligned address?  LDS holds u16 value = its own element index.  Prints, per lane, the four element indices read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const int* addr, uint16_t* out) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    uint32_t a = (uint32_t)(uintptr_t)lds + addr[threadIdx.x];
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a));
    out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
    out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
    int h_addr[64]; uint16_t h_out[256];
    int* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h_addr[l] = l * 8;                                         // contiguous 8 B per lane
            if (pat == 1) h_addr[l] = (l & 15) * 64 + (l >> 4) * 8;                 // lane = row of 32 elements (64 B rows), group = 4-col chunk
            if (pat == 2) h_addr[l] = ((l & 15) >> 2) * 256 + (l & 3) * 8 + (l >> 4) * 32;   // 4 rows of stride 256 B per group, 4 x 8 B per row
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("  lane %2d addr %4d(B) -> elems %4d %4d %4d %4d\n", l, h_addr[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
    }
    return 0;
}
