// Stream-ordered zero fill / device copy as KERNELS.
//
// hipMemsetAsync / hipMemcpyAsync issued inside a stream capture become memset / memcpy graph nodes, and on this platform
// (ROCm 7.0 runtime under PyTorch 2.10, MI355X) such a node is not reliably ordered against the neighbouring kernel nodes
// when the graph is replayed: `memset(out); atomic-accumulate(out)` replayed from a hipGraph lost part of the sums
// (tools/probes/graph_memset_probe.py: 4052 instead of 4096), which made every captured training step nondeterministic and
// could leave counters non-zero.  Kernel -> kernel order inside a captured stream is respected, so all fills are kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfx {

static __global__ void zero_fill_kernel(uint32_t* __restrict__ p, size_t n_words, int vec) {
    const size_t n4 = vec ? (n_words >> 2) : 0;
    uint4* p4 = reinterpret_cast<uint4*>(p);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p4[i] = uint4{0u, 0u, 0u, 0u};
    for (size_t i = (n4 << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

static __global__ void copy_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n_words) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// `bytes` must be a multiple of 4 (16-byte stores when `p` is 16-byte aligned)
static inline hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    const size_t words = bytes >> 2;
    size_t blocks = ((words >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), words,
                       (int)((reinterpret_cast<uintptr_t>(p) & 15) == 0));
    return hipGetLastError();
}

static inline hipError_t copy_async(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    const size_t words = bytes >> 2;
    size_t blocks = (words + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(dst), reinterpret_cast<const uint32_t*>(src), words);
    return hipGetLastError();
}

}  // namespace mfx
