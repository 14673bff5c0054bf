// Shared device helpers for the MonoFlex gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// ---- element types: float (parity mode) and bf16 (perf mode) -------------------------------
struct bf16_t { uint16_t v; };

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even, NaN preserved
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int ELEMS = 4;            // elements per 16-byte chunk
    static constexpr int DT = 0;
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float round(float v) { return v; }            // the value as it will be stored
    // unpack a 16-byte chunk to floats / pack floats to a chunk
    __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
        f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y);
        f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        u32x4 c; c.x = __float_as_uint(f[0]); c.y = __float_as_uint(f[1]);
        c.z = __float_as_uint(f[2]); c.w = __float_as_uint(f[3]); return c;
    }
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int ELEMS = 8;
    static constexpr int DT = 1;
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf2f(p->v); }
    __device__ static __forceinline__ void store(bf16_t* p, float v) { p->v = __builtin_bit_cast(uint16_t, (__bf16)v); }
    __device__ static __forceinline__ float round(float v) { return __uint_as_float((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)v) << 16); }
    __device__ static __forceinline__ uint32_t pack2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, bf16x2)); }
    __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
        f[0] = __uint_as_float(c.x << 16); f[1] = __uint_as_float(c.x & 0xffff0000u);
        f[2] = __uint_as_float(c.y << 16); f[3] = __uint_as_float(c.y & 0xffff0000u);
        f[4] = __uint_as_float(c.z << 16); f[5] = __uint_as_float(c.z & 0xffff0000u);
        f[6] = __uint_as_float(c.w << 16); f[7] = __uint_as_float(c.w & 0xffff0000u);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        // v_cvt_pk_bf16_f32: hardware round-to-nearest-even, two floats per instruction
        u32x4 c;
        c.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){f[0], f[1]}, bf16x2));
        c.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){f[2], f[3]}, bf16x2));
        c.z = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){f[4], f[5]}, bf16x2));
        c.w = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){f[6], f[7]}, bf16x2));
        return c;
    }
};

// IEEE fp16 activations (inference perf mode "fp16"): the same MFMA rate as bf16 with three more mantissa bits -- the forward pass stays
// 8x closer to the fp32 reference (profiles/r03_bf16_ablation.md) -- at fp16's range (65504), which post-BN activations of this network
// never approach.  Training in fp16 runs under dynamic loss scaling (engine/trainer.py); bf16 stays the default there.
struct half_t { uint16_t v; };
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <> struct ElemTraits<half_t> {
    static constexpr int ELEMS = 8;
    static constexpr int DT = 2;
    __device__ static __forceinline__ float load(const half_t* p) { return (float)__builtin_bit_cast(_Float16, p->v); }
    __device__ static __forceinline__ void store(half_t* p, float v) { p->v = __builtin_bit_cast(uint16_t, (_Float16)v); }
    __device__ static __forceinline__ float round(float v) { return (float)(_Float16)v; }
    __device__ static __forceinline__ uint32_t pack2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, f16x2)); }
    __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t d = c[i];                          // (copied out first: bit_cast of a vector-element lvalue is miscompiled by this clang)
            const f16x2 h = __builtin_bit_cast(f16x2, d);
            f[2 * i] = (float)h[0]; f[2 * i + 1] = (float)h[1];
        }
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        u32x4 c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f16x2 h = __builtin_convertvector((f32x2){f[2 * i], f[2 * i + 1]}, f16x2);      // round-to-nearest-even
            c[i] = __builtin_bit_cast(uint32_t, h);
        }
        return c;
    }
};

// Split-precision element ("f16x2", parity-grade perf mode): the value lives in HBM as plain fp32 (every non-GEMM kernel is the fp32
// one), and becomes an MFMA operand as a PAIR of IEEE halves hi = fp16(x), lo = fp16(x - hi): x = hi + lo to ~22 mantissa bits
// (fp16 subnormals are honoured by v_mfma_f32_16x16x32_f16 on gfx950: tools/probes/split_probe.hip).  A 16-byte operand chunk holds
// the same 4 elements as the fp32 chunk it replaces -- [h0 h1 h2 h3 | l0 l1 l2 l3] -- so every tile, patch and fragment layout of the
// fp32 instantiation carries over byte for byte; the conversion happens once, where an operand is written to LDS (activations) or
// packed on the host (weights).  One K chunk is two fp16 MFMAs: a.b and a.swap(b), swap = exchange of the hi and lo halves of the
// lane's chunk (a register renaming) -> hi.hi + lo.lo + hi.lo + lo.hi = the full product, fp32 accumulate: 32 matrix-pipe cycles per
// 16 K elements against 128 for v_mfma_f32_16x16x4_f32.
struct f32s_t { float v; };
template <> struct ElemTraits<f32s_t> {
    static constexpr int ELEMS = 4;
    static constexpr int DT = 3;
    __device__ static __forceinline__ float load(const f32s_t* p) { return p->v; }
    __device__ static __forceinline__ void store(f32s_t* p, float v) { p->v = v; }
    __device__ static __forceinline__ float round(float v) { return v; }
    __device__ static __forceinline__ void unpack(const u32x4& c, float* f) { ElemTraits<float>::unpack(c, f); }
    __device__ static __forceinline__ u32x4 pack(const float* f) { return ElemTraits<float>::pack(f); }
};

// value chunk (as loaded from HBM / produced in registers) -> MFMA operand chunk (as written to LDS / fed to the matrix core)
template <typename T> __device__ __forceinline__ u32x4 lds_operand(const u32x4& c) { return c; }
// Range sentinel of the split-precision mode: hi = fp16(x) overflows for |x| > 65504 (and a NaN / Inf stays one).  Every activation becomes an MFMA
// operand through lds_operand<f32s_t>, so the check lives here: one max chain per chunk and a branch that is never taken on a healthy network.
// Device globals are per translation unit without relocatable device code: each TU that converts operands has its own flag and registers an accessor
// (MFX_RANGE_FLAG_ACCESSOR); mfx_f16x2_range_check() in capi.hip ORs them.
static __device__ __attribute__((unused)) unsigned int mfx_tu_range_flag = 0;
#define MFX_RANGE_FLAG_ACCESSOR(tag)                                                                                   \
    int mfx_range_flag_##tag(int reset) {                                                                              \
        unsigned int v = 0, z = 0;                                                                                     \
        if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(mfx::mfx_tu_range_flag), sizeof(v)) != hipSuccess) return -1;           \
        if (reset && v && hipMemcpyToSymbol(HIP_SYMBOL(mfx::mfx_tu_range_flag), &z, sizeof(z)) != hipSuccess) return -1; \
        return (int)v;                                                                                                 \
    }
template <> __device__ __forceinline__ u32x4 lds_operand<f32s_t>(const u32x4& c) {
    const float x0 = __uint_as_float(c.x), x1 = __uint_as_float(c.y), x2 = __uint_as_float(c.z), x3 = __uint_as_float(c.w);
#ifndef MFX_NO_RANGE_CHECK
    // (on the bit patterns: |x| as an unsigned integer orders like the float and puts Inf / NaN above every finite value -- fmaxf would drop a NaN)
    if (max(max(c.x & 0x7fffffffu, c.y & 0x7fffffffu), max(c.z & 0x7fffffffu, c.w & 0x7fffffffu)) > 0x477fe000u) atomicOr(&mfx_tu_range_flag, 1u);
#endif
    const f16x2 h01 = __builtin_convertvector((f32x2){x0, x1}, f16x2), h23 = __builtin_convertvector((f32x2){x2, x3}, f16x2);
    const f16x2 l01 = __builtin_convertvector((f32x2){x0 - (float)h01[0], x1 - (float)h01[1]}, f16x2);
    const f16x2 l23 = __builtin_convertvector((f32x2){x2 - (float)h23[0], x3 - (float)h23[1]}, f16x2);
    u32x4 r;
    r.x = __builtin_bit_cast(uint32_t, h01); r.y = __builtin_bit_cast(uint32_t, h23);
    r.z = __builtin_bit_cast(uint32_t, l01); r.w = __builtin_bit_cast(uint32_t, l23);
    return r;
}

// ---- MFMA step over one 16-byte K-chunk per lane -------------------------------------------
// A fragment: lane l holds row (l&15), k-group (l>>4): 8 bf16 or 4 f32 consecutive in K.
// D layout (both dtypes): col = lane&15, row = (lane>>4)*4 + reg.
template <typename T> __device__ __forceinline__ void mma_chunk(const u32x4& a, const u32x4& b, f32x4& acc);
template <> __device__ __forceinline__ void mma_chunk<bf16_t>(const u32x4& a, const u32x4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                  acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_chunk<half_t>(const u32x4& a, const u32x4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
// f32: the lane's 4 consecutive k feed 4 MFMAs (element j of every lane forms one K=4 step);
// the k order inside the 16-wide block is permuted identically for A and B, so the sum is the same.
template <> __device__ __forceinline__ void mma_chunk<float>(const u32x4& a, const u32x4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

template <> __device__ __forceinline__ void mma_chunk<f32s_t>(const u32x4& a, const u32x4& b, f32x4& acc) {
    const u32x4 bs = {b.z, b.w, b.x, b.y};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bs), acc, 0, 0, 0);
}

// activation codes shared with the host (include/monoflex_hip.h)
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_DCN_OFFMASK = 3 };

__device__ __forceinline__ float apply_act(float v, int act, int n) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
    if (act == ACT_DCN_OFFMASK) return (n >= 18 && n < 27) ? 1.f / (1.f + expf(-v)) : v;
    return v;
}

// activation over one output chunk of OE channels starting at channel gn.  The DCN offset/mask conv needs a sigmoid on
// channels 18..26 only: chunks outside that range skip the exp entirely (the epilogue is serial per wave, so 32 exp
// per pixel cost as much as the whole K loop of a 64 -> 27 conv).
template <int OE> __device__ __forceinline__ void apply_act_chunk(float (&v)[OE], int act, int gn) {
    // one wave-uniform branch per chunk, then straight-line VALU.  (r05: written as a per-element call of apply_act the fully unrolled
    // loop compiled to a scalar compare-and-branch ladder PER ELEMENT -- ~10 instructions per value, the epilogue of a 16-bit conv as long
    // as its K loop; same arithmetic, same bits.)
    if (act == ACT_NONE) return;
    if (act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < OE; ++e) v[e] = fmaxf(v[e], 0.f);
        return;
    }
    if (act == ACT_LEAKY) {
#pragma unroll
        for (int e = 0; e < OE; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
        return;
    }
    if (act == ACT_DCN_OFFMASK) {
        if (gn + OE <= 18 || gn >= 27) return;
#pragma unroll
        for (int e = 0; e < OE; ++e)
            if (gn + e >= 18 && gn + e < 27) v[e] = 1.f / (1.f + __expf(-v[e]));
    }
}

// XCD-aware tile order: block b runs on XCD b%8 (observed, speed only); give every XCD a contiguous
// range of tile ids so neighbouring tiles (shared halos / shared A rows) meet in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace mfx
