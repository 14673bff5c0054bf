// Detection decode on device (reference model/layers/utils.py:39-145, model/head/detector_infer.py:77-237,
// model/anno_encoder.py:69-295): sigmoid/clamp -> 3x3 max NMS -> per-class top-K -> merge to K ->
// POI gather (free in NHWC) -> 2D box / dimensions / 4 depths / soft fusion / location / orientation.
// Latency-bound (reads 3*H*W + K*50 values per image); two launches, no host sync.
#include "../../include/monoflex_hip.h"
#include "common.h"
#include "err.h"

namespace mfx {

constexpr int kTopkThreads = 1024;
constexpr float kPi = 3.14159265358979323846f;

__device__ __forceinline__ float sigmoid_clamp(float x) {          // layers/utils.py:39-43
    const float s = 1.f / (1.f + expf(-x));
    return fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
}

// (value desc, index asc) ordering; ties go to the lower flat index
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// Exact k-th-largest selection by radix histograms over LDS-resident keys.
// Among the elements p in [0,n) with `pred(p)`, returns the key of rank `need` (1 = largest) and writes how many
// of the elements EQUAL to that key belong to the top `need` into *take_eq.  keys are < 2^32, examined in
// three digit levels (12 + 10 + 10 bits).  One workgroup; hist has 4096 ints; sh[0..1] is scratch.
// NLEV = 3: digits of 12 + 10 + 10 bits (few passes over many keys); NLEV = 4: four 8-bit digits (small key sets: the
// per-level cost is then the 256-bin scan, not the 4096-bin one).
template <int NLEV = 3, class KeyF, class PredF>
__device__ uint32_t select_kth(KeyF key, PredF pred, int n, int need, int* hist, int* sh, int* take_eq) {
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int shifts[4] = {NLEV == 3 ? 20 : 24, NLEV == 3 ? 10 : 16, NLEV == 3 ? 0 : 8, 0};
    const int nbits[4] = {NLEV == 3 ? 12 : 8, NLEV == 3 ? 10 : 8, NLEV == 3 ? 10 : 8, 8};
    uint32_t prefix = 0;
    for (int lev = 0; lev < NLEV; ++lev) {
        const int nb = 1 << nbits[lev], sft = shifts[lev];
        for (int i = tid; i < nb; i += nthreads) hist[i] = 0;
        __syncthreads();
        for (int p = tid; p < n; p += nthreads) {
            if (!pred(p)) continue;
            const uint32_t k = key(p);
            const bool match = lev == 0 || (k >> (sft + nbits[lev])) == prefix;
            if (match) atomicAdd(&hist[(k >> sft) & (nb - 1)], 1);
        }
        __syncthreads();
        if (tid < 64) {                                   // one wave walks the bins from the top
            const int per = nb / 64;
            int local = 0;
            for (int j = 0; j < per; ++j) local += hist[lane * per + j];
            int incl = local;                             // inclusive suffix sum over lanes (lane 63 = highest bins)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_down(incl, off);
                if (lane + off < 64) incl += v;
            }
            const int excl = incl - local;
            if (incl >= need && excl < need) {            // exactly one lane owns the bin
                int c = excl, bsel = lane * per;
                for (int j = per - 1; j >= 0; --j) {
                    const int h = hist[lane * per + j];
                    if (c + h >= need) { bsel = lane * per + j; break; }
                    c += h;
                }
                sh[0] = bsel; sh[1] = need - c;           // remaining rank inside the chosen bin
            }
        }
        __syncthreads();
        prefix = (prefix << nbits[lev]) | (uint32_t)sh[0];
        need = sh[1];
        __syncthreads();
    }
    *take_eq = need;
    return prefix;
}

// One workgroup per (class, image): sigmoid/clamp + 3x3 NMS into LDS, exact top-K by radix selection
// (ties toward the lower flat index), rank-sort of the K winners.
__global__ __launch_bounds__(kTopkThreads) void decode_topk_kernel(const float* hmap, long b_stride, long c_stride, long p_stride,
                                                                    int H, int W, int K, float* scores, int* index) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* nm = reinterpret_cast<float*>(smem_raw);                // [H*W]: logits, then heat after NMS
    __shared__ int hist[4096];
    __shared__ int sh[2];
    __shared__ int cnt;
    __shared__ float cand_v[256];
    __shared__ int cand_i[256];
    const int cls = blockIdx.x, b = blockIdx.y, ncls = gridDim.x, HW = H * W;
    const int tid = threadIdx.x;
    const float* src = hmap + (size_t)b * b_stride + (size_t)cls * c_stride;
    // stage the class logits in LDS: one global read per pixel (contiguous when the map is planar)
    for (int p = tid; p < HW; p += kTopkThreads) nm[p] = src[(size_t)p * p_stride];
    __syncthreads();
    // nms_hm (layers/utils.py:45-58): keep where the 3x3 max-pool equals the value (plateaus survive).
    // sigmoid/clamp is monotone: compare logits, evaluate the heat only to decide ties after rounding/clamping.
    unsigned long long keepmask = 0ull;                           // this thread's pixels p = tid + i*1024, i < 64
    int i = 0;
    for (int p = tid; p < HW; p += kTopkThreads, ++i) {
        const int y = p / W, x = p - y * W;
        const float xv = nm[p];
        float xm = -3.0e38f;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W || (dx == 0 && dy == 0)) continue;
                xm = fmaxf(xm, nm[yy * W + xx]);
            }
        }
        const bool keep = xv >= xm || sigmoid_clamp(xv) == sigmoid_clamp(xm);
        keepmask |= (unsigned long long)(keep ? 1 : 0) << i;
    }
    __syncthreads();
    i = 0;
    for (int p = tid; p < HW; p += kTopkThreads, ++i) nm[p] = ((keepmask >> i) & 1ull) ? sigmoid_clamp(nm[p]) : 0.f;
    if (tid == 0) { cnt = 0; sh[0] = 0; }
    __syncthreads();
    // ~90 % of the map is exactly 0 after the NMS: histogramming those would serialise ~27 k LDS atomics on one bin per
    // radix level (that was 80 % of this kernel's time).  Count the survivors; when at least K exist the zeros cannot be
    // among the top K and are left out of the selection.
    {
        int nz = 0;
        for (int p = tid; p < HW; p += kTopkThreads) nz += nm[p] != 0.f ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nz += __shfl_xor(nz, off);
        if ((tid & 63) == 0 && nz) atomicAdd(&sh[0], nz);
    }
    __syncthreads();
    const bool skip_zero = sh[0] >= K;
    __syncthreads();
    // K-th largest heat T; take_eq = how many of the elements equal to T are in the top K
    int take_eq = 0;
    const uint32_t T = select_kth([&](int p) { return __float_as_uint(nm[p]); }, [&](int p) { return !skip_zero || nm[p] != 0.f; },
                                  HW, K, hist, sh, &take_eq);
    // among the elements equal to T keep the take_eq lowest flat indices: k-th largest of (HW-1-p)
    int dummy = 0;
    const uint32_t I = select_kth([&](int p) { return (uint32_t)(HW - 1 - p); },
                                  [&](int p) { return __float_as_uint(nm[p]) == T; }, HW, take_eq, hist, sh, &dummy);
    for (int p = tid; p < HW; p += kTopkThreads) {
        const uint32_t kb = __float_as_uint(nm[p]);
        if (kb > T || (kb == T && (uint32_t)(HW - 1 - p) >= I)) {
            const int slot = atomicAdd(&cnt, 1);
            if (slot < 256) { cand_v[slot] = nm[p]; cand_i[slot] = p; }
        }
    }
    __syncthreads();
    const int n = cnt < K ? cnt : K;                      // == K by construction
    if (tid < n) {
        const float v = cand_v[tid]; const int i = cand_i[tid];
        int rank = 0;
        for (int u = 0; u < n; ++u) rank += better(cand_v[u], cand_i[u], v, i) ? 1 : 0;
        scores[((size_t)b * ncls + cls) * K + rank] = v;
        index[((size_t)b * ncls + cls) * K + rank] = i;
    }
}

// ---- strip-parallel form of stage 1 ---------------------------------------------------------------------------------
// The single-workgroup kernel above keeps one (class, image) map in one CU's LDS: 24 workgroups for B=8, i.e. a 256-CU
// device runs a chain of serial radix passes on 9 % of its CUs (79 us).  Here each map is cut into row strips; every strip
// (its rows + one halo row each side for the NMS) is reduced to its own exact top-K by one small workgroup, and a second
// tiny kernel merges the S*K candidates.  Any element of the global top-K is in the top-K of its strip under the same total
// order (value desc, flat index asc), so the result is identical.
#ifndef MFX_STRIP_THREADS
#define MFX_STRIP_THREADS 1024
#endif
constexpr int kStripThreads = MFX_STRIP_THREADS;

__global__ __launch_bounds__(kStripThreads) void decode_topk_strip_kernel(const float* hmap, long b_stride, long c_stride, long p_stride,
                                                                          int H, int W, int K, int rows_per, float* cand_v, int* cand_i) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int strip = blockIdx.x, S = gridDim.x, cls = blockIdx.y, ncls = gridDim.y, b = blockIdx.z;
    const int r0 = strip * rows_per, r1 = min(H, r0 + rows_per);
    const int ylo = max(r0 - 1, 0), yhi = min(r1 + 1, H);
    const int nloc = (r1 - r0) * W, nstage = (yhi - ylo) * W;
    float* lg = reinterpret_cast<float*>(smem_raw);                 // [(rows_per+2)*W] logits of rows ylo..yhi
    float* nm = lg + (rows_per + 2) * W;                            // [rows_per*W] heat after NMS
    __shared__ int hist[256];
    __shared__ int sh[2];
    __shared__ int cnt;
    __shared__ float cv[256];
    __shared__ int ci[256];
    const int tid = threadIdx.x;
    const float* src = hmap + (size_t)b * b_stride + (size_t)cls * c_stride;
    for (int q = tid; q < nstage; q += kStripThreads) lg[q] = src[(size_t)(ylo * W + q) * p_stride];
    if (tid == 0) { cnt = 0; sh[0] = 0; }
    __syncthreads();
    int nz = 0;
    for (int q = tid; q < nloc; q += kStripThreads) {               // nms_hm (layers/utils.py:45-58), same tie rule as above
        const int y = r0 + q / W, x = q % W;
        const float xv = lg[(y - ylo) * W + x];
        float xm = -3.0e38f;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W || (dx == 0 && dy == 0)) continue;
                xm = fmaxf(xm, lg[(yy - ylo) * W + xx]);
            }
        }
        const bool keep = xv >= xm || sigmoid_clamp(xv) == sigmoid_clamp(xm);
        const float h = keep ? sigmoid_clamp(xv) : 0.f;
        nm[q] = h;
        nz += h != 0.f ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nz += __shfl_xor(nz, off);
    if ((tid & 63) == 0 && nz) atomicAdd(&sh[0], nz);
    __syncthreads();
    const int need = K < nloc ? K : nloc;
    const bool skip_zero = sh[0] >= need;
    float* ov = cand_v + (((size_t)b * ncls + cls) * S + strip) * K;
    int* oi = cand_i + (((size_t)b * ncls + cls) * S + strip) * K;
    // Fast path (the usual case: the 3x3 NMS leaves ~1 pixel in 9, a few hundred per strip): when the survivors alone fill the strip's top-K and fit
    // the list below, they are compacted and every survivor COUNTS the survivors ahead of it under the same total order (value desc, flat index
    // asc: one 64-bit key, heat bits high -- heat > 0, so bit order is value order -- and ~index low).  Its count is its rank: ranks < need are the
    // strip's top-K, already sorted.  Same set as the radix selection below (which stays for dense maps and for strips that must take zeros).
    constexpr int kSurvCap = 512;
    __shared__ unsigned long long surv[kSurvCap];
    const int ns = sh[0];
    if (ns >= need && ns <= kSurvCap && need > 0) {
        for (int q = tid; q < nloc; q += kStripThreads) {
            const float h = nm[q];
            if (h != 0.f) surv[atomicAdd(&cnt, 1)] = ((unsigned long long)__float_as_uint(h) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)q);
        }
        __syncthreads();
        for (int c = tid; c < ns; c += kStripThreads) {
            const unsigned long long mine = surv[c];
            int rank = 0;
#pragma unroll 8
            for (int u = 0; u < ns; ++u) rank += surv[u] > mine ? 1 : 0;            // (every lane reads the same entry: one broadcast LDS read)
            if (rank < need) { ov[rank] = __uint_as_float((uint32_t)(mine >> 32)); oi[rank] = r0 * W + (int)(0xffffffffu - (uint32_t)mine); }
        }
        for (int t = need + tid; t < K; t += kStripThreads) { ov[t] = -1.f; oi[t] = 0x7fffffff; }
        return;
    }
    __syncthreads();
    int take_eq = 0, dummy = 0;
    const uint32_t T = select_kth<4>([&](int q) { return __float_as_uint(nm[q]); }, [&](int q) { return !skip_zero || nm[q] != 0.f; },
                                     nloc, need, hist, sh, &take_eq);
    const uint32_t I = select_kth<4>([&](int q) { return (uint32_t)(nloc - 1 - q); },
                                  [&](int q) { return __float_as_uint(nm[q]) == T; }, nloc, take_eq, hist, sh, &dummy);
    for (int q = tid; q < nloc; q += kStripThreads) {
        const uint32_t kb = __float_as_uint(nm[q]);
        if (kb > T || (kb == T && (uint32_t)(nloc - 1 - q) >= I)) {
            const int slot = atomicAdd(&cnt, 1);
            if (slot < 256) { cv[slot] = nm[q]; ci[slot] = r0 * W + q; }
        }
    }
    __syncthreads();
    const int n = cnt < need ? cnt : need;                          // == need by construction
    for (int t = tid; t < K; t += kStripThreads) {
        if (t < n) { ov[t] = cv[t]; oi[t] = ci[t]; }
        else { ov[t] = -1.f; oi[t] = 0x7fffffff; }                   // padding: below every heat value (heat >= 0)
    }
}

// Rank of every candidate among the S*K candidates of its map.  One wave per candidate: the lanes split the comparison
// range (independent LDS reads, a few per lane) and add their counts; a thread-per-candidate loop over all n candidates is a
// chain of ~2n dependent LDS round trips and took longer than the whole single-workgroup selection.
__global__ __launch_bounds__(512) void decode_topk_merge_kernel(const float* cand_v, const int* cand_i, int n, int K, float* scores, int* index) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* v = reinterpret_cast<float*>(smem_raw);
    int* ix = reinterpret_cast<int*>(v + n);
    const size_t map = (size_t)blockIdx.y * gridDim.x + blockIdx.x;   // (image, class)
    for (int t = threadIdx.x; t < n; t += blockDim.x) { v[t] = cand_v[map * n + t]; ix[t] = cand_i[map * n + t]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, nwaves = (blockDim.x >> 6) * gridDim.z, wave = blockIdx.z * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int c = wave; c < n; c += nwaves) {                        // the candidates are dealt to the waves of gridDim.z workgroups
        const float mv = v[c]; const int mi = ix[c];
        int rank = 0;
#pragma unroll 4
        for (int u = lane; u < n; u += 64) rank += better(v[u], ix[u], mv, mi) ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rank += __shfl_xor(rank, off);
        if (lane == 0 && rank < K) { scores[map * K + rank] = mv; index[map * K + rank] = mi; }
    }
}

struct DecodeConst {
    float dim_mean[3][3];     // (l,h,w) per class, config/defaults.py:206-208
    float depth_min, depth_max;
    float down_ratio, eps;
    int depth_mode;           // MFX_DEPTH_* (detector_infer.py:149-198 `output_depth`)
};

// key2channel offsets of runs/monoflex.yaml:27-28
enum { R_2D = 0, R_OFF3D = 4, R_KPT = 6, R_KPT_UNC = 26, R_DIM3D = 29, R_ORI_CLS = 32, R_ORI_OFF = 40, R_DEPTH = 48, R_DEPTH_UNC = 49, R_TOTAL = 50 };

__global__ __launch_bounds__(256) void decode_boxes_kernel(const float* hmap, int ld, int reg_off, const float* scores, const int* index,
                                                           int ncls, int H, int W, int K, const float* calib, const int* pad,
                                                           const int* img_size, float threshold, DecodeConst dc,
                                                           float* det, float* topk, int* valid) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* cs = reinterpret_cast<float*>(smem_raw);          // [ncls*K]
    int* ci = reinterpret_cast<int*>(cs + ncls * K);         // [ncls*K]
    int* slot = ci + ncls * K;                               // [K] position (in the ncls*K list) of rank j
    const int b = blockIdx.x, tid = threadIdx.x, n = ncls * K;
    for (int t = tid; t < n; t += blockDim.x) { cs[t] = scores[(size_t)b * n + t]; ci[t] = index[(size_t)b * n + t]; }
    __syncthreads();
    // select_topk stage 2 (utils.py:88-91): top-K of the concatenated list, ties -> lower position
    for (int t = tid; t < n; t += blockDim.x) {
        const float v = cs[t];
        int rank = 0;
        for (int u = 0; u < n; ++u) rank += better(cs[u], u, v, t) ? 1 : 0;
        if (rank < K) slot[rank] = t;
    }
    __syncthreads();
    if (tid >= K) return;
    const int j = tid, pos = slot[j];
    const int cls = pos / K;                                  // utils.py:91 (integer floor division)
    const float score = cs[pos];
    const int idx = ci[pos];
    const int ys = idx / W, xs = idx - ys * W;                // utils.py:80-81
    const float* r = hmap + ((size_t)b * H * W + idx) * ld + reg_off;   // POI gather: one contiguous NHWC row
    const float fu = calib[b * 6 + 0], fv = calib[b * 6 + 1], cu = calib[b * 6 + 2], cv = calib[b * 6 + 3];
    const float bx = calib[b * 6 + 4], by = calib[b * 6 + 5];
    const float padx = (float)pad[b * 2], pady = (float)pad[b * 2 + 1];
    const float px = (float)xs, py = (float)ys;

    // decode_box2d_fcos (anno_encoder.py:69-86); clamp uses image 0's padded size
    float x1 = (px - fmaxf(r[R_2D + 0], 0.f)) * dc.down_ratio - padx;
    float y1 = (py - fmaxf(r[R_2D + 1], 0.f)) * dc.down_ratio - pady;
    float x2 = (px + fmaxf(r[R_2D + 2], 0.f)) * dc.down_ratio - padx;
    float y2 = (py + fmaxf(r[R_2D + 3], 0.f)) * dc.down_ratio - pady;
    const float wmax = (float)(img_size[0] - 1), hmax = (float)(img_size[1] - 1);
    x1 = fminf(fmaxf(x1, 0.f), wmax); x2 = fminf(fmaxf(x2, 0.f), wmax);
    y1 = fminf(fmaxf(y1, 0.f), hmax); y2 = fminf(fmaxf(y2, 0.f), hmax);

    // decode_dimension (anno_encoder.py:221-243): exp(offset) * mean[cls], order (l,h,w)
    const float dl = expf(r[R_DIM3D + 0]) * dc.dim_mean[cls][0];
    const float dh = expf(r[R_DIM3D + 1]) * dc.dim_mean[cls][1];
    const float dw = expf(r[R_DIM3D + 2]) * dc.dim_mean[cls][2];

    // decode_depth inv_sigmoid (anno_encoder.py:124-140)
    float d0 = 1.f / (1.f / (1.f + expf(-r[R_DEPTH]))) - 1.f;
    d0 = fminf(fmaxf(d0, dc.depth_min), dc.depth_max);
    const float u0 = expf(r[R_DEPTH_UNC]);

    // decode_depth_from_keypoints_batch (anno_encoder.py:187-219); keypoint k = (r[6+2k], r[7+2k])
    auto ky = [&](int k) { return r[R_KPT + 2 * k + 1]; };
    auto kdepth = [&](float dy) { return fu * dh / (fmaxf(dy, 0.f) * dc.down_ratio + dc.eps); };
    float d1 = kdepth(ky(8) - ky(9));
    float d2 = (kdepth(ky(0) - ky(4)) + kdepth(ky(2) - ky(6))) / 2.f;
    float d3 = (kdepth(ky(1) - ky(5)) + kdepth(ky(3) - ky(7))) / 2.f;
    d1 = fminf(fmaxf(d1, dc.depth_min), dc.depth_max);
    d2 = fminf(fmaxf(d2, dc.depth_min), dc.depth_max);
    d3 = fminf(fmaxf(d3, dc.depth_min), dc.depth_max);
    const float u1 = expf(r[R_KPT_UNC + 0]), u2 = expf(r[R_KPT_UNC + 1]), u3 = expf(r[R_KPT_UNC + 2]);

    // which depth leaves the four estimates, and the uncertainty that scales the score with it (detector_infer.py:149-198 `output_depth`)
    float depth, sigma;
    if (dc.depth_mode == MFX_DEPTH_SOFT) {                    // 'soft' (:186-192; runs/monoflex.yaml)
        float w0 = 1.f / u0, w1 = 1.f / u1, w2 = 1.f / u2, w3 = 1.f / u3;
        const float ws = ((w0 + w1) + w2) + w3;
        w0 /= ws; w1 /= ws; w2 /= ws; w3 /= ws;
        depth = ((d0 * w0 + d1 * w1) + d2 * w2) + d3 * w3;
        sigma = ((w0 * u0 + w1 * u1) + w2 * u2) + w3 * u3;
    } else if (dc.depth_mode == MFX_DEPTH_HARD) {             // 'hard' (:180-184): the estimate of the largest weight 1 / u (first of equals, as argmax)
        const float w0 = 1.f / u0, w1 = 1.f / u1, w2 = 1.f / u2, w3 = 1.f / u3;
        depth = d0; float wb = w0;
        if (w1 > wb) { wb = w1; depth = d1; }
        if (w2 > wb) { wb = w2; depth = d2; }
        if (w3 > wb) { wb = w3; depth = d3; }
        sigma = fminf(fminf(u0, u1), fminf(u2, u3));
    } else if (dc.depth_mode == MFX_DEPTH_MEAN) {             // 'mean' (:194-198)
        depth = (((d0 + d1) + d2) + d3) / 4.f; sigma = (((u0 + u1) + u2) + u3) / 4.f;
    } else if (dc.depth_mode == MFX_DEPTH_DIRECT) {           // 'direct' (:149-152)
        depth = d0; sigma = u0;
    } else if (dc.depth_mode == MFX_DEPTH_KEYPOINTS_AVG) {    // 'keypoints_avg' (:155-157)
        depth = ((d1 + d2) + d3) / 3.f; sigma = ((u1 + u2) + u3) / 3.f;
    } else if (dc.depth_mode == MFX_DEPTH_KEYPOINTS_CENTER) { depth = d1; sigma = u1; }   // (:159-161)
    else if (dc.depth_mode == MFX_DEPTH_KEYPOINTS_02) { depth = d2; sigma = u2; }          // (:163-165)
    else { depth = d3; sigma = u3; }                                                         // 'keypoints_13' (:167-169)

    // decode_location_flatten (anno_encoder.py:142-155) + project_image_to_rect (kitti_utils.py:350-369)
    const float u = (px + r[R_OFF3D + 0]) * dc.down_ratio - padx;
    const float v = (py + r[R_OFF3D + 1]) * dc.down_ratio - pady;
    const float X = ((u - cu) * depth) / fu + bx;
    float Y = ((v - cv) * depth) / fv + by;
    const float Z = depth;

    // decode_axes_orientation, multi-bin (anno_encoder.py:245-295)
    int best = 0; float bestp = -1.f;
    for (int i = 0; i < 4; ++i) {
        const float a = r[R_ORI_CLS + 2 * i], c = r[R_ORI_CLS + 2 * i + 1];
        const float m = fmaxf(a, c), e0 = expf(a - m), e1 = expf(c - m);
        const float p1 = e1 / (e0 + e1);
        if (p1 > bestp) { bestp = p1; best = i; }
    }
    const float centers[4] = {0.f, kPi / 2.f, kPi, -kPi / 2.f};
    float alpha = atan2f(r[R_ORI_OFF + 2 * best], r[R_ORI_OFF + 2 * best + 1]) + centers[best];
    float ry = alpha + atan2f(X, Z);
    if (ry > kPi) ry -= 2.f * kPi;
    if (ry < -kPi) ry += 2.f * kPi;
    if (alpha > kPi) alpha -= 2.f * kPi;
    if (alpha < -kPi) alpha += 2.f * kPi;

    Y += dh / 2.f;                                            // detector_infer.py:215
    const float final_score = score * (1.f - fminf(fmaxf(sigma, 0.01f), 1.f));   // :225-227

    float* o = det + ((size_t)b * K + j) * 14;
    o[0] = (float)cls; o[1] = alpha; o[2] = x1; o[3] = y1; o[4] = x2; o[5] = y2;
    o[6] = dh; o[7] = dw; o[8] = dl;                          // roll(-1): (l,h,w) -> (h,w,l)
    o[9] = X; o[10] = Y; o[11] = Z; o[12] = ry; o[13] = final_score;
    float* t = topk + ((size_t)b * K + j) * 5;
    t[0] = score; t[1] = (float)idx; t[2] = (float)cls; t[3] = py; t[4] = px;
    valid[b * K + j] = score >= threshold ? 1 : 0;
}

}  // namespace mfx
using namespace mfx;

int g_opt_topk_merge_z = 16, g_opt_topk_merge_threads = 512;   // options "topk_merge_z" / "topk_merge_threads": workgroups per (class, image) map and their size in the merge
int g_opt_topk_strips = 8;       // row strips per (class, image) map when a workspace is supplied; 1 = single-workgroup kernel

extern "C" size_t mfx_decode_topk_workspace_bytes(int ncls, int B, int K) {
    return (size_t)B * ncls * 16 * K * 8;                           // up to 16 strips of K (value, index) pairs
}

extern "C" int mfx_decode_topk(const float* hmap, long b_stride, long c_stride, long p_stride, int ncls, int B, int H, int W, int K,
                               float* scores, int32_t* index, void* workspace, size_t workspace_bytes, void* stream) {
    if (!hmap || !scores || !index) return mfx_fail(MFX_ERR_ARG, "decode_topk: null pointer");
    if (K < 1 || K > H * W || K > 256) return mfx_fail(MFX_ERR_ARG, "decode_topk: need 1 <= K <= min(H*W, 256)");
    if (B * ncls == 0) return MFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int S = g_opt_topk_strips < 1 ? 1 : (g_opt_topk_strips > 16 ? 16 : g_opt_topk_strips);
    while (S > 1 && ((H + S - 1) / S) * W < K) --S;                 // every strip must be able to hold K candidates ...
    while (S > 1 && (S - 1) * ((H + S - 1) / S) >= H) --S;          // ... and no strip may be empty
    if (workspace && S > 1 && workspace_bytes >= (size_t)B * ncls * S * K * 8) {
        const int rows_per = (H + S - 1) / S;
        float* cand_v = reinterpret_cast<float*>(workspace);
        int* cand_i = reinterpret_cast<int*>(cand_v + (size_t)B * ncls * S * K);
        const size_t smem = (size_t)(2 * rows_per + 2) * W * sizeof(float);
        if (smem <= 48 * 1024) {
            hipLaunchKernelGGL(decode_topk_strip_kernel, dim3(S, ncls, B), dim3(kStripThreads), smem, st,
                               hmap, b_stride, c_stride, p_stride, H, W, K, rows_per, cand_v, cand_i);
            hipLaunchKernelGGL(decode_topk_merge_kernel, dim3(ncls, B, g_opt_topk_merge_z < 1 ? 1 : g_opt_topk_merge_z), dim3(g_opt_topk_merge_threads), (size_t)S * K * 8, st, cand_v, cand_i, S * K, K, scores, index);
            MFX_HIP_CHECK(hipGetLastError());
            return MFX_OK;
        }
    }
    const size_t smem = (size_t)H * W * sizeof(float);
    if (smem > 136 * 1024) return mfx_fail(MFX_ERR_UNSUPPORTED, "decode_topk: heat map larger than LDS (H*W <= 34816)");
    auto k = decode_topk_kernel;
    static size_t attr_smem = 0;
    if (smem > attr_smem) { MFX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); attr_smem = smem; }
    hipLaunchKernelGGL(k, dim3(ncls, B), dim3(kTopkThreads), smem, st, hmap, b_stride, c_stride, p_stride, H, W, K, scores, index);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}

extern "C" int mfx_decode_boxes(const float* hmap, int ld, int reg_off, const float* scores, const int32_t* index,
                                int ncls, int B, int H, int W, int K, const float* calib, const int32_t* pad,
                                const int32_t* img_size, float threshold, float* det, float* topk, int32_t* valid,
                                void* stream) {
    return mfx_decode_boxes_mode(hmap, ld, reg_off, scores, index, ncls, B, H, W, K, calib, pad, img_size, threshold, MFX_DEPTH_SOFT, det, topk, valid, stream);
}

extern "C" int mfx_decode_boxes_mode(const float* hmap, int ld, int reg_off, const float* scores, const int32_t* index,
                                     int ncls, int B, int H, int W, int K, const float* calib, const int32_t* pad,
                                     const int32_t* img_size, float threshold, int depth_mode, float* det, float* topk, int32_t* valid,
                                     void* stream) {
    if (depth_mode < MFX_DEPTH_SOFT || depth_mode > MFX_DEPTH_KEYPOINTS_13) return mfx_fail(MFX_ERR_ARG, "decode_boxes: depth_mode must be one of MFX_DEPTH_*");
    if (!hmap || !scores || !index || !calib || !pad || !img_size || !det || !topk || !valid)
        return mfx_fail(MFX_ERR_ARG, "decode_boxes: null pointer");
    if (K > 256 || ncls != 3) return mfx_fail(MFX_ERR_UNSUPPORTED, "decode_boxes: K <= 256, 3 classes (dimension means)");
    if (B == 0) return MFX_OK;
    DecodeConst dc = {{{3.8840f, 1.5261f, 1.6286f}, {0.8423f, 1.7607f, 0.6602f}, {1.7635f, 1.7372f, 0.5968f}},
                      0.1f, 100.f, 4.f, 1e-3f, depth_mode};
    const size_t smem = (size_t)ncls * K * 8 + (size_t)K * 4;
    hipLaunchKernelGGL(decode_boxes_kernel, dim3(B), dim3(256), smem, reinterpret_cast<hipStream_t>(stream),
                       hmap, ld, reg_off, scores, index, ncls, H, W, K, calib, pad, img_size, threshold, dc, det, topk, valid);
    MFX_HIP_CHECK(hipGetLastError());
    return MFX_OK;
}
