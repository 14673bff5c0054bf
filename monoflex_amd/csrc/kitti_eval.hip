// KITTI AP evaluation on the device (C ABI group 5).  Reference: data/datasets/evaluation/kitti_object_eval_python/eval.py
// and rotate_iou.py.  The arithmetic is in kitti_eval_math.h; this file maps threads onto it:
//   overlaps : one thread per (image, detection, ground truth) pair -- ragged images are located by binary search in pair_off,
//              so no all-pairs-of-a-50-image-part matrix is built (the reference computes ~50x more pairs than it uses);
//   pass 1/2 : one thread per (image, combination[, threshold]) matching problem -- 54 x 41 independent problems per image,
//              each a <= 64-detection greedy assignment held in two 64-bit masks;
//   thresholds: one thread per combination walking its sorted true-positive scores.
// Counts are accumulated with fp64 atomics (integer-valued, hence order-independent); only the orientation-similarity sum
// depends on the atomic order (relative 1e-16).
#include <hip/hip_runtime.h>

#include "err.h"
#include "fill.h"
#include "kitti_eval_math.h"

namespace mfx {

__device__ __forceinline__ int image_of(const int64_t* off, int B, long idx) {   // largest b with off[b] <= idx
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= idx) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256) keval_overlaps_kernel(mfx_kitti_eval_desc d) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= d.n_pairs) return;
  const int b = image_of(d.pair_off, d.B, idx);
  const int ng = d.gt_off[b + 1] - d.gt_off[b];
  const long local = idx - d.pair_off[b];
  keval::pair_overlaps(d, b, (int)(local / ng), (int)(local % ng));
}

__global__ void __launch_bounds__(256) keval_pass1_kernel(mfx_kitti_eval_desc d, int n_comb) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)d.B * n_comb) return;
  const int comb = (int)(idx % n_comb), b = (int)(idx / n_comb);
  keval::match<false>(d, b, comb, 0.0);
  int m, level, metric, k;
  keval::decode_comb(d, comb, m, level, metric, k);
  if (metric == 0 && k == 0) {                             // one thread per (image, class, level) counts the valid ground truths
    int nv = 0;
    for (int i = d.gt_off[b]; i < d.gt_off[b + 1]; ++i) nv += keval::gt_flag(d.gt + (long)i * keval::REC, d.classes[m], level) == 0;
    if (nv) atomicAdd(d.num_valid_gt + m * 3 + level, nv);
  }
}

__global__ void __launch_bounds__(64) keval_thresholds_kernel(mfx_kitti_eval_desc d, const double* sorted, int n_comb) {
  const int comb = blockIdx.x * blockDim.x + threadIdx.x;
  if (comb >= n_comb) return;
  int m, level, metric, k;
  keval::decode_comb(d, comb, m, level, metric, k);
  const double* s = sorted + (long)comb * d.n_gt;
  int lo = 0, hi = d.n_gt;                                 // descending, "no match" entries are -inf: count the real scores
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (s[mid] > keval::NO_MATCH) lo = mid + 1; else hi = mid; }
  d.num_thresholds[comb] = keval::sample_thresholds(s, lo, d.num_valid_gt[m * 3 + level], d.thresholds + (long)comb * keval::PTS);
}

__global__ void __launch_bounds__(256) keval_pass2_kernel(mfx_kitti_eval_desc d, int n_comb) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)d.B * n_comb * keval::PTS) return;
  const int t = (int)(idx % keval::PTS);
  const int comb = (int)((idx / keval::PTS) % n_comb), b = (int)(idx / ((long)keval::PTS * n_comb));
  if (t >= d.num_thresholds[comb]) return;
  const keval::Stats s = keval::match<true>(d, b, comb, d.thresholds[(long)comb * keval::PTS + t]);
  double* pr = d.pr + ((long)comb * keval::PTS + t) * 4;
  if (s.tp) atomicAdd(pr + 0, (double)s.tp);
  if (s.fp) atomicAdd(pr + 1, (double)s.fp);
  if (s.fn) atomicAdd(pr + 2, (double)s.fn);
  if (s.sim != -1.0 && s.sim != 0.0) atomicAdd(pr + 3, s.sim);
}

static int check_desc(const mfx_kitti_eval_desc* d, const char* who) {
  if (!d) return mfx_fail(MFX_ERR_ARG, "kitti_eval: null descriptor");
  if (d->B <= 0 || d->num_classes <= 0 || d->num_k <= 0 || d->n_gt < 0 || d->n_dt < 0 || d->n_pairs < 0)
    return mfx_fail(MFX_ERR_ARG, "kitti_eval: bad sizes");
  if (!d->gt_off || !d->dt_off || !d->pair_off || !d->classes || !d->min_overlaps || !d->overlaps || !d->tp_scores ||
      !d->num_valid_gt || !d->thresholds || !d->num_thresholds || !d->pr || (d->n_gt && !d->gt) || (d->n_dt && !d->dt))
    return mfx_fail(MFX_ERR_ARG, "kitti_eval: every pointer of the descriptor must be set");
  (void)who;
  return MFX_OK;
}

static inline unsigned blocks(long n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace mfx
using namespace mfx;

extern "C" int mfx_kitti_eval_overlaps(const mfx_kitti_eval_desc* d, void* stream) {
  if (int rc = check_desc(d, "overlaps")) return rc;
  if (d->n_pairs == 0) return MFX_OK;
  hipLaunchKernelGGL(keval_overlaps_kernel, dim3(blocks(d->n_pairs, 256)), dim3(256), 0, (hipStream_t)stream, *d);
  MFX_HIP_CHECK(hipGetLastError());
  return MFX_OK;
}

extern "C" int mfx_kitti_eval_match_pass1(const mfx_kitti_eval_desc* d, void* stream) {
  if (int rc = check_desc(d, "pass1")) return rc;
  const int n_comb = d->num_classes * 9 * d->num_k;
  hipStream_t st = (hipStream_t)stream;
  MFX_HIP_CHECK(mfx::zero_async(d->pr, sizeof(double) * n_comb * keval::PTS * 4, st));
  MFX_HIP_CHECK(mfx::zero_async(d->num_valid_gt, sizeof(int32_t) * d->num_classes * 3, st));
  hipLaunchKernelGGL(keval_pass1_kernel, dim3(blocks((long)d->B * n_comb, 256)), dim3(256), 0, st, *d, n_comb);
  MFX_HIP_CHECK(hipGetLastError());
  return MFX_OK;
}

extern "C" int mfx_kitti_eval_thresholds(const mfx_kitti_eval_desc* d, const double* sorted_scores, void* stream) {
  if (int rc = check_desc(d, "thresholds")) return rc;
  if (d->n_gt && !sorted_scores) return mfx_fail(MFX_ERR_ARG, "kitti_eval: sorted_scores is null");
  const int n_comb = d->num_classes * 9 * d->num_k;
  hipLaunchKernelGGL(keval_thresholds_kernel, dim3(blocks(n_comb, 64)), dim3(64), 0, (hipStream_t)stream, *d, sorted_scores, n_comb);
  MFX_HIP_CHECK(hipGetLastError());
  return MFX_OK;
}

extern "C" int mfx_kitti_eval_match_pass2(const mfx_kitti_eval_desc* d, void* stream) {
  if (int rc = check_desc(d, "pass2")) return rc;
  const int n_comb = d->num_classes * 9 * d->num_k;
  hipLaunchKernelGGL(keval_pass2_kernel, dim3(blocks((long)d->B * n_comb * keval::PTS, 256)), dim3(256), 0, (hipStream_t)stream, *d, n_comb);
  MFX_HIP_CHECK(hipGetLastError());
  return MFX_OK;
}
