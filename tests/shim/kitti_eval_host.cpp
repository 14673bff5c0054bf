// TEST-ONLY host build of monoflex_amd/csrc/kitti_eval_math.h: the four evaluator steps as plain loops, so the CPU suite can
// check the matching / threshold / accumulation logic against the oracle without a GPU.  Not loaded by the product.
#include <algorithm>
#include <functional>
#include <vector>

#include "../../monoflex_amd/csrc/kitti_eval_math.h"

using namespace mfx::keval;

static void matching_steps(const mfx_kitti_eval_desc& d);

extern "C" void shim_kitti_eval(const mfx_kitti_eval_desc* dp) {
  const mfx_kitti_eval_desc& d = *dp;
  for (int b = 0; b < d.B; ++b) {
    const int ng = d.gt_off[b + 1] - d.gt_off[b], nd = d.dt_off[b + 1] - d.dt_off[b];
    for (int j = 0; j < nd; ++j)
      for (int i = 0; i < ng; ++i) pair_overlaps(d, b, j, i);
  }
  matching_steps(d);
}

// steps 2-4 on overlap matrices supplied by the caller (lets a test feed arbitrary overlaps to the matching logic)
extern "C" void shim_kitti_eval_match_only(const mfx_kitti_eval_desc* dp) { matching_steps(*dp); }

static void matching_steps(const mfx_kitti_eval_desc& d) {
  const int n_comb = d.num_classes * 9 * d.num_k;
  for (long i = 0; i < (long)n_comb * PTS * 4; ++i) d.pr[i] = 0.0;
  for (int i = 0; i < d.num_classes * 3; ++i) d.num_valid_gt[i] = 0;
  for (int b = 0; b < d.B; ++b)
    for (int comb = 0; comb < n_comb; ++comb) {
      match<false>(d, b, comb, 0.0);
      int m, level, metric, k;
      decode_comb(d, comb, m, level, metric, k);
      if (metric == 0 && k == 0)
        for (int i = d.gt_off[b]; i < d.gt_off[b + 1]; ++i) d.num_valid_gt[m * 3 + level] += gt_flag(d.gt + (long)i * REC, d.classes[m], level) == 0;
    }
  for (int comb = 0; comb < n_comb; ++comb) {
    int m, level, metric, k;
    decode_comb(d, comb, m, level, metric, k);
    std::vector<double> s(d.tp_scores + (long)comb * d.n_gt, d.tp_scores + (long)(comb + 1) * d.n_gt);
    std::sort(s.begin(), s.end(), std::greater<double>());
    int n = 0;
    while (n < (int)s.size() && s[n] >= 0) ++n;
    d.num_thresholds[comb] = sample_thresholds(s.data(), n, d.num_valid_gt[m * 3 + level], d.thresholds + (long)comb * PTS);
  }
  for (int b = 0; b < d.B; ++b)
    for (int comb = 0; comb < n_comb; ++comb)
      for (int t = 0; t < d.num_thresholds[comb]; ++t) {
        const Stats s = match<true>(d, b, comb, d.thresholds[(long)comb * PTS + t]);
        double* pr = d.pr + ((long)comb * PTS + t) * 4;
        pr[0] += s.tp; pr[1] += s.fp; pr[2] += s.fn;
        if (s.sim != -1.0) pr[3] += s.sim;
      }
}

// intersection areas of box pairs (cx, cy, dx, dy, angle), for the geometric property test
extern "C" void shim_rotated_intersections(const float* a, const float* b, int n, float* out) {
  for (int i = 0; i < n; ++i) out[i] = rotated_intersection(a + 5 * i, b + 5 * i);
}
