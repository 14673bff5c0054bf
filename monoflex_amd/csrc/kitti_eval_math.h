// Per-pair / per-problem arithmetic of the KITTI AP evaluator (C ABI group 5, mfx_kitti_eval_*).
//
// Reference: data/datasets/evaluation/kitti_object_eval_python/eval.py:8-326 and rotate_iou.py:17-268.  Like
// kitti_encode_math.h these are plain functions of (image, detection, ground truth) or (image, combination, threshold) over
// the arrays of mfx_kitti_eval_desc; kitti_eval.hip maps GPU threads onto them and tests/ can compile them for the host.
// The rotated-rectangle intersection is float32 like the reference's CUDA kernel; everything else is float64.
#pragma once
#include <cmath>
#include <cstdint>

#include "../../include/monoflex_hip.h"

#ifndef MFX_HD
#ifdef __HIPCC__
#define MFX_HD __host__ __device__ inline
#else
#define MFX_HD inline
#endif
#endif

namespace mfx {
namespace keval {

constexpr int REC = MFX_EVAL_REC, PTS = MFX_EVAL_PTS;
enum { F_CODE = 0, F_TRUNC, F_OCC, F_ALPHA, F_X1, F_Y1, F_X2, F_Y2, F_L, F_H, F_W, F_X, F_Y, F_Z, F_RY, F_SCORE };
enum { CODE_PEDESTRIAN = 1, CODE_CAR = 0, CODE_VAN = 3, CODE_PERSON_SITTING = 4, CODE_DONTCARE = 6 };

// ---- overlaps -------------------------------------------------------------------------------------------------------------
// eval.py:83-110; a, q point at x1,y1,x2,y2. criterion -1: IoU, 0: / area(a), 1: / area(q)
MFX_HD double bbox_overlap(const double* a, const double* q, int criterion) {
  const double iw = fmin(a[2], q[2]) - fmax(a[0], q[0]);
  if (!(iw > 0)) return 0.0;
  const double ih = fmin(a[3], q[3]) - fmax(a[1], q[1]);
  if (!(ih > 0)) return 0.0;
  const double aa = (a[2] - a[0]) * (a[3] - a[1]), qa = (q[2] - q[0]) * (q[3] - q[1]);
  const double ua = criterion == -1 ? aa + qa - iw * ih : criterion == 0 ? aa : criterion == 1 ? qa : 1.0;
  return iw * ih / ua;
}

MFX_HD void corners(const float* r, float* c) {          // rotate_iou.py:205-228 (cx, cy, dx, dy, angle)
  const float ca = cosf(r[4]), sa = sinf(r[4]);
  const float xs[4] = {-r[2] / 2, -r[2] / 2, r[2] / 2, r[2] / 2}, ys[4] = {-r[3] / 2, r[3] / 2, r[3] / 2, -r[3] / 2};
  for (int i = 0; i < 4; ++i) {
    c[2 * i] = ca * xs[i] + sa * ys[i] + r[0];
    c[2 * i + 1] = -sa * xs[i] + ca * ys[i] + r[1];
  }
}

MFX_HD bool inside(float px, float py, const float* q) {  // rotate_iou.py:166-182
  const float ab0 = q[2] - q[0], ab1 = q[3] - q[1], ad0 = q[6] - q[0], ad1 = q[7] - q[1];
  const float ap0 = px - q[0], ap1 = py - q[1];
  const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
  const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
  return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

MFX_HD bool edge_cross(const float* p1, const float* p2, int i, int j, float* out) {   // rotate_iou.py:77-116
  const float* A = p1 + 2 * i; const float* B = p1 + 2 * ((i + 1) & 3);
  const float* C = p2 + 2 * j; const float* D = p2 + 2 * ((j + 1) & 3);
  const float BA0 = B[0] - A[0], BA1 = B[1] - A[1], DA0 = D[0] - A[0], CA0 = C[0] - A[0], DA1 = D[1] - A[1], CA1 = C[1] - A[1];
  const bool acd = DA1 * CA0 > CA1 * DA0;
  const bool bcd = (D[1] - B[1]) * (C[0] - B[0]) > (C[1] - B[1]) * (D[0] - B[0]);
  if (acd == bcd) return false;
  const bool abc = CA1 * BA0 > BA1 * CA0, abd = DA1 * BA0 > BA1 * DA0;
  if (abc == abd) return false;
  const float DC0 = D[0] - C[0], DC1 = D[1] - C[1];
  const float ABBA = A[0] * B[1] - B[0] * A[1], CDDC = C[0] * D[1] - D[0] * C[1];
  const float DH = BA1 * DC0 - BA0 * DC1;
  out[0] = (ABBA * DC0 - BA0 * CDDC) / DH;
  out[1] = (ABBA * DC1 - BA1 * CDDC) / DH;
  return true;
}

// Intersection area of two rotated rectangles (rotate_iou.py:185-247): vertices = corners of one inside the other + edge
// crossings, ordered around their centroid by the reference's monotone angle key, summed as a triangle fan.
MFX_HD float rotated_intersection(const float* r1, const float* r2) {
  float p1[8], p2[8], pts[16], key[8];
  corners(r1, p1); corners(r2, p2);
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    if (inside(p1[2 * i], p1[2 * i + 1], p2)) { pts[2 * n] = p1[2 * i]; pts[2 * n + 1] = p1[2 * i + 1]; ++n; }
    if (inside(p2[2 * i], p2[2 * i + 1], p1)) { pts[2 * n] = p2[2 * i]; pts[2 * n + 1] = p2[2 * i + 1]; ++n; }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float t[2];
      if (n < 8 && edge_cross(p1, p2, i, j, t)) { pts[2 * n] = t[0]; pts[2 * n + 1] = t[1]; ++n; }
    }
  if (n < 3) return 0.f;
  float cx = 0.f, cy = 0.f;
  for (int i = 0; i < n; ++i) { cx += pts[2 * i]; cy += pts[2 * i + 1]; }
  cx /= n; cy /= n;
  for (int i = 0; i < n; ++i) {
    float vx = pts[2 * i] - cx, vy = pts[2 * i + 1] - cy;
    const float dd = sqrtf(vx * vx + vy * vy);
    vx /= dd; vy /= dd;
    key[i] = vy < 0 ? -2 - vx : vx;
  }
  for (int i = 1; i < n; ++i) {                            // insertion sort by ascending key
    if (key[i - 1] > key[i]) {
      const float t = key[i], tx = pts[2 * i], ty = pts[2 * i + 1];
      int j = i;
      while (j > 0 && key[j - 1] > t) { key[j] = key[j - 1]; pts[2 * j] = pts[2 * j - 2]; pts[2 * j + 1] = pts[2 * j - 1]; --j; }
      key[j] = t; pts[2 * j] = tx; pts[2 * j + 1] = ty;
    }
  }
  float area = 0.f;
  for (int i = 0; i < n - 2; ++i) {
    const float* b = pts + 2 * i + 2; const float* c = pts + 2 * i + 4;
    area += fabsf(((pts[0] - c[0]) * (b[1] - c[1]) - (pts[1] - c[1]) * (b[0] - c[0])) / 2.0f);
  }
  return area;
}

// The three overlaps of (detection j, ground truth i) of image b (eval.py:329-401, 118-153).
MFX_HD void pair_overlaps(const mfx_kitti_eval_desc& d, int b, int j, int i) {
  const int ng = d.gt_off[b + 1] - d.gt_off[b];
  const double* D = d.dt + (long)(d.dt_off[b] + j) * REC;
  const double* G = d.gt + (long)(d.gt_off[b] + i) * REC;
  const long at = d.pair_off[b] + (long)j * ng + i;
  d.overlaps[at] = bbox_overlap(D + F_X1, G + F_X1, -1);
  const float rd[5] = {(float)D[F_X], (float)D[F_Z], (float)D[F_L], (float)D[F_W], (float)D[F_RY]};
  const float rg[5] = {(float)G[F_X], (float)G[F_Z], (float)G[F_L], (float)G[F_W], (float)G[F_RY]};
  const float inter = rotated_intersection(rg, rd);        // the reference's kernel evaluates (query = gt, box = dt)
  const float a1 = rg[2] * rg[3], a2 = rd[2] * rd[3];
  d.overlaps[d.n_pairs + at] = (double)(inter / (a1 + a2 - inter));
  float iou3 = 0.f;
  if (inter > 0) {
    const double ih = fmin(D[F_Y], G[F_Y]) - fmax(D[F_Y] - D[F_H], G[F_Y] - G[F_H]);
    if (ih > 0) {
      const double v1 = D[F_L] * D[F_H] * D[F_W], v2 = G[F_L] * G[F_H] * G[F_W];
      const double vol = ih * (double)inter;
      iou3 = (float)(vol / (v1 + v2 - vol));
    }
  }
  d.overlaps[2 * d.n_pairs + at] = (double)iou3;
}

// ---- ignore rules (eval.py:27-80) ------------------------------------------------------------------------------------------
MFX_HD int gt_flag(const double* G, int cls, int level) {
  const double min_h[3] = {40, 25, 25}, max_trunc[3] = {0.15, 0.3, 0.5};
  const int max_occ[3] = {0, 1, 2};
  const int code = (int)G[F_CODE];
  const int kind = code == cls ? 1 : ((cls == CODE_PEDESTRIAN && code == CODE_PERSON_SITTING) || (cls == CODE_CAR && code == CODE_VAN)) ? 0 : -1;
  const bool hard = G[F_OCC] > max_occ[level] || G[F_TRUNC] > max_trunc[level] || (G[F_Y2] - G[F_Y1]) <= min_h[level];
  if (kind == 1 && !hard) return 0;
  if (kind == 0 || (hard && kind == 1)) return 1;
  return -1;
}

MFX_HD int dt_flag(const double* D, int cls, int level) {
  const double min_h[3] = {40, 25, 25};
  if (fabs(D[F_Y2] - D[F_Y1]) < min_h[level]) return 1;
  return (int)D[F_CODE] == cls ? 0 : -1;
}

struct Stats { int tp, fp, fn; double sim; };

MFX_HD void decode_comb(const mfx_kitti_eval_desc& d, int comb, int& m, int& level, int& metric, int& k) {
  k = comb % d.num_k; comb /= d.num_k;
  metric = comb % 3; comb /= 3;
  level = comb % 3; m = comb / 3;
}

// Greedy assignment of the detections of image b to its ground truths in label order (eval.py:156-286).
// COUNT_FP = false: no score threshold, best-scoring candidate wins, tp_scores receives the winners' scores.
// COUNT_FP = true : detections below `thresh` are invisible, highest overlap wins, false positives / DontCare / orientation
//                   similarity are accounted.
#ifndef MFX_KEVAL_NO_MATCH
#define MFX_KEVAL_NO_MATCH
constexpr double NO_MATCH = -__builtin_huge_val();     // 'ground truth not matched' in tp_scores
#endif
template <bool COUNT_FP>
MFX_HD Stats match(const mfx_kitti_eval_desc& d, int b, int comb, double thresh) {
  int m, level, metric, k;
  decode_comb(d, comb, m, level, metric, k);
  const int cls = d.classes[m];
  const double min_ov = d.min_overlaps[((long)k * 3 + metric) * d.num_classes + m];
  const int g0 = d.gt_off[b], ng = d.gt_off[b + 1] - g0, d0 = d.dt_off[b];
  // one 64-bit mask per detection set: MFX_EVAL_MAX_DET detections per image (the offsets live on the device, so the entry
  // points cannot check them; the host packer refuses larger images, a raw caller's surplus detections are ignored)
  const int nd = (d.dt_off[b + 1] - d0) > 64 ? 64 : (d.dt_off[b + 1] - d0);
  const double* ov = d.overlaps + (long)metric * d.n_pairs + d.pair_off[b];
  uint64_t taken = 0, skip = 0, soft = 0;                  // skip: other class or below threshold; soft: flag 1 ("ignored")
  for (int j = 0; j < nd; ++j) {
    const double* D = d.dt + (long)(d0 + j) * REC;
    const int f = dt_flag(D, cls, level);
    if (f == -1 || (COUNT_FP && D[F_SCORE] < thresh)) skip |= 1ull << j;
    if (f == 1) soft |= 1ull << j;
  }
  Stats s = {0, 0, 0, 0.0};
  for (int i = 0; i < ng; ++i) {
    const double* G = d.gt + (long)(g0 + i) * REC;
    const int ig = gt_flag(G, cls, level);
    if (!COUNT_FP) d.tp_scores[(long)comb * d.n_gt + g0 + i] = NO_MATCH;      // -inf: any real score, negative ones included, sorts above it
    if (ig == -1) continue;
    int best = -1; bool found = false, from_soft = false;
    double best_score = 0.0, max_ov = 0.0;
    for (int j = 0; j < nd; ++j) {
      if (((skip | taken) >> j) & 1) continue;
      const double o = ov[(long)j * ng + i];
      if (!(o > min_ov)) continue;
      const bool is_soft = (soft >> j) & 1;
      if (!COUNT_FP) {
        const double sc = d.dt[(long)(d0 + j) * REC + F_SCORE];
        if (!found || sc > best_score) { best = j; best_score = sc; found = true; }
      } else if (!is_soft && (o > max_ov || from_soft)) {
        max_ov = o; best = j; found = true; from_soft = false;
      } else if (is_soft && !found) {
        best = j; found = true; from_soft = true;
      }
    }
    if (!found) { if (ig == 0) ++s.fn; continue; }
    taken |= 1ull << best;
    if (ig == 1 || ((soft >> best) & 1)) continue;         // matched, but neither side counts
    ++s.tp;
    if (!COUNT_FP) d.tp_scores[(long)comb * d.n_gt + g0 + i] = d.dt[(long)(d0 + best) * REC + F_SCORE];
    else if (d.compute_aos && metric == 0) s.sim += (1.0 + cos(G[F_ALPHA] - d.dt[(long)(d0 + best) * REC + F_ALPHA])) / 2.0;
  }
  if (COUNT_FP) {
    const uint64_t all = nd >= 64 ? ~0ull : ((1ull << nd) - 1);
    uint64_t open = all & ~(taken | skip | soft);          // unmatched detections that would count as false positives
    if (metric == 0)                                       // ... unless they sit on a DontCare region (eval.py:253-267)
      for (int i = 0; i < ng; ++i) {
        const double* G = d.gt + (long)(g0 + i) * REC;
        if ((int)G[F_CODE] != CODE_DONTCARE) continue;
        for (int j = 0; j < nd; ++j)
          if (((open >> j) & 1) && bbox_overlap(d.dt + (long)(d0 + j) * REC + F_X1, G + F_X1, 0) > min_ov) open &= ~(1ull << j);
      }
    int fp = 0;
    for (uint64_t v = open; v; v &= v - 1) ++fp;
    s.fp = fp;
    if (!(s.tp > 0 || s.fp > 0)) s.sim = -1.0;
  }
  return s;
}

// eval.py:8-24: scores (sorted descending, `n` of them >= 0) at which recall first reaches k/40; returns their count.
MFX_HD int sample_thresholds(const double* sorted, int n, int num_gt, double* out) {
  double cur = 0.0;
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    const double l = (double)(i + 1) / num_gt;
    const double r = i < n - 1 ? (double)(i + 2) / num_gt : l;
    if ((r - cur) < (cur - l) && i < n - 1) continue;
    if (cnt < PTS) out[cnt] = sorted[i];
    ++cnt;
    cur += 1 / (PTS - 1.0);
  }
  return cnt < PTS ? cnt : PTS;
}

}  // namespace keval
}  // namespace mfx
