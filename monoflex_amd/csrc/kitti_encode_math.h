// Per-object / per-pixel arithmetic of the KITTI target encoder (mfx_kitti_encode_targets, mfx_kitti_preprocess_u8).
//
// Everything here works on one (image, object), (image, class, pixel) or (image, edge point) and writes straight into the
// batch-stacked arrays of mfx_kitti_desc, so kitti_encode.hip only maps threads onto these functions.  The functions are
// plain C++ over <cmath>, so tests/ can also compile this header for the host and check the indexing and dtype rules on
// CPU; the shipped library only ever calls them from device code.
//
// dtype rules (reference data/datasets/kitti.py:334-494): python floats and projections are float64; `obj.t`, `obj.box2d`
// and whatever is derived from them by numpy scalar arithmetic stay float32.  Compile with -ffp-contract=off: a fused
// multiply-add would change float32 results the reference rounds twice.
#pragma once
#include <cmath>
#include <cstdint>

#include "../../include/monoflex_hip.h"

#ifdef __HIPCC__
#define MFX_HD __host__ __device__ inline
#else
#define MFX_HD inline
#endif

namespace mfx {
namespace kitti {

constexpr double PI = 3.141592653589793;
constexpr int REC = 14;   // doubles per object record

MFX_HD void status_or(int32_t* p, int v) {          // objects of one image are encoded by different threads
#if defined(__HIP_DEVICE_COMPILE__)
  atomicOr(p, v);
#else
  *p |= v;
#endif
}

MFX_HD double wrap_pi(double a) {                     // kitti_utils.py:37-38 (while loops)
  while (a > PI) a -= PI * 2;
  while (a < -PI) a += PI * 2;
  return a;
}

MFX_HD void project(const double* P, double X, double Y, double Z, double& u, double& v, double& w) {
  // [X Y Z 1] . P^T, accumulated in index order (kitti_utils.py:320-324)
  const double qu = P[0] * X + P[1] * Y + P[2] * Z + P[3];
  const double qv = P[4] * X + P[5] * Y + P[6] * Z + P[7];
  w = P[8] * X + P[9] * Y + P[10] * Z + P[11];
  u = qu / w;
  v = qv / w;
}

// model/heatmap_coder.py:37-57, evaluated in T = the dtype of the 2D box
template <typename T>
MFX_HD T gaussian_radius(T height, T width) {
  const double mo = 0.7;
  const T b1 = height + width;
  const T c1 = width * height * (T)(1 - mo) / (T)(1 + mo);
  const T sq1 = std::sqrt(b1 * b1 - (T)4 * c1);
  const T r1 = (b1 + sq1) / (T)2;
  const T b2 = (T)2 * (height + width);
  const T c2 = (T)(1 - mo) * width * height;
  const T sq2 = std::sqrt(b2 * b2 - (T)16 * c2);
  const T r2 = (b2 + sq2) / (T)2;
  const T b3 = (T)(-2 * mo) * (height + width);
  const T c3 = (T)(mo - 1) * width * height;
  const T sq3 = std::sqrt(b3 * b3 - (T)(4 * (4 * mo)) * c3);
  const T r3 = (b3 + sq3) / (T)2;
  T r = r1;                                           // python min(): first of equal values, NaN never wins
  if (r2 < r) r = r2;
  if (r3 < r) r = r3;
  return r;
}

struct BoxStage {          // what the dtype-dependent half of the encoder hands back
  bool filtered;           // dropped by FILTER_ANNOS
  double c2d[2];           // 2D box centre in image pixels (before padding)
  double lo[2], hi[2];     // box on the output grid, widened to double for the integer comparisons
  float store[4];          // box as stored in "2d_bboxes"
  bool dim_pos;            // (bbox_dim > 0).all()
  int radius;              // circular heat-map radius
};

template <typename T>
MFX_HD BoxStage box_stage(T b0, T b1, T b2, T b3, double truncation, const mfx_kitti_desc& d, int pad_x, int pad_y) {
  BoxStage s;
  const T dw = b2 - b0, dh = b3 - b1;
  const T dmin = dw < dh ? dw : dh;
  s.filtered = d.filter_trunc >= 0 && truncation >= d.filter_trunc && (double)dmin <= d.filter_size;
  s.c2d[0] = (double)((b0 + b2) / (T)2);
  s.c2d[1] = (double)((b1 + b3) / (T)2);
  b0 = (b0 + (T)pad_x) / (T)d.down;  b2 = (b2 + (T)pad_x) / (T)d.down;      // kitti.py:420-422
  b1 = (b1 + (T)pad_y) / (T)d.down;  b3 = (b3 + (T)pad_y) / (T)d.down;
  const T bw = b2 - b0, bh = b3 - b1;
  s.dim_pos = bw > (T)0 && bh > (T)0;
  s.lo[0] = (double)b0; s.lo[1] = (double)b1; s.hi[0] = (double)b2; s.hi[1] = (double)b3;
  s.store[0] = (float)b0; s.store[1] = (float)b1; s.store[2] = (float)b2; s.store[3] = (float)b3;
  const T r = gaussian_radius<T>(bh, bw);
  const int ri = (int)r;                               // int(): truncation toward zero
  s.radius = ri > 0 ? ri : 0;
  return s;
}

// data/datasets/kitti_utils.py:990-1028 with the degree-1 polyfit through two points written as the exact line
MFX_HD int intersect_center(const double pc[2], const double c2d[2], int img_w, int img_h, double out[2]) {
  if (!(c2d[0] >= 0 && c2d[1] >= 0 && c2d[0] <= img_w - 1 && c2d[1] <= img_h - 1)) return 2;
  const double a = (c2d[1] - pc[1]) / (c2d[0] - pc[0]);
  const double b = pc[1] - a * pc[0];
  double best = 0; bool have = false;
  auto consider = [&](double x, double y) {
    const double dx = x - pc[0], dy = y - pc[1];
    const double dist = std::sqrt(dx * dx + dy * dy);
    if (!have || dist < best) { best = dist; out[0] = x; out[1] = y; have = true; }
  };
  if (0 <= b && b <= img_h - 1) consider(0.0, b);
  const double right_y = (img_w - 1) * a + b;
  if (0 <= right_y && right_y <= img_h - 1) consider((double)(img_w - 1), right_y);
  const double top_x = -b / a;
  if (0 <= top_x && top_x <= img_w - 1) consider(top_x, 0.0);
  const double bottom_x = (img_h - 1 - b) / a;
  if (0 <= bottom_x && bottom_x <= img_w - 1) consider(bottom_x, (double)(img_h - 1));
  return have ? 0 : 4;
}

MFX_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
MFX_HD int ceil_div(int a, int b) { return (a + b - 1) / b; }

// One object of one image: zeroes row i of every per-object field, then fills it unless the reference skips the object.
MFX_HD void encode_object(const mfx_kitti_desc& d, int b, int i) {
  const int M = d.max_objs;
  const long row = (long)b * M + i;
  d.cls_ids[row] = 0; d.reg_mask[row] = 0; d.trunc_mask[row] = 0; d.reg_weight[row] = 0.f;
  d.rotys[row] = 0.f; d.alphas[row] = 0.f; d.occlusions[row] = 0.0; d.truncations[row] = 0.0;
  d.target_centers[row * 2] = 0; d.target_centers[row * 2 + 1] = 0;
  d.offset_3D[row * 2] = 0.f; d.offset_3D[row * 2 + 1] = 0.f;
  for (int k = 0; k < 30; ++k) d.keypoints[row * 30 + k] = 0.f;
  for (int k = 0; k < 3; ++k) { d.keypoints_depth_mask[row * 3 + k] = 0.f; d.dimensions[row * 3 + k] = 0.f; d.locations[row * 3 + k] = 0.f; }
  for (int k = 0; k < 4; ++k) { d.bboxes[row * 4 + k] = 0.f; d.gt_bboxes[row * 4 + k] = 0.f; d.heat_radius[row * 4 + k] = 0; }
  for (int k = 0; k < 8; ++k) d.orientations[row * 8 + k] = 0.f;
  if (i >= d.n_obj[b]) return;

  const double* rec = d.records + row * REC;
  const int img_w = d.img_wh[b * 2], img_h = d.img_wh[b * 2 + 1];
  const int pad_x = (d.in_w - img_w) / 2, pad_y = (d.in_h - img_h) / 2;             // kitti.py:222-223
  const int out_w = d.in_w / d.down, out_h = d.in_h / d.down;
  const bool flip = d.flip[b] != 0;
  double P[12];
  for (int k = 0; k < 12; ++k) P[k] = d.P[b * 12 + k];
  if (flip) { P[2] = img_w - P[2] - 1; P[3] = -P[3]; }                               // augmentations.py:70-74

  const int cls_id = (int)rec[0];
  const double truncation = rec[1];
  const double occlusion = (double)(int)rec[2];
  double xmin = rec[3], ymin = rec[4], xmax = rec[5], ymax = rec[6];
  const double h = rec[7], w = rec[8], l = rec[9];
  float t0 = (float)rec[10]; const float t1 = (float)rec[11], t2 = (float)rec[12];
  double ry = rec[13];
  if (flip) {                                                                        // augmentations.py:44-66
    const double bw = xmax - xmin;
    xmin = img_w - xmax - 1;
    xmax = xmin + bw;
    ry = ry < 0 ? (-PI - ry) : (PI - ry);
    ry = wrap_pi(ry);
    t0 = -t0;
  }
  const float lbl[4] = {(float)xmin, (float)ymin, (float)xmax, (float)ymax};
  const double alpha = wrap_pi(ry - std::atan2((double)t0, (double)t2));

  const float loc1 = t1 - (float)(h / 2);                                            // float32 arithmetic (kitti.py:361)
  if (t2 <= 0.f) return;

  // 8 corners (kitti_utils.py:120-131) + centres of the bottom / top faces (kitti.py:397-398)
  const double c = std::cos(ry), s = std::sin(ry);
  const double xs[4] = {l / 2, l / 2, -l / 2, -l / 2}, zs[4] = {w / 2, -w / 2, -w / 2, w / 2};
  double K3[10][3];
  for (int k = 0; k < 8; ++k) {
    const double x = xs[k & 3], y = k < 4 ? 0.0 : -h, z = zs[k & 3];
    K3[k][0] = (c * x + 0.0 * y + s * z) + (double)t0;
    K3[k][1] = (0.0 * x + 1.0 * y + 0.0 * z) + (double)t1;
    K3[k][2] = (-s * x + 0.0 * y + c * z) + (double)t2;
  }
  for (int a = 0; a < 3; ++a) {
    K3[8][a] = (((K3[0][a] + K3[1][a]) + K3[2][a]) + K3[3][a]) / 4;
    K3[9][a] = (((K3[4][a] + K3[5][a]) + K3[6][a]) + K3[7][a]) / 4;
  }
  double K2[10][2];
  for (int k = 0; k < 10; ++k) { double wq; project(P, K3[k][0], K3[k][1], K3[k][2], K2[k][0], K2[k][1], wq); }
  double p0 = K2[0][0], p1 = K2[0][1], p2 = K2[0][0], p3 = K2[0][1];
  for (int k = 1; k < 8; ++k) {
    p0 = K2[k][0] < p0 ? K2[k][0] : p0;  p2 = K2[k][0] > p2 ? K2[k][0] : p2;
    p1 = K2[k][1] < p1 ? K2[k][1] : p1;  p3 = K2[k][1] > p3 ? K2[k][1] : p3;
  }
  const bool use_proj = p0 >= 0 && p1 >= 0 && p2 <= img_w - 1 && p3 <= img_h - 1;  // kitti.py:370-374
  const BoxStage bs = use_proj ? box_stage<double>(p0, p1, p2, p3, truncation, d, pad_x, pad_y)
                               : box_stage<float>(lbl[0], lbl[1], lbl[2], lbl[3], truncation, d, pad_x, pad_y);
  if (bs.filtered) return;

  double pc[2], wq;
  project(P, (double)t0, (double)loc1, (double)t2, pc[0], pc[1], wq);
  const bool inside = pc[0] >= 0 && pc[0] <= img_w - 1 && pc[1] >= 0 && pc[1] <= img_h - 1;
  double tpc[2] = {pc[0], pc[1]};
  if (!inside) {
    const int err = intersect_center(pc, bs.c2d, img_w, img_h, tpc);
    if (err) { status_or(d.status + b, err); return; }          // the reference raises here; flag and drop the object
  }

  bool vis[10];
  for (int k = 0; k < 10; ++k)
    vis[k] = K2[k][0] >= 0 && K2[k][0] <= img_w - 1 && K2[k][1] >= 0 && K2[k][1] <= img_h - 1 && K3[k][2] > 0;
  bool mv[10];                                                                       // KEYPOINT_VISIBLE_MODIFY (kitti.py:412-414)
  for (int k = 0; k < 4; ++k) mv[k] = mv[k + 4] = vis[k] || vis[k + 4];
  mv[8] = mv[9] = vis[8] || vis[9];
  const bool dv0 = mv[8] && mv[9], dv1 = mv[0] && mv[2] && mv[4] && mv[6], dv2 = mv[1] && mv[3] && mv[5] && mv[7];

  const double down = (double)d.down;
  const int x_min = ceil_div(pad_x, d.down), y_min = ceil_div(pad_y, d.down);
  const int x_max = (pad_x + img_w - 1) / d.down, y_max = (pad_y + img_h - 1) / d.down;
  const double tx = (tpc[0] + pad_x) / down, ty = (tpc[1] + pad_y) / down;
  const double pcx = (pc[0] + pad_x) / down, pcy = (pc[1] + pad_y) / down;
  const int tc0 = clampi((int)std::rint(tx), x_min, x_max);                         // np.round: half to even
  const int tc1 = clampi((int)std::rint(ty), y_min, y_max);
  const bool pred_2d = tc0 >= bs.lo[0] && tc1 >= bs.lo[1] && tc0 <= bs.hi[0] && tc1 <= bs.hi[1];
  if (!(bs.dim_pos && tc0 >= 0 && tc0 <= out_w - 1 && tc1 >= 0 && tc1 <= out_h - 1)) return;

  int rx, ryy, circular;
  if (!inside) {                                                                     // boundary heat-map (kitti.py:444-450)
    const double a0 = tc0 - bs.lo[0], a1 = bs.hi[0] - tc0, c0 = tc1 - bs.lo[1], c1 = bs.hi[1] - tc1;
    const double bw = a0 < a1 ? a0 : a1, bh = c0 < c1 ? c0 : c1;
    rx = (int)(bw * d.edge_ratio); ryy = (int)(bh * d.edge_ratio);
    rx = rx > 0 ? rx : 0; ryy = ryy > 0 ? ryy : 0;
    circular = 0;
    if (rx > 0 && ryy > 0) { status_or(d.status + b, 8); return; }                             // the reference asserts
  } else {
    rx = ryy = bs.radius; circular = 1;
  }
  d.heat_radius[row * 4 + 0] = rx; d.heat_radius[row * 4 + 1] = ryy; d.heat_radius[row * 4 + 2] = circular; d.heat_radius[row * 4 + 3] = 1;

  d.cls_ids[row] = cls_id;
  d.target_centers[row * 2] = tc0; d.target_centers[row * 2 + 1] = tc1;
  d.offset_3D[row * 2] = (float)(pcx - tc0); d.offset_3D[row * 2 + 1] = (float)(pcy - tc1);
  for (int k = 0; k < 4; ++k) d.gt_bboxes[row * 4 + k] = lbl[k];
  if (pred_2d) for (int k = 0; k < 4; ++k) d.bboxes[row * 4 + k] = bs.store[k];
  for (int k = 0; k < 10; ++k) {
    d.keypoints[row * 30 + k * 3 + 0] = (float)((K2[k][0] + pad_x) / down - tc0);
    d.keypoints[row * 30 + k * 3 + 1] = (float)((K2[k][1] + pad_y) / down - tc1);
    d.keypoints[row * 30 + k * 3 + 2] = mv[k] ? 1.f : 0.f;
  }
  d.keypoints_depth_mask[row * 3 + 0] = dv0 ? 1.f : 0.f;
  d.keypoints_depth_mask[row * 3 + 1] = dv1 ? 1.f : 0.f;
  d.keypoints_depth_mask[row * 3 + 2] = dv2 ? 1.f : 0.f;
  d.dimensions[row * 3 + 0] = (float)l; d.dimensions[row * 3 + 1] = (float)h; d.dimensions[row * 3 + 2] = (float)w;
  d.locations[row * 3 + 0] = t0; d.locations[row * 3 + 1] = loc1; d.locations[row * 3 + 2] = t2;
  d.rotys[row] = (float)ry; d.alphas[row] = (float)alpha;
  {                                                                                  // kitti.py:181-200, 4 bins
    const double centers[4] = {0.0, PI / 2, PI, -PI / 2};
    const double bin = 2 * PI / 4, range = bin / 2 + bin * (1.0 / 6);
    for (int k = 0; k < 4; ++k) {
      double off = alpha - centers[k];
      if (off > PI) off -= 2 * PI;
      if (off < -PI) off += 2 * PI;
      if (std::fabs(off) < range) { d.orientations[row * 8 + k] = 1.f; d.orientations[row * 8 + 4 + k] = (float)off; }
    }
  }
  d.reg_mask[row] = 1; d.reg_weight[row] = 1.f; d.trunc_mask[row] = inside ? 0 : 1;
  d.occlusions[row] = occlusion; d.truncations[row] = truncation;
}

// Per-image scalars: pad, flipped P, edge-walk length, and the n_obj > max_objs check.  Call before encode_object.
MFX_HD void image_header(const mfx_kitti_desc& d, int b) {
  const int img_w = d.img_wh[b * 2], img_h = d.img_wh[b * 2 + 1];
  const int pad_x = (d.in_w - img_w) / 2, pad_y = (d.in_h - img_h) / 2;
  d.pad_size[b * 2] = pad_x; d.pad_size[b * 2 + 1] = pad_y;
  for (int k = 0; k < 12; ++k) d.P_out[b * 12 + k] = d.P[b * 12 + k];
  if (d.flip[b]) { d.P_out[b * 12 + 2] = img_w - d.P[b * 12 + 2] - 1; d.P_out[b * 12 + 3] = -d.P[b * 12 + 3]; }
  const int x0 = ceil_div(pad_x, d.down), y0 = ceil_div(pad_y, d.down);
  const int x1 = (pad_x + img_w - 1) / d.down, y1 = (pad_y + img_h - 1) / d.down;
  d.edge_len[b] = 2 * (y1 - y0) + 2 * (x1 - x0) + 1 - 1;
  d.status[b] = d.n_obj[b] > d.max_objs ? 1 : 0;
}

// k-th entry of the zero-padded border walk (kitti.py:126-179): left side downwards, bottom rightwards, right side
// upwards, top leftwards back to the start.
MFX_HD void edge_point(const mfx_kitti_desc& d, int b, int k) {
  const int img_w = d.img_wh[b * 2], img_h = d.img_wh[b * 2 + 1];
  const int pad_x = (d.in_w - img_w) / 2, pad_y = (d.in_h - img_h) / 2;
  const int x0 = ceil_div(pad_x, d.down), y0 = ceil_div(pad_y, d.down);
  const int x1 = (pad_x + img_w - 1) / d.down, y1 = (pad_y + img_h - 1) / d.down;
  const int nl = y1 - y0, nb = x1 - x0, nr = y1 - y0, nt = x1 - x0 + 1;
  const int max_edge = 2 * (d.in_w / d.down + d.in_h / d.down);
  int64_t x = 0, y = 0;
  if (k < nl) { x = x0; y = y0 + k; }
  else if (k < nl + nb) { x = x0 + (k - nl); y = y1; }
  else if (k < nl + nb + nr) { x = x1; y = y1 - (k - nl - nb); }
  else if (k < nl + nb + nr + nt) { x = x1 - (k - nl - nb - nr); y = y0; }
  int64_t* e = d.edge_indices + ((long)b * max_edge + k) * 2;
  e[0] = x; e[1] = y;
}

// Heat-map pixel (b, cls, y, x): max over the drawn objects of that class whose +-radius window covers the pixel
// (model/heatmap_coder.py:59-67 circular, :126-135 boundary form; sigma = (2r+1)/6).  Needs encode_object done.
MFX_HD float heat_pixel(const mfx_kitti_desc& d, int b, int cls, int y, int x) {
  double best = 0.0;
  const int n = d.n_obj[b] < d.max_objs ? d.n_obj[b] : d.max_objs;
  for (int i = 0; i < n; ++i) {
    const long row = (long)b * d.max_objs + i;
    const int32_t* hr = d.heat_radius + row * 4;
    if (!hr[3] || d.cls_ids[row] != cls) continue;
    const int dx = x - d.target_centers[row * 2], dy = y - d.target_centers[row * 2 + 1];
    const int rx = hr[0], ryy = hr[1];
    if (dx < -rx || dx > rx || dy < -ryy || dy > ryy) continue;
    const double fx = (double)dx, fy = (double)dy;
    double g;
    if (hr[2]) {
      const double sigma = (double)(2 * rx + 1) / 6;
      g = std::exp(-(fx * fx + fy * fy) / (2 * sigma * sigma));
    } else {
      const double sx = (double)(2 * rx + 1) / 6, sy = (double)(2 * ryy + 1) / 6;
      g = std::exp(-(fx * fx) / (2 * sx * sx) - (fy * fy) / (2 * sy * sy));
    }
    best = g > best ? g : best;
  }
  return (float)best;
}

// Output pixel (b, y, x) of the network input, all three channels (data/transforms/transforms.py:15-31).
MFX_HD void preprocess_pixel(const uint8_t* pixels, const int64_t* offsets, const int32_t* img_wh, const int32_t* flip,
                             float* out, int b, int y, int x, int in_w, int in_h, const float* mean, const float* stdv) {
  const int img_w = img_wh[b * 2], img_h = img_wh[b * 2 + 1];
  const int pad_x = (in_w - img_w) / 2, pad_y = (in_h - img_h) / 2;
  const int sy = y - pad_y;
  int sx = x - pad_x;
  const bool in = sy >= 0 && sy < img_h && sx >= 0 && sx < img_w;
  if (in && flip[b]) sx = img_w - 1 - sx;
  const uint8_t* src = pixels + offsets[b] + ((long)sy * img_w + sx) * 3;
  const long plane = (long)in_w * in_h;
  float* o = out + (long)b * 3 * plane + (long)y * in_w + x;
  for (int c = 0; c < 3; ++c) {
    const float v = in ? (float)src[c] / 255.f : 0.f;
    o[c * plane] = (v - mean[c]) / stdv[c];
  }
}

}  // namespace kitti
}  // namespace mfx
