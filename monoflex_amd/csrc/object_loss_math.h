// Per-object regression losses of the detection head and their gradient, as one function of (50 regression channels at the
// object's centre pixel, the object's target row) -- C ABI mfx_object_loss / mfx_object_loss_backward.
//
// Reference: model/head/detector_loss.py:116-482 (prepare_predictions + the nine regression terms + the logged MAEs) with
// the decoders of model/anno_encoder.py:88-295, model/layers/iou_loss.py:7-49 and data/datasets/kitti_utils.py:350-369.  The
// reference evaluates them as ~300 small tensor ops forward and ~600 backward per step; every one of them is per-object, so
// here one wavefront owns one object: lane c carries the forward-mode tangent d/d(channel c) through the whole expression
// (Dual below), which gives the exact gradient row of every loss term in the same pass -- no hand-derived backward to keep
// in sync.  Non-differentiable selections (arg max of the orientation bins, `.detach()`ed operands) simply carry a zero tangent.
// kitti_eval_math.h-style: plain functions, loss_kernels.hip maps lanes onto them, tests/shim compiles them for the host.
#pragma once
#include <cmath>

#include "../../include/monoflex_hip.h"

#ifndef MFX_HD
#ifdef __HIPCC__
#define MFX_HD __host__ __device__ inline
#else
#define MFX_HD inline
#endif
#endif

namespace mfx {
namespace oloss {

constexpr int ROW = MFX_OBJ_ROW, NTERM = MFX_OBJ_TERMS, NVAL = MFX_OBJ_VALUES, NNORM = 8;
// target row fields (float32 each; written by Loss_Computation.pack_objects)
enum { R_VALID = 0, R_CLS, R_CX, R_CY, R_BOX = 4, R_KP = 8, R_KDM = 38, R_DIMS = 41, R_DEPTH = 44, R_ROTY, R_ORI = 46, R_OFF = 54,
       R_TRUNC = 56, R_B, R_CAL = 58, R_PAD = 64, R_FU_RANK = 66 };
// loss terms (the reference's names), then the logged values
enum { T_BBOX = 0, T_DEPTH, T_OFFSET, T_TRUNC_OFFSET, T_ORIEN, T_DIMS, T_CORNER, T_KEYPOINT, T_KEYPOINT_DEPTH, T_SOFT_DEPTH,
       V_IOU2D = 10, V_REAL_DEPTH, V_VALID_KD, V_DEPTH_MAE, V_CENTER_MAE, V_02_MAE, V_13_MAE, V_LOWER_MAE, V_HARD_MAE, V_SOFT_MAE,
       V_MEAN_MAE };
enum { N_V = 0, N_V2D, N_V_INSIDE, N_TRUNC, N_KMASK, N_KD_VALID, N_KD_INVALID, N_ORI };
enum { C_2D = 0, C_OFF3D, C_CORNER, C_CORNER_UNC, C_DIM, C_ORI_CLS, C_ORI_OFF, C_DEPTH, C_DEPTH_UNC };

// ---- forward-mode value/tangent pair ------------------------------------------------------------------------------------------
struct Dual { float v, d; };
MFX_HD Dual K(float c) { return Dual{c, 0.f}; }
MFX_HD Dual detach(Dual a) { return Dual{a.v, 0.f}; }
MFX_HD Dual operator+(Dual a, Dual b) { return Dual{a.v + b.v, a.d + b.d}; }
MFX_HD Dual operator-(Dual a, Dual b) { return Dual{a.v - b.v, a.d - b.d}; }
MFX_HD Dual operator-(Dual a) { return Dual{-a.v, -a.d}; }
MFX_HD Dual operator*(Dual a, Dual b) { return Dual{a.v * b.v, a.d * b.v + a.v * b.d}; }
MFX_HD Dual operator/(Dual a, Dual b) { const float q = a.v / b.v; return Dual{q, (a.d - q * b.d) / b.v}; }
MFX_HD Dual operator+(Dual a, float b) { return Dual{a.v + b, a.d}; }
MFX_HD Dual operator-(Dual a, float b) { return Dual{a.v - b, a.d}; }
MFX_HD Dual operator*(Dual a, float b) { return Dual{a.v * b, a.d * b}; }
MFX_HD Dual operator/(Dual a, float b) { return Dual{a.v / b, a.d / b}; }
MFX_HD Dual dexp(Dual a) { const float e = expf(a.v); return Dual{e, e * a.d}; }
MFX_HD Dual dlog(Dual a) { return Dual{logf(a.v), a.d / a.v}; }
MFX_HD Dual dabs(Dual a) { return Dual{fabsf(a.v), a.v > 0.f ? a.d : (a.v < 0.f ? -a.d : 0.f)}; }          // sign(0) = 0 as torch
MFX_HD Dual drelu(Dual a) { return a.v > 0.f ? a : Dual{0.f, 0.f}; }
MFX_HD Dual dclamp(Dual a, float lo, float hi) {                                                            // gradient passes on [lo, hi]
    return a.v < lo ? Dual{lo, 0.f} : (a.v > hi ? Dual{hi, 0.f} : a);
}
MFX_HD Dual dmin(Dual a, Dual b) { return a.v < b.v ? a : (b.v < a.v ? b : Dual{a.v, 0.5f * (a.d + b.d)}); }
MFX_HD Dual dmax(Dual a, Dual b) { return a.v > b.v ? a : (b.v > a.v ? b : Dual{a.v, 0.5f * (a.d + b.d)}); }
MFX_HD Dual dsin(Dual a) { return Dual{sinf(a.v), cosf(a.v) * a.d}; }
MFX_HD Dual dcos(Dual a) { return Dual{cosf(a.v), -sinf(a.v) * a.d}; }
MFX_HD Dual datan2(Dual y, Dual x) {
    const float r2 = x.v * x.v + y.v * y.v;
    return Dual{atan2f(y.v, x.v), r2 > 0.f ? (y.d * x.v - x.d * y.v) / r2 : 0.f};
}
MFX_HD Dual dsqrt(Dual a) { const float s = sqrtf(a.v); return Dual{s, s > 0.f ? 0.5f * a.d / s : 0.f}; }
MFX_HD Dual dsigmoid(Dual a) { const float s = 1.f / (1.f + expf(-a.v)); return Dual{s, s * (1.f - s) * a.d}; }

struct Vec3 { Dual x, y, z; };

// anno_encoder.py:124-140
MFX_HD Dual decode_depth(Dual off, const mfx_object_loss_cfg& c) {
    Dual d = c.depth_mode == 0 ? dexp(off) : (c.depth_mode == 1 ? off * c.depth_ref[1] + c.depth_ref[0] : K(1.f) / dsigmoid(off) - 1.f);
    return c.has_depth_range ? dclamp(d, c.depth_range[0], c.depth_range[1]) : d;
}
// anno_encoder.py:142-156 with kitti_utils.py:350-369 (project_image_to_rect, one point)
MFX_HD Vec3 decode_location(float px, float py, Dual ox, Dual oy, Dual depth, const float* cal, const float* pad, float down) {
    const Dual u = (ox + px) * down - pad[0], v = (oy + py) * down - pad[1];
    return Vec3{(u - cal[2]) * depth / cal[0] + cal[4], (v - cal[3]) * depth / cal[1] + cal[5], depth};
}
// anno_encoder.py:88-122: corner k of the box (rotation about Y, centre at half height as the reference builds it)
MFX_HD Vec3 box_corner(int k, Dual cs, Dual sn, const Dual* dims, const Vec3& loc) {
    const float sx = (k & 2) ? 1.f : -1.f;                         // -1 -1 1 1 -1 -1 1 1
    const float sy = k < 4 ? 1.f : -1.f;                           //  1  1 1 1 -1 -1 -1 -1
    const float sz = ((k + 1) & 2) ? 1.f : -1.f;                   // -1  1 1 -1 -1 1 1 -1
    const Dual x = dims[0] * (0.5f * sx), y = dims[1] * (0.5f * sy), z = dims[2] * (0.5f * sz);
    return Vec3{cs * x + sn * z + loc.x, y + loc.y, -(sn * x) + cs * z + loc.z};
}

// One object.  X(ch) returns channel `ch` of the 50 regression channels at the object's pixel as a Dual seeded for this lane;
// `t` is the target row, `nrm` the eight batch-wide selection counts.  out[NVAL] ACCUMULATES nothing: it is overwritten with this
// object's contribution to each loss term (already weighted and divided by its count) and to each logged mean.
template <typename Reader>
MFX_HD void object_terms(const Reader& X, const float* t, const mfx_object_loss_cfg& c, const float* nrm, Dual* out) {
    for (int i = 0; i < NVAL; ++i) out[i] = K(0.f);
    if (t[R_VALID] == 0.f) return;
    const float PI = 3.14159265358979323846f;
    auto cnt = [&](int i) { return fmaxf(nrm[i], 1.f); };
    const int cls = t[R_CLS] > 0.f ? (int)t[R_CLS] : 0;
    const float px = t[R_CX], py = t[R_CY];
    const float* box = t + R_BOX;
    const float* cal = t + R_CAL;
    const float* pad = t + R_PAD;
    const float t_depth = t[R_DEPTH];
    const float inv_v = 1.f / cnt(N_V);

    // ---- 2D box: GIoU / IoU of the four ReLU'ed side distances (iou_loss.py:12-49) ---------------------------------------------
    if (box[3] - box[1] > 0.f && box[2] - box[0] > 0.f) {
        const Dual pl = drelu(X(c.ch[C_2D])), pt = drelu(X(c.ch[C_2D] + 1)), pr = drelu(X(c.ch[C_2D] + 2)), pb = drelu(X(c.ch[C_2D] + 3));
        const Dual tl = K(px - box[0]), tt = K(py - box[1]), tr = K(box[2] - px), tb = K(box[3] - py);
        const Dual t_area = (tl + tr) * (tt + tb), p_area = (pl + pr) * (pt + pb);
        const Dual w_i = dmin(pl, tl) + dmin(pr, tr), h_i = dmin(pb, tb) + dmin(pt, tt);
        const Dual g_w = dmax(pl, tl) + dmax(pr, tr), g_h = dmax(pb, tb) + dmax(pt, tt);
        const Dual ac = g_w * g_h + 1e-7f, inter = w_i * h_i, uni = t_area + p_area - inter;
        const Dual iou = (inter + 1.0f) / (uni + 1.0f);
        const Dual l = c.iou_type == 1 ? -dlog(iou) : (c.iou_type == 2 ? K(1.f) - iou : K(1.f) - (iou - (ac - uni) / ac));
        out[T_BBOX] = l * (c.w[T_BBOX] / cnt(N_V2D));
        out[V_IOU2D] = K(iou.v / cnt(N_V2D));
    }

    // ---- decoded predictions ---------------------------------------------------------------------------------------------------
    const Dual p_depth = decode_depth(X(c.ch[C_DEPTH]), c);
    const Dual d_unc = dclamp(X(c.ch[C_DEPTH_UNC]), c.unc_lo, c.unc_hi);
    Dual p_dims[3];
    for (int k = 0; k < 3; ++k) {                                   // anno_encoder.py:217-239
        Dual o = X(c.ch[C_DIM] + k);
        if (c.dim_exp) o = dexp(o);
        p_dims[k] = c.dim_use_std ? o * c.dim_std[cls * 3 + k] + c.dim_mean[cls * 3 + k] : o * c.dim_mean[cls * 3 + k];
    }
    Dual kx[10], ky[10];
    for (int j = 0; j < 10; ++j) { kx[j] = X(c.ch[C_CORNER] + 2 * j); ky[j] = X(c.ch[C_CORNER] + 2 * j + 1); }
    // depths from the three keypoint groups (anno_encoder.py:185-215); the focal length is the reference's rank-indexed one
    Dual kd[3];
    {
        const Dual fh = p_dims[1] * t[R_FU_RANK];
        auto solve = [&](Dual dh) { return fh / (drelu(dh) * c.down_ratio + c.eps); };
        kd[0] = solve(ky[8] - ky[9]);
        kd[1] = (solve(ky[0] - ky[4]) + solve(ky[2] - ky[6])) * 0.5f;
        kd[2] = (solve(ky[1] - ky[5]) + solve(ky[3] - ky[7])) * 0.5f;
        for (int g = 0; g < 3; ++g) kd[g] = dclamp(kd[g], c.depth_range[0], c.depth_range[1]);
    }
    Dual c_unc[3];
    for (int g = 0; g < 3; ++g) c_unc[g] = dclamp(X(c.ch[C_CORNER_UNC] + g), c.unc_lo, c.unc_hi);
    // uncertainty-weighted combination of the four depth estimates
    const Dual comb_depth[4] = {p_depth, kd[0], kd[1], kd[2]};
    const Dual comb_unc[4] = {dexp(d_unc), dexp(c_unc[0]), dexp(c_unc[1]), dexp(c_unc[2])};
    Dual wsum = K(0.f), soft = K(0.f);
    for (int i = 0; i < 4; ++i) wsum = wsum + K(1.f) / comb_unc[i];
    for (int i = 0; i < 4; ++i) soft = soft + comb_depth[i] * ((K(1.f) / comb_unc[i]) / wsum);
    int amin = 0;
    for (int i = 1; i < 4; ++i) if (comb_unc[i].v < comb_unc[amin].v) amin = i;
    Dual corner_depth = p_depth;                                     // CORNER_LOSS_DEPTH: direct
    if (c.corner_depth_mode == 1) corner_depth = (kd[0] + kd[1] + kd[2]) / 3.f;
    else if (c.corner_depth_mode == 2) corner_depth = soft;
    else if (c.corner_depth_mode == 3) corner_depth = comb_depth[amin];
    const Dual ox = X(c.ch[C_OFF3D]), oy = X(c.ch[C_OFF3D] + 1);
    const Vec3 p_loc = decode_location(px, py, ox, oy, corner_depth, cal, pad, c.down_ratio);
    // multi-bin yaw (anno_encoder.py:241-295): the most confident bin's residual + its centre, + the viewing-ray angle
    Dual p_roty;
    {
        int best = 0; float bconf = -1.f;
        for (int i = 0; i < 4; ++i) {
            const float a = X(c.ch[C_ORI_CLS] + 2 * i).v, b = X(c.ch[C_ORI_CLS] + 2 * i + 1).v;
            const float m = fmaxf(a, b), conf = expf(b - m) / (expf(a - m) + expf(b - m));
            if (conf > bconf) { bconf = conf; best = i; }
        }
        const float centers[4] = {0.f, PI / 2, PI, -PI / 2};
        const Dual alpha = datan2(X(c.ch[C_ORI_OFF] + 2 * best), X(c.ch[C_ORI_OFF] + 2 * best + 1)) + centers[best];
        p_roty = alpha + datan2(p_loc.x, p_loc.z);
        if (p_roty.v > PI) p_roty = p_roty - 2 * PI;
        if (p_roty.v < -PI) p_roty = p_roty + 2 * PI;
    }

    // ---- depth with aleatoric uncertainty (detector_loss.py:302-314) -----------------------------------------------------------
    {
        const Dual l1 = dabs(p_depth - t_depth) * c.w[T_DEPTH];
        out[T_DEPTH] = (l1 * dexp(-d_unc) + d_unc * c.w[T_DEPTH]) * inv_v;
        out[V_REAL_DEPTH] = K(l1.v * inv_v);
    }
    // ---- projected-centre offset; truncated objects optionally in their own (log) term -----------------------------------------
    {
        const Dual l1 = dabs(ox - t[R_OFF]) + dabs(oy - t[R_OFF + 1]);
        const bool trunc = t[R_TRUNC] != 0.f;
        if (c.separate_trunc) {
            if (trunc) out[T_TRUNC_OFFSET] = (c.trunc_log ? dlog(l1 + 1.f) : l1) * (c.w[T_TRUNC_OFFSET] / cnt(N_TRUNC));
            else out[T_OFFSET] = l1 * (c.w[T_OFFSET] / cnt(N_V_INSIDE));
        } else {
            out[T_OFFSET] = l1 * (c.w[T_OFFSET] * inv_v);
        }
    }
    // ---- multi-bin orientation (detector_loss.py:495-517) ----------------------------------------------------------------------
    {
        Dual ce = K(0.f), rg = K(0.f);
        for (int i = 0; i < 4; ++i) {
            const Dual a = X(c.ch[C_ORI_CLS] + 2 * i), b = X(c.ch[C_ORI_CLS] + 2 * i + 1);
            const float m = fmaxf(a.v, b.v);
            const Dual lse = dlog(dexp(a - m) + dexp(b - m)) + m;
            const bool is_bin = t[R_ORI + i] == 1.f;
            ce = ce + (lse - ((int)t[R_ORI + i] == 1 ? b : a));
            if (is_bin) {
                const Dual s = X(c.ch[C_ORI_OFF] + 2 * i), q = X(c.ch[C_ORI_OFF] + 2 * i + 1);
                const Dual nr = dsqrt(s * s + q * q);
                const Dual dn = nr.v > 1e-12f ? nr : K(1e-12f);        // F.normalize: x / max(||x||, 1e-12)
                rg = rg + dabs(s / dn - sinf(t[R_ORI + 4 + i])) + dabs(q / dn - cosf(t[R_ORI + 4 + i]));
            }
        }
        out[T_ORIEN] = (ce * (inv_v / 4.f) + rg / cnt(N_ORI)) * c.w[T_ORIEN];
    }
    // ---- dimensions ------------------------------------------------------------------------------------------------------------
    {
        Dual l = K(0.f);
        for (int k = 0; k < 3; ++k) l = l + dabs(p_dims[k] - t[R_DIMS + k]) * c.dim_weight[k];
        out[T_DIMS] = l * (c.w[T_DIMS] * inv_v);
    }
    // ---- eight box corners, L1 per coordinate, mean over corners (detector_loss.py:338-339) -------------------------------------
    {
        const Vec3 t_loc = decode_location(px, py, K(t[R_OFF]), K(t[R_OFF + 1]), K(t_depth), cal, pad, c.down_ratio);
        const Dual t_dims[3] = {K(t[R_DIMS]), K(t[R_DIMS + 1]), K(t[R_DIMS + 2])};
        const Dual pc = dcos(p_roty), ps = dsin(p_roty), tc = K(cosf(t[R_ROTY])), ts = K(sinf(t[R_ROTY]));
        Dual l = K(0.f);
        for (int k = 0; k < 8; ++k) {
            const Vec3 a = box_corner(k, pc, ps, p_dims, p_loc), b = box_corner(k, tc, ts, t_dims, t_loc);
            l = l + dabs(a.x - b.x) + dabs(a.y - b.y) + dabs(a.z - b.z);
        }
        out[T_CORNER] = l * (c.w[T_CORNER] * inv_v / 8.f);
    }
    // ---- keypoints (visible ones) ----------------------------------------------------------------------------------------------
    {
        Dual l = K(0.f);
        for (int j = 0; j < 10; ++j)
            l = l + (dabs(kx[j] - t[R_KP + 3 * j]) + dabs(ky[j] - t[R_KP + 3 * j + 1])) * t[R_KP + 3 * j + 2];
        out[T_KEYPOINT] = l * (c.w[T_KEYPOINT] / cnt(N_KMASK));
    }
    // ---- keypoint depths with their uncertainties; invalid groups only train the uncertainty (detector_loss.py:341-371) ---------
    {
        Dual l = K(0.f);
        float lg = 0.f;
        for (int g = 0; g < 3; ++g) {
            const float wk = c.w[T_KEYPOINT_DEPTH];
            if (t[R_KDM + g] != 0.f) {
                const Dual l1 = dabs(kd[g] - t_depth) * wk;
                l = l + (l1 * dexp(-c_unc[g]) + c_unc[g] * wk) / cnt(N_KD_VALID);
                lg += l1.v / cnt(N_KD_VALID);
            } else if (c.modify_invalid) {
                l = l + dabs(detach(kd[g]) - t_depth) * wk * dexp(-c_unc[g]) / cnt(N_KD_INVALID);
            }
        }
        out[T_KEYPOINT_DEPTH] = l;
        out[V_VALID_KD] = K(lg);
    }
    // ---- the combined depth ----------------------------------------------------------------------------------------------------
    out[T_SOFT_DEPTH] = dabs(soft - t_depth) * (c.w[T_SOFT_DEPTH] * inv_v);
    // ---- logged relative depth errors (detector_loss.py:396-482) ---------------------------------------------------------------
    {
        const float mae[4] = {fabsf(p_depth.v - t_depth) / t_depth, fabsf(kd[0].v - t_depth) / t_depth, fabsf(kd[1].v - t_depth) / t_depth,
                              fabsf(kd[2].v - t_depth) / t_depth};
        out[V_DEPTH_MAE] = K(mae[0] * inv_v); out[V_CENTER_MAE] = K(mae[1] * inv_v);
        out[V_02_MAE] = K(mae[2] * inv_v); out[V_13_MAE] = K(mae[3] * inv_v);
        out[V_LOWER_MAE] = K(fminf(fminf(mae[0], mae[1]), fminf(mae[2], mae[3])) * inv_v);
        out[V_HARD_MAE] = K(mae[amin] * inv_v);
        out[V_SOFT_MAE] = K(fabsf(soft.v - t_depth) / t_depth * inv_v);
        out[V_MEAN_MAE] = K(fabsf(0.25f * (comb_depth[0].v + comb_depth[1].v + comb_depth[2].v + comb_depth[3].v) - t_depth) / t_depth * inv_v);
    }
}

// the eight batch-wide selection counts, accumulated row by row (callers reduce over rows)
MFX_HD void row_counts(const float* t, float* n) {
    for (int i = 0; i < NNORM; ++i) n[i] = 0.f;
    if (t[R_VALID] == 0.f) return;
    const float* box = t + R_BOX;
    n[N_V] = 1.f;
    n[N_V2D] = (box[3] - box[1] > 0.f && box[2] - box[0] > 0.f) ? 1.f : 0.f;
    n[N_TRUNC] = t[R_TRUNC] != 0.f ? 1.f : 0.f;
    n[N_V_INSIDE] = 1.f - n[N_TRUNC];
    for (int j = 0; j < 10; ++j) n[N_KMASK] += t[R_KP + 3 * j + 2];
    for (int g = 0; g < 3; ++g) { n[N_KD_VALID] += t[R_KDM + g] != 0.f ? 1.f : 0.f; n[N_KD_INVALID] += t[R_KDM + g] != 0.f ? 0.f : 1.f; }
    for (int i = 0; i < 4; ++i) n[N_ORI] += t[R_ORI + i] == 1.f ? 1.f : 0.f;
}

}  // namespace oloss
}  // namespace mfx
