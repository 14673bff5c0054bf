/*
 * oracle/dcn_v2_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's modulated deformable convolution
 * (DCNv2) forward and backward.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker.
 *
 * What it restates (paths relative to /root/reference/model/backbone/DCNv2):
 *   src/cpu/dcn_v2_im2col_cpu.cpp:27-56    bilinear sample with per-corner zeroing
 *   src/cpu/dcn_v2_im2col_cpu.cpp:58-82    gradient weight of a pixel wrt a sample
 *   src/cpu/dcn_v2_im2col_cpu.cpp:84-125   d(sample)/d(coordinate)
 *   src/cpu/dcn_v2_im2col_cpu.cpp:127-196  modulated im2col
 *   src/cpu/dcn_v2_im2col_cpu.cpp:198-257  col2im (grad wrt input)
 *   src/cpu/dcn_v2_im2col_cpu.cpp:259-329  col2im_coord (grad wrt offset, mask)
 *   src/cpu/dcn_v2_cpu.cpp:17-107          forward  = bias + W * columns
 *   src/cpu/dcn_v2_cpu.cpp:109-233         backward = 2 GEMMs + 3 sampling passes + GEMV
 *
 * The reference's native extension cannot be built in this image (it includes
 * <TH/TH.h>, removed from torch long before 2.10), so this restatement is
 * pinned by the reference's own known-answer tests (testcpu.py:32-67 zero
 * offset, testcpu.py:69-97 gradcheck) -- see tests/test_oracle_dcn.py.
 *
 * Layouts are the reference's: input (B,C,H,W), offset (B,2*dg*kh*kw,Ho,Wo)
 * with channel 2k = dh and 2k+1 = dw of tap k = i*kw+j, mask (B,dg*kh*kw,Ho,Wo),
 * weight (Cout,C,kh,kw), columns (C*kh*kw, Ho*Wo) per image, all fp32.
 *
 * The loop nest is organised per (output pixel, tap) so the sampling geometry
 * is computed once and reused across channels; the per-element arithmetic and
 * its operation order follow the reference so results agree to fp32 round-off.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int C, H, W;          /* input channels / height / width            */
    int Ho, Wo;           /* output height / width                      */
    int kh, kw;           /* kernel                                      */
    int pad_h, pad_w, stride_h, stride_w, dil_h, dil_w;
    int dg;               /* deformable groups                           */
} dcn_geom;

/* im2col_cpu.cpp:27-56 : value of one plane at fractional (h,w); each of the
 * four corners contributes only if it lies inside the plane.               */
static float bilinear_at(const float *plane, int H, int W, float h, float w)
{
    int h0 = (int)floorf(h), w0 = (int)floorf(w);
    int h1 = h0 + 1, w1 = w0 + 1;
    float lh = h - h0, lw = w - w0;
    float hh = 1 - lh, hw = 1 - lw;
    float v1 = (h0 >= 0 && w0 >= 0) ? plane[h0 * W + w0] : 0.f;
    float v2 = (h0 >= 0 && w1 <= W - 1) ? plane[h0 * W + w1] : 0.f;
    float v3 = (h1 <= H - 1 && w0 >= 0) ? plane[h1 * W + w0] : 0.f;
    float v4 = (h1 <= H - 1 && w1 <= W - 1) ? plane[h1 * W + w1] : 0.f;
    float w1_ = hh * hw, w2_ = hh * lw, w3_ = lh * hw, w4_ = lh * lw;
    return (w1_ * v1 + w2_ * v2 + w3_ * v3 + w4_ * v4);
}

/* im2col_cpu.cpp:84-125 : derivative of the sample wrt h (dir 0) or w (dir 1) */
static float coord_weight(const float *plane, int H, int W, float h, float w, int dir)
{
    if (h <= -1 || h >= H || w <= -1 || w >= W) return 0.f;
    int h0 = (int)floorf(h), w0 = (int)floorf(w);
    int h1 = h0 + 1, w1 = w0 + 1;
    float acc = 0.f;
    if (dir == 0) {
        if (h0 >= 0 && w0 >= 0)         acc += -1 * (w0 + 1 - w) * plane[h0 * W + w0];
        if (h0 >= 0 && w1 <= W - 1)     acc += -1 * (w - w0) * plane[h0 * W + w1];
        if (h1 <= H - 1 && w0 >= 0)     acc += (w0 + 1 - w) * plane[h1 * W + w0];
        if (h1 <= H - 1 && w1 <= W - 1) acc += (w - w0) * plane[h1 * W + w1];
    } else {
        if (h0 >= 0 && w0 >= 0)         acc += -1 * (h0 + 1 - h) * plane[h0 * W + w0];
        if (h0 >= 0 && w1 <= W - 1)     acc += (h0 + 1 - h) * plane[h0 * W + w1];
        if (h1 <= H - 1 && w0 >= 0)     acc += -1 * (h - h0) * plane[h1 * W + w0];
        if (h1 <= H - 1 && w1 <= W - 1) acc += (h - h0) * plane[h1 * W + w1];
    }
    return acc;
}

/* Sampling position of (output pixel, tap) -- im2col_cpu.cpp:166-177 */
static inline void tap_geometry(const dcn_geom *g, const float *offset, const float *mask,
                                int grp, int ho, int wo, int i, int j,
                                float *h_im, float *w_im, float *m)
{
    const int HW = g->Ho * g->Wo, k = i * g->kw + j;
    const float *off = offset + (size_t)grp * 2 * g->kh * g->kw * HW;
    const float *msk = mask + (size_t)grp * g->kh * g->kw * HW;
    float dh = off[(size_t)(2 * k) * HW + ho * g->Wo + wo];
    float dw = off[(size_t)(2 * k + 1) * HW + ho * g->Wo + wo];
    *m = msk[(size_t)k * HW + ho * g->Wo + wo];
    *h_im = (ho * g->stride_h - g->pad_h) + i * g->dil_h + dh;
    *w_im = (wo * g->stride_w - g->pad_w) + j * g->dil_w + dw;
}

/* im2col_cpu.cpp:127-196 : columns[(c*kh*kw + k), ho*Wo+wo] = mask * sample */
void dcn_ref_im2col(const float *im, const float *offset, const float *mask,
                    const dcn_geom *g, float *columns)
{
    const int HW = g->Ho * g->Wo, KK = g->kh * g->kw, cpg = g->C / g->dg;
#pragma omp parallel for schedule(static)
    for (int ho = 0; ho < g->Ho; ++ho)
        for (int wo = 0; wo < g->Wo; ++wo)
            for (int grp = 0; grp < g->dg; ++grp)
                for (int i = 0; i < g->kh; ++i)
                    for (int j = 0; j < g->kw; ++j) {
                        float h_im, w_im, m;
                        tap_geometry(g, offset, mask, grp, ho, wo, i, j, &h_im, &w_im, &m);
                        const int inside = (h_im > -1 && w_im > -1 && h_im < g->H && w_im < g->W);
                        for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) {
                            float val = 0.f;
                            if (inside)
                                val = bilinear_at(im + (size_t)c * g->H * g->W, g->H, g->W, h_im, w_im);
                            columns[((size_t)c * KK + i * g->kw + j) * HW + ho * g->Wo + wo] = val * m;
                        }
                    }
}

/* im2col_cpu.cpp:198-257 with :58-82 folded in.  The reference visits a 5x5
 * neighbourhood of trunc(coord) and keeps the cells with |d|<1 whose
 * get_gradient_weight is non-zero -- i.e. exactly the (up to) four bilinear
 * corners, weighted (h+1-a_h)/(a_h+1-h) x (w+1-a_w)/(a_w+1-w).               */
void dcn_ref_col2im(const float *columns, const float *offset, const float *mask,
                    const dcn_geom *g, float *grad_im)
{
    const int HW = g->Ho * g->Wo, KK = g->kh * g->kw, cpg = g->C / g->dg;
    /* parallel over channels: each channel owns its grad plane -> no races */
#pragma omp parallel for schedule(static)
    for (int c = 0; c < g->C; ++c) {
        const int grp = c / cpg;
        float *gplane = grad_im + (size_t)c * g->H * g->W;
        for (int i = 0; i < g->kh; ++i)
            for (int j = 0; j < g->kw; ++j)
                for (int ho = 0; ho < g->Ho; ++ho)
                    for (int wo = 0; wo < g->Wo; ++wo) {
                        float a_h, a_w, m;
                        tap_geometry(g, offset, mask, grp, ho, wo, i, j, &a_h, &a_w, &m);
                        const float top = columns[((size_t)c * KK + i * g->kw + j) * HW + ho * g->Wo + wo] * m;
                        if (a_h <= -1 || a_h >= g->H || a_w <= -1 || a_w >= g->W) continue;
                        const int h0 = (int)floorf(a_h), w0 = (int)floorf(a_w);
                        for (int dy = 0; dy <= 1; ++dy)
                            for (int dx = 0; dx <= 1; ++dx) {
                                const int h = h0 + dy, w = w0 + dx;
                                if (h < 0 || h >= g->H || w < 0 || w >= g->W) continue;
                                if (!(fabsf(a_h - h) < 1 && fabsf(a_w - w) < 1)) continue;
                                const float wh = dy ? (a_h + 1 - h) : (h + 1 - a_h);
                                const float ww = dx ? (a_w + 1 - w) : (w + 1 - a_w);
                                gplane[h * g->W + w] += wh * ww * top;
                            }
                    }
    }
}

/* im2col_cpu.cpp:259-329 : grad wrt offsets (2 per tap) and mask (1 per tap) */
void dcn_ref_col2im_coord(const float *columns, const float *im, const float *offset,
                          const float *mask, const dcn_geom *g,
                          float *grad_offset, float *grad_mask)
{
    const int HW = g->Ho * g->Wo, KK = g->kh * g->kw, cpg = g->C / g->dg;
#pragma omp parallel for schedule(static)
    for (int ho = 0; ho < g->Ho; ++ho)
        for (int wo = 0; wo < g->Wo; ++wo)
            for (int grp = 0; grp < g->dg; ++grp)
                for (int i = 0; i < g->kh; ++i)
                    for (int j = 0; j < g->kw; ++j) {
                        float a_h, a_w, m;
                        tap_geometry(g, offset, mask, grp, ho, wo, i, j, &a_h, &a_w, &m);
                        const int k = i * g->kw + j;
                        const int outside = (a_h <= -1 || a_w <= -1 || a_h >= g->H || a_w >= g->W);
                        float gh = 0.f, gw = 0.f, gm = 0.f;
                        for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) {
                            const float *plane = im + (size_t)c * g->H * g->W;
                            const float col = columns[((size_t)c * KK + k) * HW + ho * g->Wo + wo];
                            if (!outside) {
                                gm += col * bilinear_at(plane, g->H, g->W, a_h, a_w);
                                gh += coord_weight(plane, g->H, g->W, a_h, a_w, 0) * col * m;
                                gw += coord_weight(plane, g->H, g->W, a_h, a_w, 1) * col * m;
                            }
                        }
                        const size_t obase = (size_t)grp * 2 * KK * HW;
                        grad_offset[obase + (size_t)(2 * k) * HW + ho * g->Wo + wo] = gh;
                        grad_offset[obase + (size_t)(2 * k + 1) * HW + ho * g->Wo + wo] = gw;
                        grad_mask[(size_t)grp * KK * HW + (size_t)k * HW + ho * g->Wo + wo] = gm;
                    }
}

/* ---- small dense helpers (row-major), fp32 accumulate like sgemm ---------- */
/* C[M,N] (+)= A[M,K] * B[K,N] */
static void gemm_nn(int M, int N, int K, const float *A, const float *B, float *C, int accumulate)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        float *c = C + (size_t)m * N;
        if (!accumulate) memset(c, 0, sizeof(float) * N);
        for (int k = 0; k < K; ++k) {
            const float a = A[(size_t)m * K + k];
            const float *b = B + (size_t)k * N;
            for (int n = 0; n < N; ++n) c[n] += a * b[n];
        }
    }
}
/* C[M,N] = A^T[M,K] * B[K,N] with A stored [K,M] */
static void gemm_tn(int M, int N, int K, const float *A, const float *B, float *C)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        float *c = C + (size_t)m * N;
        memset(c, 0, sizeof(float) * N);
        for (int k = 0; k < K; ++k) {
            const float a = A[(size_t)k * M + m];
            const float *b = B + (size_t)k * N;
            for (int n = 0; n < N; ++n) c[n] += a * b[n];
        }
    }
}
/* C[M,N] += A[M,K] * B^T[K,N] with B stored [N,K] */
static void gemm_nt_acc(int M, int N, int K, const float *A, const float *B, float *C)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            const float *a = A + (size_t)m * K, *b = B + (size_t)n * K;
            float s = 0.f;
            for (int k = 0; k < K; ++k) s += a[k] * b[k];
            C[(size_t)m * N + n] += s;
        }
}

static void fill_geom(dcn_geom *g, int C, int H, int W, int kh, int kw, int sh, int sw,
                      int ph, int pw, int dh, int dw, int dg)
{
    g->C = C; g->H = H; g->W = W; g->kh = kh; g->kw = kw;
    g->stride_h = sh; g->stride_w = sw; g->pad_h = ph; g->pad_w = pw;
    g->dil_h = dh; g->dil_w = dw; g->dg = dg;
    g->Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;   /* dcn_v2_cpu.cpp:59-60 */
    g->Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
}

/* dcn_v2_cpu.cpp:17-107 */
int dcn_ref_forward(const float *input, const float *weight, const float *bias,
                    const float *offset, const float *mask, float *output,
                    int B, int C, int H, int W, int Cout, int kh, int kw,
                    int sh, int sw, int ph, int pw, int dh, int dw, int dg)
{
    dcn_geom g; fill_geom(&g, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, dg);
    const int HW = g.Ho * g.Wo, K = C * kh * kw;
    float *columns = (float *)malloc(sizeof(float) * (size_t)K * HW);
    if (!columns) return -1;
    for (int b = 0; b < B; ++b) {
        float *out = output + (size_t)b * Cout * HW;
        for (int o = 0; o < Cout; ++o)                       /* :82-85 bias broadcast */
            for (int p = 0; p < HW; ++p) out[(size_t)o * HW + p] = bias[o];
        dcn_ref_im2col(input + (size_t)b * C * H * W,
                       offset + (size_t)b * 2 * dg * kh * kw * HW,
                       mask + (size_t)b * dg * kh * kw * HW, &g, columns);
        gemm_nn(Cout, HW, K, weight, columns, out, 1);       /* :101-104 out += W*col */
    }
    free(columns);
    return 0;
}

/* dcn_v2_cpu.cpp:109-233 ; grads are overwritten (the reference zero-inits them) */
int dcn_ref_backward(const float *input, const float *weight, const float *bias,
                     const float *offset, const float *mask, const float *grad_output,
                     float *grad_input, float *grad_offset, float *grad_mask,
                     float *grad_weight, float *grad_bias,
                     int B, int C, int H, int W, int Cout, int kh, int kw,
                     int sh, int sw, int ph, int pw, int dh, int dw, int dg)
{
    (void)bias;
    dcn_geom g; fill_geom(&g, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, dg);
    const int HW = g.Ho * g.Wo, K = C * kh * kw;
    float *columns = (float *)malloc(sizeof(float) * (size_t)K * HW);
    if (!columns) return -1;
    memset(grad_input, 0, sizeof(float) * (size_t)B * C * H * W);
    memset(grad_weight, 0, sizeof(float) * (size_t)Cout * K);
    memset(grad_bias, 0, sizeof(float) * Cout);
    for (int b = 0; b < B; ++b) {
        const float *in_b = input + (size_t)b * C * H * W;
        const float *off_b = offset + (size_t)b * 2 * dg * kh * kw * HW;
        const float *msk_b = mask + (size_t)b * dg * kh * kw * HW;
        const float *go_b = grad_output + (size_t)b * Cout * HW;
        gemm_tn(K, HW, Cout, weight, go_b, columns);                     /* :176-179 */
        dcn_ref_col2im_coord(columns, in_b, off_b, msk_b, &g,            /* :182-191 */
                             grad_offset + (size_t)b * 2 * dg * kh * kw * HW,
                             grad_mask + (size_t)b * dg * kh * kw * HW);
        dcn_ref_col2im(columns, off_b, msk_b, &g,                         /* :193-200 */
                       grad_input + (size_t)b * C * H * W);
        dcn_ref_im2col(in_b, off_b, msk_b, &g, columns);                 /* :203-210 */
        gemm_nt_acc(Cout, K, HW, go_b, columns, grad_weight);            /* :216-219 */
        for (int o = 0; o < Cout; ++o) {                                  /* :224-227 */
            float s = 0.f;
            for (int p = 0; p < HW; ++p) s += go_b[(size_t)o * HW + p];
            grad_bias[o] += s;
        }
    }
    free(columns);
    return 0;
}

/* im2col only, for a Python caller that does the GEMM itself (faster for big shapes) */
int dcn_ref_im2col_image(const float *im, const float *offset, const float *mask, float *columns,
                         int C, int H, int W, int kh, int kw, int sh, int sw,
                         int ph, int pw, int dh, int dw, int dg)
{
    dcn_geom g; fill_geom(&g, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, dg);
    dcn_ref_im2col(im, offset, mask, &g, columns);
    return 0;
}
