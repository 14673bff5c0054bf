/*
 * include/monoflex_hip.h -- C ABI of libmonoflex_hip.so (gfx950 / MI355X).
 *
 * Plain pointers and sizes only; no torch types.  Every pointer is a DEVICE pointer unless a
 * comment says otherwise; `stream` is a hipStream_t passed as void*.  Every entry point returns 0
 * on success or a negative error code and records a message readable with mfx_last_error().
 * Kernels are enqueued on `stream` and never synchronise (hipGraph-capturable).
 *
 * Two groups of entry points:
 *  (1) The reference's native boundary for this path -- what its pybind module `_ext` binds
 *      (/root/reference/model/backbone/DCNv2/src/vision.cpp:3-8):
 *        dcn_v2_forward   src/dcn_v2.h:9-46   -> mfx_dcn_v2_forward
 *        dcn_v2_backward  src/dcn_v2.h:48-92  -> mfx_dcn_v2_backward
 *      Same layouts as the reference (NCHW fp32; offset channel 2k = dh, 2k+1 = dw of tap k;
 *      mask channel k), same argument meaning; outputs are caller-allocated instead of ATen-allocated.
 *  (2) The NHWC operators the rest of the path is built from (they replace what the reference gets
 *      from cuDNN/cuBLAS/torch through nn.Conv2d, BatchNorm2d, MaxPool2d, ConvTranspose2d, topk, ...;
 *      call sites cited per function).
 */
#ifndef MONOFLEX_HIP_H
#define MONOFLEX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFX_ABI_VERSION 3      /* r06: mfx_dcn_desc.w_pair_f16 (fourth-generation DCN kernel).  r05: mfx_conv_desc / mfx_dcn_desc / mfx_heads_desc grew in r04 (paired fragments, fused offset conv, w2_scale, MFX_F16X2), mfx_dcn_desc again
                                * (per-axis geometry), mfx_gram_desc is new: a caller built against version 1 is rejected instead of being read past its structs */

/* element types of activations / packed weights */
enum { MFX_F32 = 0, MFX_BF16 = 1, MFX_F16 = 2 /* IEEE half: every operator that takes bf16 takes it (the fused heads and the LDS-patch DCN forward are inference
                                            * kernels in both); fp16 TRAINING needs loss scaling on the host side (engine/trainer.py) */,
       MFX_F16X2 = 3 /* split precision (inference GEMM kernels: conv2d, cat_conv1x1, dcn, heads_fused): activations are plain fp32 in memory (pass
                      * MFX_F32 to every other operator), each MFMA operand element is the pair fp16(x), fp16(x - fp16(x)) and a product is
                      * hi.hi + hi.lo + lo.hi + lo.lo on v_mfma_f32_16x16x32_f16 with fp32 accumulate: fp32-grade results (the reference computes
                      * in fp32: src/cuda/dcn_v2_cuda.cu:58) at 4x the matrix rate of MFX_F32.  Weights arrive pre-split: every 16-byte chunk of 4
                      * fp32 values of the MFX_F32 layouts is replaced by [4 hi halves | 4 lo halves] of the same 4 values. */ };
/* epilogue activations */
enum { MFX_ACT_NONE = 0, MFX_ACT_RELU = 1, MFX_ACT_LEAKY = 2 /* slope 0.01 */, MFX_ACT_DCN_OFFMASK = 3 /* sigmoid on ch 18..26 */ };
/* error codes */
enum { MFX_OK = 0, MFX_ERR_ARG = -1, MFX_ERR_UNSUPPORTED = -2, MFX_ERR_LAUNCH = -3, MFX_ERR_WORKSPACE = -4 };

int mfx_abi_version(void);
const char* mfx_last_error(void);
/* Threading: one host thread per process drives the library, like the reference (one process per GPU). Entry points
 * only enqueue work on the stream they are given and never synchronise; mfx_last_error() is thread-local; the options below and
 * the one-time kernel attribute setup are process-wide and unsynchronised -- set options before concurrent use.
 * Tuning/debug overrides: "conv_tile" | "dcn_tile" | "cat_tile" (tile id, 0 = automatic), "kc" (4 | 8 | 0),
 * "halo" (0 = generic kernel only, 1 = automatic, 2.. = force LDS-halo variant), "halo_cg", "dcn_wave", "dcn_patch" (same
 * convention), "ksplit" / "dcn_ksplit" (split-K factor), "topk_strips" (row strips of the top-K stage, 1 = single workgroup),
 * "wgrad_*" / "dcn_wgrad_m" (training GEMM partitioning), "heads_persist" (1 = one workgroup per resident slot over (tile, branch)
 * unit ranges, n > 1 = n workgroups, 0 = one workgroup per tile), "heads_planes", "heads_dbg" (timing probes: wrong results).
 * Unknown names return MFX_ERR_ARG. */
int mfx_set_option(const char* name, int value);
/* every switch back to its load-time value (what a test harness calls between tests) */
int mfx_reset_options(void);
/* the current values become the load-time values mfx_reset_options() restores (called once by the host after it applied the process's own
 * switches, e.g. MFX_OPTIONS from the environment) */
int mfx_commit_options(void);
/* Dispatch counters since process start, so a test can assert WHICH kernel variant a call took: "dcn_bt_fused" = launches of the
 * fused sample + weight-gradient kernel of mfx_dcn_backward_v2 (selected for bf16 / fp16, C = Cout = 64, W % 32 == 0 and at least
 * option "dcn_bt_fuse_min_chunks" (default 1024) 32-pixel chunks).  Unknown names return MFX_ERR_ARG (negative). */
long mfx_get_counter(const char* name);

/* Range sentinel of the split-precision mode (dtype MFX_F16X2: fp32 activations become fp16 (hi, lo) MFMA operand pairs; hi overflows above
 * 65504): 1 if any activation converted since the last reset was outside fp16's range or not finite, 0 if none, < 0 on error; `reset` clears the
 * flag.  Blocking device read: synchronise the streams that ran the kernels first. */
int mfx_f16x2_range_check(int reset);

/* ------------------------------------------------------------------------------------------
 * (1) reference `_ext` boundary
 * ------------------------------------------------------------------------------------------ */

/* Scratch bytes mfx_dcn_v2_forward / _backward need for the given shape: an upper bound over every deformable_group. */
size_t mfx_dcn_v2_workspace_bytes(int B, int C, int H, int W, int Cout, int kh, int kw,
                                  int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                  int backward);
/* The same for a known deformable_group -- what the entries check against: one group needs none of the group loop's temporaries (about half). */
size_t mfx_dcn_v2_workspace_bytes_g(int B, int C, int H, int W, int Cout, int kh, int kw,
                                    int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                    int deformable_group, int backward);

/* output (B,Cout,Ho,Wo) = bias + W * (mask . bilinear(input @ offsets))   -- src/dcn_v2.h:9-23, as general as the reference's entry:
 * any deformable_group dividing C (channel group g is sampled with its own 2*kh*kw offset and kh*kw mask channels,
 * src/cuda/dcn_v2_im2col_cuda.cu:147-156: the groups are looped INSIDE this call, their outputs summed), per-axis stride / padding / dilation.
 * kh * kw <= 9 (the offset row of the NHWC kernels holds nine taps). */
int mfx_dcn_v2_forward(const float* input, const float* weight, const float* bias,
                       const float* offset, const float* mask, float* output,
                       int B, int C, int H, int W, int Cout, int kh, int kw,
                       int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                       int deformable_group, void* workspace, size_t workspace_bytes, void* stream);

/* grads wrt input, offset, mask, weight, bias (all overwritten)            -- src/dcn_v2.h:48-59 */
int mfx_dcn_v2_backward(const float* input, const float* weight, const float* bias,
                        const float* offset, const float* mask, const float* grad_output,
                        float* grad_input, float* grad_offset, float* grad_mask,
                        float* grad_weight, float* grad_bias,
                        int B, int C, int H, int W, int Cout, int kh, int kw,
                        int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                        int deformable_group, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * (2) NHWC operators
 * ------------------------------------------------------------------------------------------ */

#define MFX_MAX_SEG 9

/* Implicit-GEMM convolution, NHWC in / NHWC out, fused y = act(conv(x)*scale + shift (+ res)).
 * Replaces nn.Conv2d + BatchNorm2d(eval) + ReLU (+ residual add) of dla_dcn.py:84-98,195-203,
 * 268-276,312-322, the DCN offset/mask conv (dcn_v2.py:104-122) and the head convs.
 * Weights are pre-packed [Cout_pad][K_pad], K = (tap, channel) with Ck elements per tap. */
typedef struct {
    const void* x;            /* input [B][H][W][x_pixstride]                                  */
    const void* w;            /* packed weights [Cout_pad][K_pad], element type = dtype        */
    const void* w_frag;       /* optional fragment-major copy [Cout_pad/16][K_pad/(64 B)][4 kq][16 n][16 B]: lets the
                                 3x3/stride-1 LDS-halo kernel fetch each MFMA weight fragment as one contiguous KiB */
    const float* scale;       /* [Cout_pad] or NULL (=1)                                        */
    const float* shift;       /* [Cout_pad] or NULL (=0)                                        */
    const void* res;          /* residual [M][ldres] (element type = dtype) or NULL            */
    void* y;                  /* output [M][ldy], element type = out_dtype                      */
    const int32_t* rowmap;    /* optional [M]: output row -> pixel index b*Ho*Wo+oh*Wo+ow, -1 = zero row */
    int32_t B, H, W;
    int32_t x_pixstride;      /* elements between consecutive input pixels                      */
    int32_t Ck;               /* K elements per tap: power of two, >= 16 bytes worth            */
    int32_t kh, kw, stride, pad_h, pad_w, dil_w;
    int32_t Ho, Wo;
    int32_t M;                /* rows: B*Ho*Wo, or the rowmap length                            */
    int32_t Cout;             /* valid output channels (multiple of 4 for f32 out, 8 for bf16 / fp16)  */
    int32_t Cout_pad;         /* weight rows; multiple of 16, and of 64 when > 32                */
    int32_t K_pad;            /* multiple of 64 bytes of K                                      */
    int32_t ldy, ldres;
    int32_t act;
    int32_t dtype, out_dtype;
    void* workspace;          /* optional scratch for split-K (small-M / long-K layers): fp32, >= ksplit*M*Cout_pad*4 bytes  */
    int64_t workspace_bytes;  /* 0 / NULL: never split                                           */
    /* optional: train-mode BN statistics of the OUTPUT accumulated by the conv's own epilogue.  stats = the BN layer's scratch
     * ([stats_ncopy][2*Cout] fp32 sums | sums of squares of the values as stored, see mfx_bn_train_fwd; stats_ncopy = mfx_bn_ncopy(Cout));
     * *stats_done is set to 1 when the kernel that ran supports it (the LDS-halo 3x3 kernels), else to 0 and nothing is added. */
    float* stats;
    int stats_ncopy;
    int* stats_done;
    /* optional, dtype MFX_F16X2 only: w_frag with its K steps re-packed in pairs, [Cout_pad/16][K_pad/(128 B)][hi | lo][64 lanes][16 B], a lane's
     * chunk = [its two hi (lo) dwords of step 2p | of step 2p+1] (one 8-element fp16 MFMA operand).  Lets the LDS-halo kernel form three
     * products per step pair instead of four (Ck >= 32). */
    const void* w_frag_pair;
} mfx_conv_desc;
int mfx_conv2d_nhwc(const mfx_conv_desc* d, void* stream);

/* 1x1 convolution over a virtual channel concat of up to MFX_MAX_SEG sources (DLA Root,
 * dla_dcn.py:195-203): no concatenated tensor is materialised.  Every segment contributes
 * Cseg channels (power of two) read at src[s] + pixel*stride[s] + off[s]. */
typedef struct {
    const void* src[MFX_MAX_SEG];
    int32_t stride[MFX_MAX_SEG];
    int32_t off[MFX_MAX_SEG];
    int32_t nseg, Cseg;
    const void* w; const float* scale; const float* shift; const void* res; void* y;
    int32_t M, Cout, Cout_pad, K_pad, ldy, ldres, act, dtype;
} mfx_cat_desc;
int mfx_cat_conv1x1_nhwc(const mfx_cat_desc* d, void* stream);

/* Fused modulated deformable convolution, NHWC: bilinear gather straight into the LDS A tile,
 * MFMA against the packed weights, + scale/shift (bias and BN folded) + activation; the reference's
 * `columns` buffer (src/cuda/dcn_v2_cuda.cu:139-163) is never written.
 * offmask: fp32 [B*Ho*Wo][32]: ch 2k = dh, 2k+1 = dw, 18+k = mask (already sigmoided), kh*kw <= 9. */
typedef struct {
    const void* x; const float* offmask; const void* w; const float* scale; const float* shift; void* y;
    const void* w_frag;       /* optional fragment-major weights (see mfx_conv_desc.w_frag): enables the 2nd-generation kernel */
    int32_t B, H, W, C;       /* C: power of two >= 64 (16-bit) / 16 (f32) elements                  */
    int32_t kh, kw, stride, pad, dil;
    int32_t Ho, Wo, Cout, Cout_pad, K_pad, ldy, act, dtype;
    const void* w_frag_f16;   /* optional (16-bit modes): the fragment-major weights as IEEE fp16: enables the LDS-patch kernel,
                                 which samples and multiplies in fp16 (3x3, stride 1, pad 1, C % 64 == 0)            */
    void* workspace;          /* optional fp32 scratch for split-K on small maps (>= 9*M*Cout_pad*4 bytes to allow every split) */
    int64_t workspace_bytes;
    /* The module's own offset/mask conv (reference dcn_v2.py:118-122: 3x3 / stride 1 / pad 1 on the SAME input, 27 channels padded to 32,
     * sigmoid on channels 18..26), optional: where the LDS-patch kernel runs it computes the offsets itself -- no separate conv launch, no
     * offmask round trip (mfx_dcn_fuses_offset_conv tells; `offmask` may then be NULL).  Elsewhere these fields are ignored and
     * `offmask` must hold the conv's output. */
    const void* off_w_frag_f16;   /* fragment-major IEEE fp16 weights [2][18][64 lanes][16 B] of that conv (K = 9 * 64)                      */
    const float* off_shift;       /* its bias, fp32 [32] (27 values, then zeros)                                                          */
    float* offmask_out;           /* optional: the fused kernel also writes the rows it computed, fp32 [B*H*W][32] (the backward pass reads them) */
    /* per-axis geometry (reference src/dcn_v2.h:9-23 takes stride_h/w, pad_h/w, dilation_h/w): nonsquare = 1 -> `stride`, `pad`, `dil` are the
     * ROW values and the three fields below the COLUMN values (generic gather kernel); 0 -> they are ignored (square geometry) */
    int32_t nonsquare, stride_w, pad_w, dil_w;
    /* optional (16-bit modes, 3x3 / stride 1 / pad 1, Cout_pad == 64): the weights as IEEE fp16 in the fourth-generation LDS kernel's K order --
     * 16-channel slices, five k-steps per slice, k-step j = taps (2j, 2j + 1) x 16 channels (tap 9 = zeros): [Cout_pad / 16][C / 16 * 5][64 lanes][16 B],
     * lane (kq, n) holding tap 2j + (kq >> 1), channels 16 s + 8 (kq & 1) .. + 7 of output channel 16 nf + n (ops.dcn_pair_fragments).  Needs
     * w_frag_f16 as well (samples that leave the LDS patch are multiplied with that one). */
    const void* w_pair_f16;
} mfx_dcn_desc;
int mfx_dcn_nhwc(const mfx_dcn_desc* d, void* stream);
/* 1 when mfx_dcn_nhwc(d) will compute the offsets inside the kernel from off_w_frag_f16 / off_shift (d->offmask is not read), else 0 */
int mfx_dcn_fuses_offset_conv(const mfx_dcn_desc* d);

/* DCNv2 as "project, then sample" (r06; 3x3 / stride 1 / pad 1 / dilation 1, 16-bit maps), the sampling half.  Bilinear interpolation commutes with the
 * contraction over input channels, so the module (dcn_v2_im2col_cuda.cu:125-195 + dcn_v2_cuda.cu:139-163) is
 *     y[m][n] = act(scale[n] * sum_tap mask[m,tap] * bilinear(P[:, :, tap*N + n] @ p(m,tap)) + shift[n]),    P = the 1x1 convolution of the input with the
 * DCN weights regrouped as [(tap, n)][c] (one mfx_conv2d_nhwc call, C -> 9 N, no scale / shift / activation).  P: [B*H*W][9*N] in `dtype`;
 * offmask as for mfx_dcn_nhwc; N in {64, 128, 256}; corners outside the image contribute zero; fp32 accumulation. */
int mfx_dcn_sample_nhwc(const void* P, const float* offmask, const float* scale, const float* shift, void* y,
                        int B, int H, int W, int N, int ldy, int act, int dtype, void* stream);
/* The projection half as a kernel of its own (csrc/gemm_as.hip): y[m][n] = sum_k x[m][k] * w[n][k], x [M][ldx >= K] and y [M][ldy >= N] in `dtype`
 * (MFX_BF16 / MFX_F16), w [N][K] K-contiguous (mfx_conv2d_nhwc's 1x1 weight layout), K in {64, 128, 256, 512}, N a multiple of 64; fp32 accumulate, no
 * epilogue.  Activation-stationary: a workgroup keeps 128 pixels' rows in registers and streams the output channels (the map is store-bound). */
int mfx_project_nhwc(const void* x, const void* w, void* y, int M, int K, int N, int ldx, int ldy, int dtype, void* stream);

/* DLA stem in bf16 mode: 7x7 / stride 1 / pad 3 conv of the fp32 NCHW image batch (B,3,H,W) -> NHWC (B,H,W,16) bf16 with
 * scale/shift (folded BN) + activation (dla_dcn.py:268-272).  Reads the image planes directly (no padded NHWC copy);
 * w: bf16 [16][K_pad], K = 7 rows x 4 super-taps x (2 pixels x 4 channels) as built by the host packer. */
int mfx_stem_conv7x7_nchw(const float* images, const void* w, const float* scale, const float* shift, void* y,
                          int B, int H, int W, int Cout, int K_pad, int act, int dtype, void* stream);

/* DLA F1 in one kernel (inference, 16-bit maps): stem 7x7 (3 -> 16) -> level0 3x3 (16 -> 16) -> level1 3x3 / stride 2 (16 -> 32), each + folded
 * BN + ReLU (dla_dcn.py:268-276, 312-331); both full-resolution 16-channel maps stay in LDS, only the (B, H/2, W/2, 32) map is written.
 * w_stem [16][stem_kpad] (the stem kernel's super-tap order, 224 used), w_l0 [16][160], w_l1 [32][160] with k = tap*16 + c (zero beyond 144), all in `dtype`
 * (MFX_BF16 / MFX_F16); H, W even. */
int mfx_f1_fused(const float* images, const void* w_stem, const float* sc_stem, const float* sh_stem,
                 const void* w_l0, const float* sc_l0, const float* sh_l0, const void* w_l1, const float* sc_l1, const float* sh_l1,
                 void* y, int B, int H, int W, int stem_kpad /* row length of w_stem, >= 224 */, int dtype, void* stream);

/* 2x2/stride-2 max pooling (dla_dcn.py:237-238), NHWC, C % (16 bytes) == 0 */
int mfx_maxpool2x2_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);

/* depthwise ConvTranspose2d(k=2f, stride=f, pad=f/2) + skip add (dla_dcn.py:409-411,419-425):
 * y[b,oh,ow,c] = sum x[b,ih,iw,c]*w[c][kh][kw] + skip[b,oh,ow,c];  w fp32 [2f*2f][C] (tap-major); skip may be NULL */
int mfx_upsample_add_nhwc(const void* x, const float* w, const void* skip, void* y,
                          int B, int H, int W, int C, int f, int dtype, void* stream);

/* layout helpers for the NCHW fp32 boundary */
int mfx_nchw_to_nhwc(const float* x, void* y, int B, int C, int H, int W, int ldy, int dtype, void* stream);
int mfx_nhwc_to_nchw(const void* x, float* y, int B, int C, int H, int W, int ldx, int dtype, void* stream);
/* network input: NCHW fp32 (B,3,H,W) -> zero-padded NHWC4 [B][H+2*pad][W+2*pad_w][4] (stem, dla_dcn.py:268-272) */
int mfx_pack_image_nhwc4(const float* x, void* y, int B, int H, int W, int pad_h, int pad_w_left, int pad_w_right,
                         int dtype, void* stream);

/* Nine head branches fused (detector_predictor.py:47-96,125-134): per branch
 * 3x3 conv 64->256 (no bias) -> BN(+leaky 0.01) -> 1x1 conv 256->c_k (+bias); the 256-ch trunks
 * never reach HBM.  out: fp32 [M][ld_out]; branch k writes c_out[k] channels at ch_off[k]. */
typedef struct {
    const void* x;            /* feature [B][H][W][64]                                          */
    const void* w1;           /* [nbranch*256][K_pad] packed 3x3 weights                        */
    const float* scale1; const float* shift1;   /* [nbranch*256] folded BN                      */
    const void* w2;           /* [nbranch][32][256] packed 1x1 weights (rows >= c_out zero)     */
    const float* bias2;       /* [nbranch][32]                                                   */
    float* out;               /* [M][ld_out] fp32                                                */
    float* planar;            /* optional fp32 [B][planar_c][H*W]: branch 0's channels again, class-planar */
    int32_t B, H, W, nbranch, K_pad, ld_out, dtype, planar_c;
    int32_t ch_off[16]; int32_t c_out[16];
    float w2_scale[16];       /* per branch: the 1x1 sums are multiplied by this before the bias (0 = 1): MFX_F16X2 weights are packed
                               * times a power of two so that their lo halves are normal fp16 numbers, and un-scaled here */
    /* optional (MFX_BF16 / MFX_F16; option "heads_mfma32"): the packs of the v_mfma_f32_32x32x16 form of the kernel --
     * w1_32 [nbranch][wn 4][K-step 36][rb 2][64 lanes][8]: lane (row = lane & 31, h = lane >> 5) = 3x3 weights of trunk channel 64 wn + 32 rb + row, k = 16 s + 8 h ..;
     * w2_32 [nbranch][wn 4][rb 2][t 2][64 lanes][8]: lane (o, h) = W2[o][64 wn + 32 rb + 16 t + 8 (e >> 2) + 4 h + (e & 3)] (rows >= c_out zero) */
    const void* w1_32; const void* w2_32;
} mfx_heads_desc;
int mfx_heads_fused(const mfx_heads_desc* d, void* stream);

/* Edge fusion tail (detector_predictor.py:152-158): out[b, y, x, ch_off + c] += v[b][i][c] for the
 * first edge_len[b] border points i of image b.  v: fp32 [B][L][ldv], edge_xy: int32 [B][L][2]. */
int mfx_edge_scatter_add(float* out, int ld_out, int ch_off, int C, const float* v, int ldv,
                         const int32_t* edge_xy, const int32_t* edge_len, int B, int L, int H, int W,
                         float* planar /* optional [B][C][H*W], updated too */, void* stream);

/* Decode stage 1 (layers/utils.py:39-58,61-77): per (image, class) sigmoid+clamp, 3x3 max NMS, top-K.
 * Class logit of (image b, class c, pixel p) is hmap[b*b_stride + c*c_stride + p*p_stride] (elements): either the
 * NHWC head map (c_stride 1, p_stride ld) or a planar (B,ncls,H*W) copy (c_stride H*W, p_stride 1: coalesced).
 * Outputs [B][ncls][K], sorted by descending score.  Ties are broken towards the lower flat index
 * (torch.topk leaves the order unspecified).
 * With a workspace of mfx_decode_topk_workspace_bytes() each map is cut into row strips reduced by separate workgroups and
 * merged by a second launch (same result, ~4x shorter on a 256-CU device); without one a single workgroup per map does it. */
size_t mfx_decode_topk_workspace_bytes(int ncls, int B, int K);
int mfx_decode_topk(const float* hmap, long b_stride, long c_stride, long p_stride, int ncls, int B, int H, int W, int K,
                    float* scores, int32_t* index, void* workspace, size_t workspace_bytes, void* stream);

/* Decode stage 2 (layers/utils.py:88-100,120-145; detector_infer.py:96-232; anno_encoder.py:69-295):
 * merge ncls*K -> K, gather the 50 regression channels, 3D box recovery.
 * calib: fp32 [B][6] = f_u,f_v,c_u,c_v,b_x,b_y; pad: int32 [B][2]; img_size: int32 [2] = (W,H) of image 0.
 * det: fp32 [B][K][14] rows [cls,alpha,x1,y1,x2,y2,h,w,l,x,y,z,ry,score], sorted by merged score;
 * topk: fp32 [B][K][5] = score, flat index, cls, y, x;  valid: int32 [B][K] (score >= threshold). */
int mfx_decode_boxes(const float* hmap, int ld, int reg_off, const float* scores, const int32_t* index,
                     int ncls, int B, int H, int W, int K, const float* calib, const int32_t* pad,
                     const int32_t* img_size, float threshold, float* det, float* topk, int32_t* valid,
                     void* stream);
/* The same with the reference's other `output_depth` settings (detector_infer.py:149-198; engine/inference.py:154 walks them for
 * `eval_all_depths`): which of the four depth estimates (direct, keypoint centre / 02 / 13) becomes the box depth, and which uncertainty
 * scales the score.  mfx_decode_boxes is MFX_DEPTH_SOFT (runs/monoflex.yaml).  'oracle' needs ground truth and is not a decode mode here. */
enum { MFX_DEPTH_SOFT = 0, MFX_DEPTH_HARD = 1, MFX_DEPTH_MEAN = 2, MFX_DEPTH_DIRECT = 3, MFX_DEPTH_KEYPOINTS_AVG = 4,
       MFX_DEPTH_KEYPOINTS_CENTER = 5, MFX_DEPTH_KEYPOINTS_02 = 6, MFX_DEPTH_KEYPOINTS_13 = 7 };
int mfx_decode_boxes_mode(const float* hmap, int ld, int reg_off, const float* scores, const int32_t* index,
                          int ncls, int B, int H, int W, int K, const float* calib, const int32_t* pad,
                          const int32_t* img_size, float threshold, int depth_mode, float* det, float* topk, int32_t* valid,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * (3) training path (reference: autograd over nn.Conv2d / BatchNorm2d / MaxPool2d / ConvTranspose2d and
 *     `_ext.dcn_v2_backward`, driven by engine/trainer.py:116-117).  Data gradients of convolutions reuse
 *     mfx_conv2d_nhwc with host-packed flipped/transposed weights; everything else is below.
 * ------------------------------------------------------------------------------------------ */

/* conv weight gradient dw fp32 [Cout][kh*kw][Ck] (overwritten): dw[o][tap][c] = sum_m dy[m][o] * x[pixel(m,tap)][c] */
int mfx_conv_wgrad_nhwc(const void* x, const void* dy, float* dw, int B, int H, int W, int x_pixstride, int Ck,
                        int kh, int kw, int stride, int pad_h, int pad_w, int Ho, int Wo, int Cout, int ldy,
                        int dtype, void* stream);
/* same with a tap dilation along W (the bf16 stem: 8-element super-taps = two 4-channel pixels, kw = 4, dil_w = 2, pixel
 * stride 4 elements -- see mfx_stem_conv7x7_nchw / the host packer) */
int mfx_conv_wgrad_nhwc_dil(const void* x, const void* dy, float* dw, int B, int H, int W, int x_pixstride, int Ck,
                            int kh, int kw, int stride, int pad_h, int pad_w, int dil_w, int Ho, int Wo, int Cout, int ldy,
                            int dtype, void* stream);
/* same, written straight in the parameter's layout: dw fp32 (Cout_real, Cin_real, kh, kw); channels of dy beyond Cout_real
 * and of x beyond Cin_real (padding) are dropped */
int mfx_conv_wgrad_oihw(const void* x, const void* dy, float* dw, int B, int H, int W, int x_pixstride, int Ck,
                        int kh, int kw, int stride, int pad_h, int pad_w, int Ho, int Wo, int Cout, int ldy,
                        int Cout_real, int Cin_real, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* Stem weight gradient (7x7 / stride 1 / pad 3, 3 -> 16 channels, bf16; model/backbone/dla_dcn.py:268-272 base_layer conv): xp is
 * the zero-padded NHWC4 image the forward stem reads ((B, H+6, W+8, 4), mfx_pack_image_nhwc4), dy (B,H,W,16) bf16; dw fp32
 * [16][7][32] with dw[o][th][dx*4 + c] = d/dw[o][c][th][dx] (dx = 7 and c = 3 are padding).  workspace: >= 14336 bytes per
 * workgroup used (768 by default). */
int mfx_stem_wgrad_bf16(const void* xp, const void* dy, float* dw, int B, int H, int W, int Hp, int Wp, void* workspace,
                        size_t workspace_bytes, void* stream);
/* the same for either 16-bit activation type: dtype = MFX_BF16 or MFX_F16 */
int mfx_stem_wgrad_16(const void* xp, const void* dy, float* dw, int B, int H, int W, int Hp, int Wp, int dtype, void* workspace,
                      size_t workspace_bytes, void* stream);
/* (workspace: optional fp32 scratch; with it the bf16 kernel writes per-slab partial tiles and sums them in a second pass
 *  instead of accumulating with atomics, which lets it use 4x more workgroups) */
/* fp32 OIHW parameter -> packed operand [rows_pad][K_pad] of `dtype` (+ optional fragment-major copy, see mfx_conv_desc.w_frag).
 * mode 0: forward weights, row = o, k = tap*ck + c (ck >= Cin).  mode 1: data-gradient weights (kernel rotated by 180 degrees,
 * in/out swapped): row = c, k = tap'*ck + o (ck >= Cout = channels of dy).  Padding rows/columns are zero. */
int mfx_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int kh, int kw, int mode, void* packed, void* frag,
                         int rows_pad, int K_pad, int ck, int dtype, void* stream);
/* The same packing for MANY operands in one launch (every conv operand of a training step: the weights change once per step,
 * so they are re-packed once per step).  descs / prefix live in device memory.  Every operand is cut into chunks of
 * mfx_pack_chunk_elems() elements (of its rows_pad * K_pad); prefix[i] = first chunk of descriptor i in the concatenated chunk
 * index space, prefix[n] = total_chunks.  All operands of one call share `dtype`. */
int mfx_pack_chunk_elems(void);
typedef struct mfx_pack_desc {
    const float* w;               /* (Cout, Cin, kh, kw) fp32 parameter */
    void* packed;                 /* [rows_pad][K_pad] */
    void* frag;                   /* fragment-major copy, or NULL */
    int Cout, Cin, kh, kw, mode, rows_pad, K_pad, ck;
} mfx_pack_desc;
int mfx_pack_conv_weights_batched(const mfx_pack_desc* descs_dev, const long long* prefix_dev, int n, long long total_chunks,
                                  int dtype, void* stream);
/* AdamW over every parameter tensor in one launch (csrc/adamw.hip; reference solver/__init__.py:10-60 builds torch.optim.AdamW, whose step this
 * replaces on the device path: engine/trainer.py:121).  fp32 tensors; `descs_dev` / `prefix_dev` / `groups_dev` live on the device: tensor i owns the
 * chunks [prefix[i], prefix[i+1]) of mfx_adamw_chunk_elems() elements, prefix has n + 1 entries.  Arithmetic of torch's `_fused_adamw_` (capturable):
 * step counters (fp32 device scalars, one per tensor) are advanced first, learning rates are device scalars, bias corrections in double, decoupled
 * weight decay.  `found_inf` (device fp32 scalar or NULL): when non-zero the call changes nothing (the fp16 loss scaler's skipped step). */
typedef struct mfx_adamw_desc {
    void* p;                      /* parameter, updated in place */
    const void* g;                /* gradient */
    void* m;                      /* exp_avg */
    void* v;                      /* exp_avg_sq */
    float* step;                  /* step counter (device scalar) */
    long long numel;
    int group;                    /* index into the group table */
    int pad_;
} mfx_adamw_desc;
typedef struct mfx_adamw_group {
    const float* lr;              /* device scalar */
    float beta1, beta2, eps, weight_decay;
} mfx_adamw_group;
int mfx_adamw_chunk_elems(void);
int mfx_adamw_multi(const mfx_adamw_desc* descs_dev, const long long* prefix_dev, int n, long long total_chunks,
                    const mfx_adamw_group* groups_dev, const float* found_inf, void* stream);
/* out[c] = sum_m x[m*ld + c]  (bias gradients) */
int mfx_colsum(const void* x, float* out, long M, int C, int ld, int dtype, void* stream);
/* out[c] += sum_m x[m*ld + c]: the same without the zero fill (a caller that carves many `out` vectors from one arena zeroes the arena once) */
int mfx_colsum_add(const void* x, float* out, long M, int C, int ld, int dtype, void* stream);
/* train-mode BatchNorm over [M][C]: per-channel sum and sum of squares (fp32, overwritten) */
int mfx_bn_stats(const void* x, float* sum, float* sumsq, long M, int C, int dtype, void* stream);
/* per-channel epilogue of mfx_bn_stats: mean = sum/count, biased var, rstd, scale = gamma*rstd, shift = beta - mean*scale, and
 * (optional) running_mean/var <- (1-momentum)*old + momentum*(mean / unbiased var), like nn.BatchNorm2d in training mode.
 * gamma/beta fp32.  count = rows that entered the sums (all ranks when the sums were all-reduced). */
int mfx_bn_finalize(const float* sum, const float* sumsq, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, long count, float* mean, float* rstd, float* scale,
                    float* shift, int C, void* stream);
/* y = act(x*scale[c] + shift[c] (+ res)) */
int mfx_bn_act_fwd(const void* x, const float* scale, const float* shift, const void* res, void* y,
                   long M, int C, int act, int dtype, void* stream);
/* backward of y = act(gamma*(x-mean)*rstd + beta (+res)) given da = dL/dy and the saved output a = y:
 * sg[c] = sum g (= dbeta), sgx[c] = sum g*xhat (= dgamma), dx, and dres = g (optional); g = da * act'(a) */
int mfx_bn_act_bwd(const void* x, const void* a, const void* da, const float* mean, const float* rstd, const float* gamma,
                   float* sg, float* sgx, void* dx, void* dres, long M, int C, int act, int dtype, void* stream);
/* Two-launch train-mode BatchNorm (+act, +residual) and its backward for the single-process case: the statistics / gradient sums
 * are accumulated in a persistent per-layer `scratch` (mfx_bn_scratch_bytes() bytes, ZERO before the first call; every call leaves
 * it zero again), the finalize step (mean/rstd, running statistics with momentum and the unbiased variance, num_batches_tracked += 1;
 * torch.nn.BatchNorm2d train-mode semantics, model/backbone/dla_dcn.py BatchNorm calls) happens in the prologue of the apply
 * kernel.  mean/rstd (C floats each) are outputs for the backward.  Replaces stats + finalize + act_fwd (and reduce + apply +
 * two gradient copies) with two launches each and no zero-fill launches.
 * r06: where the map fits the registers of one co-resident grid (<= 32 MB per operand) and is large enough for it to pay, both entries run as ONE
 * launch -- sums, a grid barrier on words of the same scratch, then the element-wise pass from registers (options "bn_onepass" bit 0 backward /
 * bit 1 forward, "bn_onepass_min_chunks", "bn_onepass_fwd_min_chunks", "bn_onepass_grid"; not in deterministic mode).  Same expressions, same
 * outputs to rounding (only the summation order of the column sums differs).  Such a launch spins on other workgroups of ITSELF: at most one of
 * them may be in flight on a device, i.e. issue mfx_bn_train_* of one device on one stream (the training step does). */
size_t mfx_bn_scratch_bytes(void);
/* 1 if a one-pass launch gave up at its barrier since the last reset (bounded spin; its outputs are then wrong), 0 if none, < 0 on error.
 * Cannot happen with one training process per device.  Synchronises the device; engine/trainer.py asks wherever it reads the loss. */
int mfx_bn_onepass_stuck(int reset);
int mfx_bn_ncopy(int C);      /* copies of the [2C] sums inside the scratch (the conv epilogue adds into copy workgroup % ncopy) */
/* stats_done != 0: the producing conv already added the statistics of x to the scratch (mfx_conv_desc.stats): no statistics launch */
int mfx_bn_train_fwd(const void* x, const void* res, void* y, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, long long* num_batches_tracked, float momentum, float eps, long M, int C, int act,
                     int dtype, float* scratch, float* mean, float* rstd, int stats_done, void* stream);
int mfx_bn_train_bwd(const void* x, const void* a, const void* da, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, long M, int C, int act, int dtype,
                     float* scratch, void* stream);
/* (`a` = the forward output, read only for the activation's derivative.  When no residual entered the activation it may be
 * NULL: the sign of x*scale + shift is then recomputed from x, gamma, beta, mean, rstd with the forward's expression, which
 * drops one of the three input streams of both backward launches.) */
/* the two halves of mfx_bn_act_bwd, for synchronised BN (reference tools/plain_train_net.py:131-132, SyncBatchNorm):
 * reduce -> all-reduce(sg, sgx) across ranks -> apply with M_total = rows summed over ranks */
int mfx_bn_bwd_reduce(const void* x, const void* a, const void* da, const float* mean, const float* rstd,
                      float* sg, float* sgx, long M, int C, int act, int dtype, void* stream);
int mfx_bn_bwd_apply(const void* x, const void* a, const void* da, const float* mean, const float* rstd, const float* gamma,
                     const float* sg, const float* sgx, void* dx, void* dres, long M, long M_total, int C, int act,
                     int dtype, void* stream);
int mfx_maxpool2x2_bwd_nhwc(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int dtype, void* stream);
/* depthwise deconv backward: dx (input-sized) and dw fp32 [2f*2f][C] (overwritten) */
size_t mfx_upsample_bwd_workspace_bytes(int B, int H, int C, int f);
/* `workspace` (optional, >= mfx_upsample_bwd_workspace_bytes): per-workgroup partial weight gradients summed in a fixed order (no
 * atomics, no zero-fill, bit-reproducible); without it the workgroups add into dw with fp32 atomics */
int mfx_upsample_bwd_nhwc(const void* x, const float* w, const void* dy, void* dx, float* dw,
                          int B, int H, int W, int C, int f, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* the same with dw written in the parameter's own layout (C, 1, 2f, 2f) -- no transposition left for the caller; needs the workspace */
int mfx_upsample_bwd_nhwc_oihw(const void* x, const float* w, const void* dy, void* dx, float* dw,
                          int B, int H, int W, int C, int f, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* up[b,2oh,2ow,:] = dy[b,oh,ow,:], zeros elsewhere (data gradient of a stride-2 conv = stride-1 conv of `up`) */
int mfx_zero_insert2_nhwc(const void* dy, void* up, int B, int Ho, int Wo, int C, int H, int W, int dtype, void* stream);
/* DCNv2 backward on NHWC fp32 activations (no layout transforms): see dcn_bwd.hip */
size_t mfx_dcn_backward_nhwc_workspace_bytes(int B, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil);
int mfx_dcn_backward_nhwc(const float* x, const float* offmask, const float* weight, const float* dy,
                          float* dx, float* d_offmask, float* dweight, float* dbias,
                          int B, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil,
                          void* workspace, size_t workspace_bytes, void* stream);
/* bf16 activations (x, dy bf16): d(columns) by the bf16 MFMA GEMM; all gradient outputs fp32 (dx is accumulated with fp32
 * atomics; the caller narrows it).  Same workspace size.  C >= 64. */
int mfx_dcn_backward_nhwc_bf16(const void* x, const float* offmask, const float* weight_oihw, const void* dy,
                               float* dx, float* d_offmask, float* dweight, float* dbias,
                               int B, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil,
                               void* workspace, size_t workspace_bytes, void* stream);
/* Second-generation DCNv2 backward (dcn_bwd_tile.hip): 3x3 / stride 1 / pad 1 / dilation 1, C and Cout powers of two >= 64,
 * x / dy / dx in `dtype` (MFX_F32, MFX_BF16 or MFX_F16).  grad_input is accumulated per tile in LDS and written once in the
 * activation dtype (no global atomics for offsets within 8 pixels of the sampling pixel's tile); `d_raw` is the fp32
 * gradient of the RAW 27(32)-channel offset/mask conv output (B,H,W,32): offsets 0..17, mask LOGITS 18..26 (the sigmoid
 * derivative is applied here), channels 27..31 zero.  `offmask` holds the offsets and the POST-sigmoid mask, as the forward
 * kernel consumed them.  dweight fp32 (Cout,C,3,3), dbias fp32 (Cout); all outputs overwritten. */
size_t mfx_dcn_backward_v2_workspace_bytes(int B, int C, int H, int W, int Cout, int dtype);
int mfx_dcn_backward_v2(const void* x, const float* offmask, const float* weight_oihw, const void* dy, void* dx,
                        float* d_raw, float* dweight, float* dbias, int B, int C, int H, int W, int Cout, int dtype,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The same with `d_raw` written in the activation dtype when `raw_in_act_dtype` != 0 (16-bit layers: the rows the offset conv's own
 * backward pass consumes, without the fp32 map and its cast in between; each value is the fp32 result rounded once).  MFX_F32: fp32 either way. */
int mfx_dcn_backward_v2_rt(const void* x, const float* offmask, const float* weight_oihw, const void* dy, void* dx,
                           void* d_raw, int raw_in_act_dtype, float* dweight, float* dbias, int B, int C, int H, int W, int Cout, int dtype,
                           void* workspace, size_t workspace_bytes, void* stream);
/* Heat-map loss (penalty-reduced focal loss, model/layers/focal_loss.py:29-55 on sigmoid_hm(logits), layers/utils.py:39-42)
 * in one pass: logits fp32 NHWC (B,H,W,ncls), target fp32 NCHW (B,ncls,H,W) -> sums2 = [loss_sum, num_pos] (overwritten) and
 * dlogits (B,H,W,ncls) = d(loss_sum)/d(logit), the clamp of the sigmoid included. */
int mfx_focal_loss(const float* logits_nhwc, const float* heat_nchw, int B, int H, int W, int ncls, float alpha, float beta,
                   float* sums2, float* dlogits_nhwc, void* stream);

/* Per-object regression losses and their gradient (model/head/detector_loss.py:116-482: prepare_predictions, the nine
 * regression terms, the logged MAEs; decoders model/anno_encoder.py:88-295, model/layers/iou_loss.py:7-49).  One wavefront per
 * object row; lane c carries d/d(channel c) through the expression in forward mode, so the value of every term and its
 * gradient row come out of the same launch.
 *   rows  fp32 [N][MFX_OBJ_ROW]: one row per (image, object slot), N = B * MAX_OBJECTS (layout: csrc/object_loss_math.h R_*)
 *   reg   fp32 NHWC map, pixel stride `ld`, the 50 regression channels at [ch_off, ch_off + 50) of every pixel
 *   vals  fp32 [MFX_OBJ_VALUES] (overwritten): [0, MFX_OBJ_TERMS) the weighted loss terms bbox, depth, offset, trunc_offset, orien,
 *         dims, corner, keypoint, keypoint_depth, weighted_avg_depth; then the logged means (2D_IoU, depth_loss, keypoint_depth_loss,
 *         depth / center / 02 / 13 / lower / hard / soft / mean MAE)
 *   G     fp32 [N][MFX_OBJ_TERMS][64] (overwritten): d(term)/d(channel) at the object's pixel
 * mfx_object_loss_backward ADDS sum_t gout[t] * G[n][t][c] into dreg (same geometry as reg; the caller zero-fills it): objects
 * sharing a centre pixel accumulate, as the gather's backward does.
 * B = 0 selects the GATHERED form: reg / dreg are [N][ld] tables, row n belonging to object row n (mfx_head_sparse_fwd's output). */
#define MFX_OBJ_ROW 72
#define MFX_OBJ_TERMS 10
#define MFX_OBJ_VALUES 24
typedef struct mfx_object_loss_cfg {
    float w[MFX_OBJ_TERMS];               /* INIT_LOSS_WEIGHT of the ten terms, in the order above */
    float dim_mean[9], dim_std[9], dim_weight[3];
    float depth_ref[2], depth_range[2];
    float unc_lo, unc_hi, down_ratio, eps;
    int depth_mode;                       /* 0 exp, 1 linear, 2 inv_sigmoid */
    int has_depth_range, dim_exp, dim_use_std;
    int iou_type;                         /* 0 giou, 1 iou, 2 linear_iou */
    int corner_depth_mode;                /* 0 direct, 1 keypoint_mean, 2 soft_combine, 3 hard_combine */
    int separate_trunc, trunc_log, modify_invalid;
    int ch[9];                            /* first channel of 2d_dim, 3d_offset, corner_offset, corner_uncertainty, 3d_dim, ori_cls,
                                             ori_offset, depth, depth_uncertainty within the 50 */
} mfx_object_loss_cfg;
int mfx_object_loss(const float* reg_nhwc, int B, int H, int W, int ld, int ch_off, const float* rows, int N,
                    const mfx_object_loss_cfg* cfg, float* vals, float* G, void* stream);
int mfx_object_loss_backward(const float* G, const float* gout_terms, const float* rows, int N, int B, int H, int W,
                             float* dreg_nhwc, int ld, int ch_off, void* stream);

/* Regression branches of the training step evaluated at the object centres only (csrc/head_sparse.hip; reference
 * model/head/detector_predictor.py:125-169 + the gather of model/layers/utils.py:120-145).  Per branch i: y[i] = the dense trunk
 * conv output (B,H,W,256) in `dtype`, mean/rstd = its batch statistics, gamma/beta = the ABN parameters, w2 (k,256) / b2 (k) = the
 * stacked 1x1 heads.  rows = the object table of mfx_object_loss (valid flag, image, centre).
 *   forward : out[n][out_off[i] + j] for every row (zeros for empty slots), out fp32 [N][ld_out]
 *   backward: dout [N][ld_out] -> dx[i] (dense, overwritten, `dtype`), sums[i] = [sum g | sum g*xhat] (= dbeta | dgamma),
 *             dw2[i] (k,256), db2[i] (k) fp32.  sums / dw2 / db2 of all branches must lie inside ONE buffer `arena` (zero-filled
 *             by the call); g = scratch fp32 [nbranch][N][256]. */
#define MFX_HEAD_MAX_BRANCH 8
typedef struct mfx_head_sparse_desc {
    int nbranch, N, B, H, W, C, dtype, ld_out;
    const float* rows;
    const void* y[MFX_HEAD_MAX_BRANCH];
    const float* mean[MFX_HEAD_MAX_BRANCH]; const float* rstd[MFX_HEAD_MAX_BRANCH];
    const float* gamma[MFX_HEAD_MAX_BRANCH]; const float* beta[MFX_HEAD_MAX_BRANCH];
    const float* w2[MFX_HEAD_MAX_BRANCH]; const float* b2[MFX_HEAD_MAX_BRANCH];
    int k[MFX_HEAD_MAX_BRANCH], out_off[MFX_HEAD_MAX_BRANCH];
    float* out;
    const float* dout;
    float* g;
    float* sums[MFX_HEAD_MAX_BRANCH]; float* dw2[MFX_HEAD_MAX_BRANCH]; float* db2[MFX_HEAD_MAX_BRANCH];
    void* dx[MFX_HEAD_MAX_BRANCH];
    void* arena; size_t arena_bytes;
} mfx_head_sparse_desc;
int mfx_head_sparse_fwd(const mfx_head_sparse_desc* d, void* stream);
/* The same regression branches WITHOUT their dense trunk maps: batch statistics from the patch Gram matrix of the shared 64-channel input,
 * trunk values at the object rows (and, for branch `extra_branch`, at `Ne` edge-sequence pixels) only -- csrc/gram_heads.hip, the hand-written
 * forward and backward of monoflex_amd/gram_heads.py (reference model/head/detector_predictor.py:125-165).  One descriptor for the whole node;
 * the host calls the phases in order and runs the library's own launches between them:
 *   0: A = gathered 3x3 patches of [frame rows | object rows | edge rows] ([F+N+Ne][576], activation type), Wkc / WkT = the trunk weights as one matrix
 *      -> host: P = A_f^T A_f (mfx_conv_wgrad_oihw), csA = column sums of A_f, R5 = the displacement rows -2, -1, 0 of the 5x5 autocorrelation of x
 *         ([64][64][3][5]; the other two rows follow from R[a][b][d] = R[b][a][-d]), S0 = column sums of x
 *   1: G, m, Tm = Wk G (f32 MFMA), sums = [Wk m | diag(Wk G Wk^T)]          -> host: all-reduce of `sums` when the ABNs are synchronised
 *   2: statistics + running statistics, Y / act rows, out = 1x1 heads
 *   3: row gradients (d y as hi + lo halves), d W2 / d b2, d gamma / d beta, ds = [d s1 | d s2]      -> host: all-reduce of `ds` (SyncBN)
 *   4: scal[0] = max |d s2|, Dw16 = d s2 / scal[0] * Wk, d m                -> host: dGs = Dw16^T Wkc (mfx_conv_wgrad_oihw)
 *   5: 5x5 kernel Kx (normalised, OIHW fp32) + its epilogue vectors, d A_frame, d A rows      -> host: pack Kx, dx = conv5x5(x), dwo / dwe = d y^T A
 *   6: dx += border pixels (fixed order) + object / edge rows (packed 16-bit atomics)
 *   7: trunk weight gradients dwt[b] (256, 64, 3, 3) fp32
 * dtype MFX_BF16 / MFX_F16 (fp32 activations keep the torch form). */
typedef struct mfx_gram_desc {
    const void* x; const float* rows; const long long* extra_rows;
    int B, H, W, C, N, Ne, F, nbranch, extra_branch, ld_out, dtype, nring, ring_width, arena_bytes;   /* arena: [dsum 2 CH | scal 4 | d W2 | d b2] fp32, contiguous from `dsum`, cleared by phase 3 */
    float momentum, Mt;
    const void* wk[MFX_HEAD_MAX_BRANCH];                       /* packed trunk weights [256][576], k = tap * 64 + c */
    const float* gamma[MFX_HEAD_MAX_BRANCH]; const float* beta[MFX_HEAD_MAX_BRANCH];
    const float* w2[MFX_HEAD_MAX_BRANCH]; const float* b2[MFX_HEAD_MAX_BRANCH];
    float* run_mean[MFX_HEAD_MAX_BRANCH]; float* run_var[MFX_HEAD_MAX_BRANCH]; long long* nbt[MFX_HEAD_MAX_BRANCH];
    float* dgamma[MFX_HEAD_MAX_BRANCH]; float* dbeta[MFX_HEAD_MAX_BRANCH]; float* dw2[MFX_HEAD_MAX_BRANCH]; float* db2[MFX_HEAD_MAX_BRANCH];
    float* dwt[MFX_HEAD_MAX_BRANCH];
    float eps[MFX_HEAD_MAX_BRANCH];
    int k[MFX_HEAD_MAX_BRANCH], off[MFX_HEAD_MAX_BRANCH];
    /* forward buffers */
    void* A; void* Wkc; void* WkT;
    const float* R5; const float* S0; const float* P; const float* csA;
    float* G; float* m; float* Tm; float* sums; float* stat; float* Y; float* Ye; float* act; void* act_e; float* out;
    /* backward buffers */
    const float* dout; const void* dact_e;
    void* dYh; void* dYl; void* dYeh; void* dYel;
    float* dsum; float* ds; float* scal; void* Dw16; float* dm; const float* dGs; float* Kx; void* Gn16; float* cscale; float* cshift;
    float* dAf; float* dArows; void* dx; const long long* ring_inv; const long long* ring_idx; const float* dwo; const float* dwe;
} mfx_gram_desc;
int mfx_gram_heads(const mfx_gram_desc* d, int phase, void* stream);
int mfx_head_sparse_bwd(const mfx_head_sparse_desc* d, void* stream);
/* Batch statistics of a train-mode BN without applying it: mean / rstd (C floats each), running statistics and
 * num_batches_tracked updated as mfx_bn_train_fwd does; `scratch` as there (zero before, zero after). */
int mfx_bn_train_stats(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       long long* num_batches_tracked, float momentum, float eps, long M, int C, int dtype, float* scratch,
                       float* mean, float* rstd, int stats_done, void* stream);

/* ---- (4) input pipeline: KITTI sample -> network input + training targets, on the device ------------------------------
 * Replaces the per-sample numpy/PIL work of the reference's dataset (data/datasets/kitti.py:231-525 __getitem__,
 * data/augmentations/augmentations.py:33-78 flip, data/transforms/transforms.py:15-31 ToTensor+Normalize,
 * model/heatmap_coder.py:37-124 Gaussian rasterisation). The host only parses the text files. */

/* Batch of raw samples -> every field of the reference's training `target` (kitti.py:496-523), batch-stacked.
 * records: one row per object of the DETECT_CLASSES, in label order (kitti_utils.py:64-92):
 *   [cls_id, truncation, occlusion, xmin, ymin, xmax, ymax, h, w, l, tx, ty, tz, ry]  (float64, the parsed text values)
 * Arithmetic follows the reference's dtypes: float64 throughout, float32 where its numpy code is float32 (location,
 * label 2D box and everything derived from it). Objects the reference skips leave all-zero rows. All outputs are
 * fully overwritten. status[b] != 0 flags inputs on which the reference raises (bit 0: n_obj > max_objs, bit 1: truncated
 * object whose 2D-box centre lies outside the image, bit 2: no border intersection, bit 3: both boundary radii > 0). */
typedef struct {
  const double* records;     /* (B, max_objs, 14) */
  const int32_t* n_obj;      /* (B) */
  const double* P;           /* (B, 3, 4) camera matrix as read from calib (kitti_utils.py:186-187) */
  const int32_t* img_wh;     /* (B, 2) original image width, height */
  const int32_t* flip;       /* (B) 1 = apply the horizontal flip augmentation */
  float* hm;                 /* (B, num_classes, out_h, out_w) */
  int32_t* cls_ids;          /* (B, max_objs) */
  int32_t* target_centers;   /* (B, max_objs, 2) */
  float* keypoints;          /* (B, max_objs, 10, 3) */
  float* keypoints_depth_mask; /* (B, max_objs, 3) */
  float* dimensions;         /* (B, max_objs, 3) l, h, w */
  float* locations;          /* (B, max_objs, 3) */
  uint8_t* reg_mask;         /* (B, max_objs) */
  float* reg_weight;         /* (B, max_objs) */
  float* offset_3D;          /* (B, max_objs, 2) */
  float* bboxes;             /* (B, max_objs, 4) field "2d_bboxes" */
  float* gt_bboxes;          /* (B, max_objs, 4) */
  float* rotys;              /* (B, max_objs) */
  uint8_t* trunc_mask;       /* (B, max_objs) */
  float* alphas;             /* (B, max_objs) */
  float* orientations;       /* (B, max_objs, 8) multi-bin: 4 flags + 4 residuals */
  double* occlusions;        /* (B, max_objs) */
  double* truncations;       /* (B, max_objs) */
  int64_t* pad_size;         /* (B, 2) */
  int64_t* edge_indices;     /* (B, 2*(out_w+out_h), 2) */
  int64_t* edge_len;         /* (B) point count - 1 (kitti.py:284) */
  double* P_out;             /* (B, 3, 4) P after the flip */
  int32_t* heat_radius;      /* (B, max_objs, 4) auxiliary: rx, ry, circular(1)/boundary(0), drawn(1) */
  int32_t* status;           /* (B) */
  int32_t B, max_objs, in_w, in_h, down, num_classes;
  double filter_trunc, filter_size;   /* DATASETS.FILTER_ANNOS; filter_trunc < 0 disables the filter */
  double edge_ratio;                  /* INPUT.HEATMAP_RATIO */
} mfx_kitti_desc;
int mfx_kitti_encode_targets(const mfx_kitti_desc* d, void* stream);

/* uint8 RGB images of different sizes (packed back to back, HWC) -> (B,3,in_h,in_w) float32 NCHW: optional left-right
 * flip, centre zero padding, /255, (x-mean)/std; the padding is zero BEFORE normalisation (kitti.py:218-228). */
int mfx_kitti_preprocess_u8(const uint8_t* pixels, const int64_t* offsets, const int32_t* img_wh, const int32_t* flip,
                            float* out, int B, int in_w, int in_h, const float* mean3, const float* std3, void* stream);

/* ---- (5) KITTI AP evaluation on the device -----------------------------------------------------------------------------
 * Replaces the numba CPU loops and the numba.cuda rotated-IoU kernel of the reference's evaluator
 * (data/datasets/evaluation/kitti_object_eval_python/eval.py:27-286,448-570, rotate_iou.py:17-333). The host parses the
 * label / result text, sorts nothing and decides nothing: it ships one record per box, calls the four entries in order and
 * turns the accumulated (tp, fp, fn, similarity) table into precision curves and AP numbers.
 *
 * Box record (MFX_EVAL_REC doubles): [name code, truncated, occluded, alpha, x1, y1, x2, y2, l, h, w, x, y, z, ry, score].
 * Name codes: 0 car, 1 pedestrian, 2 cyclist, 3 van, 4 person_sitting, 5 truck, 6 DontCare, 7 anything else.
 * Images are ragged: boxes of image b are rows [off[b], off[b+1]) of `gt` / `dt`; its overlap block starts at pair_off[b]
 * and is laid out [detection][ground truth]. At most MFX_EVAL_MAX_DET (64) detections per image: the offsets are device
 * data, so the entries cannot validate them -- the caller must (detections past the 64th of an image are ignored).
 * A "combination" c indexes (class m, difficulty l, metric, overlap set k) as ((m*3 + l)*3 + metric)*num_k + k. */
#define MFX_EVAL_REC 16
#define MFX_EVAL_PTS 41
#define MFX_EVAL_MAX_DET 64
typedef struct {
  const double* gt;            /* (n_gt, 16) */
  const double* dt;            /* (n_dt, 16) */
  const int32_t* gt_off;       /* (B+1) */
  const int32_t* dt_off;       /* (B+1) */
  const int64_t* pair_off;     /* (B+1) */
  const int32_t* classes;      /* (num_classes) evaluated name codes, e.g. {0,1,2} */
  const double* min_overlaps;  /* (num_k, 3 metrics, num_classes) */
  double* overlaps;            /* (3, n_pairs): bbox IoU, BEV rotated IoU, 3D IoU */
  double* tp_scores;           /* (n_comb, n_gt): score of the detection matched to that ground truth, -inf if none (any score sign is kept) */
  int32_t* num_valid_gt;       /* (num_classes, 3) */
  double* thresholds;          /* (n_comb, 41) */
  int32_t* num_thresholds;     /* (n_comb) */
  double* pr;                  /* (n_comb, 41, 4): tp, fp, fn, orientation similarity; zeroed by mfx_kitti_eval_match_pass1 */
  int32_t B, n_gt, n_dt, num_classes, num_k, compute_aos;
  int64_t n_pairs;
} mfx_kitti_eval_desc;
/* 1: all three overlap matrices (eval.py:83-153, rotate_iou.py). */
int mfx_kitti_eval_overlaps(const mfx_kitti_eval_desc* d, void* stream);
/* 2: matching without score threshold -> tp_scores, num_valid_gt (eval.py:497-512 first loop); zeroes pr. */
int mfx_kitti_eval_match_pass1(const mfx_kitti_eval_desc* d, void* stream);
/* 3: `sorted_scores` = tp_scores sorted descending per combination (the caller sorts: any device sort) ->
 *    the <= 41 recall sample thresholds per combination (eval.py:8-24). */
int mfx_kitti_eval_thresholds(const mfx_kitti_eval_desc* d, const double* sorted_scores, void* stream);
/* 4: matching at every threshold with false-positive / DontCare / orientation accounting -> pr (eval.py:289-326). */
int mfx_kitti_eval_match_pass2(const mfx_kitti_eval_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MONOFLEX_HIP_H */
