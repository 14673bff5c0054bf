// KITTI input pipeline on the device (C ABI group 4): label records -> training targets, uint8 frames -> network input.
// Reference: data/datasets/kitti.py:231-525, data/augmentations/augmentations.py:33-78, model/heatmap_coder.py:37-124,
// data/transforms/transforms.py:15-31.  The arithmetic lives in kitti_encode_math.h; this file maps threads onto it.
//
// Work per batch is tiny next to the network step (<= 40 objects and 92k heat-map pixels per image; the frame conversion
// moves 1.4 MB in / 5.9 MB out per image and is HBM-bound), so the kernels are sized for latency: one wave per image for
// the objects, one block per heat-map row, one thread per output pixel of the frame.
#include <hip/hip_runtime.h>

#include "err.h"
#include "kitti_encode_math.h"

namespace mfx {

__global__ void __launch_bounds__(64) kitti_objects_kernel(mfx_kitti_desc d) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) kitti::image_header(d, b);
  __syncthreads();                                   // status[b] is initialised before any object ORs into it
  for (int i = threadIdx.x; i < d.max_objs; i += blockDim.x) kitti::encode_object(d, b, i);
  const int max_edge = 2 * (d.in_w / d.down + d.in_h / d.down);
  for (int k = threadIdx.x; k < max_edge; k += blockDim.x) kitti::edge_point(d, b, k);
}

__global__ void __launch_bounds__(256) kitti_heatmap_kernel(mfx_kitti_desc d) {
  const int out_w = d.in_w / d.down, out_h = d.in_h / d.down;
  const int y = blockIdx.x % out_h, cls = (blockIdx.x / out_h) % d.num_classes, b = blockIdx.x / (out_h * d.num_classes);
  float* row = d.hm + (((long)b * d.num_classes + cls) * out_h + y) * out_w;
  for (int x = threadIdx.x; x < out_w; x += blockDim.x) row[x] = kitti::heat_pixel(d, b, cls, y, x);
}

struct Norm3 { float mean[3], stdv[3]; };

__global__ void __launch_bounds__(256) kitti_preprocess_kernel(const uint8_t* pixels, const int64_t* offsets, const int32_t* img_wh,
                                                               const int32_t* flip, float* out, int in_w, int in_h, Norm3 nm) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x < in_w) kitti::preprocess_pixel(pixels, offsets, img_wh, flip, out, b, y, x, in_w, in_h, nm.mean, nm.stdv);
}

}  // namespace mfx

extern "C" int mfx_kitti_encode_targets(const mfx_kitti_desc* d, void* stream) {
  using namespace mfx;
  if (!d) return mfx_fail(MFX_ERR_ARG, "mfx_kitti_encode_targets: null descriptor");
  if (d->B <= 0 || d->max_objs <= 0 || d->num_classes <= 0 || d->down <= 0 || d->in_w % d->down || d->in_h % d->down)
    return mfx_fail(MFX_ERR_ARG, "mfx_kitti_encode_targets: bad sizes (B, max_objs, num_classes, down must be positive; input size divisible by down)");
  const void* ptrs[] = {d->records, d->n_obj, d->P, d->img_wh, d->flip, d->hm, d->cls_ids, d->target_centers, d->keypoints,
                        d->keypoints_depth_mask, d->dimensions, d->locations, d->reg_mask, d->reg_weight, d->offset_3D, d->bboxes,
                        d->gt_bboxes, d->rotys, d->trunc_mask, d->alphas, d->orientations, d->occlusions, d->truncations,
                        d->pad_size, d->edge_indices, d->edge_len, d->P_out, d->heat_radius, d->status};
  for (const void* p : ptrs)
    if (!p) return mfx_fail(MFX_ERR_ARG, "mfx_kitti_encode_targets: every input and output pointer of the descriptor must be set");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(kitti_objects_kernel, dim3(d->B), dim3(64), 0, st, *d);
  const int out_h = d->in_h / d->down;
  hipLaunchKernelGGL(kitti_heatmap_kernel, dim3(d->B * d->num_classes * out_h), dim3(256), 0, st, *d);
  MFX_HIP_CHECK(hipGetLastError());
  return MFX_OK;
}

extern "C" int mfx_kitti_preprocess_u8(const uint8_t* pixels, const int64_t* offsets, const int32_t* img_wh, const int32_t* flip,
                                       float* out, int B, int in_w, int in_h, const float* mean3, const float* std3, void* stream) {
  using namespace mfx;
  if (!pixels || !offsets || !img_wh || !flip || !out || !mean3 || !std3)
    return mfx_fail(MFX_ERR_ARG, "mfx_kitti_preprocess_u8: null pointer");
  if (B <= 0 || in_w <= 0 || in_h <= 0) return mfx_fail(MFX_ERR_ARG, "mfx_kitti_preprocess_u8: bad sizes");
  Norm3 nm;
  for (int c = 0; c < 3; ++c) { nm.mean[c] = mean3[c]; nm.stdv[c] = std3[c]; }
  hipLaunchKernelGGL(kitti_preprocess_kernel, dim3((in_w + 255) / 256, in_h, B), dim3(256), 0, (hipStream_t)stream,
                     pixels, offsets, img_wh, flip, out, in_w, in_h, nm);
  MFX_HIP_CHECK(hipGetLastError());
  return MFX_OK;
}
