// TEST-ONLY host build of monoflex_amd/csrc/kitti_encode_math.h (the header kitti_encode.hip maps GPU threads onto).
// It lets the CPU test suite check the encoder's indexing, dtype rules and control flow against the oracle without a GPU.
// Nothing in the product loads this: libmonoflex_hip.so calls the same functions from device code only.
#include "../../monoflex_amd/csrc/kitti_encode_math.h"

extern "C" void shim_kitti_encode(const mfx_kitti_desc* dp) {
  const mfx_kitti_desc& d = *dp;
  const int out_w = d.in_w / d.down, out_h = d.in_h / d.down, max_edge = 2 * (out_w + out_h);
  for (int b = 0; b < d.B; ++b) {
    mfx::kitti::image_header(d, b);
    for (int i = 0; i < d.max_objs; ++i) mfx::kitti::encode_object(d, b, i);
    for (int k = 0; k < max_edge; ++k) mfx::kitti::edge_point(d, b, k);
  }
  for (int b = 0; b < d.B; ++b)
    for (int c = 0; c < d.num_classes; ++c)
      for (int y = 0; y < out_h; ++y)
        for (int x = 0; x < out_w; ++x)
          d.hm[(((long)b * d.num_classes + c) * out_h + y) * out_w + x] = mfx::kitti::heat_pixel(d, b, c, y, x);
}

extern "C" void shim_kitti_preprocess(const uint8_t* pixels, const int64_t* offsets, const int32_t* img_wh, const int32_t* flip,
                                      float* out, int B, int in_w, int in_h, const float* mean3, const float* std3) {
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < in_h; ++y)
      for (int x = 0; x < in_w; ++x)
        mfx::kitti::preprocess_pixel(pixels, offsets, img_wh, flip, out, b, y, x, in_w, in_h, mean3, std3);
}
