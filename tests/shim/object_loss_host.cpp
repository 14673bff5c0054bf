// TEST-ONLY host build of monoflex_amd/csrc/object_loss_math.h: the wavefront-per-object kernel as plain loops (object, lane), so
// the CPU suite can check every loss term and its gradient row against the reference goldens without a GPU.  Not loaded by the product.
#include <algorithm>
#include <cstddef>

#include "../../monoflex_amd/csrc/object_loss_math.h"

using namespace mfx::oloss;

namespace {
struct PixelReader {
    const float* p; int seed;
    Dual operator()(int ch) const { return Dual{p[ch], ch == seed ? 1.f : 0.f}; }
};
const float* object_pixel(const float* base, const float* t, int B, int H, int W, int ld, int ch_off) {
    const int b = std::min(std::max((int)t[R_B], 0), B - 1), cx = std::min(std::max((int)t[R_CX], 0), W - 1), cy = std::min(std::max((int)t[R_CY], 0), H - 1);
    return base + ((size_t)(b * H + cy) * W + cx) * ld + ch_off;
}
}  // namespace

extern "C" void shim_object_loss(const float* reg, int B, int H, int W, int ld, int ch_off, const float* rows, int N,
                                 const mfx_object_loss_cfg* cfg, float* vals, float* G) {
    float cn[NNORM] = {0};
    for (int r = 0; r < N; ++r) {
        float q[NNORM];
        row_counts(rows + (size_t)r * ROW, q);
        for (int i = 0; i < NNORM; ++i) cn[i] += q[i];
    }
    for (int k = 0; k < NVAL; ++k) vals[k] = 0.f;
    for (int n = 0; n < N; ++n) {
        const float* t = rows + (size_t)n * ROW;
        for (int lane = 0; lane < 64; ++lane) {
            Dual out[NVAL];
            const PixelReader X{object_pixel(reg, t, B, H, W, ld, ch_off), lane};
            object_terms(X, t, *cfg, cn, out);
            for (int k = 0; k < NTERM; ++k) G[((size_t)n * NTERM + k) * 64 + lane] = out[k].d;
            if (lane == 0 && t[R_VALID] != 0.f)
                for (int k = 0; k < NVAL; ++k) vals[k] += out[k].v;
        }
    }
}
