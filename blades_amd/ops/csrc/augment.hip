// Batch-level image augmentation, one launch for all clients of a chunk.
//
// x [n, B, C, H, W] fp32 -> out [n, B, C, S, S]: per batch b (= one client's batch; the reference applies
// its transform to the whole batch tensor, fllib/datasets/fldataset.py:247-249) one square crop (i, j, s)
// resized to SxS with bilinear interpolation (align_corners = false, PyTorch's convention), optional
// horizontal flip, per-channel normalisation and an optional zeroed erase rectangle.
// params [n, 9] = (i, j, s, flip, erase, ei, ej, eh, ew) as floats; see datasets/transforms.py.
#include "common.h"

namespace {
struct NormC { float mean[4]; float inv_std[4]; };

__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ x, const float* __restrict__ params,
                                                      float* __restrict__ out, int n, int B, int C, int H, int W,
                                                      int S, NormC nc) {
  const long total = (long)n * B * C * S * S;
  const long stride = (long)gridDim.x * 256;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const int ox = (int)(idx % S);
    const int oy = (int)((idx / S) % S);
    const int c = (int)((idx / ((long)S * S)) % C);
    const long img = idx / ((long)S * S * C);  // nb * B + b
    const int nb = (int)(img / B);
    const float* p = params + nb * 9;
    const int ci = (int)p[0], cj = (int)p[1], cs = (int)p[2];
    const bool flip = p[3] > 0.f, erase = p[4] > 0.f;
    const int ei = (int)p[5], ej = (int)p[6], eh = (int)p[7], ew = (int)p[8];
    float v = 0.f;
    if (!(erase && oy >= ei && oy < ei + eh && ox >= ej && ox < ej + ew)) {
      const int sx = flip ? (S - 1 - ox) : ox;
      const float scale = (float)cs / (float)S;
      float fy = ((float)oy + 0.5f) * scale - 0.5f; if (fy < 0.f) fy = 0.f;
      float fx = ((float)sx + 0.5f) * scale - 0.5f; if (fx < 0.f) fx = 0.f;
      int y0 = (int)fy, x0 = (int)fx;
      if (y0 > cs - 1) y0 = cs - 1;
      if (x0 > cs - 1) x0 = cs - 1;
      const int y1 = y0 + (y0 < cs - 1), x1 = x0 + (x0 < cs - 1);
      const float ly = fy - (float)y0, lx = fx - (float)x0;
      const float* src = x + (img * C + c) * (long)H * W;
      const float v00 = src[(ci + y0) * W + cj + x0], v01 = src[(ci + y0) * W + cj + x1];
      const float v10 = src[(ci + y1) * W + cj + x0], v11 = src[(ci + y1) * W + cj + x1];
      v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
      v = (v - nc.mean[c & 3]) * nc.inv_std[c & 3];
    }
    out[idx] = v;
  }
}
}  // namespace

BLADES_API int blades_augment_batch(const float* x, const float* params, float* out, int n, int B, int C, int H,
                                    int W, int S, const float* mean, const float* std, hipStream_t st) {
  if (C > 4) return 1;
  NormC nc;
  for (int c = 0; c < 4; ++c) { nc.mean[c] = c < C ? mean[c] : 0.f; nc.inv_std[c] = c < C ? 1.f / std[c] : 1.f; }
  const long total = (long)n * B * C * S * S;
  if (total <= 0) return 0;
  int g = cdiv(total, 256); if (g > 8192) g = 8192;
  hipLaunchKernelGGL(augment_kernel, dim3(g), dim3(256), 0, st, x, params, out, n, B, C, H, W, S, nc);
  return 0;
}
