// im2col of a first-layer ("stem") convolution input: x [N][H][W][Ci] fp32 (channels-last, Ci * k * k <= 32) ->
// out [N][H][W][32], channel ci*k*k + r*k + s = x[n][h + r - pad][w + s - pad][ci], zero beyond Ci*k*k and outside
// the image — the row order of weight.reshape(Co, Ci*k*k).  The engine then runs the stem as a 1x1 convolution
// over this tensor (one 32-wide K-step instead of k*k padded ones) and its weight gradient as a plain GEMM
// (engine/fused.py::_PCConvFn; reference layer: fllib/models/cifar10/resnet_cifar.py conv1).
// One thread per (pixel, 4-channel group): 16-byte stores, a pixel's 128 bytes are written by 8 adjacent lanes.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, float* __restrict__ out, long npix,
                                                          int H, int W, int Ci, int k, int pad) {
  const int kk = k * k, K = Ci * kk;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix * 8; i += (long)gridDim.x * 256) {
    const long pix = i >> 3;
    const int g = (int)(i & 7);
    const int w = (int)(pix % W);
    const long t = pix / W;
    const int h = (int)(t % H);
    const long n = t / H;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = g * 4 + j;
      v[j] = 0.f;
      if (c < K) {
        const int ci = c / kk, rs = c - ci * kk, r = rs / k, s = rs - r * k;
        const int ih = h + r - pad, iw = w + s - pad;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v[j] = x[((n * H + ih) * W + iw) * Ci + ci];
      }
    }
    *reinterpret_cast<float4*>(out + pix * 32 + g * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

}  // namespace

// returns 0 ok; 1 = unsupported shape
BLADES_API int blades_stem_im2col(const float* x, float* out, int N, int H, int W, int Ci, int k, int pad,
                                  hipStream_t st) {
  if (N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || k <= 0 || Ci * k * k > 32 || pad < 0) return 1;
  const long npix = (long)N * H * W;
  long blocks = (npix * 8 + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(stem_im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, out, npix, H, W, Ci, k, pad);
  return 0;
}
