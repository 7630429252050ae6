// im2col of a first-layer ("stem") convolution input: x [N][H][W][Ci] fp32 (channels-last, Ci * k * k <= 32) ->
// out [N][H][W][Wd] (Wd = 16 when Ci * k * k fits — single-channel images — else 32), channel ci*k*k + r*k + s =
// x[n][h + r - pad][w + s - pad][ci], zero beyond Ci*k*k and outside the image — the row order of weight.reshape(Co, Ci*k*k).  The engine then runs the stem as a 1x1 convolution
// over this tensor (one 32-wide K-step instead of k*k padded ones) and its weight gradient as a plain GEMM
// (engine/fused.py::_PCConvFn; reference layer: fllib/models/cifar10/resnet_cifar.py conv1).
// One thread per (pixel, 4-channel group): 16-byte stores, a pixel's Wd * 4 bytes are written by adjacent lanes.
#include "common.h"

namespace {

template <int GROUPS>  // Wd / 4
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, float* __restrict__ out, long npix,
                                                          int H, int W, int Ci, int k, int pad) {
  const int kk = k * k, K = Ci * kk;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix * GROUPS; i += (long)gridDim.x * 256) {
    const long pix = i / GROUPS;
    const int g = (int)(i % GROUPS);
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
    *reinterpret_cast<float4*>(out + pix * (GROUPS * 4) + g * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// General im2col: x [N][H][W][Ci] (channels-last, fp32 or bf16) -> col [N*Ho*Wo][Kp], column ci*kh*kw + r*kw + s =
// x[n][oh*sh + r - ph][ow*sw + s - pw][ci] (zero outside the image and in the padding columns K .. Kp-1): the row
// order of weight.reshape(Co, Ci*kh*kw).  With it the per-client weight gradient of ANY convolution geometry is the
// grouped GEMM gy^T col of linear_wgrad.hip — the route for the kernel sizes / strides conv_wgrad.hip has no tile
// schedule for (4x4 stride-4 patch embedding of the CVT models, 7x7 / 5x5 tokenizers; reference
// fllib/models/backbones/cctnets/cvt.py, cct.py).  One thread per output element, columns fastest.
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, T* __restrict__ col, long total, int H, int W,
                                                     int Ci, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph,
                                                     int pw, int K, int Kp) {
  const int kk = kh * kw;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Kp);
    long pix = i / Kp;
    const int ow = (int)(pix % Wo);
    long t = pix / Wo;
    const int oh = (int)(t % Ho);
    const long n = t / Ho;
    T v = (T)0;
    if (c < K) {
      const int ci = c / kk, rs = c - ci * kk, r = rs / kw, s = rs - r * kw;
      const int ih = oh * sh + r - ph, iw = ow * sw + s - pw;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[((n * H + ih) * W + iw) * Ci + ci];
    }
    col[i] = v;
  }
}

}  // namespace

// col: [N*Ho*Wo][Kp] with Kp >= Ci*kh*kw.  esize 4 (fp32) or 2 (bf16 bits).  returns 0 ok.
BLADES_API int blades_im2col(const void* x, void* col, long N, int H, int W, int Ci, int Ho, int Wo, int kh, int kw, int sh,
                             int sw, int ph, int pw, int Kp, int esize, hipStream_t st) {
  const int K = Ci * kh * kw;
  if (N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || Kp < K || Ho <= 0 || Wo <= 0)
    return 1;
  const long total = N * Ho * Wo * Kp;
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (esize == 4)
    hipLaunchKernelGGL((im2col_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (float*)col, total,
                       H, W, Ci, Ho, Wo, kh, kw, sh, sw, ph, pw, K, Kp);
  else if (esize == 2)
    hipLaunchKernelGGL((im2col_kernel<unsigned short>), dim3((unsigned)blocks), dim3(256), 0, st, (const unsigned short*)x,
                       (unsigned short*)col, total, H, W, Ci, Ho, Wo, kh, kw, sh, sw, ph, pw, K, Kp);
  else
    return 2;
  return 0;
}

// returns 0 ok; 1 = unsupported shape.  wd: channels of `out` per pixel, 16 or 32 (>= Ci * k * k)
BLADES_API int blades_stem_im2col_w(const float* x, float* out, int N, int H, int W, int Ci, int k, int pad, int wd,
                                    hipStream_t st) {
  if (N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || k <= 0 || Ci * k * k > wd || pad < 0 || (wd != 16 && wd != 32)) return 1;
  const long npix = (long)N * H * W;
  long blocks = (npix * (wd / 4) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (wd == 16)
    hipLaunchKernelGGL(stem_im2col_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, x, out, npix, H, W, Ci, k, pad);
  else
    hipLaunchKernelGGL(stem_im2col_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, x, out, npix, H, W, Ci, k, pad);
  return 0;
}

BLADES_API int blades_stem_im2col(const float* x, float* out, int N, int H, int W, int Ci, int k, int pad,
                                  hipStream_t st) {
  return blades_stem_im2col_w(x, out, N, H, W, Ci, k, pad, 32, st);
}
