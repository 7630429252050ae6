// First layer of the single-channel models (Fashion-MNIST / MNIST CNN: 1 -> 32 channels, 3x3; reference
// fllib/models/fashionmnist/cnn.py) on the vector ALUs, in plain fp32.
//
// After im2col the layer is  y[r, o] = b[o] + sum_k x[r, k] w[o, k]  with 16 reduction elements (9 real) and 32 outputs
// per pixel r: 512 multiply-adds per pixel against 192 bytes of traffic.  On the matrix-core kernels a tile has a single
// K-step — all prologue and epilogue — and the 247-register kernel keeps two workgroups per CU: 370 us forward and
// 445 us for the weight gradient of 48 clients x 64 images, against ~100 us of memory time.  At this size the fp32 FMA
// rate is not the limit (1.2 G FMA per pass = 16 us at the vector peak), so the layer runs as two streaming kernels
// with exact fp32 products (no plane split needed):
//
//   stem_fwd   thread (row lane, 4 outputs) with its 4 x 16 weights in registers over the rows of a chunk: the lanes of
//              a row share its 64 input bytes and write 128 contiguous bytes (a thread per pixel with scalar-loaded
//              weights wrote 16 bytes into 64 different lines per store instruction: 295 us instead of ~110)
//   stem_wgrad dW_c[o, k] = sum_r gy[r, o] x[r, k]: thread (row lane, 4 outputs) keeps 4 x 16 accumulators over its
//              rows of a chunk; the 32 row lanes are combined through LDS in lane order, chunks by a second kernel in
//              chunk order (bitwise reproducible), scaled and written into the update rows as [o][k < kreal]
//
// Shapes: WD = 16 input columns, Co a multiple of 4 up to 32, rows per client a multiple of 64.
#include "common.h"

namespace {

constexpr int WD = 16;
constexpr int COMAX = 32;

// thread (row lane, 4 outputs): its 4 x 16 weights and 4 bias values stay in registers over the rows of a chunk; the
// Co / 4 lanes of a row read the same 64 input bytes and write Co * 4 contiguous output bytes
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, long w_cs,
                                                       const float* __restrict__ bias, long b_cs, float* __restrict__ y,
                                                       long R, int Co, int rows_per_chunk) {
  const int c = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
  const int nob = Co >> 2, ob = t % nob, rl = t / nob, nrl = 256 / nob;
  const long r0 = (long)chunk * rows_per_chunk;
  long r1 = r0 + rows_per_chunk;
  if (r1 > R) r1 = R;
  const float* wc = w + (size_t)c * w_cs + (size_t)ob * 4 * WD;
  float wv[4][WD], bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bv[j] = bias ? bias[(size_t)c * b_cs + ob * 4 + j] : 0.f;
#pragma unroll
    for (int q = 0; q < WD / 4; ++q) {
      const float4 v = reinterpret_cast<const float4*>(wc + j * WD)[q];
      wv[j][4 * q] = v.x; wv[j][4 * q + 1] = v.y; wv[j][4 * q + 2] = v.z; wv[j][4 * q + 3] = v.w;
    }
  }
  const float* xb = x + (size_t)c * R * WD;
  float* yb = y + (size_t)c * R * Co;
  for (long r = r0 + rl; r < r1; r += nrl) {
    float xv[WD];
#pragma unroll
    for (int q = 0; q < WD / 4; ++q) {
      const float4 v = reinterpret_cast<const float4*>(xb + r * WD)[q];
      xv[4 * q] = v.x; xv[4 * q + 1] = v.y; xv[4 * q + 2] = v.z; xv[4 * q + 3] = v.w;
    }
    float a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = bv[j];
#pragma unroll
      for (int k = 0; k < WD; ++k) s = fmaf(xv[k], wv[j][k], s);
      a[j] = s;
    }
    *reinterpret_cast<float4*>(yb + r * Co + ob * 4) = make_float4(a[0], a[1], a[2], a[3]);
  }
}

// partial[(c * nchunk + chunk) * Co * WD + o * WD + k]
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy, long R,
                                                         int Co, int rows_per_chunk, int nchunk, float* __restrict__ partial) {
  __shared__ float red[4352];  // one of a block's four outputs at a time: [row lanes][Co / 4 * 16 + 1] (largest: 256 x 17)
  const int c = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
  const int nob = Co >> 2;            // output blocks of 4
  const int ob = t % nob, rl = t / nob;
  const int nrl = 256 / nob;          // row lanes (32 for Co = 32)
  const long r0 = (long)chunk * rows_per_chunk;
  long r1 = r0 + rows_per_chunk;
  if (r1 > R) r1 = R;
  const float* xb = x + (size_t)c * R * WD;
  const float* gb = gy + (size_t)c * R * Co;
  float acc[4][WD];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < WD; ++k) acc[j][k] = 0.f;
  if (rl < nrl)
    for (long r = r0 + rl; r < r1; r += nrl) {
      const float4 g4 = *reinterpret_cast<const float4*>(gb + r * Co + ob * 4);
      const float g[4] = {g4.x, g4.y, g4.z, g4.w};
      float xv[WD];
#pragma unroll
      for (int q = 0; q < WD / 4; ++q) {
        const float4 v = reinterpret_cast<const float4*>(xb + r * WD)[q];
        xv[4 * q] = v.x; xv[4 * q + 1] = v.y; xv[4 * q + 2] = v.z; xv[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < WD; ++k) acc[j][k] = fmaf(g[j], xv[k], acc[j][k]);
    }
  // combine the row lanes, one of the four outputs of a block at a time: red[rl][ob * 16 + k]
  float* out = partial + ((size_t)c * nchunk + chunk) * Co * WD;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __syncthreads();
    if (rl < nrl) {
#pragma unroll
      for (int k = 0; k < WD; ++k) red[rl * (nob * WD + 1) + ob * WD + k] = acc[j][k];
    }
    __syncthreads();
    for (int i = t; i < nob * WD; i += 256) {
      float s = 0.f;
      for (int l = 0; l < nrl; ++l) s += red[l * (nob * WD + 1) + i];  // lane order
      const int o = (i / WD) * 4 + j, k = i % WD;
      out[o * WD + k] = s;
    }
  }
}

__global__ __launch_bounds__(256) void stem_wgrad_final_kernel(const float* __restrict__ partial, int Kc, int Co, int nchunk,
                                                               int kreal, const float* __restrict__ coef,
                                                               float* __restrict__ U, long ld, long off, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Kc * Co * WD) return;
  const int c = i / (Co * WD), e = i % (Co * WD), o = e / WD, k = e % WD;
  if (k >= kreal) return;
  float s = 0.f;
  for (int z = 0; z < nchunk; ++z) s += partial[((size_t)c * nchunk + z) * Co * WD + e];  // chunk order
  const float v = (coef ? coef[c] : 1.f) * s;
  float* dst = U + (size_t)c * ld + off + (size_t)o * kreal + k;
  *dst = accumulate ? *dst + v : v;
}

inline bool shape_ok(long M, long R, int Co) {
  return M > 0 && R > 0 && M % R == 0 && R % 64 == 0 && Co >= 4 && Co <= COMAX && Co % 4 == 0 && 256 % (Co / 4) == 0;
}

}  // namespace

BLADES_API int blades_stem_valu_ok(long M, long R, int wd, int Co) { return wd == WD && shape_ok(M, R, Co) ? 1 : 0; }

// x: [M, 16] fp32, w: client 0's [Co, 16] (client c at + c * w_cs; w_cs = 0: shared), bias likewise or null, y: [M, Co].
// R = rows per client (M for shared weights).  returns 0 ok.
BLADES_API int blades_stem_fwd_valu(const float* x, const float* w, long w_cs, const float* bias, long b_cs, float* y, long M,
                                    long R, int Co, hipStream_t st) {
  if (!shape_ok(M, R, Co) || (uintptr_t)x % 16 || (uintptr_t)y % 16 || (uintptr_t)w % 16 || w_cs % 4) return 1;
  const long Kc = M / R, chunks = (R + 2047) / 2048;
  if (Kc > 65535 || chunks > 0x7fffffffL) return 2;
  hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)chunks, (unsigned)Kc), dim3(256), 0, st, x, w, w_cs, bias, b_cs, y, R, Co,
                     2048);
  return 0;
}

BLADES_API int blades_stem_wgrad_nchunk(long R) { return (int)((R + 2047) / 2048); }

// U[c, off + o * kreal + k] (+)= coef[c] * sum_r gy[c, r, o] * x[c, r, k], k < kreal.  partial: Kc * nchunk * Co * 16 floats.
BLADES_API int blades_stem_wgrad_valu(const float* x, const float* gy, int Kc, long R, int Co, int kreal, const float* coef,
                                      float* U, long ld, long off, int accumulate, float* partial, hipStream_t st) {
  if (!shape_ok((long)Kc * R, R, Co) || kreal < 1 || kreal > WD || Kc > 65535 || (uintptr_t)x % 16 || (uintptr_t)gy % 16)
    return 1;
  const int nchunk = blades_stem_wgrad_nchunk(R);
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nchunk, Kc), dim3(256), 0, st, x, gy, R, Co, 2048, nchunk, partial);
  hipLaunchKernelGGL(stem_wgrad_final_kernel, dim3(cdiv((long)Kc * Co * WD, 256)), dim3(256), 0, st, partial, Kc, Co, nchunk,
                     kreal, coef, U, ld, off, accumulate);
  return 0;
}
