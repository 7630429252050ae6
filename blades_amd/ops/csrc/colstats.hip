// Column statistics of the update matrix U[K, Dl] (row-major, leading dim ld).
//
// All kernels here stream U exactly once: lane t owns 4 adjacent columns
// (one float4 = 16 B per lane, 1 KiB per wave per row — the widest coalesced
// access on CDNA4) and walks down the K rows keeping its running statistics in
// registers.  They are HBM-bound; the K-loop is unrolled so several row loads
// are in flight per lane.  Row weights / masks are wave-uniform loads.
#include "common.h"

namespace {

constexpr int TPB = 256;

// out[j] = sum_i w[i] * U[i, j]
__global__ __launch_bounds__(TPB) void weighted_col_sum_kernel(const float* __restrict__ U, long ld, int K, long Dl,
                                                               const float* __restrict__ w,
                                                               float* __restrict__ out) {
  const long c = ((long)blockIdx.x * TPB + threadIdx.x) * 4;
  if (c >= Dl) return;
  if (c + 4 <= Dl && (ld & 3) == 0) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p = U + c;
    int i = 0;
    for (; i + 4 <= K; i += 4) {
      const float4 a = *reinterpret_cast<const float4*>(p + (long)(i + 0) * ld);
      const float4 b = *reinterpret_cast<const float4*>(p + (long)(i + 1) * ld);
      const float4 d = *reinterpret_cast<const float4*>(p + (long)(i + 2) * ld);
      const float4 e = *reinterpret_cast<const float4*>(p + (long)(i + 3) * ld);
      const float w0 = w[i], w1 = w[i + 1], w2 = w[i + 2], w3 = w[i + 3];
      acc.x += w0 * a.x; acc.y += w0 * a.y; acc.z += w0 * a.z; acc.w += w0 * a.w;
      acc.x += w1 * b.x; acc.y += w1 * b.y; acc.z += w1 * b.z; acc.w += w1 * b.w;
      acc.x += w2 * d.x; acc.y += w2 * d.y; acc.z += w2 * d.z; acc.w += w2 * d.w;
      acc.x += w3 * e.x; acc.y += w3 * e.y; acc.z += w3 * e.z; acc.w += w3 * e.w;
    }
    for (; i < K; ++i) {
      const float4 a = *reinterpret_cast<const float4*>(p + (long)i * ld);
      const float wi = w[i];
      acc.x += wi * a.x; acc.y += wi * a.y; acc.z += wi * a.z; acc.w += wi * a.w;
    }
    *reinterpret_cast<float4*>(out + c) = acc;
  } else {
    for (long cc = c; cc < Dl && cc < c + 4; ++cc) {
      float acc = 0.f;
      for (int i = 0; i < K; ++i) acc += w[i] * U[(long)i * ld + cc];
      out[cc] = acc;
    }
  }
}

// One pass over the masked rows: Welford mean / M2, min, max per column.
// Epilogue (FORGE): v = a*mean + z*std*sign is written to `vec` and to every
// row listed in mal_rows — ALIE (a=1, z=z_max) and IPM (a=-scale, z=0) are
// this kernel with different scalars.
struct ColAcc {
  float mean, m2, mn, mx;
};

template <bool WANT_STD, bool WANT_MINMAX, bool FORGE>
__global__ __launch_bounds__(TPB) void col_stats_kernel(float* __restrict__ U, long ld, int K, long Dl,
                                                        const float* __restrict__ mask,  // may be null
                                                        float* __restrict__ o_mean, float* __restrict__ o_std,
                                                        float* __restrict__ o_min, float* __restrict__ o_max,
                                                        float a_coef, float z_coef,
                                                        const float* __restrict__ sign,  // may be null
                                                        const long* __restrict__ mal_rows, int n_mal,
                                                        float* __restrict__ o_vec) {
  const long c0 = ((long)blockIdx.x * TPB + threadIdx.x) * 4;
  if (c0 >= Dl) return;
  const int nc = (int)((Dl - c0) < 4 ? (Dl - c0) : 4);
  const bool vec4 = (nc == 4) && ((ld & 3) == 0);
  float mean[4] = {0, 0, 0, 0}, m2[4] = {0, 0, 0, 0};
  float mn[4], mx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { mn[j] = INFINITY; mx[j] = -INFINITY; }
  float n = 0.f;
  for (int i = 0; i < K; ++i) {
    if (mask != nullptr && mask[i] == 0.f) continue;  // wave-uniform branch
    float x[4] = {0, 0, 0, 0};
    if (vec4) {
      const float4 v = *reinterpret_cast<const float4*>(U + (long)i * ld + c0);
      x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else {
      for (int j = 0; j < nc; ++j) x[j] = U[(long)i * ld + c0 + j];
    }
    n += 1.f;
    const float inv = 1.f / n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = x[j] - mean[j];
      mean[j] += d * inv;
      if (WANT_STD) m2[j] += d * (x[j] - mean[j]);
      if (WANT_MINMAX) { mn[j] = fminf(mn[j], x[j]); mx[j] = fmaxf(mx[j], x[j]); }
    }
  }
  float sd[4] = {0, 0, 0, 0}, v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (WANT_STD) sd[j] = (n > 1.f) ? sqrtf(m2[j] / (n - 1.f)) : NAN;  // unbiased, like torch.std
    if (FORGE) {
      float s = sd[j];
      if (sign != nullptr && j < nc) s *= sign[c0 + j];
      v[j] = a_coef * mean[j] + (z_coef != 0.f ? z_coef * s : 0.f);
    }
  }
  for (int j = 0; j < nc; ++j) {
    if (o_mean) o_mean[c0 + j] = mean[j];
    if (WANT_STD && o_std) o_std[c0 + j] = sd[j];
    if (WANT_MINMAX) { o_min[c0 + j] = mn[j]; o_max[c0 + j] = mx[j]; }
    if (FORGE && o_vec) o_vec[c0 + j] = v[j];
  }
  if (FORGE) {
    for (int r = 0; r < n_mal; ++r) {
      float* dst = U + mal_rows[r] * ld + c0;
      if (vec4) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int j = 0; j < nc; ++j) dst[j] = v[j];
      }
    }
  }
}

// U[rows[r], :] = v
__global__ __launch_bounds__(TPB) void set_rows_kernel(float* __restrict__ U, long ld, long Dl,
                                                       const long* __restrict__ rows, int R,
                                                       const float* __restrict__ v) {
  const long c = (long)blockIdx.x * TPB + threadIdx.x;
  if (c >= Dl) return;
  const float x = v[c];
  for (int r = 0; r < R; ++r) U[rows[r] * ld + c] = x;
}

// U[i, :] *= s[i]   (grid.y = row)
__global__ __launch_bounds__(TPB) void row_scale_kernel(float* __restrict__ U, long ld, long Dl,
                                                        const float* __restrict__ s) {
  const int row = blockIdx.y;
  const float f = s[row];
  if (f == 1.0f) return;
  float* p = U + (long)row * ld;
  const long stride = (long)gridDim.x * TPB * 4;
  for (long c = ((long)blockIdx.x * TPB + threadIdx.x) * 4; c < Dl; c += stride) {
    if (c + 4 <= Dl && (ld & 3) == 0) {
      float4 v = *reinterpret_cast<float4*>(p + c);
      v.x *= f; v.y *= f; v.z *= f; v.w *= f;
      *reinterpret_cast<float4*>(p + c) = v;
    } else {
      for (long cc = c; cc < Dl && cc < c + 4; ++cc) p[cc] *= f;
    }
  }
}

}  // namespace

BLADES_API void blades_weighted_col_sum(const float* U, long ld, int K, long Dl, const float* w, float* out,
                                        hipStream_t st) {
  if (Dl <= 0) return;
  hipLaunchKernelGGL(weighted_col_sum_kernel, dim3(cdiv(cdiv(Dl, 4), TPB)), dim3(TPB), 0, st, U, ld, K, Dl, w, out);
}

// mode bit0: std, bit1: minmax, bit2: forge
BLADES_API void blades_col_stats(float* U, long ld, int K, long Dl, const float* mask, float* o_mean, float* o_std,
                                 float* o_min, float* o_max, int mode, float a_coef, float z_coef,
                                 const float* sign, const long* mal_rows, int n_mal, float* o_vec,
                                 hipStream_t st) {
  if (Dl <= 0) return;
  dim3 g(cdiv(cdiv(Dl, 4), TPB)), b(TPB);
#define LAUNCH(S, M, F)                                                                                        \
  hipLaunchKernelGGL((col_stats_kernel<S, M, F>), g, b, 0, st, U, ld, K, Dl, mask, o_mean, o_std, o_min, o_max, \
                     a_coef, z_coef, sign, mal_rows, n_mal, o_vec)
  switch (mode) {
    case 0: LAUNCH(false, false, false); break;
    case 1: LAUNCH(true, false, false); break;
    case 2: LAUNCH(false, true, false); break;
    case 3: LAUNCH(true, true, false); break;
    case 4: LAUNCH(false, false, true); break;
    case 5: LAUNCH(true, false, true); break;
    default: LAUNCH(true, true, true); break;
  }
#undef LAUNCH
}

BLADES_API void blades_set_rows(float* U, long ld, long Dl, const long* rows, int R, const float* v,
                                hipStream_t st) {
  if (Dl <= 0 || R <= 0) return;
  hipLaunchKernelGGL(set_rows_kernel, dim3(cdiv(Dl, TPB)), dim3(TPB), 0, st, U, ld, Dl, rows, R, v);
}

BLADES_API void blades_row_scale(float* U, long ld, int K, long Dl, const float* s, hipStream_t st) {
  if (Dl <= 0 || K <= 0) return;
  int gx = cdiv(cdiv(Dl, 4), TPB);
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(row_scale_kernel, dim3(gx, K), dim3(TPB), 0, st, U, ld, Dl, s);
}
