// Per-client LayerNorm over the last dimension (CCT / CVT transformer blocks, reference
// fllib/models/backbones/cctnets/utils/transformers.py:17-47 runs nn.LayerNorm one client at a time).
//
//   x: [Kc][R][C] rows (client-major), y = (x - mean_row) * rstd_row * gamma + beta
//   backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = gy * gamma
//             dgamma_c = sum_{rows of client c} gy * xhat,  dbeta_c = sum gy   -> coef[c] * (...) into the update rows
//
// One wave per row (lane l owns channels l, l + 64, ...: coalesced 256-byte accesses, row statistics by DPP/shuffle
// reductions, no LDS), fp32 math, fp32 or bf16 I/O.  The per-client affine gradients are reduced in two fixed-order
// stages (per-workgroup partials, then one thread per (client, channel)): bitwise reproducible, no float atomics.
#include "common.h"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const unsigned short* p) { return __uint_as_float(((unsigned)*p) << 16); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(unsigned short* p, float v) {
  const __bf16 b = (__bf16)v;
  *p = __builtin_bit_cast(unsigned short, b);
}

template <typename T, int NE>
__global__ __launch_bounds__(TPB) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, long rows, int C, float eps,
                                                     T* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, long gstride, long bstride, int R) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ga[NE], be[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int ch = j * 64 + lane;
    ga[j] = (gamma && ch < C) ? gamma[ch] : 1.f;
    be[j] = (beta && ch < C) ? beta[ch] : 0.f;
  }
  // gstride / bstride != 0: one affine pair per client (local steps > 1: row r belongs to client r / R, whose gamma
  // starts gstride floats behind the previous client's — rows of the engine's per-client weight matrix)
  const bool per_client = gstride != 0 || bstride != 0;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    if (per_client) {
      const long cl = r / R;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int ch = j * 64 + lane;
        ga[j] = (gamma && ch < C) ? gamma[cl * gstride + ch] : 1.f;
        be[j] = (beta && ch < C) ? beta[cl * bstride + ch] : 0.f;
      }
    }
    const T* xr = x + r * C;
    float v[NE], s = 0.f;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int ch = j * 64 + lane;
      v[j] = ch < C ? ldf(xr + ch) : 0.f;
      s += v[j];
    }
    const float mu = wave_sum(s) / C;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const float d = (j * 64 + lane) < C ? v[j] - mu : 0.f;
      ss += d * d;
    }
    const float rs = rsqrtf(wave_sum(ss) / C + eps);
    T* yr = y + r * C;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int ch = j * 64 + lane;
      if (ch < C) stf(yr + ch, (v[j] - mu) * rs * ga[j] + be[j]);
    }
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
  }
}

// grid (nchunk, Kc): rows [chunk*rpc, ...) of client blockIdx.y.  partial[((c*nchunk + chunk)*C + ch)*2 + {0,1}]
template <typename T, int NE>
__global__ __launch_bounds__(TPB) void ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, int R, int C, int rpc, int nchunk,
                                                     T* __restrict__ dx, float* __restrict__ partial, long gstride) {
  __shared__ float red[4][NE * 64 * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y, chunk = blockIdx.x;
  float ga[NE], dg[NE], db[NE];
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int ch = j * 64 + lane;
    ga[j] = (gamma && ch < C) ? gamma[(long)c * gstride + ch] : 1.f;
    dg[j] = 0.f; db[j] = 0.f;
  }
  const int r_lo = chunk * rpc;
  int r_hi = r_lo + rpc;
  if (r_hi > R) r_hi = R;
  for (int rr = r_lo + wave; rr < r_hi; rr += 4) {
    const long r = (long)c * R + rr;
    const T* xr = x + r * C;
    const T* gr = gy + r * C;
    const float mu = mean[r], rs = rstd[r];
    float xh[NE], g[NE], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int ch = j * 64 + lane;
      const bool ok = ch < C;
      const float gv = ok ? ldf(gr + ch) : 0.f;
      xh[j] = ok ? (ldf(xr + ch) - mu) * rs : 0.f;
      g[j] = gv * ga[j];
      db[j] += gv;
      dg[j] += gv * xh[j];
      s1 += g[j];
      s2 += g[j] * xh[j];
    }
    const float m1 = wave_sum(s1) / C, m2 = wave_sum(s2) / C;
    T* dr = dx + r * C;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int ch = j * 64 + lane;
      if (ch < C) stf(dr + ch, rs * (g[j] - m1 - xh[j] * m2));
    }
  }
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    red[wave][(j * 64 + lane) * 2 + 0] = dg[j];
    red[wave][(j * 64 + lane) * 2 + 1] = db[j];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < C * 2; t += TPB)
    partial[((size_t)(c * nchunk + chunk) * C) * 2 + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

__global__ __launch_bounds__(TPB) void ln_finalize_kernel(const float* __restrict__ partial, int Kc, int C, int nchunk,
                                                          const float* __restrict__ coef, float* __restrict__ U,
                                                          long ld, long off_gamma, long off_beta, int accumulate) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= Kc * C) return;
  const int c = i / C, ch = i % C;
  float sg = 0.f, sb = 0.f;
  for (int k = 0; k < nchunk; ++k) {
    const float* q = partial + ((size_t)(c * nchunk + k) * C + ch) * 2;
    sg += q[0]; sb += q[1];
  }
  float* row = U + (size_t)c * ld;
  const float s = coef ? coef[c] : 1.f;
  if (off_gamma >= 0) row[off_gamma + ch] = accumulate ? row[off_gamma + ch] + s * sg : s * sg;
  if (off_beta >= 0) row[off_beta + ch] = accumulate ? row[off_beta + ch] + s * sb : s * sb;
}

inline int rows_per_chunk(int R) { return R >= 512 ? 128 : (R >= 64 ? 32 : 8); }

template <typename T, int NE>
void run_fwd(const void* x, const float* gamma, const float* beta, long rows, int C, float eps, void* y, float* mean,
             float* rstd, long gstride, long bstride, int R, hipStream_t st) {
  int g = cdiv(rows, 4);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL((ln_fwd_kernel<T, NE>), dim3(g), dim3(TPB), 0, st, (const T*)x, gamma, beta, rows, C, eps, (T*)y,
                     mean, rstd, gstride, bstride, R);
}
template <typename T, int NE>
void run_bwd(const void* x, const void* gy, const float* gamma, const float* mean, const float* rstd, int Kc, int R,
             int C, void* dx, float* partial, long gstride, hipStream_t st) {
  const int rpc = rows_per_chunk(R), nch = cdiv(R, rpc);
  hipLaunchKernelGGL((ln_bwd_kernel<T, NE>), dim3(nch, Kc), dim3(TPB), 0, st, (const T*)x, (const T*)gy, gamma, mean,
                     rstd, R, C, rpc, nch, (T*)dx, partial, gstride);
}

}  // namespace

BLADES_API int blades_ln_nchunk(int R) { return cdiv(R, rows_per_chunk(R)); }

#define LN_DISPATCH(FN, ...)                                                   \
  do {                                                                         \
    const int ne = (C + 63) / 64;                                              \
    if (esize == 4) {                                                          \
      if (ne <= 2) FN<float, 2>(__VA_ARGS__);                                  \
      else if (ne <= 4) FN<float, 4>(__VA_ARGS__);                             \
      else if (ne <= 8) FN<float, 8>(__VA_ARGS__);                             \
      else FN<float, 16>(__VA_ARGS__);                                         \
    } else {                                                                   \
      if (ne <= 2) FN<unsigned short, 2>(__VA_ARGS__);                         \
      else if (ne <= 4) FN<unsigned short, 4>(__VA_ARGS__);                    \
      else if (ne <= 8) FN<unsigned short, 8>(__VA_ARGS__);                    \
      else FN<unsigned short, 16>(__VA_ARGS__);                                \
    }                                                                          \
  } while (0)

// gstride / bstride: floats between two clients' gamma / beta (0 = one pair shared by all rows); R = rows per client.
BLADES_API int blades_ln_forward2(const void* x, const float* gamma, const float* beta, long rows, int C, float eps,
                                  void* y, float* mean, float* rstd, int esize, long gstride, long bstride, int R,
                                  hipStream_t st) {
  if (rows <= 0 || C <= 0 || C > 1024 || (esize != 2 && esize != 4)) return 1;
  if ((gstride != 0 || bstride != 0) && (R <= 0 || rows % R != 0)) return 3;
  LN_DISPATCH(run_fwd, x, gamma, beta, rows, C, eps, y, mean, rstd, gstride, bstride, R, st);
  return 0;
}

BLADES_API int blades_ln_forward(const void* x, const float* gamma, const float* beta, long rows, int C, float eps,
                                 void* y, float* mean, float* rstd, int esize, hipStream_t st) {
  return blades_ln_forward2(x, gamma, beta, rows, C, eps, y, mean, rstd, esize, 0, 0, 1, st);
}

// partial: Kc * nchunk * C * 2 floats.  off_gamma / off_beta < 0: that gradient is not written (no affine parameter).
BLADES_API int blades_ln_backward2(const void* x, const void* gy, const float* gamma, const float* mean,
                                   const float* rstd, int Kc, int R, int C, const float* coef, float* U, long ld,
                                   long off_gamma, long off_beta, int accumulate, float* partial, void* dx, int esize,
                                   long gstride, hipStream_t st) {
  if (Kc <= 0 || R <= 0 || C <= 0 || C > 1024 || Kc > 65535 || (esize != 2 && esize != 4)) return 1;
  LN_DISPATCH(run_bwd, x, gy, gamma, mean, rstd, Kc, R, C, dx, partial, gstride, st);
  if (U != nullptr && (off_gamma >= 0 || off_beta >= 0)) {
    const int nch = cdiv(R, rows_per_chunk(R));
    hipLaunchKernelGGL(ln_finalize_kernel, dim3(cdiv((long)Kc * C, TPB)), dim3(TPB), 0, st, partial, Kc, C, nch, coef, U,
                       ld, off_gamma, off_beta, accumulate);
  }
  return 0;
}

BLADES_API int blades_ln_backward(const void* x, const void* gy, const float* gamma, const float* mean,
                                  const float* rstd, int Kc, int R, int C, const float* coef, float* U, long ld,
                                  long off_gamma, long off_beta, int accumulate, float* partial, void* dx, int esize,
                                  hipStream_t st) {
  return blades_ln_backward2(x, gy, gamma, mean, rstd, Kc, R, C, coef, U, ld, off_gamma, off_beta, accumulate, partial,
                             dx, esize, 0, st);
}
