// Per-client batch normalisation (batch statistics only) for channels-last bf16 activations.
//
// x is [Kc][P][C] (client-major NHWC: P = B*H*W pixels of one client's batch, C channels
// contiguous); statistics are per (client, channel) — the group structure the reference gets
// implicitly by training one client at a time with BatchNorm2d(track_running_stats=False)
// (fllib/models/cifar10/resnet_cifar.py:14-30).  fp32 math, bf16 I/O.
//
// Forward:  stats (partials -> finalize: scale = gamma*invstd, shift = beta - mean*scale)
//           apply: y = scale*x + shift (+ residual) (ReLU)            — fused epilogue
// Backward: reduce: dbeta = sum g, dgamma = sum g*xhat with g = gy * [y > 0]
//           finalize: writes coef*dgamma / coef*dbeta straight into the update-matrix rows
//           apply: dx = scale*(g - dbeta/m - xhat*dgamma/m) (+ g as the residual-branch grad)
//
// Every pass is a pure stream (16 B per lane, all lanes of a pixel row contiguous) => HBM-bound.
// Reductions are two-stage with fixed order (no float atomics): bitwise reproducible.
#include "common.h"
#include <hip/hip_bf16.h>
#include <cstdlib>

namespace {

constexpr int TPB = 256;
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  __hip_bfloat16 h = __float2bfloat16(f);
  return *reinterpret_cast<unsigned short*>(&h);
}

// 8 consecutive channels of one pixel <-> 8 floats; T = unsigned short (bf16 bits, 16 B) or float (2 x 16 B)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<unsigned short>(const unsigned short* p, float (&v)[8]) {
  const u16x8 t = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = bf2f(t[j]);
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
// stores the values rounded to T and returns them (as the next layer will read them) in v
template <typename T> __device__ __forceinline__ void store8(T* p, float (&v)[8]);
template <> __device__ __forceinline__ void store8<unsigned short>(unsigned short* p, float (&v)[8]) {
  u16x8 t;
#pragma unroll
  for (int j = 0; j < 8; ++j) { t[j] = f2bf(v[j]); v[j] = bf2f(t[j]); }
  *reinterpret_cast<u16x8*>(p) = t;
}
template <> __device__ __forceinline__ void store8<float>(float* p, float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// partial[(c * nchunk + chunk) * C * 2 + ch * 2 + {0,1}]
// MODE 0: (sum x, sum x^2).  MODE 1: (sum g, sum g * xhat), g = gy gated by the ReLU bitmask (if any)
template <int MODE, typename T>
__global__ __launch_bounds__(TPB) void pcbn_reduce_kernel(const T* __restrict__ x,
                                                          const T* __restrict__ gy,
                                                          const unsigned char* __restrict__ mask,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, int P, int C,
                                                          int pix_per_chunk, int nchunk,
                                                          float* __restrict__ partial) {
  extern __shared__ float red[];  // [TPB][16]
  const int client = blockIdx.y, chunk = blockIdx.x;
  const int lpp = C >> 3;  // lanes per pixel
  const int cg = threadIdx.x % lpp, prow = threadIdx.x / lpp, rows = TPB / lpp;
  const int p0 = chunk * pix_per_chunk;
  int p1 = p0 + pix_per_chunk;
  if (p1 > P) p1 = P;
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = 0.f; b[j] = 0.f; }
  float mu[8], is[8];
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = mean[client * C + cg * 8 + j]; is[j] = invstd[client * C + cg * 8 + j]; }
  }
  const size_t base = (size_t)client * P * C + (size_t)cg * 8;
  for (int p = p0 + prow; p < p1; p += rows) {
    const size_t off = base + (size_t)p * C;
    float xv[8];
    load8<T>(x + off, xv);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float v = xv[j]; a[j] += v; b[j] += v * v; }
    } else {
      float gv[8];
      load8<T>(gy + off, gv);
      // one mask byte per lane: bit j = (output j of this lane's 8 channels was > 0 in the forward pass)
      const unsigned mb = mask != nullptr ? mask[off >> 3] : 0xFFu;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float g = gv[j];
        if (!((mb >> j) & 1u)) g = 0.f;
        a[j] += g;
        b[j] += g * (xv[j] - mu[j]) * is[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[threadIdx.x * 16 + j] = a[j]; red[threadIdx.x * 16 + 8 + j] = b[j]; }
  __syncthreads();
  // threads [0, lpp*16): one (channel-group, slot) each, fixed-order sum over the pixel rows
  for (int t = threadIdx.x; t < lpp * 16; t += TPB) {
    const int g = t / 16, s = t % 16;
    float acc = 0.f;
    for (int r = 0; r < rows; ++r) acc += red[(r * lpp + g) * 16 + s];
    const int ch = g * 8 + (s & 7), which = s >> 3;
    partial[((size_t)(client * nchunk + chunk) * C + ch) * 2 + which] = acc;
  }
}

// Sum of the per-chunk partials of 16 channels of one client by 16 thread groups (chunks k = part, part + 16, ...),
// combined in a fixed order through LDS: with the statistics coming from a convolution's epilogue a client has one
// chunk per output tile (256 on the 32 x 32 layers) and one thread per (client, channel) walked them serially —
// 36 us per call, 3 % of the headline round for two tiny kernels per BatchNorm.
constexpr int FIN_CH = 16, FIN_PARTS = TPB / FIN_CH;
__device__ __forceinline__ bool finalize_sums(const float* __restrict__ partial, int Kc, int C, int nchunk, int& c, int& ch,
                                              double& s0, double& s1) {
  __shared__ double red[FIN_PARTS][FIN_CH][2];
  const int groups = (C + FIN_CH - 1) / FIN_CH;
  c = blockIdx.x / groups;
  const int chl = threadIdx.x & (FIN_CH - 1), part = threadIdx.x / FIN_CH;
  ch = (blockIdx.x - c * groups) * FIN_CH + chl;
  double a = 0.0, b = 0.0;
  if (ch < C)
    for (int k = part; k < nchunk; k += FIN_PARTS) {
      const float2 q = *reinterpret_cast<const float2*>(partial + ((size_t)(c * nchunk + k) * C + ch) * 2);
      a += (double)q.x; b += (double)q.y;
    }
  red[part][chl][0] = a; red[part][chl][1] = b;
  __syncthreads();
  if (part != 0 || ch >= C) return false;
  s0 = 0.0; s1 = 0.0;
#pragma unroll
  for (int p = 0; p < FIN_PARTS; ++p) { s0 += red[p][chl][0]; s1 += red[p][chl][1]; }
  return true;
}

// forward finalize, parallel over chunks: one workgroup per (client, 16 channels)
__global__ __launch_bounds__(TPB) void pcbn_fwd_finalize_par_kernel(const float* __restrict__ partial, int Kc, int C,
                                                                    int nchunk, int P, float eps,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int gstride,
                                                                    float* __restrict__ mean, float* __restrict__ invstd,
                                                                    float* __restrict__ scale, float* __restrict__ shift) {
  int c, ch;
  double s, ss;
  if (!finalize_sums(partial, Kc, C, nchunk, c, ch, s, ss)) return;
  const int i = c * C + ch;
  const double m = s / P;
  double var = ss / P - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[(size_t)c * gstride + ch] : 1.f, b = beta ? beta[(size_t)c * gstride + ch] : 0.f;
  mean[i] = (float)m; invstd[i] = is;
  scale[i] = g * is; shift[i] = b - (float)m * g * is;
}

// forward finalize: one thread per (client, channel)
__global__ __launch_bounds__(TPB) void pcbn_fwd_finalize_kernel(const float* __restrict__ partial, int Kc, int C, int nchunk,
                                                                int P, float eps, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int gstride,
                                                                float* __restrict__ mean,
                                                                float* __restrict__ invstd, float* __restrict__ scale,
                                                                float* __restrict__ shift) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= Kc * C) return;
  const int c = i / C, ch = i % C;
  double s = 0.0, ss = 0.0;
  for (int k = 0; k < nchunk; ++k) {
    const float* q = partial + ((size_t)(c * nchunk + k) * C + ch) * 2;
    s += (double)q[0]; ss += (double)q[1];
  }
  const double m = s / P;
  double var = ss / P - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  // gstride = 0: one affine pair shared by all clients (FedSGD); = C: per-client pairs [Kc][C] (local steps > 1)
  const float g = gamma ? gamma[(size_t)c * gstride + ch] : 1.f, b = beta ? beta[(size_t)c * gstride + ch] : 0.f;
  mean[i] = (float)m; invstd[i] = is;
  scale[i] = g * is; shift[i] = b - (float)m * g * is;
}

template <typename T>
__global__ __launch_bounds__(TPB) void pcbn_fwd_apply_kernel(const T* __restrict__ x,
                                                             const T* __restrict__ res,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int P, int C, int relu,
                                                             T* __restrict__ y,
                                                             unsigned char* __restrict__ mask) {
  const int client = blockIdx.y;
  const int lpp = C >> 3;
  const size_t nvec = (size_t)P * lpp;
  const size_t base = (size_t)client * P * C;
  // TPB and the grid stride are multiples of lpp, so a thread keeps its channel group:
  // the per-(client, channel) tables live in registers for the whole sweep.
  const int cg = threadIdx.x % lpp;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = scale[client * C + cg * 8 + j]; sh[j] = shift[client * C + cg * 8 + j]; }
  for (size_t v = (size_t)blockIdx.x * TPB + threadIdx.x; v < nvec; v += (size_t)gridDim.x * TPB) {
    const size_t off = base + v * 8;
    float xv[8], rv[8], out[8];
    load8<T>(x + off, xv);
    if (res != nullptr) load8<T>(res + off, rv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float val = xv[j] * sc[j] + sh[j];
      if (res != nullptr) val += rv[j];
      if (relu) val = fmaxf(val, 0.f);
      out[j] = val;
    }
    store8<T>(y + off, out);
    unsigned mb = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) mb |= (out[j] > 0.f ? 1u : 0u) << j;  // gate on the value the next layer actually sees
    // ReLU gate for the backward pass, 1 bit per element instead of re-reading y (2 bytes per element)
    if (mask != nullptr) mask[off >> 3] = (unsigned char)mb;
  }
}

// backward finalize, parallel over chunks (see finalize_sums)
__global__ __launch_bounds__(TPB) void pcbn_bwd_finalize_par_kernel(const float* __restrict__ partial, int Kc, int C,
                                                                    int nchunk, int P, const float* __restrict__ gamma,
                                                                    int gstride, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd,
                                                                    const float* __restrict__ coef, float* __restrict__ U,
                                                                    long ld, long off_gamma, long off_beta, int accumulate,
                                                                    float* __restrict__ k_scale, float* __restrict__ k_b,
                                                                    float* __restrict__ k_c, int raw) {
  int c, ch;
  double sg, sgx;
  if (!finalize_sums(partial, Kc, C, nchunk, c, ch, sg, sgx)) return;
  const int i = c * C + ch;
  if (raw) sgx = (double)invstd[i] * (sgx - (double)mean[i] * sg);
  const float g = gamma ? gamma[(size_t)c * gstride + ch] : 1.f;
  if (U != nullptr) {
    float* row = U + (size_t)c * ld;
    const float s = coef[c];
    if (accumulate) { row[off_gamma + ch] += s * (float)sgx; row[off_beta + ch] += s * (float)sg; }
    else { row[off_gamma + ch] = s * (float)sgx; row[off_beta + ch] = s * (float)sg; }
  }
  const float ks = g * invstd[i], kb = (float)(sg / P), kc = (float)(sgx / P);
  k_scale[i] = ks;
  k_b[i] = -ks * invstd[i] * kc;
  k_c[i] = ks * (mean[i] * invstd[i] * kc - kb);
}

// backward finalize: dgamma/dbeta -> update-matrix rows (scaled), coefficient tables for apply
__global__ __launch_bounds__(TPB) void pcbn_bwd_finalize_kernel(const float* __restrict__ partial, int Kc, int C, int nchunk,
                                                                int P, const float* __restrict__ gamma, int gstride,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ coef,  // [Kc] sink scale
                                                                float* __restrict__ U, long ld, long off_gamma,
                                                                long off_beta, int accumulate,
                                                                float* __restrict__ k_scale,  // gamma*invstd
                                                                float* __restrict__ k_b,      // dbeta/m
                                                                float* __restrict__ k_c,      // dgamma/m
                                                                int raw) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= Kc * C) return;
  const int c = i / C, ch = i % C;
  double sg = 0.0, sgx = 0.0;
  for (int k = 0; k < nchunk; ++k) {
    const float* q = partial + ((size_t)(c * nchunk + k) * C + ch) * 2;
    sg += (double)q[0]; sgx += (double)q[1];
  }
  // raw: the partials are (sum g, sum g * x) from the epilogue of the convolution that produced the gradient
  // (conv_igemm.hip, bstat_x): sum g * xhat = invstd * (sum g x - mean * sum g), in double
  if (raw) sgx = (double)invstd[i] * (sgx - (double)mean[i] * sg);
  const float g = gamma ? gamma[(size_t)c * gstride + ch] : 1.f;
  if (U != nullptr) {
    float* row = U + (size_t)c * ld;
    const float s = coef[c];
    if (accumulate) { row[off_gamma + ch] += s * (float)sgx; row[off_beta + ch] += s * (float)sg; }
    else { row[off_gamma + ch] = s * (float)sgx; row[off_beta + ch] = s * (float)sg; }
  }
  // dx = ks*(g - db/m - xhat*dg/m) = A*g + Bx*x + Cc with xhat = (x - mean)*invstd
  const float ks = g * invstd[i], kb = (float)(sg / P), kc = (float)(sgx / P);
  k_scale[i] = ks;                              // A
  k_b[i] = -ks * invstd[i] * kc;                // Bx
  k_c[i] = ks * (mean[i] * invstd[i] * kc - kb);  // Cc
}

template <typename T>
__global__ __launch_bounds__(TPB) void pcbn_bwd_apply_kernel(const T* __restrict__ x,
                                                             const T* __restrict__ gy,
                                                             const unsigned char* __restrict__ mask,
                                                             const float* __restrict__ k_a,
                                                             const float* __restrict__ k_bx,
                                                             const float* __restrict__ k_cc, int P, int C,
                                                             T* __restrict__ dx,
                                                             T* __restrict__ gres) {
  const int client = blockIdx.y;
  const int lpp = C >> 3;
  const size_t nvec = (size_t)P * lpp;
  const size_t base = (size_t)client * P * C;
  const int cg = threadIdx.x % lpp;  // loop-invariant, see pcbn_fwd_apply_kernel
  float ka[8], kb[8], kc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = client * C + cg * 8 + j;
    ka[j] = k_a[ch]; kb[j] = k_bx[ch]; kc[j] = k_cc[ch];
  }
  for (size_t v = (size_t)blockIdx.x * TPB + threadIdx.x; v < nvec; v += (size_t)gridDim.x * TPB) {
    const size_t off = base + v * 8;
    float xv[8], gv[8], o[8], r[8];
    load8<T>(x + off, xv);
    load8<T>(gy + off, gv);
    const unsigned mb = mask != nullptr ? mask[off >> 3] : 0xFFu;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float g = gv[j];
      if (!((mb >> j) & 1u)) g = 0.f;
      o[j] = ka[j] * g + kb[j] * xv[j] + kc[j];
      r[j] = g;
    }
    store8<T>(dx + off, o);
    if (gres != nullptr) store8<T>(gres + off, r);
  }
}

// chunk-parallel finalize where a client has enough chunks to share out (BLADES_PCBN_FINALIZE=serial: never, for A/B)
inline bool par_finalize(int nchunk) {
  static const bool off = [] { const char* e = getenv("BLADES_PCBN_FINALIZE"); return e && e[0] == 's'; }();
  return !off && nchunk >= 8;
}

inline int chunk_pixels(int C, int esize = 2) {
  int p = 262144 / (C * esize);
  if (p < 64) p = 64;
  return p;
}

template <typename T>
int pcbn_forward_t(const void* x, const void* res, const float* gamma, const float* beta, int gstride, int Kc, int P, int C, float eps,
                   int relu, float* partial, float* mean, float* invstd, float* scale, float* shift, void* y,
                   void* relu_mask, int pre_nchunk, hipStream_t st) {
  const int ppc = chunk_pixels(C, sizeof(T));
  // pre_nchunk > 0: `partial` already holds the sums, in pre_nchunk chunks per client — written by the epilogue of
  // the convolution that produced x (conv_igemm.hip), so x is read once here instead of twice
  const int nch = pre_nchunk > 0 ? pre_nchunk : cdiv(P, ppc);
  if (pre_nchunk <= 0)
    hipLaunchKernelGGL((pcbn_reduce_kernel<0, T>), dim3(nch, Kc), dim3(TPB), TPB * 16 * sizeof(float), st, (const T*)x,
                       (const T*)nullptr, nullptr, nullptr, nullptr, P, C, ppc, nch, partial);
  if (par_finalize(nch))
    hipLaunchKernelGGL(pcbn_fwd_finalize_par_kernel, dim3(Kc * cdiv(C, FIN_CH)), dim3(TPB), 0, st, partial, Kc, C, nch,
                       P, eps, gamma, beta, gstride, mean, invstd, scale, shift);
  else
    hipLaunchKernelGGL(pcbn_fwd_finalize_kernel, dim3(cdiv((long)Kc * C, TPB)), dim3(TPB), 0, st, partial, Kc, C, nch, P,
                       eps, gamma, beta, gstride, mean, invstd, scale, shift);
  int gx = cdiv((long)P * (C / 8), TPB);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL((pcbn_fwd_apply_kernel<T>), dim3(gx, Kc), dim3(TPB), 0, st, (const T*)x, (const T*)res, scale,
                     shift, P, C, relu, (T*)y, (unsigned char*)relu_mask);
  return 0;
}

template <typename T>
int pcbn_backward_t(const void* x, const void* gy, const void* relu_mask, const float* gamma, int gstride, const float* mean,
                    const float* invstd, int Kc, int P, int C, const float* coef, float* U, long ld, long off_gamma,
                    long off_beta, int accumulate, float* partial, float* k_scale, float* k_b, float* k_c, void* dx,
                    void* gres, int pre_nchunk, hipStream_t st) {
  // pre_nchunk > 0: `partial` already holds the raw sums, written by the data-gradient kernel that produced gy
  const int ppc = chunk_pixels(C, sizeof(T)), nch = pre_nchunk > 0 ? pre_nchunk : cdiv(P, ppc);
  if (pre_nchunk <= 0)
    hipLaunchKernelGGL((pcbn_reduce_kernel<1, T>), dim3(nch, Kc), dim3(TPB), TPB * 16 * sizeof(float), st, (const T*)x,
                       (const T*)gy, (const unsigned char*)relu_mask, mean, invstd, P, C, ppc, nch, partial);
  if (par_finalize(nch))
    hipLaunchKernelGGL(pcbn_bwd_finalize_par_kernel, dim3(Kc * cdiv(C, FIN_CH)), dim3(TPB), 0, st, partial, Kc, C, nch,
                       P, gamma, gstride, mean, invstd, coef, U, ld, off_gamma, off_beta, accumulate, k_scale, k_b, k_c,
                       pre_nchunk > 0 ? 1 : 0);
  else
    hipLaunchKernelGGL(pcbn_bwd_finalize_kernel, dim3(cdiv((long)Kc * C, TPB)), dim3(TPB), 0, st, partial, Kc, C, nch, P,
                       gamma, gstride, mean, invstd, coef, U, ld, off_gamma, off_beta, accumulate, k_scale, k_b, k_c,
                       pre_nchunk > 0 ? 1 : 0);
  int gx = cdiv((long)P * (C / 8), TPB);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL((pcbn_bwd_apply_kernel<T>), dim3(gx, Kc), dim3(TPB), 0, st, (const T*)x, (const T*)gy,
                     (const unsigned char*)relu_mask, k_scale, k_b, k_c, P, C, (T*)dx, (T*)gres);
  return 0;
}

}  // namespace

// esize: bytes per activation element (2 = bf16, 4 = fp32)
BLADES_API int blades_pcbn_nchunk(int P, int C, int esize) { return cdiv(P, chunk_pixels(C, esize)); }

// workspace `partial`: Kc * nchunk * C * 2 floats (pre_nchunk > 0: input, see pcbn_forward_t).
BLADES_API int blades_pcbn_forward(const void* x, const void* res, const float* gamma, const float* beta, int Kc,
                                   int P, int C, float eps, int relu, float* partial, float* mean, float* invstd,
                                   float* scale, float* shift, void* y, void* relu_mask, int esize, int gstride,
                                   int pre_nchunk, hipStream_t st) {
  if (C % 8 != 0 || C > 2048 || (TPB % (C / 8)) != 0 || Kc <= 0 || P <= 0) return 1;
  if (esize == 4)
    return pcbn_forward_t<float>(x, res, gamma, beta, gstride, Kc, P, C, eps, relu, partial, mean, invstd, scale, shift, y,
                                 relu_mask, pre_nchunk, st);
  if (esize == 2)
    return pcbn_forward_t<unsigned short>(x, res, gamma, beta, gstride, Kc, P, C, eps, relu, partial, mean, invstd, scale,
                                          shift, y, relu_mask, pre_nchunk, st);
  return 2;
}

// workspace: partial (as forward), k_scale/k_b/k_c: Kc*C floats each.
// relu_mask: the bitmask written by the forward pass (Kc*P*C/8 bytes), or nullptr when there was no ReLU.
BLADES_API int blades_pcbn_backward2(const void* x, const void* gy, const void* relu_mask, const float* gamma,
                                     const float* mean, const float* invstd, int Kc, int P, int C, const float* coef,
                                     float* U, long ld, long off_gamma, long off_beta, int accumulate, float* partial,
                                     float* k_scale, float* k_b, float* k_c, void* dx, void* gres, int esize,
                                     int gstride, int pre_nchunk, hipStream_t st) {
  if (C % 8 != 0 || C > 2048 || (TPB % (C / 8)) != 0 || Kc <= 0 || P <= 0) return 1;
  if (esize == 4)
    return pcbn_backward_t<float>(x, gy, relu_mask, gamma, gstride, mean, invstd, Kc, P, C, coef, U, ld, off_gamma, off_beta,
                                  accumulate, partial, k_scale, k_b, k_c, dx, gres, pre_nchunk, st);
  if (esize == 2 && pre_nchunk <= 0)
    return pcbn_backward_t<unsigned short>(x, gy, relu_mask, gamma, gstride, mean, invstd, Kc, P, C, coef, U, ld, off_gamma,
                                           off_beta, accumulate, partial, k_scale, k_b, k_c, dx, gres, 0, st);
  return 2;
}

BLADES_API int blades_pcbn_backward(const void* x, const void* gy, const void* relu_mask, const float* gamma,
                                    const float* mean, const float* invstd, int Kc, int P, int C, const float* coef,
                                    float* U, long ld, long off_gamma, long off_beta, int accumulate, float* partial,
                                    float* k_scale, float* k_b, float* k_c, void* dx, void* gres, int esize,
                                    int gstride, hipStream_t st) {
  return blades_pcbn_backward2(x, gy, relu_mask, gamma, mean, invstd, Kc, P, C, coef, U, ld, off_gamma, off_beta,
                               accumulate, partial, k_scale, k_b, k_c, dx, gres, esize, gstride, 0, st);
}
