// Streaming kernels of the fused executor that are not matrix work:
//
//   bias_grad     per-client bias gradient  U[c, off + ch] (+)= coef[c] * sum_r g[c, r, ch]     (two-stage, fixed order)
//   maxpool       channels-last max-pool forward / backward with the preceding ReLU folded in   (relu o maxpool =
//                 maxpool o relu); the forward pass leaves the window position of the maximum (one byte per output
//                 element; 255 = nothing flows back), the backward pass is a gather — deterministic for overlapping
//                 windows (the CCT tokenizers pool 3x3 / stride 2)
//   avgpool       global average pool of a channels-last map, forward / backward (the ResNet head)
//   gelu / relu   forward and backward (erf GELU of the compact transformers' MLP; stand-alone ReLU of the MLP)
//   ce_clamp      the reference task's loss, per client: clamp(mean_b CE(z_cb, y_cb), 0, 1e6), and the gradient of
//                 sum_c loss_c with respect to the logits in the same kernel (fllib/tasks/mnist.py:8-18)
//
// fp32 or bf16 tensors (template T), fp32 arithmetic, 16 bytes per lane wherever the layout allows it.  Every reduction
// has a fixed order: bitwise reproducible.
#include "common.h"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const unsigned short* p) { return __uint_as_float((unsigned)(*p) << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(unsigned short* p, float v) {
  const __bf16 b = (__bf16)v;
  *p = __builtin_bit_cast(unsigned short, b);
}

template <typename T> struct Vec;  // EPV elements = 16 bytes
template <> struct Vec<float> {
  static constexpr int EPV = 4;
  typedef float4 raw;
  __device__ static void load(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ static void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// bias gradient.  g: [Kc][R][C] contiguous.  Stage 1: block (chunk, client) sums its rows per channel — thread (rl, cl):
// rows rl, rl + RL, ... of the chunk, channels cl, cl + CL, ...; the RL partial rows are combined through LDS in row-lane
// order.  Stage 2: one thread per (client, channel) adds the chunks in order, scales, writes the update row.
template <typename T>
__global__ __launch_bounds__(TPB) void bias_grad_partial_kernel(const T* __restrict__ g, int R, int C, int rows_per_chunk,
                                                                int nchunk, int CL, float* __restrict__ partial) {
  extern __shared__ float red[];  // [RL][CL] per channel pass
  const int client = blockIdx.y, chunk = blockIdx.x;
  const int RL = TPB / CL, cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int r0 = chunk * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  if (r1 > R) r1 = R;
  const T* base = g + (size_t)client * R * C;
  for (int c0 = 0; c0 < C; c0 += CL) {
    const int ch = c0 + cl;
    float a = 0.f;
    if (ch < C)
      for (int r = r0 + rl; r < r1; r += RL) a += ld1(base + (size_t)r * C + ch);
    red[rl * CL + cl] = a;
    __syncthreads();
    if (rl == 0 && ch < C) {
      float s = 0.f;
      for (int j = 0; j < RL; ++j) s += red[j * CL + cl];
      partial[((size_t)client * nchunk + chunk) * C + ch] = s;
    }
    __syncthreads();
  }
}

// fp32, C % 4 == 0: four channels (16 bytes) per lane, so a wave's load covers whole 128-byte lines of several rows
__global__ __launch_bounds__(TPB) void bias_grad_partial4_kernel(const float* __restrict__ g, int R, int C, int rows_per_chunk,
                                                                 int nchunk, int CL, float* __restrict__ partial) {
  extern __shared__ float red[];  // [RL][CL][4] per channel pass
  const int client = blockIdx.y, chunk = blockIdx.x;
  const int RL = TPB / CL, cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int r0 = chunk * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  if (r1 > R) r1 = R;
  const float* base = g + (size_t)client * R * C;
  for (int c0 = 0; c0 < C; c0 += CL * 4) {
    const int ch = c0 + cl * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < C)
      for (int r = r0 + rl; r < r1; r += RL) {
        const float4 t = *reinterpret_cast<const float4*>(base + (size_t)r * C + ch);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
    reinterpret_cast<float4*>(red)[rl * CL + cl] = a;
    __syncthreads();
    if (rl == 0 && ch < C) {
      float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = 0; j < RL; ++j) {
        const float4 t = reinterpret_cast<const float4*>(red)[j * CL + cl];
        s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w;
      }
      *reinterpret_cast<float4*>(partial + ((size_t)client * nchunk + chunk) * C + ch) = s4;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(TPB) void bias_grad_final_kernel(const float* __restrict__ partial, int Kc, int C, int nchunk,
                                                              const float* __restrict__ coef, float* __restrict__ U,
                                                              long ld, long off, int accumulate) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= Kc * C) return;
  const int c = i / C, ch = i % C;
  float s = 0.f;
  for (int k = 0; k < nchunk; ++k) s += partial[((size_t)c * nchunk + k) * C + ch];
  float* dst = U + (size_t)c * ld + off + ch;
  const float v = coef[c] * s;
  *dst = accumulate ? *dst + v : v;
}

// ---------------------------------------------------------------------------------------------------------------
// max-pool, channels-last.  x [N][H][W][C] -> y [N][Ho][Wo][C], idx [N][Ho][Wo][C] (uint8: r * kw + s of the first
// maximum in row-major window order, like ATen; 255 = the folded ReLU clipped it or the window was empty).
template <typename T>
__global__ __launch_bounds__(TPB) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          unsigned char* __restrict__ idx, long total, int H, int W, int C,
                                                          int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw,
                                                          int relu) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int c = (int)(i % C);
    long t = i / C;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const long n = t / Ho;
    float best = -INFINITY;
    int bi = 255;
    for (int r = 0; r < kh; ++r) {
      const int ih = oh * sh - ph + r;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int s = 0; s < kw; ++s) {
        const int iw = ow * sw - pw + s;
        if ((unsigned)iw >= (unsigned)W) continue;
        const float v = ld1(x + ((n * H + ih) * W + iw) * C + c);
        if (v > best || bi == 255) { best = v; bi = r * kw + s; }
      }
    }
    if (relu && !(best > 0.f)) { best = 0.f; bi = 255; }
    st1(y + i, best);
    idx[i] = (unsigned char)bi;
  }
}

// fp32, C % 4 == 0: four channels per lane (16-byte loads / stores, one 4-byte load of the window positions); same
// comparisons and the same order of additions per element as the one-channel kernels below
__global__ __launch_bounds__(TPB) void maxpool_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           unsigned char* __restrict__ idx, long total4, int H, int W, int C4,
                                                           int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw,
                                                           int relu) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long)gridDim.x * TPB) {
    const int c4 = (int)(i % C4);
    long t = i / C4;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const long n = t / Ho;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {255, 255, 255, 255};
    for (int r = 0; r < kh; ++r) {
      const int ih = oh * sh - ph + r;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int s = 0; s < kw; ++s) {
        const int iw = ow * sw - pw + s;
        if ((unsigned)iw >= (unsigned)W) continue;
        const float4 q = reinterpret_cast<const float4*>(x)[((n * H + ih) * W + iw) * C4 + c4];
        const float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (v[j] > best[j] || bi[j] == 255) { best[j] = v[j]; bi[j] = r * kw + s; }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (relu && !(best[j] > 0.f)) { best[j] = 0.f; bi[j] = 255; }
    reinterpret_cast<float4*>(y)[i] = make_float4(best[0], best[1], best[2], best[3]);
    reinterpret_cast<unsigned*>(idx)[i] = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
  }
}

__global__ __launch_bounds__(TPB) void maxpool_bwd4_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                           float* __restrict__ dx, long total4, int H, int W, int C4, int Ho,
                                                           int Wo, int kh, int kw, int sh, int sw, int ph, int pw) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long)gridDim.x * TPB) {
    const int c4 = (int)(i % C4);
    long t = i / C4;
    const int iw = (int)(t % W); t /= W;
    const int ih = (int)(t % H);
    const long n = t / H;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int oh_lo = (ih + ph - kh + 1 + sh - 1);
    oh_lo = oh_lo <= 0 ? 0 : oh_lo / sh;
    int oh_hi = (ih + ph) / sh;
    if (oh_hi > Ho - 1) oh_hi = Ho - 1;
    int ow_lo = (iw + pw - kw + 1 + sw - 1);
    ow_lo = ow_lo <= 0 ? 0 : ow_lo / sw;
    int ow_hi = (iw + pw) / sw;
    if (ow_hi > Wo - 1) ow_hi = Wo - 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int r = ih - (oh * sh - ph);
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const unsigned want = (unsigned)(r * kw + (iw - (ow * sw - pw)));
        const long o = ((n * Ho + oh) * Wo + ow) * C4 + c4;
        const unsigned m = reinterpret_cast<const unsigned*>(idx)[o];
        if (((m & 0xffu) == want) | (((m >> 8) & 0xffu) == want) | (((m >> 16) & 0xffu) == want) | ((m >> 24) == want)) {
          const float4 q = reinterpret_cast<const float4*>(gy)[o];
          if ((m & 0xffu) == want) a[0] += q.x;
          if (((m >> 8) & 0xffu) == want) a[1] += q.y;
          if (((m >> 16) & 0xffu) == want) a[2] += q.z;
          if ((m >> 24) == want) a[3] += q.w;
        }
      }
    }
    reinterpret_cast<float4*>(dx)[i] = make_float4(a[0], a[1], a[2], a[3]);
  }
}

// dx[n, ih, iw, c] = sum of gy over the output positions whose window has its maximum here (gather: no atomics)
template <typename T>
__global__ __launch_bounds__(TPB) void maxpool_bwd_kernel(const T* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                          T* __restrict__ dx, long total, int H, int W, int C, int Ho,
                                                          int Wo, int kh, int kw, int sh, int sw, int ph, int pw) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int c = (int)(i % C);
    long t = i / C;
    const int iw = (int)(t % W); t /= W;
    const int ih = (int)(t % H);
    const long n = t / H;
    float a = 0.f;
    // output rows oh with oh*sh - ph <= ih <= oh*sh - ph + kh - 1
    int oh_lo = (ih + ph - kh + 1 + sh - 1);
    oh_lo = oh_lo <= 0 ? 0 : oh_lo / sh;
    int oh_hi = (ih + ph) / sh;
    if (oh_hi > Ho - 1) oh_hi = Ho - 1;
    int ow_lo = (iw + pw - kw + 1 + sw - 1);
    ow_lo = ow_lo <= 0 ? 0 : ow_lo / sw;
    int ow_hi = (iw + pw) / sw;
    if (ow_hi > Wo - 1) ow_hi = Wo - 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int r = ih - (oh * sh - ph);
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int s = iw - (ow * sw - pw);
        const long o = ((n * Ho + oh) * Wo + ow) * C + c;
        if (idx[o] == (unsigned char)(r * kw + s)) a += ld1(gy + o);
      }
    }
    st1(dx + i, a);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// global average pool: x [N][P][C] -> y [N][C];  backward: dx[n, p, c] = gy[n, c] / P
template <typename T>
__global__ __launch_bounds__(TPB) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long NC, int P, int C) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= NC) return;
  const long n = i / C;
  const int c = (int)(i % C);
  float a = 0.f;
  for (int p = 0; p < P; ++p) a += ld1(x + (n * P + p) * C + c);
  st1(y + i, a / (float)P);
}

template <typename T>
__global__ __launch_bounds__(TPB) void avgpool_bwd_kernel(const T* __restrict__ gy, T* __restrict__ dx, long total, int P,
                                                          int C) {
  const float inv = 1.f / (float)P;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int c = (int)(i % C);
    const long n = i / ((long)P * C);
    st1(dx + i, ld1(gy + n * C + c) * inv);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// activations.  MODE 0: GELU (erf form, nn.GELU / F.gelu default)   1: ReLU
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_d(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

template <typename T, int MODE>
__global__ __launch_bounds__(TPB) void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
    const float v = ld1(x + i);
    st1(y + i, MODE == 0 ? gelu_f(v) : fmaxf(v, 0.f));
  }
}
// backward from the saved INPUT x (GELU) or OUTPUT y (ReLU; y > 0 <=> x > 0)
template <typename T, int MODE>
__global__ __launch_bounds__(TPB) void act_bwd_kernel(const T* __restrict__ s, const T* __restrict__ gy, T* __restrict__ dx,
                                                      long n) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
    const float v = ld1(s + i), g = ld1(gy + i);
    st1(dx + i, MODE == 0 ? g * gelu_d(v) : (v > 0.f ? g : 0.f));
  }
}
// fp32, 16 bytes per lane (n % 4 == 0, aligned)
template <int MODE>
__global__ __launch_bounds__(TPB) void act_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long)gridDim.x * TPB) {
    float v[4];
    Vec<float>::load(x + i * 4, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = MODE == 0 ? gelu_f(v[j]) : fmaxf(v[j], 0.f);
    Vec<float>::store(y + i * 4, v);
  }
}
template <int MODE>
__global__ __launch_bounds__(TPB) void act_bwd4_kernel(const float* __restrict__ s, const float* __restrict__ gy,
                                                       float* __restrict__ dx, long n4) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long)gridDim.x * TPB) {
    float v[4], g[4];
    Vec<float>::load(s + i * 4, v);
    Vec<float>::load(gy + i * 4, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = MODE == 0 ? g[j] * gelu_d(v[j]) : (v[j] > 0.f ? g[j] : 0.f);
    Vec<float>::store(dx + i * 4, g);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// clamped cross-entropy per client + gradient.  logits [Kc][B][C] fp32 (row stride ld), labels [Kc][B] int64.
//   loss[c]   = clamp(mean_b (lse(z_cb) - z_cb[y_cb]), lo, hi)
//   dz[c,b,:] = gl[c] * gate_c * (softmax(z_cb) - onehot(y_cb)) / B,   gate_c = [lo <= mean <= hi]  (torch.clamp)
// One block per client: a lane per sample row for the row pass, the block mean in fixed order, then the gradient pass.
__global__ __launch_bounds__(TPB) void ce_clamp_kernel(const float* __restrict__ z, long ld, const long* __restrict__ labels,
                                                       int B, int C, float lo, float hi, const float* __restrict__ gl,
                                                       float* __restrict__ loss, float* __restrict__ dz) {
  __shared__ float red[TPB];
  __shared__ float s_gate;
  const int c = blockIdx.x;
  const float* zc = z + (size_t)c * B * ld;
  const long* yc = labels + (size_t)c * B;
  float part = 0.f;
  for (int b = threadIdx.x; b < B; b += TPB) {
    const float* row = zc + (size_t)b * ld;
    long y = yc[b];
    if (y < 0 || y >= C) y = 0;  // host validates; stay in bounds regardless
    float m = -INFINITY;
    for (int k = 0; k < C; ++k) m = fmaxf(m, row[k]);
    float s = 0.f;
    for (int k = 0; k < C; ++k) s += __expf(row[k] - m);
    part += m + __logf(s) - row[y];
  }
  red[threadIdx.x] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    const int n = B < TPB ? B : TPB;
    for (int j = 0; j < n; ++j) t += red[j];
    const float mean = t / (float)B;
    const float cl = fminf(fmaxf(mean, lo), hi);
    loss[c] = cl;
    s_gate = (mean >= lo && mean <= hi) ? 1.f : 0.f;
  }
  __syncthreads();
  if (dz == nullptr) return;
  const float scale = s_gate * (gl ? gl[c] : 1.f) / (float)B;
  float* dc = dz + (size_t)c * B * C;
  for (int b = threadIdx.x; b < B; b += TPB) {
    const float* row = zc + (size_t)b * ld;
    long y = yc[b];
    if (y < 0 || y >= C) y = 0;
    float m = -INFINITY;
    for (int k = 0; k < C; ++k) m = fmaxf(m, row[k]);
    float s = 0.f;
    for (int k = 0; k < C; ++k) s += __expf(row[k] - m);
    const float inv = 1.f / s;
    for (int k = 0; k < C; ++k)
      dc[(size_t)b * C + k] = scale * (__expf(row[k] - m) * inv - (k == y ? 1.f : 0.f));
  }
}

inline int grid_for(long n, int cap = 4096) {
  long g = (n + TPB - 1) / TPB;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

// ---------------------------------------------------------------------------------------------------------------
// sequence pooling of the compact transformers (reference fllib/models/backbones/cctnets/utils/transformers.py:
// softmax(attention_pool(x), dim = tokens)^T x):  y[b, c] = sum_t p[b, t] x[b, t, c],  p = softmax_t(logit[b, .]).
// One workgroup per sample.  Backward: dp_t = gy . x[t], dlogit_t = p_t (dp_t - sum_u p_u dp_u), dx[t, c] = p_t gy[c].
constexpr int SP_TMAX = 1024;

__global__ __launch_bounds__(TPB) void seqpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ logit,
                                                          float* __restrict__ y, float* __restrict__ p, int T, int C) {
  __shared__ float pt[SP_TMAX];
  __shared__ float red[4];
  const long b = blockIdx.x;
  const int t = threadIdx.x;
  const float* lg = logit + b * T;
  float m = -INFINITY;
  for (int i = t; i < T; i += TPB) m = fmaxf(m, lg[i]);
  m = wave_max(m);
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int i = t; i < T; i += TPB) {
    const float e = __expf(lg[i] - m);
    pt[i] = e;
    s += e;
  }
  s = block_sum_256(s, red);
  const float inv = 1.f / s;
  for (int i = t; i < T; i += TPB) {
    const float v = pt[i] * inv;
    pt[i] = v;
    p[b * T + i] = v;
  }
  __syncthreads();
  const float* xb = x + b * (long)T * C;
  for (int c = t; c < C; c += TPB) {
    float a = 0.f;
    for (int i = 0; i < T; ++i) a = fmaf(pt[i], xb[(long)i * C + c], a);
    y[b * C + c] = a;
  }
}

__global__ __launch_bounds__(TPB) void seqpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ p,
                                                          const float* __restrict__ gy, float* __restrict__ dx,
                                                          float* __restrict__ dlogit, int T, int C) {
  __shared__ float dp[SP_TMAX];
  __shared__ float red[4];
  const long b = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* xb = x + b * (long)T * C;
  const float* g = gy + b * C;
  const float* pb = p + b * T;
  for (int i = wave; i < T; i += 4) {  // a wave per token: dp_i = gy . x[i]
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a = fmaf(g[c], xb[(long)i * C + c], a);
    a = wave_sum(a);
    if (lane == 0) dp[i] = a;
  }
  __syncthreads();
  float s = 0.f;
  for (int i = t; i < T; i += TPB) s += pb[i] * dp[i];
  s = block_sum_256(s, red);
  for (int i = t; i < T; i += TPB) dlogit[b * T + i] = pb[i] * (dp[i] - s);
  float* db = dx + b * (long)T * C;
  const long n = (long)T * C;
  for (long e = t; e < n; e += TPB) db[e] = pb[e / C] * g[e % C];
}

}  // namespace

// rows of one client per stage-1 block
BLADES_API int blades_bias_grad_nchunk(int R, int C) {
  int rpc = 65536 / (C > 0 ? C : 1);
  if (rpc < 64) rpc = 64;
  if (rpc > 4096) rpc = 4096;
  return cdiv(R, rpc);
}

// g: [Kc*R][C] (esize 4: fp32, 2: bf16).  partial: Kc * nchunk * C floats.
BLADES_API int blades_bias_grad(const void* g, int Kc, int R, int C, const float* coef, float* U, long ld, long off,
                                int accumulate, float* partial, int esize, hipStream_t st) {
  if (Kc <= 0 || R <= 0 || C <= 0) return 1;
  int rpc = 65536 / C;
  if (rpc < 64) rpc = 64;
  if (rpc > 4096) rpc = 4096;
  const int nch = cdiv(R, rpc);
  int CL = 1;
  while (CL < C && CL < TPB) CL <<= 1;  // channel lanes: a power of two <= 256
  const size_t lds = TPB * sizeof(float);
  if (esize == 4 && C % 4 == 0 && (uintptr_t)g % 16 == 0 && (uintptr_t)partial % 16 == 0) {
    int CL4 = 1;
    while (CL4 * 4 < C && CL4 < 64) CL4 <<= 1;  // lanes of four channels: at least four rows per block pass
    hipLaunchKernelGGL(bias_grad_partial4_kernel, dim3(nch, Kc), dim3(TPB), 4 * lds, st, (const float*)g, R, C, rpc, nch,
                       CL4, partial);
  } else if (esize == 4)
    hipLaunchKernelGGL((bias_grad_partial_kernel<float>), dim3(nch, Kc), dim3(TPB), lds, st, (const float*)g, R, C, rpc,
                       nch, CL, partial);
  else if (esize == 2)
    hipLaunchKernelGGL((bias_grad_partial_kernel<unsigned short>), dim3(nch, Kc), dim3(TPB), lds, st,
                       (const unsigned short*)g, R, C, rpc, nch, CL, partial);
  else
    return 2;
  hipLaunchKernelGGL(bias_grad_final_kernel, dim3(cdiv((long)Kc * C, TPB)), dim3(TPB), 0, st, partial, Kc, C, nch, coef, U,
                     ld, off, accumulate);
  return 0;
}

BLADES_API int blades_maxpool_forward(const void* x, void* y, unsigned char* idx, long N, int H, int W, int C, int Ho,
                                      int Wo, int kh, int kw, int sh, int sw, int ph, int pw, int relu, int esize,
                                      hipStream_t st) {
  if (kh * kw > 254 || kh < 1 || kw < 1 || sh < 1 || sw < 1 || N <= 0) return 1;
  const long total = N * Ho * Wo * C;
  if (esize == 4 && C % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)idx % 4 == 0)
    hipLaunchKernelGGL(maxpool_fwd4_kernel, dim3(grid_for(total / 4, 16384)), dim3(TPB), 0, st, (const float*)x, (float*)y, idx,
                       total / 4, H, W, C / 4, Ho, Wo, kh, kw, sh, sw, ph, pw, relu);
  else if (esize == 4)
    hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(grid_for(total, 16384)), dim3(TPB), 0, st, (const float*)x,
                       (float*)y, idx, total, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw, relu);
  else if (esize == 2)
    hipLaunchKernelGGL((maxpool_fwd_kernel<unsigned short>), dim3(grid_for(total, 16384)), dim3(TPB), 0, st,
                       (const unsigned short*)x, (unsigned short*)y, idx, total, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw,
                       relu);
  else
    return 2;
  return 0;
}

BLADES_API int blades_maxpool_backward(const void* gy, const unsigned char* idx, void* dx, long N, int H, int W, int C,
                                       int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw, int esize,
                                       hipStream_t st) {
  if (kh * kw > 254 || kh < 1 || kw < 1 || sh < 1 || sw < 1 || N <= 0) return 1;
  const long total = N * H * W * C;
  if (esize == 4 && C % 4 == 0 && (uintptr_t)gy % 16 == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)idx % 4 == 0)
    hipLaunchKernelGGL(maxpool_bwd4_kernel, dim3(grid_for(total / 4, 16384)), dim3(TPB), 0, st, (const float*)gy, idx,
                       (float*)dx, total / 4, H, W, C / 4, Ho, Wo, kh, kw, sh, sw, ph, pw);
  else if (esize == 4)
    hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(grid_for(total, 16384)), dim3(TPB), 0, st, (const float*)gy, idx,
                       (float*)dx, total, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw);
  else if (esize == 2)
    hipLaunchKernelGGL((maxpool_bwd_kernel<unsigned short>), dim3(grid_for(total, 16384)), dim3(TPB), 0, st,
                       (const unsigned short*)gy, idx, (unsigned short*)dx, total, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw);
  else
    return 2;
  return 0;
}

BLADES_API int blades_avgpool_forward(const void* x, void* y, long N, int P, int C, int esize, hipStream_t st) {
  if (N <= 0 || P <= 0 || C <= 0) return 1;
  const long NC = N * C;
  if (esize == 4)
    hipLaunchKernelGGL((avgpool_fwd_kernel<float>), dim3(cdiv(NC, TPB)), dim3(TPB), 0, st, (const float*)x, (float*)y, NC, P, C);
  else if (esize == 2)
    hipLaunchKernelGGL((avgpool_fwd_kernel<unsigned short>), dim3(cdiv(NC, TPB)), dim3(TPB), 0, st,
                       (const unsigned short*)x, (unsigned short*)y, NC, P, C);
  else
    return 2;
  return 0;
}

BLADES_API int blades_avgpool_backward(const void* gy, void* dx, long N, int P, int C, int esize, hipStream_t st) {
  if (N <= 0 || P <= 0 || C <= 0) return 1;
  const long total = N * P * C;
  if (esize == 4)
    hipLaunchKernelGGL((avgpool_bwd_kernel<float>), dim3(grid_for(total, 16384)), dim3(TPB), 0, st, (const float*)gy,
                       (float*)dx, total, P, C);
  else if (esize == 2)
    hipLaunchKernelGGL((avgpool_bwd_kernel<unsigned short>), dim3(grid_for(total, 16384)), dim3(TPB), 0, st,
                       (const unsigned short*)gy, (unsigned short*)dx, total, P, C);
  else
    return 2;
  return 0;
}

// mode 0: GELU, 1: ReLU.  backward: `saved` = the forward INPUT for GELU, the forward OUTPUT (or input) for ReLU.
BLADES_API int blades_act_forward(const void* x, void* y, long n, int mode, int esize, hipStream_t st) {
  if (n <= 0 || mode < 0 || mode > 1) return 1;
  const bool v4 = esize == 4 && n % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
  if (v4) {
    if (mode == 0) hipLaunchKernelGGL((act_fwd4_kernel<0>), dim3(grid_for(n / 4, 16384)), dim3(TPB), 0, st, (const float*)x, (float*)y, n / 4);
    else hipLaunchKernelGGL((act_fwd4_kernel<1>), dim3(grid_for(n / 4, 16384)), dim3(TPB), 0, st, (const float*)x, (float*)y, n / 4);
    return 0;
  }
  if (esize == 4) {
    if (mode == 0) hipLaunchKernelGGL((act_fwd_kernel<float, 0>), dim3(grid_for(n, 16384)), dim3(TPB), 0, st, (const float*)x, (float*)y, n);
    else hipLaunchKernelGGL((act_fwd_kernel<float, 1>), dim3(grid_for(n, 16384)), dim3(TPB), 0, st, (const float*)x, (float*)y, n);
  } else if (esize == 2) {
    if (mode == 0) hipLaunchKernelGGL((act_fwd_kernel<unsigned short, 0>), dim3(grid_for(n, 16384)), dim3(TPB), 0, st, (const unsigned short*)x, (unsigned short*)y, n);
    else hipLaunchKernelGGL((act_fwd_kernel<unsigned short, 1>), dim3(grid_for(n, 16384)), dim3(TPB), 0, st, (const unsigned short*)x, (unsigned short*)y, n);
  } else return 2;
  return 0;
}

BLADES_API int blades_act_backward(const void* saved, const void* gy, void* dx, long n, int mode, int esize, hipStream_t st) {
  if (n <= 0 || mode < 0 || mode > 1) return 1;
  const bool v4 = esize == 4 && n % 4 == 0 && ((uintptr_t)saved % 16 == 0) && ((uintptr_t)gy % 16 == 0) && ((uintptr_t)dx % 16 == 0);
  if (v4) {
    if (mode == 0) hipLaunchKernelGGL((act_bwd4_kernel<0>), dim3(grid_for(n / 4, 16384)), dim3(TPB), 0, st, (const float*)saved, (const float*)gy, (float*)dx, n / 4);
    else hipLaunchKernelGGL((act_bwd4_kernel<1>), dim3(grid_for(n / 4, 16384)), dim3(TPB), 0, st, (const float*)saved, (const float*)gy, (float*)dx, n / 4);
    return 0;
  }
  if (esize == 4) {
    if (mode == 0) hipLaunchKernelGGL((act_bwd_kernel<float, 0>), dim3(grid_for(n, 16384)), dim3(TPB), 0, st, (const float*)saved, (const float*)gy, (float*)dx, n);
    else hipLaunchKernelGGL((act_bwd_kernel<float, 1>), dim3(grid_for(n, 16384)), dim3(TPB), 0, st, (const float*)saved, (const float*)gy, (float*)dx, n);
  } else if (esize == 2) {
    if (mode == 0) hipLaunchKernelGGL((act_bwd_kernel<unsigned short, 0>), dim3(grid_for(n, 16384)), dim3(TPB), 0, st, (const unsigned short*)saved, (const unsigned short*)gy, (unsigned short*)dx, n);
    else hipLaunchKernelGGL((act_bwd_kernel<unsigned short, 1>), dim3(grid_for(n, 16384)), dim3(TPB), 0, st, (const unsigned short*)saved, (const unsigned short*)gy, (unsigned short*)dx, n);
  } else return 2;
  return 0;
}

// logits [Kc][B] rows of C fp32 (row stride ld), labels [Kc*B] int64; loss [Kc]; dz [Kc][B][C] or null; gl [Kc] or null.
BLADES_API int blades_ce_clamp(const float* z, long ld, const long* labels, int Kc, int B, int C, float lo, float hi,
                               const float* gl, float* loss, float* dz, hipStream_t st) {
  if (Kc <= 0 || B <= 0 || C <= 0 || ld < C) return 1;
  hipLaunchKernelGGL(ce_clamp_kernel, dim3(Kc), dim3(TPB), 0, st, z, ld, labels, B, C, lo, hi, gl, loss, dz);
  return 0;
}

// x [B, T, C], logit [B, T] -> y [B, C], p [B, T] (the softmax weights, kept for the backward pass).  fp32.
BLADES_API int blades_seqpool_forward(const float* x, const float* logit, float* y, float* p, long B, int T, int C,
                                      hipStream_t st) {
  if (B <= 0 || B > 0x7fffffffL || T < 1 || T > SP_TMAX || C < 1) return 1;
  hipLaunchKernelGGL(seqpool_fwd_kernel, dim3((unsigned)B), dim3(TPB), 0, st, x, logit, y, p, T, C);
  return 0;
}

BLADES_API int blades_seqpool_backward(const float* x, const float* p, const float* gy, float* dx, float* dlogit, long B, int T,
                                       int C, hipStream_t st) {
  if (B <= 0 || B > 0x7fffffffL || T < 1 || T > SP_TMAX || C < 1) return 1;
  hipLaunchKernelGGL(seqpool_bwd_kernel, dim3((unsigned)B), dim3(TPB), 0, st, x, p, gy, dx, dlogit, T, C);
  return 0;
}
