// Single-workgroup solvers on the K x K Gram matrix for small client counts (K <= 128: the reference's own experiments
// run 60 clients, blades/tuned_examples/*.yaml).  At that size the iterations of the Gram-space aggregators are a few
// thousand multiply-adds each, and a Python loop of library calls spends 10 us of launch overhead on every one of them
// (GeoMed: ~20 tiny kernels x up to 100 Weiszfeld iterations = up to 10 ms on a 34 ms ResNet-10 round; DnC: 0.7 ms per
// power iteration call, five calls per round).  Here the whole iteration runs in one launch with G resident in LDS.
//
//   power_top   top eigenvector of a symmetric PSD G by `steps` normalised power steps from the reference's start
//               vector (ops/reference.py::power_iteration_top does 6 squarings + 8 steps = the direction G^72 v0)
//   weiszfeld   geometric median in coefficient space (aggregators/simple.py::GeoMed._aggregate_gram; reference
//               blades/aggregators/aggregators.py:51-110): alpha <- normalise(w0 / max(dist, eps)),
//               dist_i = sqrt(max(G_ii - 2 (G alpha)_i + alpha^T G alpha, 0)), until |obj - obj'| <= ftol * obj'
//               or maxiter; fp64 arithmetic on the fp32 Gram entries, like the torch form
//
//   cclip       centred clipping in coefficient space (aggregators/centeredclipping.py; reference
//               blades/aggregators/centeredclipping.py): every iterate is m_t = c0 m_0 + sum_j c_j x_j, so with
//               G = X X^T, p = X m_0 and |m_0|^2 the distances ||x_i - m_t|| and the update
//               m <- (1 - S/K) m + (1/K) sum_i s_i x_i, s_i = min(1, tau / ||x_i - m||), need no pass over X
//
// All loops have a fixed upper bound; every reduction has a fixed order (bitwise reproducible).
#include "common.h"

namespace {

constexpr int KMAX = 128;
constexpr int TPB = 256;  // four waves: block_sum adds four wave sums

// fixed-order sum over the 256 threads: butterfly inside each wave (shuffles, no barrier), then the four wave sums in
// wave order through LDS — 2 barriers instead of the 10 of an LDS tree (a Weiszfeld iteration has three of these and a
// solver runs up to a hundred iterations: the barriers were most of its time)
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const T r = ((red[0] + red[1]) + red[2]) + red[3];
  __syncthreads();  // (red is reused by the next reduction)
  return r;
}

// P = 2, 4 or 8 neighbouring lanes share a row of the matrix-vector product (lane `sub` takes columns sub, sub + P, ...;
// butterfly over the P lanes): a thread per row was a chain of K dependent multiply-adds per step, 80 us for the 72
// steps of DnC at 60 clients
__global__ __launch_bounds__(TPB) void power_top_kernel(const float* __restrict__ G, int K, int steps, float* __restrict__ out) {
  extern __shared__ float lds[];
  float* g = lds;                    // [K][K + 1]
  float* v = g + K * (K + 1);        // [K]
  float* red = v + KMAX;             // [TPB]
  const int t = threadIdx.x;
  G += (size_t)blockIdx.x * K * K;  // one matrix per workgroup
  out += (size_t)blockIdx.x * K;
  for (int i = t; i < K * K; i += TPB) g[(i / K) * (K + 1) + (i % K)] = G[i];
  if (t < K) v[t] = 1.0f / sqrtf((float)K) + (float)t * (1e-3f / (float)K);
  const int P = K <= 32 ? 8 : K <= 64 ? 4 : 2;  // P * K <= 256
  const int row = t / P, sub = t - row * P;
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    float w = 0.f;
    if (row < K) {
      const float* gr = g + row * (K + 1);
      for (int j = sub; j < K; j += P) w += gr[j] * v[j];
    }
    for (int o = 1; o < P; o <<= 1) w += __shfl_xor(w, o);  // (the P lanes of a row are neighbours inside one wave)
    const float n2 = block_sum<float>(row < K && sub == 0 ? w * w : 0.f, red);  // (barriers inside: every thread has read v)
    const float nrm = fmaxf(sqrtf(n2), 1e-30f);
    if (row < K && sub == 0) v[row] = w / nrm;
    __syncthreads();
  }
  if (t < K) out[t] = v[t];
}

__global__ __launch_bounds__(TPB) void weiszfeld_kernel(const float* __restrict__ G, const float* __restrict__ w0, int K,
                                                        int maxiter, double eps, double ftol, float* __restrict__ alpha_out,
                                                        int* __restrict__ iters_out) {
  extern __shared__ float lds[];
  float* g = lds;                                                   // [K][K + 1]
  double* a = reinterpret_cast<double*>(g + ((K * (K + 1) + 1) & ~1));  // [KMAX] coefficients
  double* red = a + KMAX;                                           // [TPB]
  const int t = threadIdx.x;
  for (int i = t; i < K * K; i += TPB) g[(i / K) * (K + 1) + (i % K)] = G[i];
  const double wt = t < K ? (double)w0[t] : 0.0;
  __syncthreads();
  const double wsum = block_sum<double>(wt, red);
  const double diag = t < K ? (double)g[t * (K + 1) + t] : 0.0;
  // dist of the coefficient vector in `a` (every thread gets its own entry; obj = sum_i dist_i w0_i / sum w0)
  auto dist_of = [&](double& obj) {
    double ga = 0.0;
    if (t < K) {
      const float* row = g + t * (K + 1);
      for (int j = 0; j < K; ++j) ga += (double)row[j] * a[j];
    }
    const double q = block_sum<double>(t < K ? a[t] * ga : 0.0, red);
    const double d2 = diag - 2.0 * ga + q;
    const double d = t < K ? sqrt(d2 > 0.0 ? d2 : 0.0) : 0.0;
    obj = block_sum<double>(d * wt, red) / wsum;
    return d;
  };
  if (t < K) a[t] = wt / wsum;
  __syncthreads();
  double obj;
  double dist = dist_of(obj);
  int it = 0;
  for (; it < maxiter; ++it) {
    const double cw = t < K ? wt / (dist > eps ? dist : eps) : 0.0;
    const double cs = block_sum<double>(cw, red);
    if (t < K) a[t] = cw / cs;
    __syncthreads();
    double cobj;
    dist = dist_of(cobj);
    const bool conv = fabs(obj - cobj) <= ftol * cobj;  // the same value in every thread
    obj = cobj;
    if (conv) { ++it; break; }
  }
  if (t < K) alpha_out[t] = (float)a[t];
  if (t == 0) *iters_out = it;
}

// Bisection of the min-max attack (adversaries/optimized.py::MinMaxAdversary; reference
// blades/adversaries/minmax_adversary.py:37-63): the largest gamma in [lo, hi] (halving until the interval is <= tol)
// with sqrt(max_i max(a_i - 2 gamma b_i + gamma^2 c, 0)) < threshold.  Same fp32 operations, in the same order, as
// the tensor form it replaces (ten tiny launches per halving step).  Any n; one workgroup.
__global__ __launch_bounds__(TPB) void minmax_bisect_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                            const float* __restrict__ c_ptr,
                                                            const float* __restrict__ thr_ptr, float lo, float hi,
                                                            float tol, float* __restrict__ out) {
  __shared__ float red[TPB];
  const int t = threadIdx.x;
  const float c = *c_ptr, thr = *thr_ptr;
  float low = lo, high = hi, mid = __fmul_rn(__fadd_rn(low, high), 0.5f);
  float lo_f = lo, hi_f = hi;
  while (fabsf(hi_f - lo_f) > tol) {
    mid = __fmul_rn(__fadd_rn(low, high), 0.5f);
    const float two_mid = __fmul_rn(2.f, mid), mid2c = __fmul_rn(__fmul_rn(mid, mid), c);
    float m = 0.f;  // = clamp_min(., 0) of the maximum
    for (int i = t; i < n; i += TPB) m = fmaxf(m, __fadd_rn(__fadd_rn(a[i], -__fmul_rn(two_mid, b[i])), mid2c));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    const bool ok = sqrtf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))) < thr;
    __syncthreads();
    if (ok) low = mid; else high = mid;
    hi_f = lo_f + (hi_f - lo_f) * 0.5f;
  }
  if (t == 0) *out = mid;
}

// out[0] = c0, out[1 + j] = c_j (fp64)
__global__ __launch_bounds__(TPB) void cclip_kernel(const float* __restrict__ G, const float* __restrict__ pvec,
                                                    const float* __restrict__ mm_ptr, int K, int n_iter, double tau,
                                                    double* __restrict__ out) {
  extern __shared__ float lds[];
  float* g = lds;
  double* c = reinterpret_cast<double*>(g + ((K * (K + 1) + 1) & ~1));  // [KMAX]
  double* red = c + KMAX;                                               // [TPB]
  const int t = threadIdx.x;
  for (int i = t; i < K * K; i += TPB) g[(i / K) * (K + 1) + (i % K)] = G[i];
  if (t < K) c[t] = 0.0;
  __syncthreads();
  const double diag = t < K ? (double)g[t * (K + 1) + t] : 0.0;
  const double p = t < K ? (double)pvec[t] : 0.0;
  const double mm = (double)*mm_ptr;
  double c0 = 1.0;
  for (int it = 0; it < n_iter; ++it) {
    double gc = 0.0;
    if (t < K) {
      const float* row = g + t * (K + 1);
      for (int j = 0; j < K; ++j) gc += (double)row[j] * c[j];
    }
    const double cp = block_sum<double>(t < K ? c[t] * p : 0.0, red);
    const double cgc = block_sum<double>(t < K ? c[t] * gc : 0.0, red);
    const double m2 = c0 * c0 * mm + 2.0 * c0 * cp + cgc;
    const double d2 = diag - 2.0 * (c0 * p + gc) + m2;
    // tau / 0 = inf -> 1, like the tensor form
    const double sc = t < K ? (d2 > 0.0 ? fmin(1.0, tau / sqrt(d2)) : 1.0) : 0.0;
    const double S = block_sum<double>(sc, red);
    const double keep = 1.0 - S / (double)K;
    if (t < K) c[t] = c[t] * keep + sc / (double)K;
    c0 *= keep;
    __syncthreads();
  }
  if (t == 0) out[0] = c0;
  if (t < K) out[1 + t] = c[t];
}

}  // namespace

// G: [K, K]; p: [K] = X m_0; mm: one float = |m_0|^2 (device); out: 1 + K doubles (c0, c).
BLADES_API int blades_cclip(const float* G, const float* p, const float* mm, int K, int n_iter, double tau, double* out,
                            hipStream_t st) {
  if (K < 1 || K > KMAX || n_iter < 0 || n_iter > 100000 || !(tau > 0.0)) return 1;
  const size_t gwords = ((size_t)K * (K + 1) + 1) & ~(size_t)1;
  const size_t lds = gwords * sizeof(float) + (KMAX + TPB) * sizeof(double);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cclip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (KMAX * (KMAX + 1) + 2) * (int)sizeof(float) + (KMAX + TPB) * (int)sizeof(double));
    attr = true;
  }
  hipLaunchKernelGGL(cclip_kernel, dim3(1), dim3(TPB), lds, st, G, p, mm, K, n_iter, tau, out);
  return 0;
}

// a, b: [n] fp32; c, thr: one fp32 each (device); out: one fp32 (the last midpoint tried, as the tensor loop returns it)
BLADES_API int blades_minmax_bisect(const float* a, const float* b, int n, const float* c, const float* thr, float lo,
                                    float hi, float tol, float* out, hipStream_t st) {
  if (n < 1 || !(hi > lo) || !(tol > 0.f) || (hi - lo) / tol > 1e9f) return 1;
  hipLaunchKernelGGL(minmax_bisect_kernel, dim3(1), dim3(TPB), 0, st, a, b, n, c, thr, lo, hi, tol, out);
  return 0;
}

BLADES_API int blades_smallk_max() { return KMAX; }

// G: nb matrices [K, K] fp32 contiguous, symmetric.  out: [nb, K].  returns 0 ok.
BLADES_API int blades_power_top(const float* G, int K, int nb, int steps, float* out, hipStream_t st) {
  if (K < 1 || K > KMAX || steps < 1 || steps > 100000 || nb < 1 || nb > 65535) return 1;
  const size_t lds = ((size_t)K * (K + 1) + KMAX + TPB) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(power_top_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (KMAX * (KMAX + 1) + KMAX + TPB) * (int)sizeof(float));
    attr = true;
  }
  hipLaunchKernelGGL(power_top_kernel, dim3(nb), dim3(TPB), lds, st, G, K, steps, out);
  return 0;
}

// alpha_out: [K] fp32 coefficients of the geometric median; iters_out: one int (Weiszfeld iterations run).
BLADES_API int blades_weiszfeld(const float* G, const float* w0, int K, int maxiter, double eps, double ftol,
                                float* alpha_out, int* iters_out, hipStream_t st) {
  if (K < 1 || K > KMAX || maxiter < 0 || maxiter > 1000000) return 1;
  const size_t gwords = ((size_t)K * (K + 1) + 1) & ~(size_t)1;
  const size_t lds = gwords * sizeof(float) + (KMAX + TPB) * sizeof(double);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(weiszfeld_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (KMAX * (KMAX + 1) + 2) * (int)sizeof(float) + (KMAX + TPB) * (int)sizeof(double));
    attr = true;
  }
  hipLaunchKernelGGL(weiszfeld_kernel, dim3(1), dim3(TPB), lds, st, G, w0, K, maxiter, eps, ftol, alpha_out, iters_out);
  return 0;
}
