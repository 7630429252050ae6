// Fused optimizer steps on flat fp32 parameter storage.
//
// server_step:      g = -agg; buf = first ? g : mu*buf + g; theta -= lr*buf      (torch.optim.SGD semantics,
//                   reference fllib/algorithms/server.py:109-130) on the rank's D-slice.
// client_sgd_step:  the same update for Kc client replicas at once, W[Kc, D], with a per-row gradient
//                   coefficient (sign flip x clip) folded in — one launch instead of Kc optimizers.
#include "common.h"

namespace {
constexpr int TPB = 256;

__global__ __launch_bounds__(TPB) void server_step_kernel(float* __restrict__ theta, float* __restrict__ buf,
                                                          const float* __restrict__ agg, long n, float lr, float mu,
                                                          int first) {
  const long stride = (long)gridDim.x * TPB;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
    const float g = -agg[i];
    float b = g;
    if (mu != 0.f) {
      if (!first) b = mu * buf[i] + g;
      buf[i] = b;
    }
    theta[i] -= lr * b;
  }
}

// Non-finite guard of the server step (algorithms/server.py): `bad` (> 0: the aggregate holds an inf / nan somewhere;
// summed over the ranks by the caller) turns the step into one with a zero aggregate and is counted in `skipped`.
// As ATen calls the guard was eight launches (isfinite, all, not, float, zeros_like, where, clamp, add): 0.1 ms of
// launch latency on the 0.65 ms round of the 60-client MLP.
__global__ __launch_bounds__(TPB) void nonfinite_flag_kernel(const float* __restrict__ x, long n, float* __restrict__ flag) {
  const long stride = (long)gridDim.x * TPB;
  bool bad = false;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
    const float v = x[i];
    bad = bad || !(fabsf(v) <= 3.402823466e+38f);  // inf and nan fail the comparison
  }
  if (bad) *flag = 1.0f;  // (every writer stores the same value)
}

__global__ __launch_bounds__(TPB) void server_step_guarded_kernel(float* __restrict__ theta, float* __restrict__ buf,
                                                                  const float* __restrict__ agg, long n, float lr, float mu,
                                                                  int first, const float* __restrict__ bad,
                                                                  float* __restrict__ skipped) {
  const bool skip = *bad > 0.f;
  const long stride = (long)gridDim.x * TPB;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
    const float g = skip ? 0.f : -agg[i];
    float b = g;
    if (mu != 0.f) {
      if (!first) b = mu * buf[i] + g;
      buf[i] = b;
    }
    theta[i] -= lr * b;
  }
  if (skip && skipped != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *skipped += 1.0f;
}

__global__ __launch_bounds__(TPB) void client_sgd_kernel(float* __restrict__ W, const float* __restrict__ G,
                                                         float* __restrict__ M, long D, long ldw, long ldg, long ldm,
                                                         float lr, float mu, int first,
                                                         const float* __restrict__ gscale) {
  const int row = blockIdx.y;
  const float gs = gscale ? gscale[row] : 1.f;
  float* w = W + (long)row * ldw;
  const float* g = G + (long)row * ldg;
  float* m = M ? M + (long)row * ldm : nullptr;
  const long stride = (long)gridDim.x * TPB;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < D; i += stride) {
    float b = g[i] * gs;
    if (mu != 0.f) {
      if (!first) b = mu * m[i] + b;
      m[i] = b;
    }
    w[i] -= lr * b;
  }
}
}  // namespace

BLADES_API void blades_server_step(float* theta, float* buf, const float* agg, long n, float lr, float mu, int first,
                                   hipStream_t st) {
  if (n <= 0) return;
  int g = cdiv(n, TPB);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(server_step_kernel, dim3(g), dim3(TPB), 0, st, theta, buf, agg, n, lr, mu, first);
}

// flag: one float, set to 1 when x holds a non-finite value (zeroed here first)
BLADES_API void blades_nonfinite_flag(const float* x, long n, float* flag, hipStream_t st) {
  (void)hipMemsetAsync(flag, 0, sizeof(float), st);
  if (n <= 0) return;
  int g = cdiv(n, TPB * 4);
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(g), dim3(TPB), 0, st, x, n, flag);
}

// the server step with `bad` (one float on the device) deciding on the device whether the aggregate counts
BLADES_API void blades_server_step_guarded(float* theta, float* buf, const float* agg, long n, float lr, float mu, int first,
                                           const float* bad, float* skipped, hipStream_t st) {
  if (n <= 0) return;
  int g = cdiv(n, TPB);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(server_step_guarded_kernel, dim3(g), dim3(TPB), 0, st, theta, buf, agg, n, lr, mu, first, bad, skipped);
}

BLADES_API void blades_client_sgd_step(float* W, const float* G, float* M, int Kc, long D, long ldw, long ldg,
                                       long ldm, float lr, float mu, int first, const float* gscale,
                                       hipStream_t st) {
  if (D <= 0 || Kc <= 0) return;
  int gx = cdiv(D, TPB);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(client_sgd_kernel, dim3(gx, Kc), dim3(TPB), 0, st, W, G, M, D, ldw, ldg, ldm, lr, mu, first,
                     gscale);
}
