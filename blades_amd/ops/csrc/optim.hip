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

BLADES_API void blades_client_sgd_step(float* W, const float* G, float* M, int Kc, long D, long ldw, long ldg,
                                       long ldm, float lr, float mu, int first, const float* gscale,
                                       hipStream_t st) {
  if (D <= 0 || Kc <= 0) return;
  int gx = cdiv(D, TPB);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(client_sgd_kernel, dim3(gx, Kc), dim3(TPB), 0, st, W, G, M, D, ldw, ldg, ldm, lr, mu, first,
                     gscale);
}
