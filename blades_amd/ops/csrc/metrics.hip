// Fused evaluation metrics: summed cross-entropy and top-k hit counts of a logits batch in one pass
// (replaces torchmetrics Accuracy(top_k) + the CustomCrossEntropy metric of the reference,
// fllib/tasks/task.py:104-121).
//
// One lane per sample: a running max / sum-exp gives the log-sum-exp, and the rank of the target class is
// the number of classes that beat it (ties broken towards the lower class index, i.e. the order a stable
// descending sort would give).  hit@k = rank < k.  Block sums are written to a partial table and added up in
// block order by a second, single-block kernel, so the result does not depend on scheduling.
#include "common.h"

namespace {

constexpr int MAXK = 8;

struct TopK { int n; int k[MAXK]; };

__global__ __launch_bounds__(256) void ce_topk_partial_kernel(const float* __restrict__ logits, long ld,
                                                              const long* __restrict__ labels, long N, int C,
                                                              TopK tk, double* __restrict__ part) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  double v[1 + MAXK];
#pragma unroll
  for (int j = 0; j <= MAXK; ++j) v[j] = 0.0;
  if (i < N) {
    const float* row = logits + i * ld;
    long y = labels[i];
    if (y < 0 || y >= C) y = 0;  // host validates; stay in bounds regardless
    const float ly = row[y];
    float m = -INFINITY, s = 0.f;
    int rank = 0;
    for (int c = 0; c < C; ++c) {
      const float x = row[c];
      rank += (x > ly) || (x == ly && c < y);
      if (x > m) { s = s * __expf(m - x) + 1.f; m = x; }
      else s += __expf(x - m);
    }
    v[0] = (double)(m + __logf(s) - ly);
#pragma unroll
    for (int j = 0; j < MAXK; ++j)
      if (j < tk.n) v[1 + j] = rank < tk.k[j] ? 1.0 : 0.0;
  }
  __shared__ double sh[4][1 + MAXK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j <= MAXK; ++j) {
    double t = v[j];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) sh[wave][j] = t;
  }
  __syncthreads();
  if (threadIdx.x <= MAXK)
    part[(size_t)blockIdx.x * (1 + MAXK) + threadIdx.x] =
        (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

__global__ __launch_bounds__(64) void ce_topk_final_kernel(const double* __restrict__ part, int nblk, int nout,
                                                           int accumulate, double* __restrict__ out) {
  const int j = threadIdx.x;
  if (j >= nout) return;
  double t = 0.0;
  for (int b = 0; b < nblk; ++b) t += part[(size_t)b * (1 + MAXK) + j];
  out[j] = accumulate ? out[j] + t : t;
}

}  // namespace

// out[0] = sum CE, out[1 + j] = #samples whose target is within the top ks[j];  part: >= cdiv(N,256)*9 doubles.
BLADES_API int blades_ce_topk(const float* logits, long ld, const long* labels, long N, int C, const int* ks, int nk,
                              double* part, int accumulate, double* out, hipStream_t st) {
  if (nk < 0 || nk > MAXK || C < 1) return 1;
  if (N <= 0) return 0;
  TopK tk;
  tk.n = nk;
  for (int j = 0; j < MAXK; ++j) tk.k[j] = j < nk ? ks[j] : 0;
  const int nblk = (int)cdiv(N, 256L);
  hipLaunchKernelGGL(ce_topk_partial_kernel, dim3(nblk), dim3(256), 0, st, logits, ld, labels, N, C, tk, part);
  hipLaunchKernelGGL(ce_topk_final_kernel, dim3(1), dim3(64), 0, st, part, nblk, 1 + nk, accumulate, out);
  return 0;
}
