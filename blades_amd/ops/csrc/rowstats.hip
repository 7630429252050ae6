// Per-row reductions over the columns of U[K, Dl]: squared norm, sign counts,
// squared distance to a vector, dot with a vector.
//
// Two deterministic stages (no float atomics, so results are bitwise
// reproducible — SURVEY §5.2): grid (chunks, K) blocks reduce CHUNK columns of
// one row into partial[row][chunk]; a second tiny kernel sums the partials of a
// row in fixed order.  Stage 1 streams U once with float4 loads (HBM-bound).
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int CHUNK = TPB * 4 * 8;  // 8192 columns per block

// kind 0: sum x^2 (+ optional counts), 1: sum (x - m)^2, 2: sum x * m
template <int KIND, bool COUNTS>
__global__ __launch_bounds__(TPB) void row_partial_kernel(const float* __restrict__ U, long ld, long Dl,
                                                          long valid_cols, const float* __restrict__ m,
                                                          float* __restrict__ part, int* __restrict__ cpart,
                                                          int nchunks) {
  __shared__ float red[4];
  __shared__ int redi[12];
  // rows vary fastest over the grid: the workgroups in flight share a few column chunks, so the slice of m they
  // read (kinds 1, 2) stays in the XCDs' L2 — chunk-fastest order re-read m from the fabric for every row (K = 1000:
  // as many bytes again as the matrix, sqdist_to at 3.4 TB/s against the 5 TB/s of the m-free kinds)
  const int row = blockIdx.x, chunk = blockIdx.y;
  const float* p = U + (long)row * ld;
  const long c_begin = (long)chunk * CHUNK;
  const long lim = KIND == 0 ? (valid_cols < Dl ? valid_cols : Dl) : Dl;
  float acc = 0.f;
  int pos = 0, neg = 0, zer = 0;
  const bool al = (ld & 3) == 0;
#pragma unroll 4
  for (int it = 0; it < 8; ++it) {
    const long c = c_begin + ((long)it * TPB + threadIdx.x) * 4;
    if (c >= lim) break;
    float x[4], mv[4] = {0, 0, 0, 0};
    int nc = 4;
    if (c + 4 <= lim && al) {
      const float4 t = *reinterpret_cast<const float4*>(p + c);
      x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
      if (KIND != 0) {
        const float4 q = *reinterpret_cast<const float4*>(m + c);
        mv[0] = q.x; mv[1] = q.y; mv[2] = q.z; mv[3] = q.w;
      }
    } else {
      nc = (int)((lim - c) < 4 ? (lim - c) : 4);
      for (int j = 0; j < 4; ++j) {
        x[j] = j < nc ? p[c + j] : 0.f;
        if (KIND != 0) mv[j] = j < nc ? m[c + j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KIND == 0) acc += x[j] * x[j];
      if (KIND == 1) { const float d = x[j] - mv[j]; acc += d * d; }
      if (KIND == 2) acc += x[j] * mv[j];
      if (COUNTS && j < nc) { pos += x[j] > 0.f; neg += x[j] < 0.f; zer += x[j] == 0.f; }
    }
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[(long)row * nchunks + chunk] = s;
  if (COUNTS) {
    pos = wave_sum_i(pos); neg = wave_sum_i(neg); zer = wave_sum_i(zer);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { redi[w * 3] = pos; redi[w * 3 + 1] = neg; redi[w * 3 + 2] = zer; }
    __syncthreads();
    if (threadIdx.x < 3) {
      const int t = threadIdx.x;
      cpart[((long)row * nchunks + chunk) * 3 + t] = redi[t] + redi[3 + t] + redi[6 + t] + redi[9 + t];
    }
  }
}

template <bool COUNTS>
__global__ __launch_bounds__(64) void row_final_kernel(const float* __restrict__ part, const int* __restrict__ cpart,
                                                       int nchunks, float* __restrict__ out,
                                                       long long* __restrict__ counts) {
  const int row = blockIdx.x, lane = threadIdx.x;
  // fixed summation order: lane-strided partial sums, then the wave butterfly
  double acc = 0.0;
  long long c0 = 0, c1 = 0, c2 = 0;
  for (int i = lane; i < nchunks; i += 64) {
    acc += (double)part[(long)row * nchunks + i];
    if (COUNTS) {
      const int* q = cpart + ((long)row * nchunks + i) * 3;
      c0 += q[0]; c1 += q[1]; c2 += q[2];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    acc += __shfl_xor(acc, o, 64);
    if (COUNTS) { c0 += __shfl_xor(c0, o, 64); c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
  }
  if (lane == 0) {
    out[row] = (float)acc;
    if (COUNTS) { counts[row * 3] = c0; counts[row * 3 + 1] = c1; counts[row * 3 + 2] = c2; }
  }
}

}  // namespace

BLADES_API int blades_row_chunks(long Dl) { return cdiv(Dl, CHUNK) > 0 ? cdiv(Dl, CHUNK) : 1; }

// kind 0 sqnorm (+counts if counts != null), 1 sqdist to m, 2 dot with m.
// part: float [K * nchunks] workspace, cpart: int [K * nchunks * 3] workspace (kind 0 with counts).
BLADES_API void blades_row_reduce(const float* U, long ld, int K, long Dl, long valid_cols, const float* m, int kind,
                                  float* part, int* cpart, float* out, long long* counts, hipStream_t st) {
  if (K <= 0) return;
  const int nch = blades_row_chunks(Dl);
  if (nch > 65535) return;  // (checked by the caller)
  dim3 g(K, nch), b(TPB);  // (nch <= 65535: Dl < 2^29 columns per rank)
  if (kind == 0 && counts != nullptr) {
    hipLaunchKernelGGL((row_partial_kernel<0, true>), g, b, 0, st, U, ld, Dl, valid_cols, m, part, cpart, nch);
    hipLaunchKernelGGL((row_final_kernel<true>), dim3(K), dim3(64), 0, st, part, cpart, nch, out, counts);
    return;
  }
  if (kind == 0) hipLaunchKernelGGL((row_partial_kernel<0, false>), g, b, 0, st, U, ld, Dl, valid_cols, m, part, cpart, nch);
  else if (kind == 1) hipLaunchKernelGGL((row_partial_kernel<1, false>), g, b, 0, st, U, ld, Dl, valid_cols, m, part, cpart, nch);
  else hipLaunchKernelGGL((row_partial_kernel<2, false>), g, b, 0, st, U, ld, Dl, valid_cols, m, part, cpart, nch);
  hipLaunchKernelGGL((row_final_kernel<false>), dim3(K), dim3(64), 0, st, part, cpart, nch, out, counts);
}
