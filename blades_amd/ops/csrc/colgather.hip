// Column sub-sample of the update matrix for DnC (reference blades/aggregators/aggregators.py:113-151: np.random.choice of
// sub_dim coordinates, centred sub-matrix, top singular vector), all iterations in one launch:
//
//   out[it, k, j] = U[k, idx[it, j]] - mean_k' U[k', idx[it, j]]
//
// replaces index_select + mean + subtract (three passes and two temporaries of the sub-matrix): K strided reads of a
// selected column for the mean (fixed order: bitwise reproducible), K more (L2 hits) for the centred values, written
// with j fastest so that the rows of `out` are contiguous for the Gram kernel.
#include "common.h"

namespace {

// workgroup = 32 selected columns x 8 row lanes (lane l owns rows l, l + 8, ...): a thread per column alone left K = 1000
// with 40 workgroups of 2000 dependent-latency loads each (17.6 ms for DnC against 13.3 with index_select)
__global__ __launch_bounds__(256) void gather_center_kernel(const float* __restrict__ U, long ld, int K, long Dl,
                                                            const long* __restrict__ idx, long n, long total,
                                                            float* __restrict__ out) {
  __shared__ float part[8][33];
  const int e = threadIdx.x & 31, l = threadIdx.x >> 5;
  const long t = (long)blockIdx.x * 32 + e;
  const bool live = t < total;
  const long it = live ? t / n : 0, j = live ? t - it * n : 0;
  const long col = live ? idx[t] : -1;
  const bool ok = col >= 0 && col < Dl;  // never read outside the matrix: a zero column
  const float* src = U + (ok ? col : 0);
  // K <= 128: a lane's (up to 16) values stay in registers between the two passes — a selected column is a 4-byte read
  // out of a separate line per row, the second round of those reads was half of the kernel's time at 60 clients
  const bool small = K <= 128;  // (block-uniform)
  float v[16];
  float s = 0.f;
  if (small) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = l + 8 * i;
      v[i] = ok && k < K ? src[(size_t)k * ld] : 0.f;
      s += v[i];  // (rows l, l + 8, ... in order, zeros beyond K: the sum of the loop below)
    }
  } else if (ok) {
    for (int k = l; k < K; k += 8) s += src[(size_t)k * ld];
  }
  part[l][e] = s;
  __syncthreads();
  float tot = part[0][e];
#pragma unroll
  for (int q = 1; q < 8; ++q) tot += part[q][e];  // lane order: bitwise reproducible
  const float mean = tot / (float)K;
  if (!live) return;
  float* dst = out + (size_t)it * K * n + j;
  if (small) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = l + 8 * i;
      if (k < K) dst[(size_t)k * n] = ok ? v[i] - mean : 0.f;
    }
  } else {
    for (int k = l; k < K; k += 8) dst[(size_t)k * n] = ok ? src[(size_t)k * ld] - mean : 0.f;
  }
}

// ---- column ids without replacement: a keyed bijection of [0, D) ------------------------------------------------------
// DnC draws n distinct coordinates per iteration (reference: np.random.choice(d, sub_dim, replace=False)).  A draw
// without replacement on the host is O(n) hash-set work plus an upload per iteration; rejection sampling on the GPU needs
// a sort to find duplicates (0.38 ms for five sets of 10 000 — more than the rest of DnC at 60 clients).  Here set `it`
// is the image of 0 .. n-1 under a pseudo-random permutation of [0, D): a balanced Feistel network on b = 2 h bits
// (2^b >= D > 2^(b-2)), eight rounds of a 32-bit mixing function keyed by (seed, it, round), and cycle walking
// (re-apply until the value falls below D; < 4 applications expected).  Every id is computed independently, the ids of a
// set are distinct because the map is a bijection, and ops/reference.py::sample_distinct computes the same numbers on
// the CPU (tests compare them bit for bit; every rank and both devices see the same sets).
constexpr int FEISTEL_ROUNDS = 8;

__host__ __device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x *= 0xC2B2AE35u;
  return x ^ (x >> 16);
}

__global__ __launch_bounds__(256) void sample_distinct_kernel(long D, long n, long total, int h, unsigned long long seed,
                                                              long* __restrict__ out) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long it = t / n, j = t - it * n;
  unsigned key[FEISTEL_ROUNDS];
#pragma unroll
  for (int r = 0; r < FEISTEL_ROUNDS; ++r)
    key[r] = (unsigned)splitmix64(seed + 0x632BE59BD9B4E019ull * (unsigned long long)(it * FEISTEL_ROUNDS + r + 1));
  const unsigned mask = (1u << h) - 1u;
  unsigned long long x = (unsigned long long)j;
  for (int walk = 0; walk < 512; ++walk) {  // (every walk ends: P(512 misses) < 0.75^512)
    unsigned L = (unsigned)(x >> h), R = (unsigned)x & mask;
#pragma unroll
    for (int r = 0; r < FEISTEL_ROUNDS; ++r) {
      const unsigned nr = L ^ (mix32(R + key[r]) & mask);
      L = R; R = nr;
    }
    x = ((unsigned long long)L << h) | R;
    if (x < (unsigned long long)D) break;
  }
  out[t] = (long)(x < (unsigned long long)D ? x : x % (unsigned long long)D);
}

}  // namespace

// out[it, j] = the j-th id of set `it`: [iters, n] int64, n <= D <= 2^40, ids of a set distinct.  returns 0 ok.
BLADES_API int blades_sample_distinct(long D, long n, long iters, unsigned long long seed, long* out, hipStream_t st) {
  if (D < 1 || n < 1 || n > D || iters < 1 || D > (1L << 40) || iters * n > (1L << 40)) return 1;
  int b = 2;
  while ((1L << b) < D) ++b;
  if (b & 1) ++b;
  const long total = iters * n;
  hipLaunchKernelGGL(sample_distinct_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, D, n, total, b / 2, seed, out);
  return 0;
}

// U: [K, Dl] fp32 with row stride ld; idx: [iters, n] int64 column ids (an id outside [0, Dl) gives a zero column);
// out: [iters, K, n] fp32.  returns 0 ok.
BLADES_API int blades_gather_center(const float* U, long ld, int K, long Dl, const long* idx, long iters, long n, float* out,
                                    hipStream_t st) {
  if (K < 1 || iters < 1 || n < 1 || iters * n > (1L << 40)) return 1;
  const long total = iters * n;
  hipLaunchKernelGGL(gather_center_kernel, dim3((unsigned)cdiv(total, 32)), dim3(256), 0, st, U, ld, K, Dl, idx, n, total, out);
  return 0;
}
