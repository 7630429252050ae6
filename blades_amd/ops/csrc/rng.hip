// Counter-based RNG kernels (Philox4x32-10), bit-compatible with ops/reference.py.
// Element index e is global (row_id * d_total + column), so draws do not depend
// on how the matrix is sharded over GPUs.
#include "common.h"

namespace {
constexpr int TPB = 256;

__device__ __forceinline__ float u24(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ float philox_uniform_at(unsigned long long e, uint32_t k0, uint32_t k1, uint32_t stream) {
  const unsigned long long blk = e >> 2;
  const Philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), stream, 0u, k0, k1);
  const int lane = (int)(e & 3);
  const uint32_t x = lane == 0 ? r.x : lane == 1 ? r.y : lane == 2 ? r.z : r.w;
  return u24(x);
}

__device__ __forceinline__ float philox_normal_at(unsigned long long e, uint32_t k0, uint32_t k1, uint32_t stream) {
  const unsigned long long blk = e >> 1;
  const Philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), stream, 1u, k0, k1);
  const bool odd = e & 1;
  const uint32_t a = odd ? r.z : r.x, b = odd ? r.w : r.y;
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = u24(b);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

__global__ __launch_bounds__(TPB) void uniform_kernel(float* out, long n, unsigned long long offset, uint32_t k0,
                                                      uint32_t k1, uint32_t stream) {
  const long stride = (long)gridDim.x * TPB;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride)
    out[i] = philox_uniform_at(offset + i, k0, k1, stream);
}

__global__ __launch_bounds__(TPB) void normal_kernel(float* out, long n, unsigned long long offset, uint32_t k0,
                                                     uint32_t k1, uint32_t stream, float mean, float std) {
  const long stride = (long)gridDim.x * TPB;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride)
    out[i] = philox_normal_at(offset + i, k0, k1, stream) * std + mean;
}

// rows: [R,2] (local row, global row id)
__global__ __launch_bounds__(TPB) void normal_rows_kernel(float* U, long ld, long Dl, const long* rows, long col_offset,
                                                          long d_total, uint32_t k0, uint32_t k1, uint32_t stream,
                                                          float mean, float std, int overwrite) {
  const long lr = rows[blockIdx.y * 2], g = rows[blockIdx.y * 2 + 1];
  float* p = U + lr * ld;
  const unsigned long long base = (unsigned long long)g * (unsigned long long)d_total + (unsigned long long)col_offset;
  const long stride = (long)gridDim.x * TPB;
  for (long c = (long)blockIdx.x * TPB + threadIdx.x; c < Dl; c += stride) {
    const float z = philox_normal_at(base + c, k0, k1, stream) * std + mean;
    p[c] = overwrite ? z : p[c] + z;
  }
}
// ---- adversary forges (one pass over the column shard each; reference blades/adversaries/adaptive_adversary.py:32-65,
// signguard_adversary.py:39-67 build them from half a dozen D-sized torch.where / arithmetic kernels) ----------------
// Fang-style adaptive attack on Median / TrimmedMean: s = sign(mean), u ~ U(0,1) per coordinate (Philox, global
// coordinate index), b = 2.
__global__ __launch_bounds__(TPB) void adaptive_forge_kernel(const float* __restrict__ mean, const float* __restrict__ mn,
                                                             const float* __restrict__ mx, float* __restrict__ out, long n,
                                                             unsigned long long offset, uint32_t k0, uint32_t k1,
                                                             uint32_t stream, float b) {
  const long stride = (long)gridDim.x * TPB;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
    const float u = philox_uniform_at(offset + i, k0, k1, stream);
    const float m = mean[i], lo = mn[i], hi = mx[i];
    float o = 0.f;
    if (m < 0.f) {
      if (hi > 0.f) o = hi * (1.f + u * (b - 1.f));
      else if (hi < 0.f) o = hi * (1.f + u * (1.f / b - 1.f));
    } else if (m > 0.f) {
      if (lo > 0.f) o = lo * (1.f / b + u * (1.f - 1.f / b));
      else if (lo < 0.f) o = lo * (b + u * (1.f - b));
    } else if (m == 0.f) {
      o = m;  // (a NaN mean matches no case and gives 0, like the reference's chain of torch.where)
    }
    out[i] = o;
  }
}

// Sort keys of the SignGuard attack's random permutation: coordinate e gets (31 random bits : 32-bit global index),
// unique, so "the p coordinates with the smallest keys" is an exact-count uniformly random subset.
__global__ __launch_bounds__(TPB) void perm_keys_kernel(long* __restrict__ keys, long n, unsigned long long offset,
                                                        unsigned long long d_valid_end, uint32_t k0, uint32_t k1,
                                                        uint32_t stream) {
  const long stride = (long)gridDim.x * TPB;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
    const unsigned long long e = offset + i;
    const unsigned long long blk = e >> 2;
    const Philox4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), stream, 0u, k0, k1);
    const int lane = (int)(e & 3);
    const uint32_t x = lane == 0 ? r.x : lane == 1 ? r.y : lane == 2 ? r.z : r.w;
    // padding columns of a shard (e >= d_valid_end) sort behind every real coordinate
    keys[i] = e < d_valid_end ? (long)(((unsigned long long)(x >> 1) << 32) | (e & 0xFFFFFFFFull)) : 0x7FFFFFFFFFFFFFFFll;
  }
}

// out[i] = +u if key < t[0], -u if key < t[1], else 0 (t = the keys of global rank pos and pos + neg), u ~ U(0,1)
__global__ __launch_bounds__(TPB) void signguard_forge_kernel(const long* __restrict__ keys, const long* __restrict__ t,
                                                              float* __restrict__ out, long n, unsigned long long offset,
                                                              uint32_t k0, uint32_t k1, uint32_t stream) {
  const long t1 = t[0], t2 = t[1];
  const long stride = (long)gridDim.x * TPB;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
    const float u = philox_uniform_at(offset + i, k0, k1, stream);
    const long k = keys[i];
    out[i] = k < t1 ? u : (k < t2 ? -u : 0.f);
  }
}

// Client-keyed dropout multipliers: element e of client uid draws lane e & 3 of Philox counter
// (e >> 2, uid, tag, c3) — a function of (seed, client, round/step tag, layer) only, so the masks a client sees do
// not depend on which rank or chunk trains it.  `tag` is read from device memory: a captured HIP graph replays
// the launch with the tag of the current round.
__global__ __launch_bounds__(TPB) void client_dropout_kernel(float* __restrict__ out, const long* __restrict__ uids,
                                                             long n, const long* __restrict__ tag, uint32_t c3,
                                                             uint32_t k0, uint32_t k1, float p, float keep_scale) {
  const long uid = uids[blockIdx.y];
  const uint32_t t = (uint32_t)tag[0];
  float* o = out + (size_t)blockIdx.y * n;
  const long stride = (long)gridDim.x * TPB;
  for (long e = (long)blockIdx.x * TPB + threadIdx.x; e < n; e += stride) {
    const Philox4 r = philox4x32_10((uint32_t)(e >> 2), (uint32_t)uid, t, c3, k0, k1);
    const int lane = (int)(e & 3);
    const uint32_t x = lane == 0 ? r.x : lane == 1 ? r.y : lane == 2 ? r.z : r.w;
    o[e] = u24(x) >= p ? keep_scale : 0.f;
  }
}
// Shuffle keys of the device batch sampler (engine/sampler.py): position `pos` of client `uid` in epoch `ep` gets the
// 64-bit key (w0 : w1) of Philox counter (pos, uid, ep, 7), as a double in [-0.5, 0.5); positions past the client's shard
// get 2.0 so they sort last.  One launch instead of the ~50 elementwise kernels of the torch formulation.
__global__ __launch_bounds__(TPB) void shuffle_keys_kernel(double* __restrict__ keys, const long* __restrict__ uids,
                                                           const long* __restrict__ epochs, const long* __restrict__ lens,
                                                           int nmax, uint32_t k0, uint32_t k1) {
  const int row = blockIdx.y;
  const uint32_t uid = (uint32_t)uids[row], ep = (uint32_t)epochs[row];
  const long len = lens[row];
  for (int pos = blockIdx.x * TPB + threadIdx.x; pos < nmax; pos += gridDim.x * TPB) {
    const Philox4 r = philox4x32_10((uint32_t)pos, uid, ep, 7u, k0, k1);
    // the word pair read as a *signed* 64-bit integer, like the int64 arithmetic of the oracle: keys in [-0.5, 0.5)
    const double key = (double)(long long)(((unsigned long long)r.x << 32) | r.y) * (1.0 / 18446744073709551616.0);
    keys[(size_t)row * nmax + pos] = pos < len ? key : 2.0;
  }
}
}  // namespace

BLADES_API void blades_shuffle_keys(double* keys, const long* uids, const long* epochs, const long* lens, int rows,
                                    int nmax, unsigned long long seed, hipStream_t st) {
  if (rows <= 0 || nmax <= 0) return;
  int g = cdiv(nmax, TPB); if (g > 64) g = 64;
  for (int r0 = 0; r0 < rows; r0 += 65535) {
    const int nr = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL(shuffle_keys_kernel, dim3(g, nr), dim3(TPB), 0, st, keys + (size_t)r0 * nmax, uids + r0,
                       epochs + r0, lens + r0, nmax, (uint32_t)seed, (uint32_t)(seed >> 32));
  }
}

BLADES_API void blades_client_dropout_mult(float* out, const long* uids, int Kc, long n, const long* tag, unsigned c3,
                                           unsigned long long seed, float p, hipStream_t st) {
  if (n <= 0 || Kc <= 0) return;
  int g = cdiv(n, TPB); if (g > 1024) g = 1024;
  hipLaunchKernelGGL(client_dropout_kernel, dim3(g, Kc), dim3(TPB), 0, st, out, uids, n, tag, c3, (uint32_t)seed,
                     (uint32_t)(seed >> 32), p, 1.0f / (1.0f - p));
}

BLADES_API void blades_philox_uniform(float* out, long n, unsigned long long seed, unsigned stream,
                                      unsigned long long offset, hipStream_t st) {
  if (n <= 0) return;
  int g = cdiv(n, TPB); if (g > 4096) g = 4096;
  hipLaunchKernelGGL(uniform_kernel, dim3(g), dim3(TPB), 0, st, out, n, offset, (uint32_t)seed, (uint32_t)(seed >> 32), stream);
}
BLADES_API void blades_philox_normal(float* out, long n, unsigned long long seed, unsigned stream,
                                     unsigned long long offset, float mean, float std, hipStream_t st) {
  if (n <= 0) return;
  int g = cdiv(n, TPB); if (g > 4096) g = 4096;
  hipLaunchKernelGGL(normal_kernel, dim3(g), dim3(TPB), 0, st, out, n, offset, (uint32_t)seed, (uint32_t)(seed >> 32), stream, mean, std);
}
BLADES_API void blades_normal_rows(float* U, long ld, long Dl, const long* rows, int R, long col_offset, long d_total,
                                   unsigned long long seed, unsigned stream, float mean, float std, int overwrite,
                                   hipStream_t st) {
  if (Dl <= 0 || R <= 0) return;
  int g = cdiv(Dl, TPB); if (g > 2048) g = 2048;
  hipLaunchKernelGGL(normal_rows_kernel, dim3(g, R), dim3(TPB), 0, st, U, ld, Dl, rows, col_offset, d_total,
                     (uint32_t)seed, (uint32_t)(seed >> 32), stream, mean, std, overwrite);
}

BLADES_API void blades_adaptive_forge(const float* mean, const float* mn, const float* mx, float* out, long n,
                                      unsigned long long seed, unsigned stream, unsigned long long offset, float b,
                                      hipStream_t st) {
  if (n <= 0) return;
  int g = cdiv(n, TPB); if (g > 4096) g = 4096;
  hipLaunchKernelGGL(adaptive_forge_kernel, dim3(g), dim3(TPB), 0, st, mean, mn, mx, out, n, offset, (uint32_t)seed,
                     (uint32_t)(seed >> 32), stream, b);
}
BLADES_API void blades_perm_keys(long* keys, long n, unsigned long long seed, unsigned stream, unsigned long long offset,
                                 unsigned long long d_valid_end, hipStream_t st) {
  if (n <= 0) return;
  int g = cdiv(n, TPB); if (g > 4096) g = 4096;
  hipLaunchKernelGGL(perm_keys_kernel, dim3(g), dim3(TPB), 0, st, keys, n, offset, d_valid_end, (uint32_t)seed,
                     (uint32_t)(seed >> 32), stream);
}
BLADES_API void blades_signguard_forge(const long* keys, const long* thresholds, float* out, long n,
                                       unsigned long long seed, unsigned stream, unsigned long long offset, hipStream_t st) {
  if (n <= 0) return;
  int g = cdiv(n, TPB); if (g > 4096) g = 4096;
  hipLaunchKernelGGL(signguard_forge_kernel, dim3(g), dim3(TPB), 0, st, keys, thresholds, out, n, offset, (uint32_t)seed,
                     (uint32_t)(seed >> 32), stream);
}
