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
}  // namespace

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
