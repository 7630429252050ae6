// What does the matrix pipe sustain on this part under load?  Register-resident MFMA loops on random bf16 data
// (the chip lowers its clock under MFMA load, and trivial operands would hide that): v_mfma_f32_16x16x32_bf16 with 12
// independent accumulators and v_mfma_f32_32x32x16_bf16 with 4, at 1, 2 and 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak.bin mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k16(const i32x4* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  i32x4 a = in[lane], b[12];
  f32x4 acc[12];
  for (int j = 0; j < 12; ++j) { b[j] = in[64 + j * 64 + lane]; acc[j] = f32x4{0, 0, 0, 0}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 12; ++j)
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b[j]), acc[j], 0, 0, 0);
  }
  float s = 0;
  for (int j = 0; j < 12; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(const i32x4* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  i32x4 a = in[lane], b[4];
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) { b[j] = in[64 + j * 64 + lane]; for (int e = 0; e < 16; ++e) acc[j][e] = 0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b[j]), acc[j], 0, 0, 0);
  }
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  std::vector<unsigned short> h(13 * 64 * 8);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.25f; unsigned u; __builtin_memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  i32x4* in; float* out;
  hipMalloc(&in, h.size() * 2); hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&out, 4 * 256 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape = 0; shape < 2; ++shape)
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
      const int grid = 256 * bpc, iters = shape == 0 ? 100000 : 150000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) k16<<<grid, 256>>>(in, out, iters); else k32<<<grid, 256>>>(in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)grid * 4 * iters * (shape == 0 ? 12 * 16384.0 : 4 * 32768.0);
      printf("%s  %d wave(s)/SIMD: %8.2f ms  %6.0f TFLOP/s (%3.0f %% of 2500)\n", shape == 0 ? "16x16x32 x12 acc" : "32x32x16 x4 acc ", bpc, ms, fl / ms / 1e9, fl / ms / 1e9 / 25.0);
    }
  return 0;
}
