// Probe: semantics of global_load_lds_dwordx4 on gfx950 with partially masked lanes.
// Expectation: active lane L writes its 16 bytes at lds_base + L*16; inactive lanes leave LDS untouched.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float4* __restrict__ src, float4* dst) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  reinterpret_cast<float4*>(smem)[threadIdx.x] = float4{-1.f, -1.f, -1.f, -1.f};
  __syncthreads();
  const float4* g = src + (255 - threadIdx.x);  // arbitrary per-lane gather
  if (lane & 1)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  dst[threadIdx.x] = reinterpret_cast<float4*>(smem)[threadIdx.x];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float4 *s, *d;
  hipMalloc(&s, 4096); hipMalloc(&d, 4096);
  hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, s, d);
  std::vector<float> o(1024);
  hipMemcpy(o.data(), d, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) {
    const float want = (t & 1) ? (float)((255 - t) * 4) : -1.f;
    if (o[t * 4] != want) { if (bad < 8) printf("thread %d: got %g want %g\n", t, o[t * 4], want); ++bad; }
  }
  printf("lds dma probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
