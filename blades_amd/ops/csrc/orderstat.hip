// Coordinate-wise order statistics (Median, TrimmedMean) of U[K, Dl].
//
// D independent selections over K values each (K = 60 … 2048, D up to 11 M) —
// SURVEY §7.3 item 4.  Structure per workgroup (256 threads = 4 waves):
//
//   1. stage a [K x C] column tile through LDS with float4 row-segment loads
//      (coalesced along D), stored with an odd row stride (C+1) so that the
//      column reads of step 2 hit 64 distinct banks;
//   2. each wave sorts one column at a time with a bitonic network held
//      entirely in registers: lane l keeps R = ceil(K/64) values, element
//      e = r*64 + l.  Exchange distances < 64 cross lanes (one shuffle per
//      value), distances >= 64 are register-to-register — no LDS traffic, no
//      barriers inside the sort;
//   3. the lower/upper medians (or the trimmed sum) are read out of the sorted
//      registers with one cross-lane broadcast.
//
// Masked-out rows and the padding up to 64*R are +inf and sort to the end.
#include "common.h"

namespace {

template <int R> struct Tile { static constexpr int C = (R <= 2) ? 64 : (R <= 8) ? 32 : (R <= 16) ? 16 : 8; };

template <int R, int J, int KK>
__device__ __forceinline__ void stage_lane(float (&v)[R], int lane) {
  // exchange distance J < 64 (cross-lane), merge size KK
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float o = __shfl_xor(v[r], J, 64);
    bool asc;
    if (KK < 64) asc = (lane & KK) == 0;
    else asc = (((r * 64) & KK) == 0);
    const bool lower = (lane & J) == 0;
    const float mn = fminf(v[r], o), mx = fmaxf(v[r], o);
    v[r] = (lower == asc) ? mn : mx;
  }
}

template <int R, int JR, int KK>
__device__ __forceinline__ void stage_reg(float (&v)[R]) {
  // exchange distance JR*64 (register-to-register), merge size KK >= 128
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if ((r & JR) == 0) {
      const int p = r | JR;
      const bool asc = (((r * 64) & KK) == 0);
      const float a = v[r], b = v[p];
      const float mn = fminf(a, b), mx = fmaxf(a, b);
      v[r] = asc ? mn : mx;
      v[p] = asc ? mx : mn;
    }
  }
}

template <int R, int KK, int J>
struct MergeSteps {
  __device__ __forceinline__ static void run(float (&v)[R], int lane) {
    if constexpr (J >= 64) stage_reg<R, J / 64, KK>(v);
    else stage_lane<R, J, KK>(v, lane);
    if constexpr (J > 1) MergeSteps<R, KK, J / 2>::run(v, lane);
  }
};

template <int R, int KK>
struct SortSteps {
  __device__ __forceinline__ static void run(float (&v)[R], int lane) {
    if constexpr (KK > 2) SortSteps<R, KK / 2>::run(v, lane);
    MergeSteps<R, KK, KK / 2>::run(v, lane);
  }
};

// mode 0: median (mean of lower & upper), mode 1: trimmed mean dropping b per side
template <int R>
__global__ __launch_bounds__(256) void col_order_stat_kernel(const float* __restrict__ U, long ld, int K, long Dl,
                                                             const float* __restrict__ mask, int mode, int b,
                                                             float* __restrict__ out) {
  constexpr int C = Tile<R>::C;
  constexpr int LDS_STRIDE = C + 1;
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [K][C+1]
  const long c_base = (long)blockIdx.x * C;
  const int tid = threadIdx.x;
  // ---- 1. stage the tile --------------------------------------------------
  constexpr int LPR = C / 4;  // lanes per row (float4 each)
  const bool vec_ok = ((ld & 3) == 0) && (c_base + C <= Dl);
  for (int i = tid / LPR; i < K; i += 256 / LPR) {
    const int c4 = (tid % LPR) * 4;
    float x0, x1, x2, x3;
    const bool dead = (mask != nullptr) && (mask[i] == 0.f);
    if (dead) {
      x0 = x1 = x2 = x3 = INFINITY;
    } else if (vec_ok) {
      const float4 t = *reinterpret_cast<const float4*>(U + (long)i * ld + c_base + c4);
      x0 = t.x; x1 = t.y; x2 = t.z; x3 = t.w;
    } else {
      const float* p = U + (long)i * ld + c_base + c4;
      const long rem = Dl - (c_base + c4);
      x0 = rem > 0 ? p[0] : 0.f; x1 = rem > 1 ? p[1] : 0.f; x2 = rem > 2 ? p[2] : 0.f; x3 = rem > 3 ? p[3] : 0.f;
    }
    float* d = tile + i * LDS_STRIDE + c4;
    d[0] = x0; d[1] = x1; d[2] = x2; d[3] = x3;
  }
  __syncthreads();
  // ---- 2./3. per-wave register sort + readout -------------------------------
  const int lane = tid & 63, wave = tid >> 6;
  for (int c = wave; c < C; c += 4) {
    if (c_base + c >= Dl) break;  // wave-uniform
    float v[R];
    int nvalid = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int e = r * 64 + lane;
      float x = INFINITY;
      if (e < K) x = tile[e * LDS_STRIDE + c];
      v[r] = x;
      nvalid += (e < K) && (mask == nullptr || mask[e] != 0.f);
    }
    const int n = wave_sum_i(nvalid);
    SortSteps<R, 64 * R>::run(v, lane);
    float res;
    if (mode == 0) {
      const int e_lo = (n - 1) >> 1, e_hi = n >> 1;
      float lo = 0.f, hi = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float a = __shfl(v[r], e_lo & 63, 64);
        const float bb = __shfl(v[r], e_hi & 63, 64);
        if ((e_lo >> 6) == r) lo = a;
        if ((e_hi >> 6) == r) hi = bb;
      }
      res = (lo + hi) * 0.5f;
    } else {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int e = r * 64 + lane;
        if (e >= b && e < n - b) s += v[r];
      }
      res = wave_sum(s) / (float)(n - 2 * b);
    }
    if (lane == 0) out[c_base + c] = res;
  }
}

template <int R>
void launch(const float* U, long ld, int K, long Dl, const float* mask, int mode, int b, float* out,
            hipStream_t st) {
  constexpr int C = Tile<R>::C;
  const size_t lds = (size_t)K * (C + 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(col_order_stat_kernel<R>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((col_order_stat_kernel<R>), dim3(cdiv(Dl, C)), dim3(256), lds, st, U, ld, K, Dl, mask, mode, b,
                     out);
}

}  // namespace

// returns 0 on success, 1 if K is out of the supported range (<= 2048)
BLADES_API int blades_col_order_stat(const float* U, long ld, int K, long Dl, const float* mask, int mode, int b,
                                     float* out, hipStream_t st) {
  if (Dl <= 0 || K <= 0) return 0;
  if (K <= 64) launch<1>(U, ld, K, Dl, mask, mode, b, out, st);
  else if (K <= 128) launch<2>(U, ld, K, Dl, mask, mode, b, out, st);
  else if (K <= 256) launch<4>(U, ld, K, Dl, mask, mode, b, out, st);
  else if (K <= 512) launch<8>(U, ld, K, Dl, mask, mode, b, out, st);
  else if (K <= 1024) launch<16>(U, ld, K, Dl, mask, mode, b, out, st);
  else if (K <= 2048) launch<32>(U, ld, K, Dl, mask, mode, b, out, st);
  else return 1;
  return 0;
}
