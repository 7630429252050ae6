// Coordinate-wise order statistics (Median, TrimmedMean) of U[K, Dl].
//
// D independent selections over K values each (K = 4 … 2048, D up to 11 M) —
// SURVEY §7.3 item 4.  The op is VALU-bound (≈50 compare-exchanges per element
// against 4 bytes of HBM traffic), so the design minimises vector instructions
// per element and keeps LDS and barriers out of the picture entirely:
//
//   * a column is sorted by L = N/64 neighbouring lanes that keep R = 64 values
//     (128 for K > 512) each in registers (N = next power of two ≥ K).  Sort position
//     p = s*R + r (s = lane within the group, r = register): the low six
//     position bits live in registers, so 21+ of the network's stages are pure
//     register-to-register v_min/v_max, and only log2(L) bits cross lanes;
//   * the network is the all-ascending "flip" form of the bitonic sorter: the
//     first stage of every merge pairs p with its mirror image in the block,
//     later stages pair p with p^J.  Mirrors over 2/4/8/16 lanes are a single
//     DPP move (quad_perm / row_half_mirror / row_mirror); xor 1/2 are one
//     quad_perm, xor 4/8 two DPP moves.  A cross-lane compare-exchange is
//     v_mov_dpp + v_med3_f32(v, other, ∓inf) — the ∓inf bound (fixed per lane
//     and stage) turns med3 into the min or the max this lane has to keep;
//   * median mode never finishes the sort: rows are padded with alternating
//     ∓inf so that the lower/upper medians sit at positions N/2-1 and N/2, and
//     after the first stage of the last merge those are the max of the lower
//     half and the min of the upper half — one med3 reduction instead of
//     log2(N)-1 more stages;
//   * loads go straight from global memory into the sort registers (a wave
//     reads L rows x 64/L columns per instruction, the four waves of a block
//     cover 128 contiguous bytes of every row at L = 8); blockIdx is remapped
//     so that neighbouring column groups run on the same XCD and share L2 lines.
//
// `fill` (optional, N + 1 floats) marks dead rows: 0 = live row, otherwise the pad
// value to sort instead of the row's data; slot N carries the live-row count.  The host side builds it from the
// row mask; without a mask only rows >= K are padded and that is done inline.
#include "common.h"

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}

// value held by lane (lane ^ M), M in {1,2,3,4,7,8,15,16,31}
template <int M>
__device__ __forceinline__ float lane_xor(float x) {
  if constexpr (M == 1) return dpp<0xB1>(x);        // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return dpp<0x4E>(x);   // quad_perm [2,3,0,1]
  else if constexpr (M == 3) return dpp<0x1B>(x);   // quad_perm [3,2,1,0]
  else if constexpr (M == 7) return dpp<0x141>(x);  // row_half_mirror
  else if constexpr (M == 15) return dpp<0x140>(x); // row_mirror
  else if constexpr (M == 4) return dpp<0x1B>(dpp<0x141>(x));
  else if constexpr (M == 8) return dpp<0x141>(dpp<0x140>(x));
  else {
    static_assert(M == 16 || M == 31, "unsupported lane mask");
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x1F | (M << 10)));
  }
}

__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// ---- one stage of the network -------------------------------------------------
// merge block size KK, exchange distance J (J == KK/2 is the mirror stage)
template <int L, int R, int KK, int J>
__device__ __forceinline__ void stage(float (&v)[R], int s) {
  constexpr bool FLIP = (J * 2 == KK);
  if constexpr (J < R) {
    // register-to-register
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if ((r & J) == 0) {
        const int p = FLIP ? (r ^ (KK - 1)) : (r | J);
        const float a = v[r], b = v[p];
        v[r] = vmin(a, b);
        v[p] = vmax(a, b);
      }
    }
  } else {
    constexpr int X = J / R;  // lane bit that separates the two partners
    const float bound = (s & X) ? INFINITY : -INFINITY;  // keep max : keep min
    if constexpr (FLIP) {
      constexpr int M = 2 * X - 1;
#pragma unroll
      for (int r = 0; r < R / 2; ++r) {
        const float a = v[r], b = v[R - 1 - r];
        const float oa = lane_xor<M>(b), ob = lane_xor<M>(a);
        v[r] = med3(a, oa, bound);
        v[R - 1 - r] = med3(b, ob, bound);
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) v[r] = med3(v[r], lane_xor<X>(v[r]), bound);
    }
  }
}

template <int L, int R, int KK, int J>
struct Merge {
  __device__ __forceinline__ static void run(float (&v)[R], int s) {
    stage<L, R, KK, J>(v, s);
    if constexpr (J > 1) Merge<L, R, KK, J / 2>::run(v, s);
  }
};

// sorts blocks of KK ascending
template <int L, int R, int KK>
struct Sort {
  __device__ __forceinline__ static void run(float (&v)[R], int s) {
    if constexpr (KK > 2) Sort<L, R, KK / 2>::run(v, s);
    Merge<L, R, KK, KK / 2>::run(v, s);
  }
};

// mode 0: median (mean of lower & upper), mode 1: trimmed mean dropping b per side
template <int LG, int R, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(R == 128 ? 3 : 1))) void
col_order_stat_kernel(const float* __restrict__ U, long ld, int K, long Dl,
                                                             const float* __restrict__ fill, int n, int b,
                                                             float* __restrict__ out, int nblk) {
  constexpr int L = 1 << LG, CW = 64 >> LG, N = R * L;
  // XCD-contiguous block order (grid is a multiple of 8)
  const int per = gridDim.x >> 3;
  const int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (blk >= nblk) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & (L - 1), cs = lane >> LG;
  const long col = (long)blk * (4 * CW) + wave * CW + cs;
  const bool col_ok = col < Dl;
  const long colc = col_ok ? col : Dl - 1;
  const unsigned boff = (unsigned)((long)s * ld + colc) * 4u;  // host checks L*ld*4 < 2^32

  float v[R];
  if (fill == nullptr) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int e0 = r * L;  // wave-uniform
      const char* rowp = reinterpret_cast<const char*>(U + (long)e0 * ld);
      if (e0 + L <= K) {
        v[r] = *reinterpret_cast<const float*>(rowp + boff);
      } else {
        const int e = e0 + s;
        float x = (MODE == 0 && ((e - K) & 1)) ? -INFINITY : INFINITY;
        if (e < K) x = *reinterpret_cast<const float*>(rowp + boff);
        v[r] = x;
      }
    }
  } else {
    n = (int)fill[N];  // live-row count rides behind the table
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int e0 = r * L;
      const int e = e0 + s;
      const float f = fill[e];
      float x = f;
      if (f == 0.f && e < K) x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(U + (long)e0 * ld) + boff);
      v[r] = x;
    }
  }

  float res;
  if constexpr (MODE == 0) {
    if constexpr (N > 2) Sort<L, R, N / 2>::run(v, s);
    stage<L, R, N, N / 2>(v, s);  // lower half of the positions now holds the N/2 smallest
    float lo, hi;
    if constexpr (L == 1) {
      float a[8], c[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = v[i]; c[i] = v[R / 2 + i]; }
#pragma unroll
      for (int r = 8; r < R / 2; ++r) { a[r & 7] = vmax(a[r & 7], v[r]); c[r & 7] = vmin(c[r & 7], v[R / 2 + r]); }
#pragma unroll
      for (int w = 4; w >= 1; w >>= 1)
#pragma unroll
        for (int i = 0; i < w; ++i) { a[i] = vmax(a[i], a[i + w]); c[i] = vmin(c[i], c[i + w]); }
      lo = a[0]; hi = c[0];
    } else {
      const float bound = (s & (L / 2)) ? -INFINITY : INFINITY;  // upper lanes: min, lower lanes: max
      float a[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = v[i];
#pragma unroll
      for (int r = 8; r < R; ++r) a[r & 7] = med3(a[r & 7], v[r], bound);
#pragma unroll
      for (int w = 4; w >= 1; w >>= 1)
#pragma unroll
        for (int i = 0; i < w; ++i) a[i] = med3(a[i], a[i + w], bound);
      float m = a[0];
      if constexpr (L >= 4) m = med3(m, lane_xor<1>(m), bound);
      if constexpr (L >= 8) m = med3(m, lane_xor<2>(m), bound);
      if constexpr (L >= 16) m = med3(m, lane_xor<4>(m), bound);
      if constexpr (L >= 32) m = med3(m, lane_xor<8>(m), bound);
      lo = m;
      hi = lane_xor<L / 2>(m);  // meaningful on s == 0
    }
    res = (n & 1) ? lo : 0.5f * (lo + hi);
  } else {
    Sort<L, R, N>::run(v, s);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int p0 = s * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int p = p0 + r;
      if (p >= b && p < n - b) acc[r & 3] += v[r];
    }
    float t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if constexpr (L >= 2) t += lane_xor<1>(t);
    if constexpr (L >= 4) t += lane_xor<2>(t);
    if constexpr (L >= 8) t += lane_xor<4>(t);
    if constexpr (L >= 16) t += lane_xor<8>(t);
    if constexpr (L >= 32) t += lane_xor<16>(t);
    res = t / (float)(n - 2 * b);
  }
  if (s == 0 && col_ok) out[col] = res;
}

template <int LG, int R, int MODE>
void launch(const float* U, long ld, int K, long Dl, const float* fill, int n, int b, float* out, hipStream_t st) {
  constexpr int CW = 64 >> LG;
  const int nblk = (int)cdiv(Dl, (long)(4 * CW));
  const int grid = (nblk + 7) / 8 * 8;
  hipLaunchKernelGGL((col_order_stat_kernel<LG, R, MODE>), dim3(grid), dim3(256), 0, st, U, ld, K, Dl, fill, n, b, out,
                     nblk);
}

template <int LG, int R>
void launch_mode(const float* U, long ld, int K, long Dl, const float* fill, int n, int mode, int b, float* out,
                 hipStream_t st) {
  if (mode == 0) launch<LG, R, 0>(U, ld, K, Dl, fill, n, b, out, st);
  else launch<LG, R, 1>(U, ld, K, Dl, fill, n, b, out, st);
}

}  // namespace

// fill: nullptr, or N + 1 floats with N = 64 * 2^LG (0 = live row, else the pad value; slot N = live-row count,
// which then overrides n).
// returns 0 on success, 1 if K is out of the supported range (<= 2048), 2 if the row stride is too wide,
// 3 on inconsistent n / b.
BLADES_API int blades_col_order_stat(const float* U, long ld, int K, long Dl, const float* fill, int n, int mode, int b,
                                     float* out, hipStream_t st) {
  if (Dl <= 0 || K <= 0) return 0;
  if (K > 2048) return 1;
  if (fill == nullptr && (n != K || (mode == 1 && (b < 0 || 2 * b >= n)))) return 3;
  int lg = 0;
  while ((64 << lg) < K) ++lg;
  if (((long)(1 << lg) * ld + Dl) * 4 >= (1L << 32)) return 2;
  switch (lg) {
    case 0: launch_mode<0, 64>(U, ld, K, Dl, fill, n, mode, b, out, st); break;
    case 1: launch_mode<1, 64>(U, ld, K, Dl, fill, n, mode, b, out, st); break;
    case 2: launch_mode<2, 64>(U, ld, K, Dl, fill, n, mode, b, out, st); break;
    case 3: launch_mode<3, 64>(U, ld, K, Dl, fill, n, mode, b, out, st); break;
    // K > 512: 128 values per lane keep a wave's row segments at >= 16 bytes (measured 20.6 vs 36.6 ms at K = 1000)
    case 4: launch_mode<3, 128>(U, ld, K, Dl, fill, n, mode, b, out, st); break;
    default: launch_mode<4, 128>(U, ld, K, Dl, fill, n, mode, b, out, st); break;
  }
  return 0;
}
