// Fused small-sequence multi-head attention for the compact transformers (CCT / CVT: 64 tokens x 128 dims x 2 heads
// in the shipped configs; reference fllib/models/backbones/cctnets/utils/transformers.py:17-47 runs
// softmax(q k^T * scale) -> dropout -> @ v as separate ATen kernels, one client at a time).
//
//   forward :  O = dropout(softmax(Q K^T * scale)) V            one workgroup per (sample, head), N <= 64, head dim 64
//   backward:  dQ, dK, dV from Q, K, V, dO, O and the saved row log-sum-exp (scores are recomputed, never stored)
//
// CDNA4 mapping.  Everything is v_mfma_f32_16x16x32_bf16 on 16x16 tiles of the 64x64 score matrix; fp32 tensors are
// split exactly into NPL bf16 planes in registers (3 planes / 6 products = fp32 grade, like the convolution kernels).
//  * Operands whose reduction index is the head dimension (Q, K, V, dO rows) are read straight from global memory:
//    a lane's MFMA fragment is 8 consecutive elements of one row.
//  * An accumulator tile D[4g + r][lane & 15] is re-used as the A operand of the next GEMM without leaving
//    registers: the A fragment of lane (g, l) for a 32-deep step is {D_tile(2s)[.., l], D_tile(2s+1)[.., l]}, i.e. the
//    reduction index is visited in the order 32s + 16 (e >> 2) + 4g + (e & 3).  ds_read_b64_tr_b16 on a
//    [rows][16 columns] LDS image returns exactly that order for the B operand (rows 16 h + 4g .. + 3 of column l), so
//    the second operand (V, K, Q or dO, reduction index = token) is staged once per workgroup as such images.
//  * Forward: wave w owns queries 16w .. 16w + 15: S^T tiles (rows = keys, columns = queries) put one query's 64
//    scores in 16 registers of 4 lanes -> softmax needs two xor-shuffles; P is then the A operand of P V.
//  * Backward pass A: wave w owns keys 16w ..: S and dP tiles with rows = queries, columns = keys become the A
//    operands of dV = Pd^T dO and dK = dS^T Q (reduction over all queries inside the wave, no cross-wave sums).
//    Pass B: wave w owns queries again, dS^T tiles -> dQ = dS K.  Softmax statistics come from the saved L.
//  * Dropout is the engine's client-keyed Philox mask (rng.hip client_dropout_kernel): element
//    e = ((b H + h) N + i) N + j of client uid draws lane e & 3 of counter (e >> 2, uid, tag, c3) — forward and both
//    backward passes regenerate it, nothing is stored.
#include "common.h"
#include <type_traits>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64;              // head dimension
constexpr int NT = 64;              // tokens per workgroup tile (N <= NT)
constexpr int BLK = NT * 32 + 32;   // one 16-column block image [64 rows][16 cols] bf16; +32 B: the four block images a
                                    // staging store instruction touches land on different banks (see conv_wgrad.hip)
constexpr int IMG = 4 * BLK;        // one precision plane of a [64][64] matrix

struct MhaParams {
  const void* qkv;   // [R][N][3][H][64]
  const void* o;     // forward: output [R][N][H*64]; backward: the forward output
  const void* d_o;   // backward: gradient of o
  void* dqkv;        // backward: [R][N][3][H][64]
  float* lse;        // [R][H][N] row log-sum-exp of the scaled scores
  const long* uids;  // [R / B] client ids (dropout key), or null
  const long* tag;   // device scalar: round / step tag of the dropout key
  int R, B, N, H;
  float scale, p_drop, keep;
  uint32_t c3, k0, k1;
};

template <int NPL> struct Frag { i32x4 p[NPL]; };

template <int NPL>
__device__ __forceinline__ void split8(const u32x4 lo, const u32x4 hi, u32x4 (&pl)[NPL]) {
  f32x4 r0 = __builtin_bit_cast(f32x4, lo), r1 = __builtin_bit_cast(f32x4, hi);
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    const bf16x4v b0 = __builtin_convertvector(r0, bf16x4v), b1 = __builtin_convertvector(r1, bf16x4v);
    const u32x2v u0 = __builtin_bit_cast(u32x2v, b0), u1 = __builtin_bit_cast(u32x2v, b1);
    pl[q] = u32x4{u0[0], u0[1], u1[0], u1[1]};
    if (q + 1 < NPL) { r0 -= __builtin_convertvector(b0, f32x4); r1 -= __builtin_convertvector(b1, f32x4); }
  }
}

// 8 fp32 values (two accumulator quads) -> MFMA fragment planes
template <int NPL>
__device__ __forceinline__ Frag<NPL> pack8(const f32x4 a, const f32x4 b) {
  u32x4 pl[NPL];
  split8<NPL>(__builtin_bit_cast(u32x4, a), __builtin_bit_cast(u32x4, b), pl);
  Frag<NPL> f;
#pragma unroll
  for (int q = 0; q < NPL; ++q) f.p[q] = __builtin_bit_cast(i32x4, pl[q]);
  return f;
}

// 8 consecutive elements of a row in global memory (bf16 as stored, fp32 split into planes); !valid -> zeros
template <int NPL, typename T>
__device__ __forceinline__ Frag<NPL> load_row8(const T* ptr, bool valid) {
  Frag<NPL> f;
  if constexpr (NPL == 1) {
    const u32x4 v = valid ? *reinterpret_cast<const u32x4*>(ptr) : u32x4{0, 0, 0, 0};
    f.p[0] = __builtin_bit_cast(i32x4, v);
  } else {
    u32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
    if (valid) { lo = *reinterpret_cast<const u32x4*>(ptr); hi = *reinterpret_cast<const u32x4*>(ptr + 4); }
    u32x4 pl[NPL];
    split8<NPL>(lo, hi, pl);
#pragma unroll
    for (int q = 0; q < NPL; ++q) f.p[q] = __builtin_bit_cast(i32x4, pl[q]);
  }
  return f;
}

__device__ __forceinline__ i32x2 tr_read(const unsigned char* lds_ptr) {
  return __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                       (__attribute__((address_space(3))) s16x4*)(lds_ptr)));
}

// B fragment from an LDS image: column block `img_lane` (lane offset applied), reduction rows row0 .. of step s
template <int NPL>
__device__ __forceinline__ Frag<NPL> tr_frag(const unsigned char* img_lane, int row0) {
  Frag<NPL> f;
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    const i32x2 lo = tr_read(img_lane + q * IMG + row0 * 32), hi = tr_read(img_lane + q * IMG + (row0 + 16) * 32);
    f.p[q] = i32x4{lo[0], lo[1], hi[0], hi[1]};
  }
  return f;
}

template <int NPL>
__device__ __forceinline__ f32x4 mma(const Frag<NPL>& a, const Frag<NPL>& b, f32x4 d) {
#define MM(x, y) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.p[x]), \
                                                              __builtin_bit_cast(bf16x8, b.p[y]), d, 0, 0, 0)
  if constexpr (NPL == 3) { MM(1, 1); MM(0, 2); MM(2, 0); }
  if constexpr (NPL >= 2) { MM(0, 1); MM(1, 0); }
  MM(0, 0);
#undef MM
  return d;
}

// [N rows][64 cols] slice at src (row stride rs elements) -> LDS images, rows >= N zero.  256 threads.
template <int NPL, typename T>
__device__ __forceinline__ void stage(const T* src, long rs, int N, unsigned char* img) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int idx = threadIdx.x + 256 * u, row = idx >> 3, chunk = idx & 7;
    const Frag<NPL> f = load_row8<NPL, T>(src + (size_t)row * rs + chunk * 8, row < N);
    unsigned char* d = img + (chunk >> 1) * BLK + row * 32 + (chunk & 1) * 16;
#pragma unroll
    for (int q = 0; q < NPL; ++q) *reinterpret_cast<i32x4*>(d + q * IMG) = f.p[q];
  }
}

__device__ __forceinline__ float drop_mult(const MhaParams& p, long uid, uint32_t tag, long e) {
  const Philox4 r = philox4x32_10((uint32_t)(e >> 2), (uint32_t)uid, tag, p.c3, p.k0, p.k1);
  const int ln = (int)(e & 3);
  const uint32_t x = ln == 0 ? r.x : ln == 1 ? r.y : ln == 2 ? r.z : r.w;
  return ((float)(x >> 8) * (1.0f / 16777216.0f)) >= p.p_drop ? p.keep : 0.f;
}

__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(unsigned short* p, float v) {
  const __bf16 b = (__bf16)v;
  *p = __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const unsigned short* p) { return __uint_as_float(((unsigned)*p) << 16); }

// ---------------------------------------------------------------------------------------------------------------
template <int NPL>
__global__ __launch_bounds__(256) void mha_fwd_kernel(const MhaParams p) {
  using T = std::conditional_t<NPL == 1, unsigned short, float>;
  __shared__ __attribute__((aligned(16))) unsigned char sV[NPL * IMG];
  const int r = blockIdx.x / p.H, h = blockIdx.x % p.H, N = p.N;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  const long rs = 3L * p.H * HD;
  const T* q0 = reinterpret_cast<const T*>(p.qkv) + ((size_t)r * N * 3 * p.H + h) * HD;
  const T* k0 = q0 + (size_t)p.H * HD;
  const T* v0 = k0 + (size_t)p.H * HD;
  stage<NPL, T>(v0, rs, N, sV);
  __syncthreads();
  const int i0 = 16 * w;
  if (i0 >= N) return;  // wave-uniform; no barrier follows
  const bool drop = p.p_drop > 0.f && p.uids != nullptr;
  const long uid = drop ? p.uids[r / p.B] : 0;
  const uint32_t tag = drop ? (uint32_t)p.tag[0] : 0u;
  const long ebase = ((long)(r % p.B) * p.H + h) * N * N;

  const int i = i0 + li;
  Frag<NPL> qf[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) qf[kk] = load_row8<NPL, T>(q0 + (size_t)i * rs + 32 * kk + 8 * g, i < N);
  f32x4 st[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const int j = 16 * jt + li;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      acc = mma<NPL>(load_row8<NPL, T>(k0 + (size_t)j * rs + 32 * kk + 8 * g, j < N), qf[kk], acc);
    st[jt] = acc;  // S^T[j = 16 jt + 4g + rr][i]
  }
  float mx = -INFINITY;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float s = (16 * jt + 4 * g + rr) < N ? st[jt][rr] * p.scale : -INFINITY;
      st[jt][rr] = s;
      mx = fmaxf(mx, s);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float e = expf(st[jt][rr] - mx);
      st[jt][rr] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  if (g == 0 && i < N) p.lse[((size_t)r * p.H + h) * N + i] = mx + logf(sum);
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      float pv = st[jt][rr] * inv;
      const int j = 16 * jt + 4 * g + rr;
      if (drop && i < N && j < N) pv *= drop_mult(p, uid, tag, ebase + (long)i * N + j);
      st[jt][rr] = pv;
    }
  f32x4 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* vl = sV + (li >> 2) * 32 + (li & 3) * 8;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const Frag<NPL> pa = pack8<NPL>(st[2 * s], st[2 * s + 1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = mma<NPL>(pa, tr_frag<NPL>(vl + dt * BLK, 32 * s + 4 * g), oacc[dt]);
  }
  T* o0 = reinterpret_cast<T*>(const_cast<void*>(p.o)) + ((size_t)r * N * p.H + h) * HD;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int io = i0 + 4 * g + rr;
      if (io < N) stf(o0 + (size_t)io * p.H * HD + 16 * dt + li, oacc[dt][rr]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <int NPL>
__global__ __launch_bounds__(256, 2) void mha_bwd_kernel(const MhaParams p) {
  using T = std::conditional_t<NPL == 1, unsigned short, float>;
  __shared__ __attribute__((aligned(16))) unsigned char sDO[NPL * IMG];
  __shared__ __attribute__((aligned(16))) unsigned char sQ[NPL * IMG];
  __shared__ __attribute__((aligned(16))) unsigned char sK[NPL * IMG];
  __shared__ float sL[NT], sDelta[NT];
  const int r = blockIdx.x / p.H, h = blockIdx.x % p.H, N = p.N;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  const long rs = 3L * p.H * HD, os = (long)p.H * HD;
  const T* q0 = reinterpret_cast<const T*>(p.qkv) + ((size_t)r * N * 3 * p.H + h) * HD;
  const T* k0 = q0 + (size_t)p.H * HD;
  const T* v0 = k0 + (size_t)p.H * HD;
  const T* do0 = reinterpret_cast<const T*>(p.d_o) + ((size_t)r * N * p.H + h) * HD;
  const T* o0 = reinterpret_cast<const T*>(p.o) + ((size_t)r * N * p.H + h) * HD;
  T* dq0 = reinterpret_cast<T*>(p.dqkv) + ((size_t)r * N * 3 * p.H + h) * HD;
  T* dk0 = dq0 + (size_t)p.H * HD;
  T* dv0 = dk0 + (size_t)p.H * HD;
  stage<NPL, T>(do0, os, N, sDO);
  stage<NPL, T>(q0, rs, N, sQ);
  stage<NPL, T>(k0, rs, N, sK);
  {  // delta_i = sum_d dO[i][d] * O[i][d]  (= sum_j dP_ij P_ij), L_i; rows past N: P = exp(.. - inf) = 0
    const int i = threadIdx.x >> 2, part = threadIdx.x & 3;
    float acc = 0.f;
    if (i < N)
#pragma unroll
      for (int d = 0; d < 16; ++d) acc += ldf(do0 + (size_t)i * os + part * 16 + d) * ldf(o0 + (size_t)i * os + part * 16 + d);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) {
      sDelta[i] = acc;
      sL[i] = i < N ? p.lse[((size_t)r * p.H + h) * N + i] : INFINITY;
    }
  }
  __syncthreads();
  const bool drop = p.p_drop > 0.f && p.uids != nullptr;
  const long uid = drop ? p.uids[r / p.B] : 0;
  const uint32_t tag = drop ? (uint32_t)p.tag[0] : 0u;
  const long ebase = ((long)(r % p.B) * p.H + h) * N * N;
  const int lane_off = (li >> 2) * 32 + (li & 3) * 8;

  // ---- pass A: this wave's 16 keys, all queries -> dV, dK -------------------------------------------------------
  const int j0 = 16 * w;
  if (j0 < N) {
    const int j = j0 + li;
    Frag<NPL> kb[2], vb[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      kb[kk] = load_row8<NPL, T>(k0 + (size_t)j * rs + 32 * kk + 8 * g, j < N);
      vb[kk] = load_row8<NPL, T>(v0 + (size_t)j * rs + 32 * kk + 8 * g, j < N);
    }
    f32x4 pd[4], ds[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int ia = 16 * it + li;
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        s = mma<NPL>(load_row8<NPL, T>(q0 + (size_t)ia * rs + 32 * kk + 8 * g, ia < N), kb[kk], s);
        dp = mma<NPL>(load_row8<NPL, T>(do0 + (size_t)ia * os + 32 * kk + 8 * g, ia < N), vb[kk], dp);
      }
      // D[row i = 16 it + 4g + rr][col j]
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int i = 16 * it + 4 * g + rr;
        const float P = j < N ? expf(s[rr] * p.scale - sL[i]) : 0.f;
        const float M = (drop && i < N && j < N) ? drop_mult(p, uid, tag, ebase + (long)i * N + j) : 1.f;
        pd[it][rr] = P * M;
        ds[it][rr] = P * (dp[rr] * M - sDelta[i]) * p.scale;
      }
    }
    f32x4 dv[4], dk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const Frag<NPL> pa = pack8<NPL>(pd[2 * s2], pd[2 * s2 + 1]);
      const Frag<NPL> da = pack8<NPL>(ds[2 * s2], ds[2 * s2 + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = mma<NPL>(pa, tr_frag<NPL>(sDO + lane_off + dt * BLK, 32 * s2 + 4 * g), dv[dt]);
        dk[dt] = mma<NPL>(da, tr_frag<NPL>(sQ + lane_off + dt * BLK, 32 * s2 + 4 * g), dk[dt]);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int jo = j0 + 4 * g + rr;
        if (jo < N) {
          stf(dv0 + (size_t)jo * rs + 16 * dt + li, dv[dt][rr]);
          stf(dk0 + (size_t)jo * rs + 16 * dt + li, dk[dt][rr]);
        }
      }
  }

  // ---- pass B: this wave's 16 queries, all keys -> dQ ------------------------------------------------------------
  const int i0 = 16 * w;
  if (i0 < N) {
    const int i = i0 + li;
    Frag<NPL> qb[2], ob[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      qb[kk] = load_row8<NPL, T>(q0 + (size_t)i * rs + 32 * kk + 8 * g, i < N);
      ob[kk] = load_row8<NPL, T>(do0 + (size_t)i * os + 32 * kk + 8 * g, i < N);
    }
    const float Li = sL[i], Di = sDelta[i];
    f32x4 ds[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int ja = 16 * jt + li;
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        s = mma<NPL>(load_row8<NPL, T>(k0 + (size_t)ja * rs + 32 * kk + 8 * g, ja < N), qb[kk], s);
        dp = mma<NPL>(load_row8<NPL, T>(v0 + (size_t)ja * rs + 32 * kk + 8 * g, ja < N), ob[kk], dp);
      }
      // D[row j = 16 jt + 4g + rr][col i]
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int j = 16 * jt + 4 * g + rr;
        const float P = j < N ? expf(s[rr] * p.scale - Li) : 0.f;
        const float M = (drop && i < N && j < N) ? drop_mult(p, uid, tag, ebase + (long)i * N + j) : 1.f;
        ds[jt][rr] = P * (dp[rr] * M - Di) * p.scale;
      }
    }
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const Frag<NPL> da = pack8<NPL>(ds[2 * s2], ds[2 * s2 + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = mma<NPL>(da, tr_frag<NPL>(sK + lane_off + dt * BLK, 32 * s2 + 4 * g), dq[dt]);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int io = i0 + 4 * g + rr;
        if (io < N) stf(dq0 + (size_t)io * rs + 16 * dt + li, dq[dt][rr]);
      }
  }
}

int fill(MhaParams& p, const void* qkv, const void* o, const void* d_o, void* dqkv, float* lse, const long* uids,
         const long* tag, int R, int B, int N, int H, float scale, float p_drop, unsigned c3, unsigned long long seed) {
  if (R <= 0 || B <= 0 || R % B || N < 1 || N > NT || H < 1 || !(p_drop >= 0.f && p_drop < 1.f)) return 1;
  p.qkv = qkv; p.o = o; p.d_o = d_o; p.dqkv = dqkv; p.lse = lse; p.uids = uids; p.tag = tag;
  p.R = R; p.B = B; p.N = N; p.H = H; p.scale = scale; p.p_drop = p_drop; p.keep = 1.f / (1.f - p_drop);
  p.c3 = c3; p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32);
  return 0;
}

}  // namespace

// qkv [R, N, 3, H, 64] (the output of the fused qkv projection), R = Kc * B samples, client-major; npl 1: bf16,
// 2 / 3: fp32 split into bf16 planes.  o [R, N, H * 64], lse [R, H, N].  uids null or p_drop 0: no dropout.
BLADES_API int blades_mha_forward(const void* qkv, void* o, float* lse, const long* uids, const long* tag, int R, int B,
                                  int N, int H, float scale, float p_drop, unsigned c3, unsigned long long seed, int npl,
                                  hipStream_t st) {
  MhaParams p;
  if (fill(p, qkv, o, nullptr, nullptr, lse, uids, tag, R, B, N, H, scale, p_drop, c3, seed) || npl < 1 || npl > 3) return 1;
  const dim3 grid(R * H);
  if (npl == 3) hipLaunchKernelGGL(mha_fwd_kernel<3>, grid, dim3(256), 0, st, p);
  else if (npl == 2) hipLaunchKernelGGL(mha_fwd_kernel<2>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(mha_fwd_kernel<1>, grid, dim3(256), 0, st, p);
  return 0;
}

BLADES_API int blades_mha_backward(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv,
                                   const long* uids, const long* tag, int R, int B, int N, int H, float scale,
                                   float p_drop, unsigned c3, unsigned long long seed, int npl, hipStream_t st) {
  MhaParams p;
  if (fill(p, qkv, o, d_o, dqkv, const_cast<float*>(lse), uids, tag, R, B, N, H, scale, p_drop, c3, seed) || npl < 1 ||
      npl > 3)
    return 1;
  const dim3 grid(R * H);
  if (npl == 3) hipLaunchKernelGGL(mha_bwd_kernel<3>, grid, dim3(256), 0, st, p);
  else if (npl == 2) hipLaunchKernelGGL(mha_bwd_kernel<2>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(mha_bwd_kernel<1>, grid, dim3(256), 0, st, p);
  return 0;
}
