// Per-client weight gradient of a Linear layer, written straight into the update matrix:
//
//   U[c*ld + off + o*In + i] (+)= coef[c] * sum_r gy[c, r, o] * x[c, r, i]        r = the client's R rows
//
// (R = batch size for an MLP / CNN head, batch x tokens for a transformer block).  This is the grouped GEMM that
// engine/fused.py used to hand to torch.bmm + a scaled copy (reference: the per-client backward of nn.Linear in
// fllib/models/mnist/mlp.py and fashionmnist/cnn.py, run one client at a time by Task.train_one_batch).  For the
// Fashion-MNIST CNN it produces 94 % of the update row (fc1: 600 x 2304), for the MLP all of it.
//
// The reduction index (rows) is the outer dimension of both operands in memory, exactly as in conv_wgrad.hip, and
// the same LDS recipe is used: images are [16-channel block][32 rows][16 ch] bf16 and both MFMA operands come out
// transposed through ds_read_b64_tr_b16.  fp32 activations are split exactly into NPL bf16 planes on the way into
// LDS (NPL = 3: 6 products, fp32 grade; 2: 3 products; 1: bf16 inputs as they are), fp32 accumulation.
//   workgroup = (client, 64 inputs i, 64 outputs o), wave w owns inputs [16w, 16w+16) x all 64 outputs;
//   x is the A operand so a lane ends up with 4 consecutive i of one output row (16-byte stores when aligned).
// The reduction is short (R = 64 for the headline batch), so the kernel is bound by writing the K x Out x In fp32
// result once; with many rows per client (token layers of the compact transformers: R = 4096) the next step's loads
// are in flight while the current step is multiplied.
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

struct LinParams {
  const void* x;    // [Kc*R, In]
  const void* gy;   // [Kc*R, Out]
  float* U;
  long ld, off;
  const float* coef;
  int Kc, R, In, Out, accumulate, vec_store;
};

__device__ __forceinline__ i32x2 tr_read(const unsigned char* lds_ptr) {
  return __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                       (__attribute__((address_space(3))) s16x4*)(lds_ptr)));
}

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

template <int NPL>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const LinParams p) {
  using elem_t = std::conditional_t<NPL == 1, unsigned short, float>;
  constexpr int TILE = 4 * 1024;  // one operand tile of one plane: [4 blocks][32 rows][16 ch] bf16
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NPL * TILE];
  unsigned char* x_lds = smem;                // [NPL][4][32][16]
  unsigned char* g_lds = smem + NPL * TILE;   // [NPL][4][32][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.y;
  const int n_it = (p.In + 63) >> 6;
  const int i0 = (blockIdx.x % n_it) * 64, o0 = (blockIdx.x / n_it) * 64;
  const elem_t* xc = reinterpret_cast<const elem_t*>(p.x) + (size_t)c * p.R * p.In;
  const elem_t* gc = reinterpret_cast<const elem_t*>(p.gy) + (size_t)c * p.R * p.Out;

  // staging role: row (tid >> 3) of the 32-row step, 8 channels (tid & 7) * 8 .. + 7 of the 64-channel tile
  const int chunk = tid & 7, row = tid >> 3;
  constexpr int VEC = NPL == 1 ? 8 : 4;  // elements per 16-byte load
  const bool x_vec = (p.In % VEC) == 0, g_vec = (p.Out % VEC) == 0;
  // a step's operands: fetched into registers one step ahead of the MFMAs that use them, split and written to LDS
  // when their turn comes (the reduction of a transformer layer is 4096 rows = 128 steps: without the look-ahead
  // every step waited for its own loads)
  auto fetch = [&](const elem_t* base, int C, int ch0, bool vec, int r, u32x4& lo, u32x4& hi) {
    lo = u32x4{0, 0, 0, 0};
    hi = u32x4{0, 0, 0, 0};
    if (r < p.R && ch0 < C) {
      const elem_t* src = base + (size_t)r * C + ch0;
      if constexpr (NPL == 1) {
        if (vec && ch0 + 8 <= C) lo = *reinterpret_cast<const u32x4*>(src);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned e = ch0 + j < C ? src[j] : 0u;
            lo[j >> 1] |= e << ((j & 1) * 16);
          }
        }
      } else {
        if (vec && ch0 + 8 <= C) {
          lo = *reinterpret_cast<const u32x4*>(src);
          hi = *reinterpret_cast<const u32x4*>(src + 4);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lo[j] = ch0 + j < C ? __float_as_uint(src[j]) : 0u;
            hi[j] = ch0 + 4 + j < C ? __float_as_uint(src[4 + j]) : 0u;
          }
        }
      }
    }
  };
  auto commit = [&](const u32x4 lo, const u32x4 hi, unsigned char* dst) {
    // 8 channels of row `row` -> every plane's [block][row][16 ch] image
    unsigned char* d = dst + (chunk >> 1) * 1024 + row * 32 + (chunk & 1) * 16;
    if constexpr (NPL == 1) {
      *reinterpret_cast<u32x4*>(d) = lo;
    } else {
      u32x4 pl[NPL];
      split8<NPL>(lo, hi, pl);
#pragma unroll
      for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x4*>(d + q * TILE) = pl[q];
    }
  };

  f32x4 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int frag = lq * 32 + lp * 8;  // this lane's address inside a 4-row quad of a [32][16] image

  u32x4 xlo, xhi, glo, ghi;
  fetch(xc, p.In, i0 + chunk * 8, x_vec, row, xlo, xhi);
  fetch(gc, p.Out, o0 + chunk * 8, g_vec, row, glo, ghi);
  for (int r0 = 0; r0 < p.R; r0 += 32) {
    __syncthreads();  // previous step's fragments are read
    commit(xlo, xhi, x_lds);
    commit(glo, ghi, g_lds);
    __syncthreads();
    if (r0 + 32 < p.R) {
      fetch(xc, p.In, i0 + chunk * 8, x_vec, r0 + 32 + row, xlo, xhi);
      fetch(gc, p.Out, o0 + chunk * 8, g_vec, r0 + 32 + row, glo, ghi);
    }
    i32x4 a[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      const unsigned char* xb = x_lds + q * TILE + wave * 1024 + frag;
      const i32x2 lo = tr_read(xb + (0 * 16 + g * 4) * 32), hi = tr_read(xb + (1 * 16 + g * 4) * 32);
      a[q] = i32x4{lo[0], lo[1], hi[0], hi[1]};
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      bf16x8 b[NPL];
#pragma unroll
      for (int q = 0; q < NPL; ++q) {
        const unsigned char* gb = g_lds + q * TILE + m * 1024 + frag;
        const i32x2 lo = tr_read(gb + (0 * 16 + g * 4) * 32), hi = tr_read(gb + (1 * 16 + g * 4) * 32);
        b[q] = __builtin_bit_cast(bf16x8, i32x4{lo[0], lo[1], hi[0], hi[1]});
      }
      f32x4 d = acc[m];
      if constexpr (NPL == 3) {
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1]), b[1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), b[2], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[2]), b[0], d, 0, 0, 0);
      }
      if constexpr (NPL >= 2) {
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), b[1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1]), b[0], d, 0, 0, 0);
      }
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), b[0], d, 0, 0, 0);
    }
  }

  // D rows = i (4 consecutive per lane: g*4 + reg), cols = o (lane & 15)
  const float cf = p.coef ? p.coef[c] : 1.f;
  float* urow = p.U + (size_t)c * p.ld + p.off;
  const int i = i0 + wave * 16 + g * 4;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int o = o0 + m * 16 + li;
    if (o >= p.Out || i >= p.In) continue;
    float* dst = urow + (size_t)o * p.In + i;
    f32x4 v = acc[m] * cf;
    if (p.vec_store && i + 4 <= p.In) {
      if (p.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j < p.In) dst[j] = p.accumulate ? dst[j] + v[j] : v[j];
    }
  }
}

}  // namespace

// x: [Kc*R, In], gy: [Kc*R, Out] contiguous; npl 1: bf16, npl 2 / 3: fp32.  returns 0 ok.
BLADES_API int blades_linear_wgrad(const void* x, const void* gy, float* U, long ld, long off, const float* coef,
                                   int accumulate, int Kc, int R, int In, int Out, int npl, hipStream_t st) {
  if (Kc <= 0 || R <= 0 || In <= 0 || Out <= 0 || npl < 1 || npl > 3) return 1;
  LinParams p;
  p.x = x; p.gy = gy; p.U = U; p.ld = ld; p.off = off; p.coef = coef;
  p.Kc = Kc; p.R = R; p.In = In; p.Out = Out; p.accumulate = accumulate;
  p.vec_store = (ld % 4 == 0) && (off % 4 == 0) && (In % 4 == 0) && ((uintptr_t)U % 16 == 0);
  const int n_it = (In + 63) / 64, n_ot = (Out + 63) / 64;
  if (Kc > 65535) return 2;
  const dim3 grid(n_it * n_ot, Kc);
  if (npl == 3) hipLaunchKernelGGL(linear_wgrad_kernel<3>, grid, dim3(256), 0, st, p);
  else if (npl == 2) hipLaunchKernelGGL(linear_wgrad_kernel<2>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(linear_wgrad_kernel<1>, grid, dim3(256), 0, st, p);
  return 0;
}
