// Linear layers with per-client weights (several local steps per round: every client has its own fp32 weight set, a
// row of the engine's W matrix), forward and data gradient, straight from the fp32 master weights:
//
//   mode 0 (forward)        y[c, r, o]  = sum_i x[c, r, i]  * w_c[o, i]  (+ bias_c[o])
//   mode 1 (data gradient)  dx[c, r, i] = sum_o gy[c, r, o] * w_c[o, i]
//
// (reference: nn.Linear forward / backward of fllib/models/mnist/mlp.py and fashionmnist/cnn.py, run per client by
// fllib/clients/client.py:55-61.)  A client's batch is short (R = 64 rows for the MLP / CNN heads), so every weight
// is used 64 times and the layer is bound by reading K x Out x In weights once per pass.  Going through conv_igemm
// needs the weights as bf16 planes: a separate pass (conv_wprep) that reads the fp32 weights and writes 1.5x their
// bytes, which the GEMM then reads again — 2.5x the traffic of the weights themselves, twice per local step.  Here the
// fp32 weights are read once, split exactly into NPL bf16 planes on their way into LDS (same split and product order
// as linear_wgrad.hip / conv_igemm.hip: NPL = 3 gives 6 MFMA products per fp32 product, error below one fp32 rounding),
// and multiplied on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
//
//   workgroup = (client, 64 rows r, 64 output features n — 32 for forward launches with few tiles), tiles of a client on
//   one XCD; 4 waves; wave w owns features [16w, 16w + 16) x all 64 rows.
//   The weights are the MFMA A operand, so a lane ends up with 4 consecutive features of one row: 16-byte stores.
//   LDS images are [64 rows or features][64 k] bf16 per plane, XOR-swizzled (lds_off) so that fragment reads and both
//   kinds of staging writes are free of bank conflicts.  In mode 1 the weights arrive with the feature index
//   contiguous; a lane loads a 4 k x 4 feature block and writes four k's per feature, i.e. transposes on the way in.
//   A staged step covers 64 reduction elements (32 KB of fp32 operands in flight per workgroup, three workgroups per CU);
//   the next step's global loads are issued before the current step's MFMAs (register prefetch; two steps ahead
//   measured the same).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

struct PcLinParams {
  const float* a;     // activations [Kc*R, Ka]  (Ka = In in mode 0, Out in mode 1)
  const float* w;     // client 0's weights [Out, In]
  const float* bias;  // client 0's bias [Out] (mode 0) or nullptr
  float* c;           // output [Kc*R, N]      (N = Out in mode 0, In in mode 1)
  long w_cs, b_cs;    // client strides of w / bias, in floats
  int R, In, Out, mode;
  int a_vec, w_vec, c_vec;  // 16-byte accesses allowed (alignment of base, strides and row lengths)
  int nt, mt, tiles;        // feature tiles, row tiles per client, nt * mt * Kc
};

constexpr int BK = 64;               // reduction elements per staged step
constexpr int PITCH = BK * 2;        // bytes per LDS row: 64 bf16, no padding (swizzled instead)
constexpr int PLANE = 64 * PITCH;    // one plane of one operand tile

// LDS address of the 16-byte granule (8 consecutive k) `gk` of row / feature `f` of a tile image.  64 banks x 4 bytes:
// address bit 7 (the row's parity) picks the bank half, the granule a group of four banks in it.  The row is moved to
// f ^ bit 2 of f and its granules are XOR-ed with ((row >> 1) ^ (row >> 4)) & 7, so that all three access patterns of
// the kernel are conflict-free: the 16-byte fragment reads of 16 consecutive rows, the 8-byte staging writes of 16
// lanes per row over four consecutive rows, and (mode 1) the 8-byte transposing writes to every fourth feature.
__device__ __forceinline__ int lds_off(int f, int gk) {
  const int row = f ^ ((f >> 2) & 1);
  return row * PITCH + ((gk ^ (((row >> 1) ^ (row >> 4)) & 7)) << 4);
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

// 4 consecutive floats
__device__ __forceinline__ u32x4 load4(const float* src, int left, bool vec) {
  u32x4 v = {0, 0, 0, 0};
  if (left <= 0) return v;
  if (vec) return *reinterpret_cast<const u32x4*>(src);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < left) v[j] = __float_as_uint(src[j]);
  return v;
}

// NF = 16-feature fragments per workgroup: 4 (wave w owns fragment w x all 64 rows) or, for forward launches with few
// tiles, 2 (wave w owns fragment w & 1 x rows [32 (w >> 1), + 32): twice the workgroups, mode 0 only)
template <int NPL, int NF>
__global__ __launch_bounds__(256) void pc_linear_kernel(const PcLinParams p) {
  constexpr int WPLANE = NF * 16 * PITCH;
  constexpr int MR = NF == 4 ? 4 : 2;  // row fragments per wave
  __shared__ __attribute__((aligned(16))) unsigned char smem[NPL * (WPLANE + PLANE)];
  unsigned char* w_lds = smem;                  // [NPL][16 NF features][PITCH]
  unsigned char* a_lds = smem + NPL * WPLANE;   // [NPL][64 rows][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): give every XCD a contiguous range of tiles,
  // feature tile fastest, so that the tiles of one client — which all read the client's activation rows — share an L2.
  const int per_xcd = gridDim.x >> 3;
  const int lid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (lid >= p.tiles) return;
  const int n0 = (lid % p.nt) * (16 * NF), r0 = ((lid / p.nt) % p.mt) * 64, c = lid / (p.nt * p.mt);
  const int N = p.mode == 0 ? p.Out : p.In, K = p.mode == 0 ? p.In : p.Out;
  const float* ac = p.a + (size_t)c * p.R * K;
  const float* wc = p.w + (size_t)c * p.w_cs;

  // staging roles.  k-contiguous operands: 4 k's from (tid & 15) * 4 of rows (tid >> 4) + 16 j, j = 0..3 — the 16 lanes of
  // a row read 256 contiguous bytes, a wave's load 4 whole row pieces.  Mode-1 weights: rows k = (tid >> 4) * 4 + i,
  // i = 0..3, 4 features from (tid & 15) * 4 (again 256 contiguous bytes per 16 lanes).
  const int q4 = tid & 15, rbase = tid >> 4;
  auto fetch_rows = [&](const float* base, int row0, int rows, int k0, bool vec, u32x4 (&r)[4], int nj) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= nj) break;
      const int row = row0 + rbase + 16 * j;
      r[j] = load4(base + (size_t)row * K + k0 + q4 * 4, row < rows ? K - (k0 + q4 * 4) : 0, vec);
    }
  };
  auto fetch = [&](u32x4 (&ra)[4], u32x4 (&rw)[4], int k0) {
    fetch_rows(ac, r0, p.R, k0, p.a_vec, ra, 4);
    if (NF == 2 || p.mode == 0) {
      fetch_rows(wc, n0, N, k0, p.w_vec, rw, NF);  // K == In: the weight rows have the same pitch as the reduction length
    } else {
      const int n = n0 + q4 * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + rbase * 4 + i;
        rw[i] = load4(wc + (size_t)k * p.In + n, k < K ? N - n : 0, p.w_vec);
      }
    }
  };
  auto commit_rows = [&](const u32x4 (&r)[4], unsigned char* lds, int plane, int nh) {
    u32x4 pl[NPL];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h >= nh) break;
      split8<NPL>(r[2 * h], r[2 * h + 1], pl);  // words 0, 1: row rbase + 32 h; words 2, 3: row rbase + 32 h + 16
#pragma unroll
      for (int q = 0; q < NPL; ++q) {
        unsigned char* d = lds + q * plane + (q4 & 1) * 8;
        *reinterpret_cast<u32x2v*>(d + lds_off(rbase + 32 * h, q4 >> 1)) = u32x2v{pl[q][0], pl[q][1]};
        *reinterpret_cast<u32x2v*>(d + lds_off(rbase + 32 * h + 16, q4 >> 1)) = u32x2v{pl[q][2], pl[q][3]};
      }
    }
  };
  auto commit = [&](const u32x4 (&ra)[4], const u32x4 (&rw)[4]) {
    commit_rows(ra, a_lds, PLANE, 2);
    if (NF == 2 || p.mode == 0) {
      commit_rows(rw, w_lds, WPLANE, NF / 2);
    } else {
      // rw[i] = row k + i, features n..n+3.  pa[q] = bf16 of (row k | row k+1), pb[q] = (row k+2 | row k+3), two
      // features per word: feature n + j gets the four k's as one 8-byte write
      u32x4 pa[NPL], pb[NPL];
      split8<NPL>(rw[0], rw[1], pa);
      split8<NPL>(rw[2], rw[3], pb);
#pragma unroll
      for (int q = 0; q < NPL; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int sh = (j & 1) * 16;
          const unsigned e0 = (pa[q][j >> 1] >> sh) & 0xffffu, e1 = (pa[q][2 + (j >> 1)] >> sh) & 0xffffu;
          const unsigned e2 = (pb[q][j >> 1] >> sh) & 0xffffu, e3 = (pb[q][2 + (j >> 1)] >> sh) & 0xffffu;
          *reinterpret_cast<u32x2v*>(w_lds + q * WPLANE + lds_off(q4 * 4 + j, rbase >> 1) + (rbase & 1) * 8) =
              u32x2v{e0 | (e1 << 16), e2 | (e3 << 16)};
        }
    }
  };

  f32x4 acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ff = NF == 4 ? wave : (wave & 1), m0 = NF == 4 ? 0 : (wave >> 1) * 2;
  const int g = lane >> 4, li = lane & 15;
  // this lane's 8 k's (granule kh * 4 + g) of row / feature 16 m + li of a tile image
  auto frag = [&](int m, int kh) { return lds_off(m * 16 + li, kh * 4 + g); };

  auto compute = [&]() {
#pragma unroll
    for (int kh = 0; kh < BK / 32; ++kh) {
      bf16x8 a[NPL];
#pragma unroll
      for (int q = 0; q < NPL; ++q)
        a[q] = *reinterpret_cast<const bf16x8*>(w_lds + q * WPLANE + frag(ff, kh));
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        bf16x8 b[NPL];
#pragma unroll
        for (int q = 0; q < NPL; ++q)
          b[q] = *reinterpret_cast<const bf16x8*>(a_lds + q * PLANE + frag(m0 + m, kh));
        f32x4 d = acc[m];
        if constexpr (NPL == 3) {
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], d, 0, 0, 0);
        }
        if constexpr (NPL >= 2) {
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], d, 0, 0, 0);
        }
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], d, 0, 0, 0);
      }
    }
  };

  // (A start step of its own per tile, wrapping around — so that concurrent tiles read different column ranges — measured
  // 5 % slower: the tiles of a client then stop sharing their activation slices in L2.)
  const int nsteps = (K + BK - 1) / BK;
  auto kof = [&](int i) { return i * BK; };
  u32x4 ra0[4], rw0[4];
  // the next step's loads are in flight while this step is multiplied
  fetch(ra0, rw0, kof(0));
  for (int i = 0; i < nsteps; ++i) {
    __syncthreads();  // the previous step's fragments are read
    commit(ra0, rw0);
    __syncthreads();
    if (i + 1 < nsteps) fetch(ra0, rw0, kof(i + 1));
    compute();
  }

  // D rows = features (4 consecutive per lane: g * 4 + reg), cols = rows r (lane & 15)
  const int n = n0 + ff * 16 + g * 4;
  if (n >= N) return;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (p.mode == 0 && p.bias != nullptr) {
    const float* bc = p.bias + (size_t)c * p.b_cs;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n + j < N) bv[j] = bc[n + j];
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int r = r0 + (m0 + m) * 16 + li;
    if (r >= p.R) continue;
    float* dst = p.c + ((size_t)c * p.R + r) * N + n;
    const f32x4 v = acc[m] + bv;
    if (p.c_vec && n + 4 <= N) *reinterpret_cast<f32x4*>(dst) = v;
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n + j < N) dst[j] = v[j];
    }
  }
}

}  // namespace

// a: [Kc*R, In] (mode 0) or [Kc*R, Out] (mode 1), contiguous fp32; w / bias: client 0's set, client c's at + c * stride.
// out: [Kc*R, Out] (mode 0) or [Kc*R, In] (mode 1).  npl 2 / 3.  returns 0 ok.
BLADES_API int blades_pc_linear(const float* a, const float* w, long w_cs, const float* bias, long b_cs, float* out, int Kc,
                                int R, int In, int Out, int mode, int npl, hipStream_t st) {
  if (Kc <= 0 || R <= 0 || In <= 0 || Out <= 0 || mode < 0 || mode > 1 || npl < 2 || npl > 3) return 1;
  PcLinParams p;
  p.a = a; p.w = w; p.bias = mode == 0 ? bias : nullptr; p.c = out; p.w_cs = w_cs; p.b_cs = b_cs;
  p.R = R; p.In = In; p.Out = Out; p.mode = mode;
  const int K = mode == 0 ? In : Out, N = mode == 0 ? Out : In;
  p.a_vec = (K % 4 == 0) && ((uintptr_t)a % 16 == 0);
  p.w_vec = (In % 4 == 0) && (w_cs % 4 == 0) && ((uintptr_t)w % 16 == 0);
  p.c_vec = (N % 4 == 0) && ((uintptr_t)out % 16 == 0);
  p.mt = (R + 63) / 64;
  // forward launches that would not fill the chip three workgroups deep take 32-feature tiles (twice the workgroups)
  const bool half = mode == 0 && (long)((N + 63) / 64) * p.mt * Kc < 768;
  p.nt = half ? (N + 31) / 32 : (N + 63) / 64;
  const long tiles = (long)p.nt * p.mt * Kc;
  if (tiles > (1L << 30)) return 2;
  p.tiles = (int)tiles;
  const dim3 grid((unsigned)((tiles + 7) / 8 * 8));
  if (npl == 3) {
    if (half) hipLaunchKernelGGL((pc_linear_kernel<3, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((pc_linear_kernel<3, 4>), grid, dim3(256), 0, st, p);
  } else {
    if (half) hipLaunchKernelGGL((pc_linear_kernel<2, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((pc_linear_kernel<2, 4>), grid, dim3(256), 0, st, p);
  }
  return 0;
}
