// Gram matrix G = U U^T of the update matrix U[K, Dl] with exact-fp32 MFMA.
//
// This is the GEMM behind every pairwise-distance / cosine statistic
// (Multi-Krum, ClippedClustering, MinMax, DnC).  fp32 inputs go
// straight to v_mfma_f32_32x32x2_f32 — bit-for-bit an fp32 fmaf chain, no
// reduced-precision detour, because the consumers subtract nearly equal numbers
// (g_ii + g_jj - 2 g_ij).
//
// Shape is "K tiny, reduction huge" (K <= ~2048 rows, Dl up to 11 M columns), so
// parallelism comes from split-K: block (pair, split) computes one BxB tile of the
// upper triangle over a column range and writes an fp32 slab; a second kernel adds
// the slabs in fixed order (deterministic) and mirrors the lower triangle.
//
// Tile: 256 threads = 2x2 waves, each wave (B/2)x(B/2) out of 32x32 MFMA blocks,
// BK = 32 columns per step staged global -> registers -> LDS with a 36-float row
// stride (conflict-free ds_read_b128, see cdna guide §2).  The k index inside a
// step is permuted identically for both operands (lane half h reads the 4
// consecutive floats 8t+4h..8t+4h+3 with one b128 and feeds them to 4 MFMAs), which
// a dot product does not care about.
#include "common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDS_LD = 36;

// Workgroup -> (tile pair, column split).  The grid is linear; hardware deals consecutive workgroups round-robin over
// the 8 XCDs, so workgroup b is given work item (b % 8) * per_xcd + b / 8: an XCD receives a contiguous range of
// (split, pair) items, i.e. all tile pairs of the same column split run on one XCD at the same time and the eight to
// nine workgroups that read the same 128 rows share them through that XCD's L2 (pair-major order over all XCDs read
// every row block 7.5x from the fabric: profiles/r3_pmc_shipping_kernels_bench_scale.txt, 41.8 GB for a 5.6 GB slab).
__device__ __forceinline__ bool gram_item(int pairs, int nsplit, int& pair, int& split) {
  const int per_xcd = gridDim.x >> 3;
  // (nsplit < 0: dispatch order as it comes, BLADES_GRAM_XCD=0, for A/B runs)
  const int id = nsplit < 0 ? (int)blockIdx.x : (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (nsplit < 0) nsplit = -nsplit;
  if (id >= pairs * nsplit) return false;
  split = id / pairs; pair = id - split * pairs;
  return true;
}

template <int B>
__global__ __launch_bounds__(256) void gram_tile_kernel(const float* __restrict__ U, long ld, int K, long Dl,
                                                        long cols_per_split, int ntile, int Kp, int nsplit,
                                                        float* __restrict__ slabs) {
  constexpr int T = B / 64;  // MFMA blocks per wave per dim
  __shared__ __attribute__((aligned(16))) float lds[2 * B * LDS_LD];
  float* As = lds;
  float* Bs = lds + B * LDS_LD;
  // decode upper-triangular tile pair
  int pair, split, ti = 0;
  if (!gram_item(ntile * (ntile + 1) / 2, nsplit, pair, split)) return;  // (whole workgroup: no barrier is skipped)
  while (pair >= ntile - ti) { pair -= ntile - ti; ++ti; }
  const int tj = ti + pair;
  const long c_begin = (long)split * cols_per_split;
  long c_end = c_begin + cols_per_split;
  if (c_end > Dl) c_end = Dl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  const bool al = (ld & 3) == 0;

  f32x16 acc[T][T];
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  constexpr int F4_PER_TILE = B * (BK / 4);      // float4 per operand tile
  constexpr int LOADS = F4_PER_TILE / 256;        // per thread per operand
  // Global loads run one K-step ahead of the MFMAs: the tile of step c0 + BK is fetched into registers while
  // the tile of step c0 (already in LDS) is being multiplied, so HBM latency hides behind ~4000 MFMA cycles.
  f32x4 ra[LOADS], rb[LOADS];
  auto fetch = [&](long c0) {
#pragma unroll
    for (int q = 0; q < LOADS; ++q) {
      const int f = q * 256 + tid;
      const int row = f / (BK / 4), k4 = (f % (BK / 4)) * 4;
      const long col = c0 + k4;
      const int ga = ti * B + row, gb = tj * B + row;
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (col + 4 <= c_end && al) {
        if (ga < K) va = *reinterpret_cast<const f32x4*>(U + (long)ga * ld + col);
        if (gb < K) vb = *reinterpret_cast<const f32x4*>(U + (long)gb * ld + col);
      } else {
        for (int j = 0; j < 4; ++j) {
          if (col + j < c_end) {
            if (ga < K) va[j] = U[(long)ga * ld + col + j];
            if (gb < K) vb[j] = U[(long)gb * ld + col + j];
          }
        }
      }
      ra[q] = va; rb[q] = vb;
    }
  };
  if (c_begin < c_end) fetch(c_begin);
  for (long c0 = c_begin; c0 < c_end; c0 += BK) {
    __syncthreads();  // previous step's reads done
#pragma unroll
    for (int q = 0; q < LOADS; ++q) {
      const int f = q * 256 + tid;
      const int row = f / (BK / 4), k4 = (f % (BK / 4)) * 4;
      *reinterpret_cast<f32x4*>(As + row * LDS_LD + k4) = ra[q];
      *reinterpret_cast<f32x4*>(Bs + row * LDS_LD + k4) = rb[q];
    }
    __syncthreads();
    if (c0 + BK < c_end) fetch(c0 + BK);
#pragma unroll
    for (int t = 0; t < BK / 8; ++t) {
      f32x4 fa[T], fb[T];
#pragma unroll
      for (int a = 0; a < T; ++a)
        fa[a] = *reinterpret_cast<const f32x4*>(As + (wr * (B / 2) + a * 32 + r32) * LDS_LD + t * 8 + h * 4);
#pragma unroll
      for (int b = 0; b < T; ++b)
        fb[b] = *reinterpret_cast<const f32x4*>(Bs + (wc * (B / 2) + b * 32 + r32) * LDS_LD + t * 8 + h * 4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
          for (int b = 0; b < T; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][s], fb[b][s], acc[a][b], 0, 0, 0);
    }
  }
  // C/D map of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* slab = slabs + (long)split * Kp * Kp;
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = ti * B + wr * (B / 2) + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const int col = tj * B + wc * (B / 2) + b * 32 + r32;
        slab[(long)row * Kp + col] = acc[a][b][q];
      }
}

// ---- the same product on the bf16 matrix cores, fp32 grade (round 3) ---------------------------------------------
// Every fp32 element is split exactly into three bf16 planes (x = h + m + l) on its way into LDS and a tile product
// is the six plane pairs hh + hm + mh + hl + lh + mm with fp32 accumulation — the scheme of conv_igemm.hip (dropped
// terms <= 2^-26 |a||b|).  The fp32 MFMA runs at 1/16 of the bf16 rate, so six bf16 products are still 2.7x cheaper.
// Identical rows still cancel exactly in g_ii + g_jj - 2 g_ij: the three entries are produced by the same
// instruction sequence on the same bits (same K-steps, same split-K ranges, same slab order), whichever tile they
// sit in, and a_h*b_l / a_l*b_h are the same numbers when a = b.
//   tile 128 x 128, 4 waves as 2 x 2, wave = 4 x 4 fragments of v_mfma_f32_16x16x32_bf16, K-step = 32 columns;
//   LDS image per operand [3 planes][128 rows][32 k] bf16 with 64-byte rows, 16-byte groups XOR-swizzled by row bit 2
//   (conflict-free ds_read_b128 fragment reads, see conv_igemm.hip); global loads run one K-step ahead in registers.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 2) void gram_planes_kernel(const float* __restrict__ U, long ld, int K, long Dl,
                                                             long cols_per_split, int ntile, int Kp, int nsplit,
                                                             float* __restrict__ slabs) {
  constexpr int B = 128, PLANE = B * 64;  // bytes of one plane image
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 3 * PLANE];
  unsigned char* As = lds;
  unsigned char* Bs = lds + 3 * PLANE;
  int pair, split, ti = 0;
  if (!gram_item(ntile * (ntile + 1) / 2, nsplit, pair, split)) return;  // (whole workgroup: no barrier is skipped)
  while (pair >= ntile - ti) { pair -= ntile - ti; ++ti; }
  const int tj = ti + pair;
  const long c_begin = (long)split * cols_per_split;
  long c_end = c_begin + cols_per_split;
  if (c_end > Dl) c_end = Dl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, kg = lane >> 4;
  const bool al = (ld & 3) == 0;
  auto swz = [](int row) { return (row & 4) << 3; };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int LOADS = B * (BK / 4) / 256;  // float4 per thread per operand = 4
  f32x4 ra[LOADS], rb[LOADS];
  auto fetch = [&](long c0) {
#pragma unroll
    for (int q = 0; q < LOADS; ++q) {
      const int f = q * 256 + tid;
      const int row = f >> 3, k4 = (f & 7) * 4;
      const long col = c0 + k4;
      const int ga = ti * B + row, gb = tj * B + row;
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (col + 4 <= c_end && al) {
        if (ga < K) va = *reinterpret_cast<const f32x4*>(U + (long)ga * ld + col);
        if (gb < K) vb = *reinterpret_cast<const f32x4*>(U + (long)gb * ld + col);
      } else {
        for (int j = 0; j < 4; ++j) {
          if (col + j < c_end) {
            if (ga < K) va[j] = U[(long)ga * ld + col + j];
            if (gb < K) vb[j] = U[(long)gb * ld + col + j];
          }
        }
      }
      ra[q] = va; rb[q] = vb;
    }
  };
  auto commit = [&](unsigned char* img, f32x4 r, int row, int k4) {
    unsigned char* dst = img + row * 64 + ((((k4 >> 3)) * 16) ^ swz(row)) + ((k4 >> 2) & 1) * 8;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bf16x4 b = __builtin_convertvector(r, bf16x4);
      *reinterpret_cast<u32x2*>(dst + q * PLANE) = __builtin_bit_cast(u32x2, b);
      if (q < 2) r -= __builtin_convertvector(b, f32x4);
    }
  };
  if (c_begin < c_end) fetch(c_begin);
  for (long c0 = c_begin; c0 < c_end; c0 += BK) {
    __syncthreads();  // previous step's fragment reads done
#pragma unroll
    for (int q = 0; q < LOADS; ++q) {
      const int f = q * 256 + tid;
      commit(As, ra[q], f >> 3, (f & 7) * 4);
      commit(Bs, rb[q], f >> 3, (f & 7) * 4);
    }
    __syncthreads();
    if (c0 + BK < c_end) fetch(c0 + BK);
    bf16x8 bf[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int row = wc * 64 + b * 16 + li;
      const unsigned char* src = Bs + row * 64 + ((kg * 16) ^ swz(row));
#pragma unroll
      for (int q = 0; q < 3; ++q) bf[q][b] = *reinterpret_cast<const bf16x8*>(src + q * PLANE);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = wr * 64 + a * 16 + li;
      const unsigned char* src = As + row * 64 + ((kg * 16) ^ swz(row));
      bf16x8 af[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) af[q] = *reinterpret_cast<const bf16x8*>(src + q * PLANE);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        // small terms first; (a plane, b plane)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[1][b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[2][b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bf[0][b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[1][b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[0][b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[0][b], acc[a][b], 0, 0, 0);
      }
    }
  }
  // C/D map of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg
  float* slab = slabs + (long)split * Kp * Kp;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // (in a diagonal tile (r, c) and (c, r) are sums in different plane order: the upper one serves both)
        const int row = ti * B + wr * 64 + a * 16 + kg * 4 + q;
        const int col = tj * B + wc * 64 + b * 16 + li;
        if (row > col) continue;
        slab[(long)row * Kp + col] = acc[a][b][q];
        if (ti == tj && row != col) slab[(long)col * Kp + row] = acc[a][b][q];
      }
}

// ---- K <= 128: the whole Gram matrix is one tile (round 3) ---------------------------------------------------------
// With a single 128-row tile the planes kernel stages the same rows twice (A and B images of a diagonal tile) and
// computes all 64 fragments although 36 are distinct; at K <= 64 the fp32-MFMA tile kernel runs at 1/16 of the bf16
// rate.  Here one image [3 planes][16 NF rows][32 k] serves both operands and the NF (NF + 1) / 2 fragments of the
// upper triangle are dealt over the four waves (NF = 8: wave w takes row band w and row band 7 - w, 9 fragments each,
// the band's a-fragments read once; NF = 4: fragment w + 4 s of the row-major upper triangle).  The mirror image is
// written by the same wave from the same registers, so G stays bitwise symmetric, and every entry is still produced
// by the one instruction sequence (identical rows cancel exactly, see above).
template <int NF>
__global__ __launch_bounds__(256, 2) void gram_small_kernel(const float* __restrict__ U, long ld, int K, long Dl,
                                                            long cols_per_split, int nsplit, float* __restrict__ slabs,
                                                            long u_bs) {
  constexpr int B = 16 * NF, PLANE = B * 64, SLOTS = (NF * (NF + 1) / 2 + 3) / 4;
  __shared__ __attribute__((aligned(16))) unsigned char As[3 * PLANE];
  const int split = blockIdx.x;
  if (split >= nsplit) return;
  U += (size_t)blockIdx.y * u_bs;  // a stack of matrices: one grid row each
  slabs += (size_t)blockIdx.y * nsplit * B * B;
  const long c_begin = (long)split * cols_per_split;
  long c_end = c_begin + cols_per_split;
  if (c_end > Dl) c_end = Dl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;
  const bool al = (ld & 3) == 0;
  auto swz = [](int row) { return (row & 4) << 3; };

  // fragment (fi, fj) of slot s; fi < 0: none
  int fi[SLOTS], fj[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    if constexpr (NF == 8) {
      const int n1 = 8 - wave;
      fi[s] = s < n1 ? wave : 7 - wave;
      fj[s] = s < n1 ? wave + s : 7 - wave + (s - n1);
    } else {
      int idx = wave + 4 * s, r = 0;
      if (idx >= NF * (NF + 1) / 2) { fi[s] = -1; fj[s] = 0; continue; }
      while (idx >= NF - r) { idx -= NF - r; ++r; }
      fi[s] = r; fj[s] = r + idx;
    }
  }
  f32x4 acc[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int LOADS = B * (BK / 4) / 256;
  f32x4 ra[LOADS];
  auto fetch = [&](long c0) {
#pragma unroll
    for (int q = 0; q < LOADS; ++q) {
      const int f = q * 256 + tid;
      const int row = f >> 3, k4 = (f & 7) * 4;
      const long col = c0 + k4;
      f32x4 va = {0.f, 0.f, 0.f, 0.f};
      if (row < K) {
        if (col + 4 <= c_end && al) va = *reinterpret_cast<const f32x4*>(U + (long)row * ld + col);
        else
          for (int j = 0; j < 4; ++j)
            if (col + j < c_end) va[j] = U[(long)row * ld + col + j];
      }
      ra[q] = va;
    }
  };
  auto frag = [&](int f, bf16x8 (&out)[3]) {
    const int row = f * 16 + li;
    const unsigned char* src = As + row * 64 + ((kg * 16) ^ swz(row));
#pragma unroll
    for (int q = 0; q < 3; ++q) out[q] = *reinterpret_cast<const bf16x8*>(src + q * PLANE);
  };
  auto six = [](f32x4 c, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    // small terms first; (a plane, b plane) — the order of gram_planes_kernel
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
    return c;
  };
  if (c_begin < c_end) fetch(c_begin);
  for (long c0 = c_begin; c0 < c_end; c0 += BK) {
    __syncthreads();  // previous step's fragment reads done
#pragma unroll
    for (int q = 0; q < LOADS; ++q) {
      const int f = q * 256 + tid;
      const int row = f >> 3, k4 = (f & 7) * 4;
      unsigned char* dst = As + row * 64 + (((k4 >> 3) * 16) ^ swz(row)) + ((k4 >> 2) & 1) * 8;
      f32x4 r = ra[q];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const bf16x4 b = __builtin_convertvector(r, bf16x4);
        *reinterpret_cast<u32x2*>(dst + p * PLANE) = __builtin_bit_cast(u32x2, b);
        if (p < 2) r -= __builtin_convertvector(b, f32x4);
      }
    }
    __syncthreads();
    if (c0 + BK < c_end) fetch(c0 + BK);
    if constexpr (NF == 8) {
      bf16x8 a1[3], a2[3];
      frag(wave, a1);
      frag(7 - wave, a2);
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        bf16x8 bfr[3];
        frag(fj[s], bfr);
        if (s < 8 - wave) acc[s] = six(acc[s], a1, bfr);
        else acc[s] = six(acc[s], a2, bfr);
      }
    } else {
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        if (fi[s] < 0) continue;
        bf16x8 afr[3], bfr[3];
        frag(fi[s], afr);
        frag(fj[s], bfr);
        acc[s] = six(acc[s], afr, bfr);
      }
    }
  }
  // C/D map of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg
  float* slab = slabs + (long)split * B * B;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    if (fi[s] < 0) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // (a diagonal fragment holds (r, c) and (c, r) as sums in different plane order: keep the upper one)
      const int row = fi[s] * 16 + kg * 4 + q, col = fj[s] * 16 + li;
      if (row > col) continue;
      slab[(long)row * B + col] = acc[s][q];
      if (row != col) slab[(long)col * B + row] = acc[s][q];
    }
  }
}

__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ slabs, int nsplit, int Kp, int K,
                                                          int B, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)K * K) return;
  slabs += (size_t)blockIdx.y * nsplit * Kp * Kp;
  out += (size_t)blockIdx.y * K * K;
  const int i = (int)(idx / K), j = (int)(idx % K);
  // only tiles with tile_row <= tile_col were computed
  int si = i, sj = j;
  if (i / B > j / B) { si = j; sj = i; }
  float acc = 0.f;
  for (int s = 0; s < nsplit; ++s) acc += slabs[(long)s * Kp * Kp + (long)si * Kp + sj];
  out[idx] = acc;
}

// Many splits, few entries (K <= 128 has up to 1024 slabs for 16 K entries, 32 splits and more take this kernel): a thread per entry walks 1024 strided
// loads on 14 workgroups — 0.25 ms of latency, more than the product itself.  Here a workgroup takes 32 entries x 8
// split lanes (lane l adds slabs l, l + 8, ... in order), and the lanes are combined in lane order: still one fixed
// summation order.
__global__ __launch_bounds__(256) void gram_reduce_lanes_kernel(const float* __restrict__ slabs, int nsplit, int Kp, int K,
                                                                int B, float* __restrict__ out) {
  __shared__ float part[8][33];
  const int e = threadIdx.x & 31, l = threadIdx.x >> 5;
  const long idx = (long)blockIdx.x * 32 + e;
  slabs += (size_t)blockIdx.y * nsplit * Kp * Kp;
  out += (size_t)blockIdx.y * K * K;
  float acc = 0.f;
  if (idx < (long)K * K) {
    int i = (int)(idx / K), j = (int)(idx % K);
    if (i / B > j / B) { const int t = i; i = j; j = t; }  // only tiles with tile_row <= tile_col were computed
    const float* src = slabs + (long)i * Kp + j;
    const long stride = (long)Kp * Kp;
    int s = l;
    for (; s + 24 < nsplit; s += 32) {
      const float a = src[s * stride], b = src[(s + 8) * stride], c = src[(s + 16) * stride], d = src[(s + 24) * stride];
      acc += a; acc += b; acc += c; acc += d;
    }
    for (; s < nsplit; s += 8) acc += src[s * stride];
  }
  part[l][e] = acc;
  __syncthreads();
  if (l == 0 && idx < (long)K * K) {
    float t = part[0][e];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += part[q][e];
    out[idx] = t;
  }
}

}  // namespace

BLADES_API void blades_gram_plan(int K, long Dl, int* tile, int* ntile, int* nsplit, long* cols_per_split) {
  const int B = K <= 64 ? 64 : 128;
  const int nt = cdiv(K, B);
  const int pairs = nt * (nt + 1) / 2;
  long want = 1024 / pairs;
  if (want < 1) want = 1;
  long steps = cdiv(Dl, BK);
  long min_steps_per_split = 16;  // >= 512 columns per block
  long max_split = steps / min_steps_per_split;
  if (max_split < 1) max_split = 1;
  if (want > max_split) want = max_split;
  long sps = cdiv(steps, want);
  *tile = B; *ntile = nt; *cols_per_split = sps * BK; *nsplit = cdiv(Dl, sps * BK);
}

namespace {

void gram_launch(const float* U, long ld, long u_bs, int nb, int K, long Dl, float* slabs, float* out, hipStream_t st) {
  int B, nt, nsplit; long cps;
  blades_gram_plan(K, Dl, &B, &nt, &nsplit, &cps);
  const int Kp = nt * B, pairs = nt * (nt + 1) / 2;
  if (Dl <= 0) { (void)hipMemsetAsync(out, 0, sizeof(float) * nb * K * K, st); return; }
  // BLADES_GRAM_IMPL=f32: the exact-fp32 MFMA kernel for every size (A/B runs); default: bf16 planes
  static const bool f32_only = [] { const char* e = getenv("BLADES_GRAM_IMPL"); return e && e[0] == 'f'; }();
  const int grid = (pairs * nsplit + 7) / 8 * 8;
  static const bool no_xcd = [] { const char* e = getenv("BLADES_GRAM_XCD"); return e && atoi(e) == 0; }();
  static const bool no_small = [] { const char* e = getenv("BLADES_GRAM_SMALL"); return e && atoi(e) == 0; }();
  const int ns = nsplit;
  if (nt == 1 && ((!f32_only && !no_small) || nb > 1)) {
    // (one launch item per column split: no XCD sharing to arrange)
    if (B == 64)
      hipLaunchKernelGGL((gram_small_kernel<4>), dim3(ns, nb), dim3(256), 0, st, U, ld, K, Dl, cps, ns, slabs, u_bs);
    else
      hipLaunchKernelGGL((gram_small_kernel<8>), dim3(ns, nb), dim3(256), 0, st, U, ld, K, Dl, cps, ns, slabs, u_bs);
  } else {
    if (no_xcd) nsplit = -nsplit;
    if (B == 64)
      hipLaunchKernelGGL((gram_tile_kernel<64>), dim3(grid), dim3(256), 0, st, U, ld, K, Dl, cps, nt, Kp, nsplit, slabs);
    else if (f32_only)
      hipLaunchKernelGGL((gram_tile_kernel<128>), dim3(grid), dim3(256), 0, st, U, ld, K, Dl, cps, nt, Kp, nsplit, slabs);
    else
      hipLaunchKernelGGL(gram_planes_kernel, dim3(grid), dim3(256), 0, st, U, ld, K, Dl, cps, nt, Kp, nsplit, slabs);
  }
  if (ns >= 32)
    hipLaunchKernelGGL(gram_reduce_lanes_kernel, dim3(cdiv((long)K * K, 32), nb), dim3(256), 0, st, slabs, ns, Kp, K, B, out);
  else
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(cdiv((long)K * K, 256), nb), dim3(256), 0, st, slabs, ns, Kp, K, B, out);
}

}  // namespace

// slabs: workspace of nsplit * Kp * Kp floats, Kp = ntile * tile
BLADES_API void blades_gram(const float* U, long ld, int K, long Dl, float* slabs, float* out, hipStream_t st) {
  if (K <= 0) return;
  gram_launch(U, ld, 0, 1, K, Dl, slabs, out, st);
}

// nb matrices of K <= 128 rows (one tile), matrix b at U + b * u_bs; slabs: nb * nsplit * Kp * Kp floats; out: [nb, K, K].
// returns 0 ok.
BLADES_API int blades_gram_batched(const float* U, long ld, long u_bs, int nb, int K, long Dl, float* slabs, float* out,
                                   hipStream_t st) {
  if (K < 1 || K > 128 || nb < 1 || nb > 65535) return 1;
  gram_launch(U, ld, u_bs, nb, K, Dl, slabs, out, st);
  return 0;
}
