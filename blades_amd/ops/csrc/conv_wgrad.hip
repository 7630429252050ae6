// Per-client convolution weight gradient, written straight into the update matrix.
//
//   dW_c[co][ci][r][s] = coef[c] * sum_{b,oh,ow} gy[c,b,oh,ow,co] * x[c,b, oh*st+r-pad, ow*st+s-pad, ci]
//
// x / gy are channels-last bf16 activations of a client-major batch ([Kc*B, H, W, C]); the
// result is fp32 in PyTorch weight order at U[c*ld + off + ((co*Cin+ci)*KS + r)*KS + s]  (Cin there = the
// layer's real channel count `cin_store`; the stem's 3 channels arrive zero-padded to 16).
// This is the one genuinely per-client GEMM of the FedSGD fast path:
// M = Cout, N = Cin*taps, reduction = the client's B*Ho*Wo pixels.
//
// CDNA4 mapping
//  * block = (client, 16*TMT output channels, 64 input channels); wave w owns input-channel
//    block w (16 ci) x all taps x all TMT co-blocks: TMT*taps accumulators of
//    v_mfma_f32_16x16x32_bf16 (D rows = co, cols = ci).
//  * The reduction index (pixels) is the *outer* dimension of both operands in memory, so both
//    MFMA operands are transposed on the way out of LDS with ds_read_b64_tr_b16: LDS images are
//    [16-channel block][pixel][16 ch] (32 B per pixel), a lane group fetches 4 consecutive
//    pixels ("quad") of its channel.  A K-step is 32 output pixels = 8 quads; one half-wave reads
//    two adjacent quads = 256 contiguous bytes -> conflict-free.
//  * x is staged as whole input rows in a circular line buffer (each input row is loaded once
//    per block and reused by all taps and by the neighbouring output rows); the 9 shifted
//    windows are just different LDS offsets.  Stride-2 convs store a row de-interleaved by
//    column parity so that 4 consecutive outputs read 4 consecutive slots.  Zero padding =
//    border slots that are never written + a per-quad row-validity mask.
//  * Loads for step s+1 are issued to registers before the MFMAs of step s and written to LDS
//    after them (T14 "issue early / write late"); one barrier per step.
//  * The host passes a per-step table (same for every client) so the kernel does no index
//    arithmetic with divisions: rows to stage, and per quad {gy pixel, first input row, ow0}.
#include "common.h"
#include <cstdlib>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int MAXP = 4;         // staging passes (32 pixels x 64 ch each) per step
constexpr int TBL = 36;         // ints per step: row_lo,row_hi, 8 quads x {gy pixel, gi0, ih0, ow0}, window_lo, pad

struct WgradParams {
  const unsigned short* x;
  const unsigned short* gy;
  float* U;
  long ld, off;
  const float* coef;
  const int* table;
  int accumulate;
  int B, H, W, Cin, Ho, Wo, Cout, stride, pad;
  int nsteps, nlines, wslots, half;
  int w_shift;          // log2(W) if W is a power of two, else -1
  int split;            // K-split factor; >1: partial sums go to `scratch`, reduced afterwards
  float* scratch;       // [split][Kc][numel]
  long numel;
  int Kc;
  long x_client_stride;  // elements between two clients' inputs
  int rmul, cmul;       // virtual pixel (row, col) -> physical pixel index row*rmul + col*cmul (1x1 strided convs
                        // are run as stride-1 on the subsampled input without copying it)
  int Cs;               // input channels actually stored (< Cin when x was zero-padded to a multiple of 16)
  int ablate;           // diagnostics only: bit0 skip staging after the prologue, bit1 skip MFMA work
};

// 4 consecutive k (pixels) of this lane's channel, as two packed-bf16 dwords (kept as 32-bit lanes so
// masking / concatenation stay whole-register ops instead of 16-bit sdwa/perm sequences)
__device__ __forceinline__ i32x2 tr_read(const unsigned char* lds_ptr) {
  return __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                       (__attribute__((address_space(3))) s16x4*)(lds_ptr)));
}

template <int TMT, int KS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int TAPS = KS * KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.y;
  const int n_ci_tiles = (p.Cin + 63) >> 6;
  const int co0 = (blockIdx.x / n_ci_tiles) * (TMT * 16);
  const int ci0 = (blockIdx.x % n_ci_tiles) * 64;
  unsigned char* gy_lds = smem;                    // [2][TMT][32 px][16 co] bf16
  unsigned char* x_lds = smem + 2 * TMT * 1024;    // [4 ci-blocks][nlines][wslots][16 ci] bf16
  const int plane_bytes = p.nlines * p.wslots * 32;
  const int line_mask = p.nlines - 1;

  for (int i = tid * 16; i < 4 * plane_bytes; i += 256 * 16) *reinterpret_cast<u32x4*>(x_lds + i) = u32x4{0, 0, 0, 0};

  const unsigned short* xc = p.x + (size_t)c * p.x_client_stride;
  const unsigned short* gc = p.gy + (size_t)c * p.B * p.Ho * p.Wo * p.Cout;

  // ---- staging roles: 8 lanes x 16 B cover one pixel's 64-channel tile --------------------
  const int chunk = tid & 7, pix = tid >> 3;
  const bool gy_lane = chunk < TMT * 2 && (co0 + chunk * 8) < p.Cout;
  const bool x_lane = (ci0 + chunk * 8) < p.Cin;
  u32x4 rg_gy = {0, 0, 0, 0};
  u32x4 rg_x[MAXP];

  auto issue = [&](int s) {
    const int* T = p.table + (size_t)s * TBL;
    const int qd = pix >> 2, q = pix & 3;
    const int gbase = T[2 + qd * 4 + 0], ow0 = T[2 + qd * 4 + 3];
    rg_gy = u32x4{0, 0, 0, 0};
    if (gy_lane && gbase >= 0 && ow0 + q < p.Wo)
      rg_gy = *reinterpret_cast<const u32x4*>(gc + (size_t)(gbase + q) * p.Cout + co0 + chunk * 8);
    const int row_lo = T[0], row_hi = T[1];
    const int npix = (row_hi - row_lo + 1) * p.W;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int e = i * 32 + pix;
      rg_x[i] = u32x4{0, 0, 0, 0};
      if (x_lane && e < npix) {
        const int er = p.w_shift >= 0 ? (e >> p.w_shift) : e / p.W;
        const size_t pixel = (size_t)(row_lo + er) * p.rmul + (size_t)(e - er * p.W) * p.cmul;
        rg_x[i] = *reinterpret_cast<const u32x4*>(xc + pixel * p.Cin + ci0 + chunk * 8);
      }
    }
  };
  auto commit = [&](int s) {
    const int* T = p.table + (size_t)s * TBL;
    if (chunk < TMT * 2)
      *reinterpret_cast<u32x4*>(gy_lds + (s & 1) * TMT * 1024 + (chunk >> 1) * 1024 + pix * 32 + (chunk & 1) * 16) = rg_gy;
    const int row_lo = T[0], row_hi = T[1];
    const int npix = (row_hi - row_lo + 1) * p.W;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int e = i * 32 + pix;
      if (e < npix) {
        const int er = p.w_shift >= 0 ? (e >> p.w_shift) : e / p.W;
        const int row = row_lo + er, iw = e - er * p.W + p.pad;
        const int slot = p.stride == 1 ? iw : (iw & 1) * p.half + (iw >> 1);
        *reinterpret_cast<u32x4*>(x_lds + (chunk >> 1) * plane_bytes + ((row & line_mask) * p.wslots + slot) * 32 +
                                  (chunk & 1) * 16) = rg_x[i];
      }
    }
  };

  f32x4 acc[TMT][TAPS];
#pragma unroll
  for (int m = 0; m < TMT; ++m)
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool wave_active = (ci0 + wave * 16) < p.Cin;  // wave-uniform
  const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const unsigned char* x_plane = x_lds + wave * plane_bytes + lq * 32 + lp * 8;

  // this block's share of the K-steps
  const int per = (p.nsteps + p.split - 1) / p.split;
  const int s_begin = blockIdx.z * per;
  const int s_end = s_begin + per < p.nsteps ? s_begin + per : p.nsteps;

  __syncthreads();  // zero fill done
  if (s_begin < s_end) {
    // prologue: the whole row window of the first step, synchronously
    const int* T = p.table + (size_t)s_begin * TBL;
    const int w_lo = T[34], w_hi = T[1];
    const int npix = (w_hi - w_lo + 1) * p.W;
    for (int e = pix; e < npix; e += 32) {
      if (x_lane) {
        const int er = p.w_shift >= 0 ? (e >> p.w_shift) : e / p.W;
        const size_t pixel = (size_t)(w_lo + er) * p.rmul + (size_t)(e - er * p.W) * p.cmul;
        const u32x4 v = *reinterpret_cast<const u32x4*>(xc + pixel * p.Cin + ci0 + chunk * 8);
        const int row = w_lo + er, iw = e - er * p.W + p.pad;
        const int slot = p.stride == 1 ? iw : (iw & 1) * p.half + (iw >> 1);
        *reinterpret_cast<u32x4*>(x_lds + (chunk >> 1) * plane_bytes + ((row & line_mask) * p.wslots + slot) * 32 +
                                  (chunk & 1) * 16) = v;
      }
    }
    {  // gy tile of the first step
      const int qd = pix >> 2, q = pix & 3;
      const int gbase = T[2 + qd * 4 + 0], ow0 = T[2 + qd * 4 + 3];
      u32x4 v = {0, 0, 0, 0};
      if (gy_lane && gbase >= 0 && ow0 + q < p.Wo)
        v = *reinterpret_cast<const u32x4*>(gc + (size_t)(gbase + q) * p.Cout + co0 + chunk * 8);
      if (chunk < TMT * 2)
        *reinterpret_cast<u32x4*>(gy_lds + (s_begin & 1) * TMT * 1024 + (chunk >> 1) * 1024 + pix * 32 +
                                  (chunk & 1) * 16) = v;
    }
  }
  // quad descriptors of the two reads of a K-step, prefetched one step ahead
  int d_gb[2], d_gi[2], d_ih[2], d_ow[2];
  auto load_desc = [&](int s) {
    const int* T = p.table + (size_t)s * TBL;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int* Q = T + 2 + (4 * t + g) * 4;
      d_gb[t] = Q[0]; d_gi[t] = Q[1]; d_ih[t] = Q[2]; d_ow[t] = Q[3];
    }
  };
  if (s_begin < s_end) load_desc(s_begin);
  __syncthreads();

  for (int s = s_begin; s < s_end; ++s) {
    int gi0[2], ih0[2], ow0[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      gi0[t] = d_gi[t];
      ih0[t] = d_gb[t] >= 0 ? d_ih[t] : -1000000;  // quads past the end: every row invalid
      ow0[t] = d_ow[t];
    }
    const bool stage = (s + 1 < s_end) && !(p.ablate & 1);
    if (s + 1 < s_end) load_desc(s + 1);
    if (stage) issue(s + 1);
    if (wave_active && !(p.ablate & 2)) {
      const unsigned char* gy_buf = gy_lds + (s & 1) * TMT * 1024 + lq * 32 + lp * 8;
      i32x4 a[TMT];
#pragma unroll
      for (int m = 0; m < TMT; ++m) {
        const i32x2 lo = tr_read(gy_buf + m * 1024 + (0 * 16 + g * 4) * 32);
        const i32x2 hi = tr_read(gy_buf + m * 1024 + (1 * 16 + g * 4) * 32);
        a[m] = i32x4{lo[0], lo[1], hi[0], hi[1]};
      }
#pragma unroll
      for (int r = 0; r < KS; ++r) {
        int line_off[2];
        int msk[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ih = ih0[t] + r;
          msk[t] = (ih >= 0 && ih < p.H) ? -1 : 0;
          line_off[t] = ((gi0[t] + r) & line_mask) * p.wslots;
        }
#pragma unroll
        for (int sx = 0; sx < KS; ++sx) {
          i32x2 h[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int slot = p.stride == 1 ? ow0[t] + sx : (sx & 1) * p.half + ow0[t] + (sx >> 1);
            h[t] = tr_read(x_plane + (line_off[t] + slot) * 32) & i32x2{msk[t], msk[t]};
          }
          const i32x4 b = {h[0][0], h[0][1], h[1][0], h[1][1]};
#pragma unroll
          for (int m = 0; m < TMT; ++m)
            acc[m][r * KS + sx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, a[m]), __builtin_bit_cast(bf16x8, b), acc[m][r * KS + sx], 0, 0, 0);
        }
      }
    }
    if (stage) commit(s + 1);
    __syncthreads();
  }

  // ---- epilogue: D rows = co (lane>>4)*4+reg, cols = ci (lane&15) ------------------------------
  if (wave_active) {
    const bool to_scratch = p.split > 1;
    const float cf = (p.coef && !to_scratch) ? p.coef[c] : 1.f;
    float* row = to_scratch ? p.scratch + ((size_t)blockIdx.z * p.Kc + c) * p.numel : p.U + (size_t)c * p.ld + p.off;
    const bool accum = p.accumulate && !to_scratch;
    const int ci = ci0 + wave * 16 + li;
#pragma unroll
    for (int m = 0; m < TMT; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = co0 + m * 16 + g * 4 + q;
        if (co < p.Cout && ci < p.Cs) {
          float* dst = row + ((size_t)co * p.Cs + ci) * TAPS;
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            const float v = cf * acc[m][t][q];
            dst[t] = accum ? dst[t] + v : v;
          }
        }
      }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ scratch, int split, int Kc,
                                                           long numel, const float* __restrict__ coef,
                                                           float* __restrict__ U, long ld, long off, int accumulate) {
  const int c = blockIdx.y;
  const float cf = coef ? coef[c] : 1.f;
  float* row = U + (size_t)c * ld + off;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long)gridDim.x * 256) {
    float acc = 0.f;
    for (int z = 0; z < split; ++z) acc += scratch[((size_t)z * Kc + c) * numel + i];  // fixed order
    row[i] = accumulate ? row[i] + cf * acc : cf * acc;
  }
}

template <int TMT, int KS>
int launch(const WgradParams& p, int Kc, hipStream_t st) {
  const size_t lds = 2 * TMT * 1024 + 4 * (size_t)p.nlines * p.wslots * 32;
  if (lds > 160 * 1024) return 3;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<TMT, KS>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int n_ci_tiles = (p.Cin + 63) / 64, n_co_tiles = p.Cout / (TMT * 16);
  hipLaunchKernelGGL((conv_wgrad_kernel<TMT, KS>), dim3(n_co_tiles * n_ci_tiles, Kc, p.split), dim3(256), lds, st, p);
  if (p.split > 1) {
    int gx = cdiv(p.numel, 256 * 4);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, Kc), dim3(256), 0, st, p.scratch, p.split, Kc, p.numel, p.coef,
                       p.U, p.ld, p.off, p.accumulate);
  }
  return 0;
}

}  // namespace

// returns 0 ok; nonzero = unsupported configuration (caller falls back)
BLADES_API int blades_conv_wgrad(const void* x, const void* gy, float* U, long ld, long off, const float* coef,
                                 int accumulate, int Kc, int B, int H, int W, int Cin, int Ho, int Wo, int Cout,
                                 int ks, int stride, int pad, const int* table, int nsteps, int nlines, int wslots,
                                 int tmt, int split, float* scratch, int rmul, int cmul,
                                 long x_client_stride, int cin_store, hipStream_t st) {
  if (!(ks == 1 || ks == 3) || !(stride == 1 || stride == 2) || Cin % 16 || Cout % 16) return 1;
  if (nlines & (nlines - 1)) return 2;
  if (cin_store < 1 || cin_store > Cin) return 7;
  WgradParams p;
  p.x = (const unsigned short*)x; p.gy = (const unsigned short*)gy; p.U = U; p.ld = ld; p.off = off;
  p.coef = coef; p.table = table; p.accumulate = accumulate;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.stride = stride; p.pad = pad;
  p.nsteps = nsteps; p.nlines = nlines; p.wslots = wslots; p.half = wslots / 2;
  p.rmul = rmul; p.cmul = cmul; p.x_client_stride = x_client_stride;
  p.w_shift = -1;
  for (int b = 0; b < 16; ++b) if ((1 << b) == W) p.w_shift = b;
  p.split = split < 1 ? 1 : split; p.scratch = scratch; p.numel = (long)Cout * cin_store * ks * ks; p.Kc = Kc;
  p.Cs = cin_store;
  if (p.split > 1 && scratch == nullptr) return 6;
  { const char* e = getenv("BLADES_WGRAD_ABLATE"); p.ablate = e ? atoi(e) : 0; }
  if (Cout % (tmt * 16)) return 4;
#define GO(T, K) return launch<T, K>(p, Kc, st)
  if (ks == 3) { if (tmt == 4) GO(4, 3); if (tmt == 2) GO(2, 3); if (tmt == 1) GO(1, 3); }
  else { if (tmt == 4) GO(4, 1); if (tmt == 2) GO(2, 1); if (tmt == 1) GO(1, 1); }
#undef GO
  return 5;
}
