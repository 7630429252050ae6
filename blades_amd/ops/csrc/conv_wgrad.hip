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
//    two adjacent quads = 256 contiguous bytes -> conflict-free (PMC: 0.7 M conflict cycles with staging off).
//  * x is staged as whole input rows in a circular line buffer (each input row is loaded once
//    per block and reused by all taps and by the neighbouring output rows); the 9 shifted
//    windows are just different LDS offsets.  Stride-2 convs store a row de-interleaved by
//    column parity so that 4 consecutive outputs read 4 consecutive slots.  Zero padding =
//    border slots that are never written; input rows outside the image (and quads past the end of
//    the pixel list) are redirected to one all-zero line, so fragments need no masking.
//  * Pipeline per K-step: B fragments are read one kernel row ahead of the MFMAs that use them;
//    global loads are staged through registers NPF steps ahead and written to LDS after the MFMAs
//    of the step before they are needed (one barrier per step).  The staged loads are issued
//    through inline asm and awaited with hand-counted s_waitcnt vmcnt(N): every wave issues exactly
//    1 + MAXP loads per step, so "step s+1 has landed" is a constant count — the compiler's own
//    waitcnt insertion degrades to vmcnt(0) around rotating registers and would drain the queue.
//  * Descriptors: for power-of-two output geometry (every ResNet layer) a quad's (image, row,
//    column), the gy pixel range and the new-row range of a step follow from the step index by
//    shifts (FAST path, nothing is loaded inside the K loop).  Other shapes read a per-step table
//    the host builds once per layer shape (48 ints per step, through the scalar cache).
//  * Grid: one workgroup per (client, co/ci tile, K split).  The split factor is chosen so the grid is
//    close to a multiple of the 512 resident workgroups (no ragged last wave); partial sums go to a
//    scratch slab and are added in fixed order (deterministic).  Workgroup ids are remapped so that
//    all tiles of a client run on one XCD and share its rows through that XCD's L2.
//  * Each workgroup starts at its own rotation point of its step range (diagnostic bit 4 disables it).
//
// Measured (MI355X, 200 clients x batch 64, tools/bench_wgrad.py): 3x3 stride-1 layers 730-810 TFLOP/s,
// stride-2 260-340, 1x1 190-220; per-layer ablations and PMC counters are in profiles/.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int TBL = 48;         // ints per step (192 B, 64 B-aligned rows): 8 quads x {gy pixel, gi0, ih0, ow0},
                                // then row_lo, row_hi, window_lo at [32], [33], [34]

struct WgradParams {
  const void* x;   // bf16 (NPL == 1) or fp32 (NPL == 2, 3: split into bf16 planes on the way into LDS)
  const void* gy;
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
  int ntiles, total_blocks;  // (co, ci) tiles per client; ntiles * Kc * split
  int rows_r, rows_off, rows_last;  // rows_r > 0: new-row range of step s is arithmetic (see fetch_info)
  int lg_qpr, lg_ho, nquads;  // FAST path (Wo % 4 == 0, Wo/4 and Ho powers of two): quad -> (b, oh, ow0) by shifts
  int Cs;               // input channels actually stored (< Cin when x was zero-padded to a multiple of 16)
  int ablate;           // diagnostics only (BLADES_WGRAD_ABLATE): bit0 skip staging after the prologue, bit1 skip
                        // the MFMA work, bit2 no rotation of the step range
  int s2_classes;       // MASKED variant: the four parity classes of a stride-2 3x3 layer in one grid (see the API)
};

// 4 consecutive k (pixels) of this lane's channel, as two packed-bf16 dwords (kept as 32-bit lanes so
// masking / concatenation stay whole-register ops instead of 16-bit sdwa/perm sequences)
__device__ __forceinline__ i32x2 tr_read(const unsigned char* lds_ptr) {
  return __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                       (__attribute__((address_space(3))) s16x4*)(lds_ptr)));
}

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// 16-byte global load the compiler does not track: completion is awaited by hand with vmcnt counting (below),
// because the compiler's own waitcnt insertion falls back to vmcnt(0) around the rotating staging registers
// and would drain the prefetch queue every K-step.  Nothing may read the result before that manual wait.
__device__ __forceinline__ u32x4 gload16_untracked(const void* ptr) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MAXP: staging passes (32 pixels x 64 channels each) a K-step may need for new input rows — 1 for every
// stride-1 layer, up to 4 for stride 2.  NPF: K-steps of global loads in flight (register-staged); the
// stride-1 variants run 3 deep, which is what hides HBM latency (measured: staging alone cost 1.4 of 1.5 ms
// per layer at depth 1).
// FAST: output geometry is all powers of two (every ResNet layer), so a quad's (image, row, column) and the gy
// pixel range of a step follow from the step index by shifts — no descriptor loads at all in the K loop; only
// the range of new input rows still comes from the table (one 8-byte scalar load, fetched a step early).
// NPL: operand planes.  1 = bf16 activations as they are.  2 / 3 = fp32 activations, split exactly into bf16 planes
// (x = h + m (+ l)) when a staged tile is committed to LDS; every plane pair that matters is multiplied
// (NPL = 3: hh + hm + mh + hl + lh + mm, fp32 grade; NPL = 2: hh + hm + mh) into the same fp32 accumulators.
// x = h + m (+ l) with bf16 planes, exact.  One v_cvt_pk_bf16_f32 per *pair* of values and plane: the packed result
// is both the plane's output dword and, shifted / masked back into fp32 position, the value to subtract for the next
// plane (the vector-convert formulation costs three conversions per pair).
template <int NPL>
__device__ __forceinline__ void split8(const u32x4 lo, const u32x4 hi, u32x4 (&pl)[NPL]) {
  typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  float r[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = __uint_as_float(lo[j]); r[4 + j] = __uint_as_float(hi[j]); }
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v{r[2 * j], r[2 * j + 1]}), bf16x2v));
      pl[q][j] = u;
      if (q + 1 < NPL) {
        r[2 * j] -= __uint_as_float(u << 16);
        r[2 * j + 1] -= __uint_as_float(u & 0xffff0000u);
      }
    }
  }
}

template <int TMT, int KS, int MAXP, int NPF, bool FAST, int NPL, bool MASKED = false>
__global__ __launch_bounds__(256, NPL == 1 ? 2 : 1) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int TAPS = KS * KS;
  using elem_t = std::conditional_t<NPL == 1, unsigned short, float>;
  constexpr int LPI = NPL == 1 ? 1 : 2;  // 16-byte loads per staged item (8 channels of one pixel)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid, XCD-aware: hardware deals consecutive workgroup ids round-robin to the 8 XCDs, so id -> id'
  // below gives every XCD one contiguous range of logical blocks.  Logical order is tile-fastest, hence all
  // (co, ci) tiles of one client's K-slice run on the same XCD at about the same time and share the client's
  // x / gy rows through that XCD's L2 instead of each fetching them from HBM.
  const int per_xcd = gridDim.x >> 3;
  const int lid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (lid >= p.total_blocks) return;
  const int n_ci_tiles = (p.Cin + 63) >> 6;
  const int tile = lid % p.ntiles, rest = lid / p.ntiles;
  const int c = rest % p.Kc, zsplit = rest / p.Kc;
  // MASKED: a (co, ci) tile exists once per parity class; class = (row parity, column parity) of the input pixels.
  // The class's input is x[:, rp::2, cp::2] (rmul / cmul address it in place); kernel row r of the stride-1 problem
  // it poses is the layer's kernel row 2r (odd rows, r = 0, 1) or 1 (even rows, r = 1) — same for columns — so the
  // class owns 4, 2, 2 or 1 of the 9 taps, computes only those and stores only those.
  const int n_cls_tiles = MASKED ? 4 * n_ci_tiles : n_ci_tiles;
  const int co0 = (tile / n_cls_tiles) * (TMT * 16);
  const int cls = MASKED ? (tile % n_cls_tiles) / n_ci_tiles : 0;
  const int ci0 = ((tile % n_cls_tiles) % n_ci_tiles) * 64;
  const int rp = cls >> 1, cp = cls & 1;
  const int row_bits = rp ? 3 : 2, col_bits = cp ? 3 : 2;  // kernel rows / columns (of the stride-1 problem) in use
  int tap_mask = 0x1ff;
  if constexpr (MASKED) {
    tap_mask = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if ((row_bits >> r) & 1) tap_mask |= col_bits << (r * 3);
  }
  auto tap_dst = [&](int t) {  // where tap t of the class goes in the layer's 3x3 filter
    const int r = t / 3, sx = t - r * 3;
    return (rp ? 2 * r : 1) * 3 + (cp ? 2 * sx : 1);
  };
  // LDS stores are banked over 128 B in groups of 8 lanes = the 8 channel chunks of one pixel, i.e. four 32-byte
  // pieces one per 16-channel block: block images sit at odd multiples of 32 B (mod 128) from each other (GYM below,
  // the +32 of plane_bytes) or all four pieces land on the same 8 banks.  Measured with the kernel's ablation
  // switches (profiles/r2_pmc_hot_kernels.txt): every bank-conflict cycle of this kernel came from these stores
  // (4-way, 24 M cycles of 58 M LDS cycles); the transposed fragment reads are conflict-free.
  constexpr int GYM = 1024 + 32;                         // one 16-co block of a gy tile: [32 px][16 co] bf16 + pad
  constexpr int GYB = TMT * GYM;                         // one gy tile of one plane
  unsigned char* gy_lds = smem;                          // [2][NPL][TMT][32 px][16 co] bf16
  unsigned char* x_lds = smem + 2 * NPL * GYB;           // [NPL][4 ci-blocks][nlines][wslots][16 ci] bf16
  const int plane_bytes = (p.nlines + 1) * p.wslots * 32 + 32;  // ring of nlines + one all-zero line (index nlines)
  const int xpl_bytes = 4 * plane_bytes;                 // one precision plane of the x image
  const int line_mask = p.nlines - 1;

  for (int i = tid * 16; i < NPL * xpl_bytes; i += 256 * 16) *reinterpret_cast<u32x4*>(x_lds + i) = u32x4{0, 0, 0, 0};

  const elem_t* xc = reinterpret_cast<const elem_t*>(p.x) + (size_t)c * p.x_client_stride +
                     (MASKED ? (size_t)(rp * (p.rmul >> 1) + cp) * p.Cin : (size_t)0);
  const elem_t* gc = reinterpret_cast<const elem_t*>(p.gy) + (size_t)c * p.B * p.Ho * p.Wo * p.Cout;

  // ---- staging roles: 8 lanes x 16 B cover one pixel's 64-channel tile --------------------
  const int chunk = tid & 7, pix = tid >> 3;
  const bool gy_lane = chunk < TMT * 2 && (co0 + chunk * 8) < p.Cout;
  const bool x_lane = (ci0 + chunk * 8) < p.Cin;
  struct Item { u32x4 v[LPI]; };
  struct Stg { Item gy; Item x[MAXP]; };
  auto fetch = [&](const elem_t* src) {
    Item it;
#pragma unroll
    for (int l = 0; l < LPI; ++l) it.v[l] = gload16_untracked(reinterpret_cast<const unsigned char*>(src) + 16 * l);
    return it;
  };
  // one staged item -> its 16 bytes in every precision plane (plane q at dst + q * plane_stride)
  auto put = [&](unsigned char* dst, int plane_stride, const Item& it) {
    if constexpr (NPL == 1) {
      *reinterpret_cast<u32x4*>(dst) = it.v[0];
    } else {
      u32x4 pl[NPL];
      split8<NPL>(it.v[0], it.v[LPI - 1], pl);
#pragma unroll
      for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x4*>(dst + q * plane_stride) = pl[q];
    }
  };
  Stg stg[NPF];
  // step table through the scalar cache: every index below is wave-uniform, so these loads never enter the
  // vector-memory queue and cannot force the in-flight staging loads to drain
  const __attribute__((address_space(4))) int* tbl =
      (const __attribute__((address_space(4))) int*)(p.table);
  const int wq = __builtin_amdgcn_readfirstlane(wave) * 2;  // the two gy quads this wave stages: wq, wq + 1

  // Staging pipeline.  Every issue() puts exactly 1 + MAXP loads per wave into the vector-memory queue (lanes
  // with nothing to fetch read 16 zero bytes that sit behind the step table), so "the loads of step s+1 have
  // landed" is the constant wait vmcnt((NPF-1) * (1 + MAXP)).  What issue() needs from the step table — the two
  // gy quads this wave stages and the range of new input rows — is fetched through the scalar cache one
  // iteration ahead (StepInfo) and kept in SGPRs, so neither its latency nor a vector-queue drain sits in
  // front of the loads.
  struct StepInfo { i32x8 q; int step, row_lo, row_hi; };
  auto fetch_info = [&](int s) {
    StepInfo f;
    f.step = s;
    if (FAST && p.rows_r > 0) {
      // regular geometry (checked on the host against the planned table): step s > 0 brings in the rows
      // (s*R + off, (s+1)*R + off], clipped to the client's last row — nothing to load at all
      const int hi = (s + 1) * p.rows_r + p.rows_off, lo = s * p.rows_r + p.rows_off;
      f.row_hi = hi < p.rows_last ? hi : p.rows_last;
      f.row_lo = (lo < p.rows_last ? lo : p.rows_last) + 1;
      return f;
    }
    const __attribute__((address_space(4))) int* T = tbl + (size_t)s * TBL;
    if constexpr (!FAST) f.q = *reinterpret_cast<const __attribute__((address_space(4))) i32x8*>(T + wq * 4);
    f.row_lo = T[32];
    f.row_hi = T[33];
    return f;
  };
  const elem_t* zeros16 = reinterpret_cast<const elem_t*>(p.table + (size_t)p.nsteps * TBL);  // >= 32 zero bytes
  int rlo[NPF], rhi[NPF];  // new-row range of the step held by each staging slot (needed again at commit)
  auto issue = [&](const StepInfo& f, Stg& rg) {
    const elem_t* ga = zeros16;
    if constexpr (FAST) {
      const int gpix = f.step * 32 + pix;  // quads tile the flattened pixel list: a step is 32 consecutive pixels
      if (gy_lane && gpix < p.nquads * 4) ga = gc + (size_t)gpix * p.Cout + co0 + chunk * 8;
    } else {
      const bool hi_q = (lane & 32) != 0;  // pix >> 2 == wq + (lane >> 5)
      const int gbase = hi_q ? f.q[4] : f.q[0];
      const int ow0 = hi_q ? f.q[7] : f.q[3];
      const int q = pix & 3;
      if (gy_lane && gbase >= 0 && ow0 + q < p.Wo) ga = gc + (size_t)(gbase + q) * p.Cout + co0 + chunk * 8;
    }
    rg.gy = fetch(ga);
    const int npix = (f.row_hi - f.row_lo + 1) * p.W;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int e = i * 32 + pix;
      const elem_t* xa = zeros16;
      if (x_lane && e < npix) {
        const int er = p.w_shift >= 0 ? (e >> p.w_shift) : e / p.W;
        const size_t pixel = (size_t)(f.row_lo + er) * p.rmul + (size_t)(e - er * p.W) * p.cmul;
        xa = xc + pixel * p.Cin + ci0 + chunk * 8;
      }
      rg.x[i] = fetch(xa);
    }
    // The table-driven variants are register-tight enough for the compiler to spill; a spill of a staging
    // register before its load has landed would save garbage, so they wait here (they are the slow general
    // path anyway).  The FAST variants must compile without spills — tools/check_wgrad_spills.py verifies it.
    if constexpr (!FAST) wait_vmcnt<0>();
  };
  auto commit = [&](int s, const Stg& rg, int row_lo, int row_hi) {
    if (chunk < TMT * 2)
      put(gy_lds + (s & 1) * NPL * GYB + (chunk >> 1) * GYM + pix * 32 + (chunk & 1) * 16, GYB, rg.gy);
    const int npix = (row_hi - row_lo + 1) * p.W;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int e = i * 32 + pix;
      if (e < npix) {
        const int er = p.w_shift >= 0 ? (e >> p.w_shift) : e / p.W;
        const int row = row_lo + er, iw = e - er * p.W + p.pad;
        const int slot = p.stride == 1 ? iw : (iw & 1) * p.half + (iw >> 1);
        put(x_lds + (chunk >> 1) * plane_bytes + ((row & line_mask) * p.wslots + slot) * 32 + (chunk & 1) * 16,
            xpl_bytes, rg.x[i]);
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
  const int blk_begin = zsplit * per;
  const int blk_end = blk_begin + per < p.nsteps ? blk_begin + per : p.nsteps;
  int s_begin = 0, s_end = 0;  // the segment of [blk_begin, blk_end) being processed (see the loop at the end)

  __syncthreads();  // zero fill done
  // quad descriptors of the two reads of a K-step, prefetched one step ahead
  int d_gb[2], d_gi[2], d_ih[2], d_ow[2];
  auto load_desc = [&](int s) {
    if constexpr (FAST) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int Q = s * 8 + 4 * t + g;
        const int qx = Q & ((1 << p.lg_qpr) - 1);
        const int oh = (Q >> p.lg_qpr) & ((1 << p.lg_ho) - 1);
        const int b = Q >> (p.lg_qpr + p.lg_ho);
        d_gb[t] = Q < p.nquads ? 0 : -1;
        d_ow[t] = qx * 4;
        d_ih[t] = oh * p.stride - p.pad;
        d_gi[t] = b * p.H + d_ih[t];
      }
      return;
    }
    // all eight quads through two 64-byte scalar loads, then each 16-lane group picks its own (g, g + 4)
    const __attribute__((address_space(4))) i32x16* T16 =
        reinterpret_cast<const __attribute__((address_space(4))) i32x16*>(tbl + (size_t)s * TBL);
    const i32x16 qa = T16[0], qb = T16[1];
    int v[2][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      int x = qa[f], y = qb[f];
      x = g == 1 ? qa[4 + f] : x;  y = g == 1 ? qb[4 + f] : y;
      x = g == 2 ? qa[8 + f] : x;  y = g == 2 ? qb[8 + f] : y;
      x = g == 3 ? qa[12 + f] : x; y = g == 3 ? qb[12 + f] : y;
      v[0][f] = x; v[1][f] = y;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) { d_gb[t] = v[t][0]; d_gi[t] = v[t][1]; d_ih[t] = v[t][2]; d_ow[t] = v[t][3]; }
  };

  int tapoff[KS];  // byte offset of kernel column sx inside a line (parity-split slots for stride 2)
#pragma unroll
  for (int sx = 0; sx < KS; ++sx) tapoff[sx] = (p.stride == 1 ? sx : (sx & 1) * p.half + (sx >> 1)) * 32;

  StepInfo nxt;  // table info of the next step to be issued (s + NPF), fetched one iteration early
  auto step = [&](int s, Stg& slot_far, int& far_lo, int& far_hi, const Stg& slot_next, const int& next_lo, const int& next_hi) {
    int gi0[2], ih0[2], ow0[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      gi0[t] = d_gi[t];
      ih0[t] = d_gb[t] >= 0 ? d_ih[t] : -1000000;  // quads past the end: every row invalid
      ow0[t] = d_ow[t];
    }
    const bool ablate_stage = (p.ablate & 1) != 0;
    if (s + NPF < s_end && !ablate_stage) {
      issue(nxt, slot_far);  // into the slot step s no longer needs
      far_lo = nxt.row_lo;
      far_hi = nxt.row_hi;
    }
    if (s + NPF + 1 < s_end) nxt = fetch_info(s + NPF + 1);  // scalar loads; consumed next iteration
    if (s + 1 < s_end) load_desc(s + 1);
    if (wave_active && !(p.ablate & 2)) {
      const unsigned char* gy_buf = gy_lds + (s & 1) * NPL * GYB + lq * 32 + lp * 8;
      // Row base pointers of the two quads this lane reads (t = 0, 1).  Input rows outside the image (and
      // quads past the end of the pixel list) point at the all-zero line, so fragments need no masking.
      const unsigned char* bp[KS][2];
#pragma unroll
      for (int r = 0; r < KS; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ih = ih0[t] + r;
          const int line = ((unsigned)ih < (unsigned)p.H) ? ((gi0[t] + r) & line_mask) : p.nlines;
          bp[r][t] = x_plane + (line * p.wslots + ow0[t]) * 32;
        }
      // B fragments are fetched one kernel row ahead of the MFMAs that use them; the A fragments and row 0
      // go first.  sched_barriers keep the compiler from sinking the reads back down to their uses.
      i32x2 hb[2][NPL][KS * 2];
#pragma unroll
      for (int q = 0; q < NPL; ++q)
#pragma unroll
        for (int sx = 0; sx < KS; ++sx)
#pragma unroll
          for (int t = 0; t < 2; ++t) hb[0][q][sx * 2 + t] = tr_read(bp[0][t] + q * xpl_bytes + tapoff[sx]);
      i32x4 a[NPL][TMT];
#pragma unroll
      for (int q = 0; q < NPL; ++q)
#pragma unroll
        for (int m = 0; m < TMT; ++m) {
          const i32x2 lo = tr_read(gy_buf + q * GYB + m * GYM + (0 * 16 + g * 4) * 32);
          const i32x2 hi = tr_read(gy_buf + q * GYB + m * GYM + (1 * 16 + g * 4) * 32);
          a[q][m] = i32x4{lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
      for (int r = 0; r < KS; ++r) {
        if (r + 1 < KS && (!MASKED || ((tap_mask >> ((r + 1) * KS)) & ((1 << KS) - 1)))) {
#pragma unroll
          for (int q = 0; q < NPL; ++q)
#pragma unroll
            for (int sx = 0; sx < KS; ++sx)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                hb[(r + 1) & 1][q][sx * 2 + t] = tr_read(bp[r + 1][t] + q * xpl_bytes + tapoff[sx]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sx = 0; sx < KS; ++sx) {
          if (MASKED && !((tap_mask >> (r * KS + sx)) & 1)) continue;  // block-uniform
          bf16x8 b[NPL];
#pragma unroll
          for (int q = 0; q < NPL; ++q) {
            const i32x2 h0 = hb[r & 1][q][sx * 2 + 0], h1 = hb[r & 1][q][sx * 2 + 1];
            b[q] = __builtin_bit_cast(bf16x8, i32x4{h0[0], h0[1], h1[0], h1[1]});
          }
#pragma unroll
          for (int m = 0; m < TMT; ++m) {
            f32x4 d = acc[m][r * KS + sx];
            // (gy plane, x plane) pairs, small terms first
            if constexpr (NPL == 3) {
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1][m]), b[1], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0][m]), b[2], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[2][m]), b[0], d, 0, 0, 0);
            }
            if constexpr (NPL >= 2) {
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0][m]), b[1], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1][m]), b[0], d, 0, 0, 0);
            }
            acc[m][r * KS + sx] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0][m]), b[0], d, 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (s + 1 < s_end && !ablate_stage) {
      // loads newer than step s+1's: NPF-1 steps' worth while the queue is full, fewer (so wait for all) at the tail
      if (s + NPF < s_end) wait_vmcnt<(NPF - 1) * (1 + MAXP) * LPI>();
      else wait_vmcnt<0>();
      commit(s + 1, slot_next, next_lo, next_hi);
    }
    __syncthreads();
  };
  // Every client's x / gy block is a power-of-two number of bytes, so blocks that walk their K range in
  // lockstep from the same relative offset hit the same HBM channels at the same time (measured: ~2 TB/s no
  // matter how deep the prefetch).  Each block therefore starts at its own rotation point of its step range
  // and wraps around: segment 0 = [blk_begin + rot, blk_end), segment 1 = [blk_begin, blk_begin + rot).
  const int nblk_steps = blk_end - blk_begin;
  const int rot = (nblk_steps > 1 && !(p.ablate & 4)) ? (int)(((unsigned)c * 7919u + (unsigned)zsplit * 104729u) % (unsigned)nblk_steps) : 0;
  for (int seg = 0; seg < 2; ++seg) {
    s_begin = seg == 0 ? blk_begin + rot : blk_begin;
    s_end = seg == 0 ? blk_end : blk_begin + rot;
    if (s_begin >= s_end) continue;
    {
      // prologue: the whole row window of the first step, synchronously
      const __attribute__((address_space(4))) int* T = tbl + (size_t)s_begin * TBL;
      const int w_lo = T[34], w_hi = T[33];
      const int npix = (w_hi - w_lo + 1) * p.W;
      for (int e = pix; e < npix; e += 32) {
        if (x_lane) {
          const int er = p.w_shift >= 0 ? (e >> p.w_shift) : e / p.W;
          const size_t pixel = (size_t)(w_lo + er) * p.rmul + (size_t)(e - er * p.W) * p.cmul;
          const elem_t* src = xc + pixel * p.Cin + ci0 + chunk * 8;
          Item it;
#pragma unroll
          for (int l = 0; l < LPI; ++l)
            it.v[l] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(src) + 16 * l);
          const int row = w_lo + er, iw = e - er * p.W + p.pad;
          const int slot = p.stride == 1 ? iw : (iw & 1) * p.half + (iw >> 1);
          put(x_lds + (chunk >> 1) * plane_bytes + ((row & line_mask) * p.wslots + slot) * 32 + (chunk & 1) * 16,
              xpl_bytes, it);
        }
      }
      {  // gy tile of the first step
        const int qd = pix >> 2, q = pix & 3;
        const int gbase = T[qd * 4 + 0], ow0 = T[qd * 4 + 3];
        Item it;
#pragma unroll
        for (int l = 0; l < LPI; ++l) it.v[l] = u32x4{0, 0, 0, 0};
        if (gy_lane && gbase >= 0 && ow0 + q < p.Wo) {
          const elem_t* src = gc + (size_t)(gbase + q) * p.Cout + co0 + chunk * 8;
#pragma unroll
          for (int l = 0; l < LPI; ++l)
            it.v[l] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(src) + 16 * l);
        }
        if (chunk < TMT * 2)
          put(gy_lds + (s_begin & 1) * NPL * GYB + (chunk >> 1) * GYM + pix * 32 + (chunk & 1) * 16, GYB, it);
      }
    }
    load_desc(s_begin);
    __syncthreads();
    // steps already in flight when the loop starts: s_begin + 1 .. s_begin + NPF - 1
  #pragma unroll
    for (int k = 1; k < NPF; ++k)
      if (s_begin + k < s_end && !(p.ablate & 1)) {
        const StepInfo f = fetch_info(s_begin + k);
        issue(f, stg[k]);
        rlo[k] = f.row_lo;
        rhi[k] = f.row_hi;
      }
    if (s_begin + NPF < s_end) nxt = fetch_info(s_begin + NPF);
    // slot of step s is (s - s_begin) % NPF; the loop is unrolled NPF-fold so every slot index is static
    for (int s0 = s_begin; s0 < s_end; s0 += NPF) {
  #pragma unroll
      for (int k = 0; k < NPF; ++k)
        if (s0 + k < s_end)
          step(s0 + k, stg[k], rlo[k], rhi[k], stg[(k + 1) % NPF], rlo[(k + 1) % NPF], rhi[(k + 1) % NPF]);
    }
  }

  // ---- epilogue: D rows = co (lane>>4)*4+reg, cols = ci (lane&15) ------------------------------
  const bool to_scratch = p.split > 1;
  const float cf = (p.coef && !to_scratch) ? p.coef[c] : 1.f;
  float* row = to_scratch ? p.scratch + ((size_t)zsplit * p.Kc + c) * p.numel : p.U + (size_t)c * p.ld + p.off;
  const bool accum = p.accumulate && !to_scratch;
  if (p.ablate & 8) return;
  // For a fixed output channel a wave's 16 input channels x TAPS form one contiguous run of the weight
  // tensor (16*TAPS floats), but in registers a lane holds TAPS floats of it — writing from there means 4-byte
  // stores 36 bytes apart.  For 3x3 kernels the tile goes through LDS instead (the staging buffers are free
  // now) and comes back out lane-contiguous: every store instruction covers 256 consecutive bytes.  Measured
  // on the 512-channel layers: epilogue 0.8 -> ~0.35 ms of a 1.8 ms kernel.
  constexpr int SEG = 16 * TAPS, SEGP = SEG + 4;
  const bool full_tile = (p.Cs == p.Cin) && (ci0 + 64 <= p.Cin);  // block-uniform
  if (TAPS > 1 && full_tile) {
    float* tile = reinterpret_cast<float*>(smem) + wave * 16 * SEGP;  // [16 co rows][SEGP] per wave
#pragma unroll
    for (int m = 0; m < TMT; ++m) {
      __syncthreads();  // previous pass read out / K loop done with the staging buffers
      if constexpr (MASKED) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < TAPS; ++t)
            if ((tap_mask >> t) & 1) tile[(g * 4 + q) * SEGP + li * TAPS + tap_dst(t)] = cf * acc[m][t][q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < TAPS; ++t) tile[(g * 4 + q) * SEGP + li * TAPS + t] = cf * acc[m][t][q];
      }
      __syncthreads();
      float* dst0 = row + ((size_t)(co0 + m * 16) * p.Cin + ci0 + wave * 16) * TAPS;
#pragma unroll
      for (int j = 0; j < 16 * SEG / 64; ++j) {
        const int f = j * 64 + lane;
        const int r = f / SEG, col = f - r * SEG;
        float* d = dst0 + (size_t)r * p.Cin * TAPS + col;
        const float v = tile[r * SEGP + col];
        if constexpr (MASKED) {  // only the taps this class owns (the other classes write the rest)
          const int k = col % TAPS, kr = k / 3, kc = k - kr * 3;
          if ((kr == 1) == (rp == 0) && (kc == 1) == (cp == 0)) *d = accum ? *d + v : v;
        } else {
          *d = accum ? *d + v : v;
        }
      }
    }
    return;
  }
  if (wave_active) {
    const int ci = ci0 + wave * 16 + li;
#pragma unroll
    for (int m = 0; m < TMT; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = co0 + m * 16 + g * 4 + q;
        if (co < p.Cout && ci < p.Cs) {
          float* dst = row + ((size_t)co * p.Cs + ci) * TAPS;
          if constexpr (MASKED) {
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
              if ((tap_mask >> t) & 1) {
                const float v = cf * acc[m][t][q];
                float* d = dst + tap_dst(t);
                *d = accum ? *d + v : v;
              }
          } else {
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
              const float v = cf * acc[m][t][q];
              dst[t] = accum ? dst[t] + v : v;
            }
          }
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Eight-wave variant for the fp32 planes (NPL = 2, 3), 3x3 stride-1 problems with power-of-two output geometry and an
// arithmetic new-row schedule (every ResNet layer, and the parity classes of the strided ones).
//
// The plane images of a 64-channel tile fill LDS, so only one workgroup fits on a CU; with four waves that is one
// wave per SIMD and nothing overlaps a wave's LDS fragment reads, plane splitting and staging stores with MFMA work
// (PMC, profiles/r2_pmc_hot_kernels.txt: per K-step 3456 MFMA cycles, 2200 exposed read cycles, 1500 exposed
// staging cycles).  Here the same tile is worked by eight waves — wave w owns input-channel block w & 3 and half
// of the output-channel blocks (w >> 2) — so every SIMD holds two waves that hide each other's non-MFMA phases.
// Waves 0-3 stage gy, waves 4-7 stage x: one 16-byte item (LPI loads) per thread and step.
template <int TMT, int NPL, bool MASKED>
__global__ __launch_bounds__(512, 1) void conv_wgrad8_kernel(const WgradParams p) {
  static_assert(NPL >= 2 && TMT >= 2, "fp32 planes, at least two output-channel blocks");
  constexpr int KS = 3, TAPS = 9, NPF = 3, LPI = 2, MH = TMT / 2;
  using elem_t = float;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cib = wave & 3, half = wave >> 2;
  const int per_xcd = gridDim.x >> 3;
  const int lid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (lid >= p.total_blocks) return;
  const int n_ci_tiles = (p.Cin + 63) >> 6;
  const int tile = lid % p.ntiles, rest = lid / p.ntiles;
  const int c = rest % p.Kc, zsplit = rest / p.Kc;
  const int n_cls_tiles = MASKED ? 4 * n_ci_tiles : n_ci_tiles;
  const int co0 = (tile / n_cls_tiles) * (TMT * 16);
  const int cls = MASKED ? (tile % n_cls_tiles) / n_ci_tiles : 0;
  const int ci0 = ((tile % n_cls_tiles) % n_ci_tiles) * 64;
  const int rp = cls >> 1, cp = cls & 1;
  const int row_bits = rp ? 3 : 2, col_bits = cp ? 3 : 2;
  int tap_mask = 0x1ff;
  if constexpr (MASKED) {
    tap_mask = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if ((row_bits >> r) & 1) tap_mask |= col_bits << (r * 3);
  }
  auto tap_dst = [&](int t) {
    const int r = t / 3, sx = t - r * 3;
    return (rp ? 2 * r : 1) * 3 + (cp ? 2 * sx : 1);
  };
  constexpr int GYM = 1024 + 32;
  constexpr int GYB = TMT * GYM;
  unsigned char* gy_lds = smem;
  unsigned char* x_lds = smem + 2 * NPL * GYB;
  const int plane_bytes = (p.nlines + 1) * p.wslots * 32 + 32;
  const int xpl_bytes = 4 * plane_bytes;
  const int line_mask = p.nlines - 1;
  for (int i = tid * 16; i < NPL * xpl_bytes; i += 512 * 16) *reinterpret_cast<u32x4*>(x_lds + i) = u32x4{0, 0, 0, 0};

  const elem_t* xc = reinterpret_cast<const elem_t*>(p.x) + (size_t)c * p.x_client_stride +
                     (MASKED ? (size_t)(rp * (p.rmul >> 1) + cp) * p.Cin : (size_t)0);
  const elem_t* gc = reinterpret_cast<const elem_t*>(p.gy) + (size_t)c * p.B * p.Ho * p.Wo * p.Cout;

  // staging roles: waves 0-3 gy, waves 4-7 x; 8 lanes x 16 B (x LPI) cover one pixel's 64-channel tile
  const bool is_x = tid >= 256;  // wave-uniform
  const int chunk = tid & 7, pix = (tid >> 3) & 31;
  const bool gy_lane = chunk < TMT * 2 && (co0 + chunk * 8) < p.Cout;
  const bool x_lane = (ci0 + chunk * 8) < p.Cin;
  struct Item { u32x4 v[LPI]; };
  auto fetch = [&](const elem_t* src) {
    Item it;
#pragma unroll
    for (int l = 0; l < LPI; ++l) it.v[l] = gload16_untracked(reinterpret_cast<const unsigned char*>(src) + 16 * l);
    return it;
  };
  auto put = [&](unsigned char* dst, int plane_stride, const Item& it) {
    u32x4 pl[NPL];
    split8<NPL>(it.v[0], it.v[1], pl);
#pragma unroll
    for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x4*>(dst + q * plane_stride) = pl[q];
  };
  const __attribute__((address_space(4))) int* tbl = (const __attribute__((address_space(4))) int*)(p.table);
  const elem_t* zeros16 = reinterpret_cast<const elem_t*>(p.table + (size_t)p.nsteps * TBL);
  // new input rows of step s > its block's first: (s R + off, (s + 1) R + off], clipped to the client's last row
  auto row_hi = [&](int s) { const int v = (s + 1) * p.rows_r + p.rows_off; return v < p.rows_last ? v : p.rows_last; };
  auto row_lo = [&](int s) { const int v = s * p.rows_r + p.rows_off; return (v < p.rows_last ? v : p.rows_last) + 1; };
  auto issue = [&](int s) {
    const elem_t* a = zeros16;
    if (!is_x) {
      const int gpix = s * 32 + pix;
      if (gy_lane && gpix < p.nquads * 4) a = gc + (size_t)gpix * p.Cout + co0 + chunk * 8;
    } else {
      const int lo = row_lo(s), npix = (row_hi(s) - lo + 1) * p.W;
      if (x_lane && pix < npix) {
        const int er = pix >> p.w_shift;
        const size_t pixel = (size_t)(lo + er) * p.rmul + (size_t)(pix - er * p.W) * p.cmul;
        a = xc + pixel * p.Cin + ci0 + chunk * 8;
      }
    }
    return fetch(a);
  };
  auto commit = [&](int s, const Item& it) {
    if (!is_x) {
      if (chunk < TMT * 2)
        put(gy_lds + (s & 1) * NPL * GYB + (chunk >> 1) * GYM + pix * 32 + (chunk & 1) * 16, GYB, it);
    } else {
      const int lo = row_lo(s), npix = (row_hi(s) - lo + 1) * p.W;
      if (pix < npix) {
        const int er = pix >> p.w_shift;
        const int row = lo + er, iw = pix - er * p.W + p.pad;
        put(x_lds + (chunk >> 1) * plane_bytes + ((row & line_mask) * p.wslots + iw) * 32 + (chunk & 1) * 16,
            xpl_bytes, it);
      }
    }
  };

  f32x4 acc[MH][TAPS];
#pragma unroll
  for (int m = 0; m < MH; ++m)
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool wave_active = (ci0 + cib * 16) < p.Cin;
  const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const unsigned char* x_plane = x_lds + cib * plane_bytes + lq * 32 + lp * 8;
  const int per = (p.nsteps + p.split - 1) / p.split;
  const int blk_begin = zsplit * per;
  const int blk_end = blk_begin + per < p.nsteps ? blk_begin + per : p.nsteps;
  int s_begin = 0, s_end = 0;
  __syncthreads();  // zero fill done

  Item stg[NPF];
  // `mid(r)` runs after the MFMAs of kernel row r have been issued: the staging commit of the next step goes there
  // (row 0 for one wave group, row 1 for the other), so on every SIMD one wave converts / stores while the other
  // keeps the matrix pipe busy — at the end of the step both would do it at the same time with the pipe idle.
  auto compute = [&](int s, auto&& mid) {
    if (!wave_active) { mid(0); mid(1); return; }
    {
      int gi0[2], ih0[2], ow0[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int Q = s * 8 + 4 * t + g;
        const int qx = Q & ((1 << p.lg_qpr) - 1);
        const int oh = (Q >> p.lg_qpr) & ((1 << p.lg_ho) - 1);
        const int b = Q >> (p.lg_qpr + p.lg_ho);
        ow0[t] = qx * 4;
        ih0[t] = Q < p.nquads ? oh - p.pad : -1000000;
        gi0[t] = b * p.H + oh - p.pad;
      }
      const unsigned char* gy_buf = gy_lds + (s & 1) * NPL * GYB + lq * 32 + lp * 8;
      const unsigned char* bp[KS][2];
#pragma unroll
      for (int r = 0; r < KS; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ih = ih0[t] + r;
          const int line = ((unsigned)ih < (unsigned)p.H) ? ((gi0[t] + r) & line_mask) : p.nlines;
          bp[r][t] = x_plane + (line * p.wslots + ow0[t]) * 32;
        }
      // x fragments of one kernel row.  MASKED: one read per (tap column, quad).  Otherwise the three tap columns are
      // windows of the same 8 slots: two reads per quad (slots ow0.. and ow0 + 4..) and the middle window by two
      // v_alignbit — a third less LDS read traffic for the same operands (WIN = reads per quad and row).
      constexpr int WIN = MASKED ? KS : 2;
      i32x2 hb[2][NPL][WIN * 2];
#pragma unroll
      for (int q = 0; q < NPL; ++q)
#pragma unroll
        for (int sx = 0; sx < WIN; ++sx)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            hb[0][q][sx * 2 + t] = tr_read(bp[0][t] + q * xpl_bytes + sx * (MASKED ? 32 : 128));
      i32x4 a[NPL][MH];
#pragma unroll
      for (int q = 0; q < NPL; ++q)
#pragma unroll
        for (int m = 0; m < MH; ++m) {
          const unsigned char* gb = gy_buf + q * GYB + (half * MH + m) * GYM;
          const i32x2 lo = tr_read(gb + (0 * 16 + g * 4) * 32), hi = tr_read(gb + (1 * 16 + g * 4) * 32);
          a[q][m] = i32x4{lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
      for (int r = 0; r < KS; ++r) {
        if (r + 1 < KS && (!MASKED || ((tap_mask >> ((r + 1) * KS)) & 7))) {
#pragma unroll
          for (int q = 0; q < NPL; ++q)
#pragma unroll
            for (int sx = 0; sx < WIN; ++sx)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                hb[(r + 1) & 1][q][sx * 2 + t] =
                    tr_read(bp[r + 1][t] + q * xpl_bytes + sx * (MASKED ? 32 : 128));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sx = 0; sx < KS; ++sx) {
          if (MASKED && !((tap_mask >> (r * KS + sx)) & 1)) continue;  // block-uniform
          bf16x8 b[NPL];
#pragma unroll
          for (int q = 0; q < NPL; ++q) {
            i32x2 h0, h1;
            if constexpr (MASKED) {
              h0 = hb[r & 1][q][sx * 2 + 0]; h1 = hb[r & 1][q][sx * 2 + 1];
            } else {
              // quad t: slots ow0 .. ow0 + 3 in A, ow0 + 4 .. ow0 + 7 in B; tap column sx = slots ow0 + sx .. + 3
              auto window = [&](const i32x2 A, const i32x2 B) {
                if (sx == 0) return A;
                if (sx == 2) return i32x2{A[1], B[0]};
                return i32x2{(int)__builtin_amdgcn_alignbit((unsigned)A[1], (unsigned)A[0], 16),
                             (int)__builtin_amdgcn_alignbit((unsigned)B[0], (unsigned)A[1], 16)};
              };
              h0 = window(hb[r & 1][q][0], hb[r & 1][q][2]);
              h1 = window(hb[r & 1][q][1], hb[r & 1][q][3]);
            }
            b[q] = __builtin_bit_cast(bf16x8, i32x4{h0[0], h0[1], h1[0], h1[1]});
          }
#pragma unroll
          for (int m = 0; m < MH; ++m) {
            f32x4 d = acc[m][r * KS + sx];
            if constexpr (NPL == 3) {
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1][m]), b[1], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0][m]), b[2], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[2][m]), b[0], d, 0, 0, 0);
            }
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0][m]), b[1], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1][m]), b[0], d, 0, 0, 0);
            acc[m][r * KS + sx] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0][m]), b[0], d, 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (r < 2) mid(r);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  const int grp = __builtin_amdgcn_readfirstlane(half);
  const bool ab_stage = (p.ablate & 1) != 0, ab_mfma = (p.ablate & 2) != 0;  // diagnostics (BLADES_WGRAD_ABLATE)
  auto step = [&](int s, Item& slot_far, const Item& slot_next) {
    if (s + NPF < s_end && !ab_stage) slot_far = issue(s + NPF);
    auto stage_next = [&](int r) {
      if (r != grp || s + 1 >= s_end || ab_stage) return;
      if (!(p.ablate & 4)) {
        if (s + NPF < s_end) wait_vmcnt<(NPF - 1) * LPI>();
        else wait_vmcnt<0>();
      }
      if (!(p.ablate & 8)) commit(s + 1, slot_next);
    };
    if (!ab_mfma) compute(s, stage_next);
    else { stage_next(0); stage_next(1); }
    __syncthreads();
  };

  const int nblk_steps = blk_end - blk_begin;
  const int rot = nblk_steps > 1 ? (int)(((unsigned)c * 7919u + (unsigned)zsplit * 104729u) % (unsigned)nblk_steps) : 0;
  for (int seg = 0; seg < 2; ++seg) {
    s_begin = seg == 0 ? blk_begin + rot : blk_begin;
    s_end = seg == 0 ? blk_end : blk_begin + rot;
    if (s_begin >= s_end) continue;
    {
      // prologue: the whole row window of the first step and its gy tile, synchronously (all 512 threads load x)
      const __attribute__((address_space(4))) int* T = tbl + (size_t)s_begin * TBL;
      const int w_lo = T[34], w_hi = T[33];
      const int npix = (w_hi - w_lo + 1) * p.W;
      for (int e = tid >> 3; e < npix; e += 64) {
        if (x_lane) {
          const int er = e >> p.w_shift;
          const size_t pixel = (size_t)(w_lo + er) * p.rmul + (size_t)(e - er * p.W) * p.cmul;
          const elem_t* src = xc + pixel * p.Cin + ci0 + chunk * 8;
          Item it;
#pragma unroll
          for (int l = 0; l < LPI; ++l)
            it.v[l] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(src) + 16 * l);
          const int row = w_lo + er, iw = e - er * p.W + p.pad;
          put(x_lds + (chunk >> 1) * plane_bytes + ((row & line_mask) * p.wslots + iw) * 32 + (chunk & 1) * 16,
              xpl_bytes, it);
        }
      }
      if (!is_x) {
        Item it;
#pragma unroll
        for (int l = 0; l < LPI; ++l) it.v[l] = u32x4{0, 0, 0, 0};
        const int gpix = s_begin * 32 + pix;
        if (gy_lane && gpix < p.nquads * 4) {
          const elem_t* src = gc + (size_t)gpix * p.Cout + co0 + chunk * 8;
#pragma unroll
          for (int l = 0; l < LPI; ++l)
            it.v[l] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(src) + 16 * l);
        }
        if (chunk < TMT * 2)
          put(gy_lds + (s_begin & 1) * NPL * GYB + (chunk >> 1) * GYM + pix * 32 + (chunk & 1) * 16, GYB, it);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 1; k < NPF; ++k)
      if (s_begin + k < s_end && !ab_stage) stg[k] = issue(s_begin + k);
    for (int s0 = s_begin; s0 < s_end; s0 += NPF) {
#pragma unroll
      for (int k = 0; k < NPF; ++k)
        if (s0 + k < s_end) step(s0 + k, stg[k], stg[(k + 1) % NPF]);
    }
  }

  // ---- epilogue: each wave's [MH x 16 co][16 ci x 9 taps] through LDS, lane-contiguous stores ------------------
  const bool to_scratch = p.split > 1;
  const float cf = (p.coef && !to_scratch) ? p.coef[c] : 1.f;
  float* row = to_scratch ? p.scratch + ((size_t)zsplit * p.Kc + c) * p.numel : p.U + (size_t)c * p.ld + p.off;
  const bool accum = p.accumulate && !to_scratch;
  constexpr int SEG = 16 * TAPS, SEGP = SEG + 4;
  const bool full_tile = (p.Cs == p.Cin) && (ci0 + 64 <= p.Cin);  // block-uniform
  if (full_tile) {
    float* tile_l = reinterpret_cast<float*>(smem) + wave * 16 * SEGP;
#pragma unroll
    for (int m = 0; m < MH; ++m) {
      __syncthreads();
      if constexpr (MASKED) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < TAPS; ++t)
            if ((tap_mask >> t) & 1) tile_l[(g * 4 + q) * SEGP + li * TAPS + tap_dst(t)] = cf * acc[m][t][q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < TAPS; ++t) tile_l[(g * 4 + q) * SEGP + li * TAPS + t] = cf * acc[m][t][q];
      }
      __syncthreads();
      float* dst0 = row + ((size_t)(co0 + (half * MH + m) * 16) * p.Cin + ci0 + cib * 16) * TAPS;
#pragma unroll
      for (int j = 0; j < 16 * SEG / 64; ++j) {
        const int f = j * 64 + lane;
        const int r = f / SEG, col = f - r * SEG;
        float* d = dst0 + (size_t)r * p.Cin * TAPS + col;
        const float v = tile_l[r * SEGP + col];
        if constexpr (MASKED) {
          const int k = col % TAPS, kr = k / 3, kc = k - kr * 3;
          if ((kr == 1) == (rp == 0) && (kc == 1) == (cp == 0)) *d = accum ? *d + v : v;
        } else {
          *d = accum ? *d + v : v;
        }
      }
    }
    return;
  }
  if (wave_active) {
    const int ci = ci0 + cib * 16 + li;
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = co0 + (half * MH + m) * 16 + g * 4 + q;
        if (co < p.Cout && ci < p.Cs) {
          float* dst = row + ((size_t)co * p.Cs + ci) * TAPS;
#pragma unroll
          for (int t = 0; t < TAPS; ++t)
            if (!MASKED || ((tap_mask >> t) & 1)) {
              const float v = cf * acc[m][t][q];
              float* d = dst + (MASKED ? tap_dst(t) : t);
              *d = accum ? *d + v : v;
            }
        }
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The parity classes of a stride-2 3x3 layer (the MASKED problems above) on 128 output channels per workgroup.
// A class owns 1, 2 or 4 of the nine taps, so per staged K-step it has a quarter of a stride-1 layer's matrix work
// and the steps are staging- and barrier-bound.  Doubling the output-channel tile doubles the MFMAs per step at the
// same x staging and halves the number of workgroup-steps; the accumulators still fit because only the 2 x 2 taps a
// class can own are kept (view rows oh - 1, oh  x  view columns ow - 1, ow).
// Staging: every thread carries one 32-byte gy item (32 pixels x 16 eight-channel groups) and one 16-byte x item
// (32 pixels x 16 four-channel groups): three loads per thread and step.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int NPL>
__device__ __forceinline__ void split4(const u32x4 v, u32x2 (&pl)[NPL]) {
  typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  float r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = __uint_as_float(v[j]);
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v{r[2 * j], r[2 * j + 1]}), bf16x2v));
      pl[q][j] = u;
      if (q + 1 < NPL) {
        r[2 * j] -= __uint_as_float(u << 16);
        r[2 * j + 1] -= __uint_as_float(u & 0xffff0000u);
      }
    }
  }
}

template <int NPL>
__global__ __launch_bounds__(512, 1) void conv_wgrad8m_kernel(const WgradParams p) {
  static_assert(NPL >= 2, "fp32 planes");
  constexpr int TMT = 8, KV = 2, TV = KV * KV, TAPS = 9, NPF = 3, LPS = 3, MH = TMT / 2;
  using elem_t = float;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cib = wave & 3, half = wave >> 2;
  const int per_xcd = gridDim.x >> 3;
  const int lid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (lid >= p.total_blocks) return;
  const int n_ci_tiles = (p.Cin + 63) >> 6;
  const int tile = lid % p.ntiles, rest = lid / p.ntiles;
  const int c = rest % p.Kc, zsplit = rest / p.Kc;
  const int n_cls_tiles = 4 * n_ci_tiles;
  const int co0 = (tile / n_cls_tiles) * (TMT * 16);
  const int cls = (tile % n_cls_tiles) / n_ci_tiles;
  const int ci0 = ((tile % n_cls_tiles) % n_ci_tiles) * 64;
  const int rp = cls >> 1, cp = cls & 1;
  // view tap (r, sx), r, sx in {0, 1}: input pixel (2 (oh - 1 + r) + rp, 2 (ow - 1 + sx) + cp) = filter tap
  // (2 r + rp - 1, 2 sx + cp - 1): row 0 exists only in the odd-row classes, column 0 only in the odd-column ones
  const int tap_mask = (rp ? 3 : 0) | (3 << 2);               // bit r * 2 + sx, before the column mask
  const int col_mask = cp ? 0xf : 0xa;                        // sx = 0 only with cp
  const int live = tap_mask & col_mask;
  auto tap_dst = [&](int t) {
    const int r = t >> 1, sx = t & 1;
    return (rp ? 2 * r : 1) * 3 + (cp ? 2 * sx : 1);
  };
  constexpr int GYM = 1024 + 32;
  constexpr int GYB = TMT * GYM;
  unsigned char* gy_lds = smem;
  unsigned char* x_lds = smem + 2 * NPL * GYB;
  const int plane_bytes = (p.nlines + 1) * p.wslots * 32 + 32;
  const int xpl_bytes = 4 * plane_bytes;
  const int line_mask = p.nlines - 1;
  for (int i = tid * 16; i < NPL * xpl_bytes; i += 512 * 16) *reinterpret_cast<u32x4*>(x_lds + i) = u32x4{0, 0, 0, 0};

  const elem_t* xc = reinterpret_cast<const elem_t*>(p.x) + (size_t)c * p.x_client_stride +
                     (size_t)(rp * (p.rmul >> 1) + cp) * p.Cin;
  const elem_t* gc = reinterpret_cast<const elem_t*>(p.gy) + (size_t)c * p.B * p.Ho * p.Wo * p.Cout;

  // staging roles of the K loop: thread -> (pixel tid >> 4, group tid & 15) of both operands
  const int grp16 = tid & 15, pix = tid >> 4;
  const bool gy_lane = (co0 + grp16 * 8) < p.Cout;
  const bool x_lane = (ci0 + grp16 * 4) < p.Cin;
  struct Item { u32x4 g[2]; u32x4 x; };
  const __attribute__((address_space(4))) int* tbl = (const __attribute__((address_space(4))) int*)(p.table);
  const elem_t* zeros16 = reinterpret_cast<const elem_t*>(p.table + (size_t)p.nsteps * TBL);
  auto row_hi = [&](int s) { const int v = (s + 1) * p.rows_r + p.rows_off; return v < p.rows_last ? v : p.rows_last; };
  auto row_lo = [&](int s) { const int v = s * p.rows_r + p.rows_off; return (v < p.rows_last ? v : p.rows_last) + 1; };
  auto issue = [&](int s) {
    const elem_t* ag = zeros16;
    const int gpix = s * 32 + pix;
    if (gy_lane && gpix < p.nquads * 4) ag = gc + (size_t)gpix * p.Cout + co0 + grp16 * 8;
    const elem_t* ax = zeros16;
    const int lo = row_lo(s), npix = (row_hi(s) - lo + 1) * p.W;
    if (x_lane && pix < npix) {
      const int er = pix >> p.w_shift;
      const size_t pixel = (size_t)(lo + er) * p.rmul + (size_t)(pix - er * p.W) * p.cmul;
      ax = xc + pixel * p.Cin + ci0 + grp16 * 4;
    }
    Item it;
    it.g[0] = gload16_untracked(ag);
    it.g[1] = gload16_untracked(reinterpret_cast<const unsigned char*>(ag) + 16);
    it.x = gload16_untracked(ax);
    return it;
  };
  auto put_gy = [&](int s, const u32x4 lo, const u32x4 hi) {
    u32x4 pl[NPL];
    split8<NPL>(lo, hi, pl);
    unsigned char* dst = gy_lds + (s & 1) * NPL * GYB + (grp16 >> 1) * GYM + pix * 32 + (grp16 & 1) * 16;
#pragma unroll
    for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x4*>(dst + q * GYB) = pl[q];
  };
  auto commit = [&](int s, const Item& it) {
    put_gy(s, it.g[0], it.g[1]);
    const int lo = row_lo(s), npix = (row_hi(s) - lo + 1) * p.W;
    if (pix < npix) {
      const int er = pix >> p.w_shift;
      const int row = lo + er, iw = pix - er * p.W + p.pad;
      u32x2 pl[NPL];
      split4<NPL>(it.x, pl);
      unsigned char* dst = x_lds + (grp16 >> 2) * plane_bytes + ((row & line_mask) * p.wslots + iw) * 32 + (grp16 & 3) * 8;
#pragma unroll
      for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2*>(dst + q * xpl_bytes) = pl[q];
    }
  };

  f32x4 acc[MH][TV];
#pragma unroll
  for (int m = 0; m < MH; ++m)
#pragma unroll
    for (int t = 0; t < TV; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool wave_active = (ci0 + cib * 16) < p.Cin;
  const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const unsigned char* x_plane = x_lds + cib * plane_bytes + lq * 32 + lp * 8;
  const int per = (p.nsteps + p.split - 1) / p.split;
  const int blk_begin = zsplit * per;
  const int blk_end = blk_begin + per < p.nsteps ? blk_begin + per : p.nsteps;
  int s_begin = 0, s_end = 0;
  __syncthreads();  // zero fill done

  Item stg[NPF];
  auto compute = [&](int s, auto&& mid) {
    if (!wave_active) { mid(0); mid(1); return; }
    int gi0[2], ih0[2], ow0[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int Q = s * 8 + 4 * t + g;
      const int qx = Q & ((1 << p.lg_qpr) - 1);
      const int oh = (Q >> p.lg_qpr) & ((1 << p.lg_ho) - 1);
      const int b = Q >> (p.lg_qpr + p.lg_ho);
      ow0[t] = qx * 4;
      ih0[t] = Q < p.nquads ? oh - p.pad : -1000000;
      gi0[t] = b * p.H + oh - p.pad;
    }
    const unsigned char* gy_buf = gy_lds + (s & 1) * NPL * GYB + lq * 32 + lp * 8;
    const unsigned char* bp[KV][2];
#pragma unroll
    for (int r = 0; r < KV; ++r)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ih = ih0[t] + r;
        const int line = ((unsigned)ih < (unsigned)p.H) ? ((gi0[t] + r) & line_mask) : p.nlines;
        bp[r][t] = x_plane + (line * p.wslots + ow0[t]) * 32;
      }
    i32x4 a[NPL][MH];
#pragma unroll
    for (int q = 0; q < NPL; ++q)
#pragma unroll
      for (int m = 0; m < MH; ++m) {
        const unsigned char* gb = gy_buf + q * GYB + (half * MH + m) * GYM;
        const i32x2 lo = tr_read(gb + (0 * 16 + g * 4) * 32), hi = tr_read(gb + (1 * 16 + g * 4) * 32);
        a[q][m] = i32x4{lo[0], lo[1], hi[0], hi[1]};
      }
#pragma unroll
    for (int r = 0; r < KV; ++r) {
      if ((live >> (r * KV)) & 3) {  // block-uniform
        i32x2 hb[NPL][KV * 2];
#pragma unroll
        for (int sx = 0; sx < KV; ++sx) {
          if (!((live >> (r * KV + sx)) & 1)) continue;  // block-uniform: the even-column classes own one column
#pragma unroll
          for (int q = 0; q < NPL; ++q)
#pragma unroll
            for (int t = 0; t < 2; ++t) hb[q][sx * 2 + t] = tr_read(bp[r][t] + q * xpl_bytes + sx * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sx = 0; sx < KV; ++sx) {
          if (!((live >> (r * KV + sx)) & 1)) continue;  // block-uniform
          bf16x8 b[NPL];
#pragma unroll
          for (int q = 0; q < NPL; ++q) {
            const i32x2 h0 = hb[q][sx * 2 + 0], h1 = hb[q][sx * 2 + 1];
            b[q] = __builtin_bit_cast(bf16x8, i32x4{h0[0], h0[1], h1[0], h1[1]});
          }
#pragma unroll
          for (int m = 0; m < MH; ++m) {
            f32x4 d = acc[m][r * KV + sx];
            if constexpr (NPL == 3) {
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1][m]), b[1], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0][m]), b[2], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[2][m]), b[0], d, 0, 0, 0);
            }
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0][m]), b[1], d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1][m]), b[0], d, 0, 0, 0);
            acc[m][r * KV + sx] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0][m]), b[0], d, 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      mid(r);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const int grp = __builtin_amdgcn_readfirstlane(half);
  const bool ab_stage = (p.ablate & 1) != 0, ab_mfma = (p.ablate & 2) != 0;  // diagnostics (BLADES_WGRAD_ABLATE)
  auto step = [&](int s, Item& slot_far, const Item& slot_next) {
    // Unconditional: every thread issues exactly LPS loads per step, so "step s + 1 has landed" is always the same
    // count, and the staging registers carry no conditional assignment (the compiler resolves those with register
    // copies, and a copy of a register whose load is still in flight reads garbage).  Steps past the range load
    // valid memory (or the zero line) that is never committed.
    slot_far = issue(s + NPF);
    // the staging commit of the next step: one wave group after the first view row, the other after the second
    auto stage_next = [&](int r) {
      if (r != grp) return;
      wait_vmcnt<(NPF - 1) * LPS>();
      if (s + 1 < s_end && !ab_stage) commit(s + 1, slot_next);
    };
    if (!ab_mfma) compute(s, stage_next);
    else { stage_next(0); stage_next(1); }
    __syncthreads();
  };

  const int nblk_steps = blk_end - blk_begin;
  const int rot = nblk_steps > 1 ? (int)(((unsigned)c * 7919u + (unsigned)zsplit * 104729u) % (unsigned)nblk_steps) : 0;
  for (int seg = 0; seg < 2; ++seg) {
    s_begin = seg == 0 ? blk_begin + rot : blk_begin;
    s_end = seg == 0 ? blk_end : blk_begin + rot;
    if (s_begin >= s_end) continue;
    {
      // prologue: the whole row window of the first step and its gy tile, synchronously
      const __attribute__((address_space(4))) int* T = tbl + (size_t)s_begin * TBL;
      const int w_lo = T[34], w_hi = T[33];
      const int npix = (w_hi - w_lo + 1) * p.W;
      for (int e = pix; e < npix; e += 32) {
        if (x_lane) {
          const int er = e >> p.w_shift;
          const size_t pixel = (size_t)(w_lo + er) * p.rmul + (size_t)(e - er * p.W) * p.cmul;
          const u32x4 v = *reinterpret_cast<const u32x4*>(xc + pixel * p.Cin + ci0 + grp16 * 4);
          const int row = w_lo + er, iw = e - er * p.W + p.pad;
          u32x2 pl[NPL];
          split4<NPL>(v, pl);
          unsigned char* dst = x_lds + (grp16 >> 2) * plane_bytes + ((row & line_mask) * p.wslots + iw) * 32 + (grp16 & 3) * 8;
#pragma unroll
          for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2*>(dst + q * xpl_bytes) = pl[q];
        }
      }
      u32x4 g0 = u32x4{0, 0, 0, 0}, g1 = u32x4{0, 0, 0, 0};
      const int gpix = s_begin * 32 + pix;
      if (gy_lane && gpix < p.nquads * 4) {
        const elem_t* src = gc + (size_t)gpix * p.Cout + co0 + grp16 * 8;
        g0 = *reinterpret_cast<const u32x4*>(src);
        g1 = *reinterpret_cast<const u32x4*>(src + 4);
      }
      put_gy(s_begin, g0, g1);
    }
    __syncthreads();
#pragma unroll
    for (int k = 1; k < NPF; ++k) stg[k] = issue(s_begin + k);
    for (int s0 = s_begin; s0 < s_end; s0 += NPF) {
#pragma unroll
      for (int k = 0; k < NPF; ++k)
        if (s0 + k < s_end) step(s0 + k, stg[k], stg[(k + 1) % NPF]);
    }
    wait_vmcnt<0>();  // the look-ahead loads past the range: their registers are about to be reused
  }

  // ---- epilogue: each wave's [MH x 16 co][16 ci x 9 taps] through LDS, lane-contiguous stores of the owned taps ----
  const bool to_scratch = p.split > 1;
  const float cf = (p.coef && !to_scratch) ? p.coef[c] : 1.f;
  float* row = to_scratch ? p.scratch + ((size_t)zsplit * p.Kc + c) * p.numel : p.U + (size_t)c * p.ld + p.off;
  const bool accum = p.accumulate && !to_scratch;
  constexpr int SEG = 16 * TAPS, SEGP = SEG + 4;
  const bool full_tile = (p.Cs == p.Cin) && (ci0 + 64 <= p.Cin);  // block-uniform
  if (full_tile) {
    float* tile_l = reinterpret_cast<float*>(smem) + wave * 16 * SEGP;
#pragma unroll
    for (int m = 0; m < MH; ++m) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < TV; ++t)
          if ((live >> t) & 1) tile_l[(g * 4 + q) * SEGP + li * TAPS + tap_dst(t)] = cf * acc[m][t][q];
      __syncthreads();
      float* dst0 = row + ((size_t)(co0 + (half * MH + m) * 16) * p.Cin + ci0 + cib * 16) * TAPS;
#pragma unroll
      for (int j = 0; j < 16 * SEG / 64; ++j) {
        const int f = j * 64 + lane;
        const int r = f / SEG, col = f - r * SEG;
        float* d = dst0 + (size_t)r * p.Cin * TAPS + col;
        const float v = tile_l[r * SEGP + col];
        const int k = col % TAPS, kr = k / 3, kc = k - kr * 3;
        if ((kr == 1) == (rp == 0) && (kc == 1) == (cp == 0)) *d = accum ? *d + v : v;
      }
    }
    return;
  }
  if (wave_active) {
    const int ci = ci0 + cib * 16 + li;
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = co0 + (half * MH + m) * 16 + g * 4 + q;
        if (co < p.Cout && ci < p.Cs) {
          float* dst = row + ((size_t)co * p.Cs + ci) * TAPS;
#pragma unroll
          for (int t = 0; t < TV; ++t)
            if ((live >> t) & 1) {
              const float v = cf * acc[m][t][q];
              float* d = dst + tap_dst(t);
              *d = accum ? *d + v : v;
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

template <int TMT, int KS, int MAXP, int NPF, bool FAST, int NPL, bool MASKED = false>
int launch(const WgradParams& p, int Kc, hipStream_t st) {
  size_t lds = NPL * (2 * TMT * (1024 + 32) + 4 * ((size_t)(p.nlines + 1) * p.wslots * 32 + 32));
  const size_t lds_epilogue = 4 * 16 * (16 * KS * KS + 4) * sizeof(float);  // LDS-transposed write-out
  if (KS > 1 && lds < lds_epilogue) lds = lds_epilogue;
  if (lds > 160 * 1024) return 3;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<TMT, KS, MAXP, NPF, FAST, NPL, MASKED>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int n_ci_tiles = (p.Cin + 63) / 64, n_co_tiles = p.Cout / (TMT * 16);
  WgradParams q = p;
  q.ntiles = n_co_tiles * n_ci_tiles * (MASKED ? 4 : 1);
  q.total_blocks = q.ntiles * Kc * p.split;
  const int grid = (q.total_blocks + 7) / 8 * 8;
  hipLaunchKernelGGL((conv_wgrad_kernel<TMT, KS, MAXP, NPF, FAST, NPL, MASKED>), dim3(grid), dim3(256), lds, st, q);
  if (p.split > 1) {
    int gx = cdiv(p.numel, 256 * 4);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, Kc), dim3(256), 0, st, p.scratch, p.split, Kc, p.numel, p.coef,
                       p.U, p.ld, p.off, p.accumulate);
  }
  return 0;
}

template <int TMT, int NPL, bool MASKED>
int launch8(const WgradParams& p, int Kc, hipStream_t st) {
  size_t lds = NPL * (2 * TMT * (1024 + 32) + 4 * ((size_t)(p.nlines + 1) * p.wslots * 32 + 32));
  const size_t lds_epilogue = 8 * 16 * (16 * 9 + 4) * sizeof(float);
  if (lds < lds_epilogue) lds = lds_epilogue;
  if (lds > 160 * 1024) return 3;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad8_kernel<TMT, NPL, MASKED>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int n_ci_tiles = (p.Cin + 63) / 64, n_co_tiles = p.Cout / (TMT * 16);
  WgradParams q = p;
  q.ntiles = n_co_tiles * n_ci_tiles * (MASKED ? 4 : 1);
  q.total_blocks = q.ntiles * Kc * p.split;
  const int grid = (q.total_blocks + 7) / 8 * 8;
  hipLaunchKernelGGL((conv_wgrad8_kernel<TMT, NPL, MASKED>), dim3(grid), dim3(512), lds, st, q);
  if (p.split > 1) {
    int gx = cdiv(p.numel, 256 * 4);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, Kc), dim3(256), 0, st, p.scratch, p.split, Kc, p.numel, p.coef,
                       p.U, p.ld, p.off, p.accumulate);
  }
  return 0;
}

template <int NPL>
int launch8m(const WgradParams& p, int Kc, hipStream_t st) {
  constexpr int TMT = 8;
  size_t lds = NPL * (2 * TMT * (1024 + 32) + 4 * ((size_t)(p.nlines + 1) * p.wslots * 32 + 32));
  const size_t lds_epilogue = 8 * 16 * (16 * 9 + 4) * sizeof(float);
  if (lds < lds_epilogue) lds = lds_epilogue;
  if (lds > 160 * 1024) return 3;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad8m_kernel<NPL>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int n_ci_tiles = (p.Cin + 63) / 64, n_co_tiles = p.Cout / (TMT * 16);
  WgradParams q = p;
  q.ntiles = n_co_tiles * n_ci_tiles * 4;
  q.total_blocks = q.ntiles * Kc * p.split;
  const int grid = (q.total_blocks + 7) / 8 * 8;
  hipLaunchKernelGGL((conv_wgrad8m_kernel<NPL>), dim3(grid), dim3(512), lds, st, q);
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
                                 int tmt, int split, float* scratch, int rmul, int cmul, int passes, int rows_r, int rows_off,
                                 long x_client_stride, int cin_store, int npl, int s2_classes, hipStream_t st) {
  // s2_classes: the call is a stride-2 3x3 pad-1 layer posed as four stride-1 problems in one grid, one per parity
  // class of the input pixels (H, W, Ho, Wo are then all the *output* size, rmul = 2 * full width, cmul = 2); every
  // class computes and stores only the taps it owns (4 + 2 + 2 + 1 = 9), so the layer costs its own MFMA work, and
  // the classes of one client run side by side on one XCD and share its gy rows through L2.  Used for the fp32
  // planes, whose strided input window does not fit in LDS.
  if (npl < 1 || npl > 3) return 8;
  const bool masked = s2_classes != 0;
  if (masked && (ks != 3 || stride != 1 || pad != 1 || passes > 1 || npl == 1 || cmul != 2)) return 9;
  if (!(ks == 1 || ks == 3) || !(stride == 1 || stride == 2) || Cin % 16 || Cout % 16) return 1;
  if (nlines & (nlines - 1)) return 2;
  if (cin_store < 1 || cin_store > Cin) return 7;
  WgradParams p;
  p.x = x; p.gy = gy; p.U = U; p.ld = ld; p.off = off;
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
  p.s2_classes = masked ? 1 : 0;
  if (Cout % (tmt * 16)) return 4;
  // power-of-two output geometry -> descriptor-free K loop
  const int qpr = (Wo + 3) / 4;
  const bool fast = (Wo % 4 == 0) && !(qpr & (qpr - 1)) && !(Ho & (Ho - 1));
  p.lg_qpr = p.lg_ho = 0;
  while ((1 << p.lg_qpr) < qpr) ++p.lg_qpr;
  while ((1 << p.lg_ho) < Ho) ++p.lg_ho;
  p.nquads = B * Ho * qpr;
  p.rows_r = rows_r; p.rows_off = rows_off; p.rows_last = B * H - 1;
#define GO2(T, K, N)                                                                         \
  do {                                                                                       \
    if (fast) return passes <= 1 ? launch<T, K, 1, 3, true, N>(p, Kc, st) : launch<T, K, 4, 1, true, N>(p, Kc, st); \
    return passes <= 1 ? launch<T, K, 1, 2, false, N>(p, Kc, st) : launch<T, K, 4, 1, false, N>(p, Kc, st);           \
  } while (0)
#define GO(T, K)                  \
  do {                            \
    if (npl == 1) GO2(T, K, 1);   \
    if (npl == 2) GO2(T, K, 2);   \
    GO2(T, K, 3);                 \
  } while (0)
#define GOM(T)                                                                                              \
  do {                                                                                                      \
    if (npl == 2) return fast ? launch<T, 3, 1, 3, true, 2, true>(p, Kc, st) : launch<T, 3, 1, 2, false, 2, true>(p, Kc, st); \
    return fast ? launch<T, 3, 1, 3, true, 3, true>(p, Kc, st) : launch<T, 3, 1, 2, false, 3, true>(p, Kc, st);               \
  } while (0)
  // fp32 planes on the regular ResNet geometry: the eight-wave kernel (BLADES_WGRAD_WAVES=4 keeps the four-wave one)
  static const bool waves8 = []() { const char* e = getenv("BLADES_WGRAD_WAVES"); return !(e && atoi(e) == 4); }();
  if (waves8 && npl >= 2 && ks == 3 && stride == 1 && fast && passes <= 1 && rows_r > 0 && p.w_shift >= 0 &&
      tmt >= 2) {
#define GO8(T, M)                                              \
  do {                                                         \
    if (npl == 2) return launch8<T, 2, M>(p, Kc, st);          \
    return launch8<T, 3, M>(p, Kc, st);                        \
  } while (0)
    if (masked && tmt == 8) {  // 128 output channels per workgroup
      const int rc = npl == 2 ? launch8m<2>(p, Kc, st) : launch8m<3>(p, Kc, st);
      if (rc != 3) return rc;
      tmt = 4;  // the plane images do not fit next to a 128-channel gy tile: 64-channel tiles
    }
    if (masked) { if (tmt == 4) GO8(4, true); GO8(2, true); }
    if (tmt == 4) GO8(4, false);
    GO8(2, false);
#undef GO8
  }
  if (masked) { if (tmt == 4) GOM(4); if (tmt == 2) GOM(2); GOM(1); }
#undef GOM
  if (ks == 3) { if (tmt == 4) GO(4, 3); if (tmt == 2) GO(2, 3); if (tmt == 1) GO(1, 3); }
  else { if (tmt == 4) GO(4, 1); if (tmt == 2) GO(2, 1); if (tmt == 1) GO(1, 1); }
#undef GO
#undef GO2
  return 5;
}
