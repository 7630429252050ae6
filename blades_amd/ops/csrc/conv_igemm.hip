// Shared-weight convolution (forward and data gradient) as an implicit GEMM on the bf16 matrix cores, with an
// fp32-faithful operand split.
//
//   out[n, sy*so+poy, sx*so+pox, co] = bias[co] + sum_t sum_ci in[n, sy*si+ay_t, sx*si+ax_t, ci] * W_t[co][ci]
//
// One "tap table" form covers every convolution of the FedSGD fast path (reference hot loop: Task.train_one_batch,
// fllib/tasks/task.py:170-186, which runs them through cuDNN):
//   * forward, any stride/padding:       si = stride, so = 1, taps = all (r, s), ay = r - pad, ax = s - pad
//   * forward, stride s > 1, H and W multiples of s: the reduction runs over *segments*, one per input parity class
//                                        (py, px): a stride-1 problem on the view in[:, py::s, px::s] with the taps
//                                        whose r - pad = s*ay + py (and likewise in x).  Only one class of the
//                                        receptive field is in LDS at a time, so the tile is as large as a stride-1
//                                        layer's (64*4 pixels) instead of a quarter of it.
//   * data gradient, stride 1:           in = gy, taps (r, s) -> ay = pad - r, ax = pad - s, W transposed
//   * data gradient, stride s > 1:       one launch per output parity class (ph, pw): so = s, taps = the (r, s)
//                                        with (ph + pad - r) % s == 0, ay = (ph + pad - r) / s   (no zero-stuffing)
// Activations are channels-last.  gfx950 has no TF32/xf32 path and its fp32 MFMA runs at 1/16 of the bf16 rate,
// so fp32 operands are split *exactly* into NPL bf16 planes while they are staged into LDS
// (x = h + m + l, 3 x 8 = 24 significant bits) and the product is evaluated as
//   NPL = 3:  hh + hm + mh + hl + lh + mm   (6 MFMAs, dropped terms <= 2^-26 |a||b|: below one fp32 rounding)
//   NPL = 2:  hh + hm + mh                  (3 MFMAs, ~2^-17 relative — "bf16x3", 64x tighter than TF32)
//   NPL = 1:  plain bf16 in / bf16 out
// always accumulating in fp32.  6 MFMAs per staged fragment pair also halve the LDS traffic per MFMA, so the
// kernel is MFMA-bound by construction.
//
// CDNA4 mapping
//   * workgroup = 4 waves = 64*MF output pixels x 64 output channels; wave w owns pixels [w*16*MF, (w+1)*16*MF)
//     as MF fragments of v_mfma_f32_16x16x32_bf16 with the *weights* as the A operand, so a lane ends up with 4
//     consecutive output channels of one pixel (16-byte channels-last stores).
//   * K-step = (tap, 32 input channels).  The input tile is staged ONCE per 32-channel chunk as the tile's
//     receptive field (rows x full width, zero halo shared between neighbouring rows / images) and every tap is
//     just a different LDS offset: a lane's fragment address is base(pixel) + tapoff(t).  Strides and parity
//     classes only change base(pixel).
//   * Two workgroups share a CU (<= 80 KB of LDS and <= 256 registers each), so one workgroup's staging, barriers,
//     prologue and epilogue run under the other's MFMAs.  Within a workgroup the global loads of the next weight
//     tile (pre-split planes, [tap][co][ci] with ci contiguous) and, at a chunk's last tap, of the next input
//     chunk are issued before the step's MFMAs and committed to LDS (split into planes there) after them.
//   * Workgroup ids are remapped so that an XCD receives a contiguous range of tiles (neighbouring tiles share
//     halo rows and all channel tiles of a pixel tile share the input through that XCD's L2).
#include "common.h"
#include <cstring>
#include <algorithm>
#include <cstdlib>

#ifndef BLADES_AF_AHEAD
#define BLADES_AF_AHEAD 1
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 32;        // input channels per K-step = K of one 16x16x32 MFMA
constexpr int BN = 64;        // output channels per workgroup
constexpr int MAXTAPS = 25;
constexpr int MAXSEG = 16;    // reduction segments (input parity classes of a strided forward convolution)
constexpr int LDS_TOTAL = 160 * 1024;
// two workgroups per CU (<= 80 KB LDS, <= 256 registers each); three smaller ones measured slower (MF = 2: 5.7 vs
// 5.2 ms on the 64-channel layer: the weight-fragment reads per MFMA double)
constexpr int wgs_per_cu(int) { return 2; }
// A-staging passes of the next chunk prefetched into registers under the MFMAs
constexpr int max_prefetch(int, bool rf = false) { return rf ? 10 : 12; }

struct IgemmParams {
  const void* in;             // [N][Hi][Wi][Ci]
  const unsigned short* wp;   // [NPL][ntw][Cop][Cip] bf16 planes, ci contiguous
  void* out;                  // [N][Hout][Wout][Co]
  const float* bias;          // [Co] or null
  int N, Hi, Wi, Ci, Cip;
  int Hout, Wout, Co, Cop;
  int Hs, Ws, so, poy, pox, si;
  int TI, TR, tiles_per_img, n_mtiles, n_ntiles;
  int IR, WS, PL, IS, NS, aymin;
  int ntaps, ntw, npass, region_px;
  float inv_ws, inv_tr, inv_wi, inv_irwi;
  int tap_off[MAXTAPS];       // slot offset of tap t relative to a pixel's base slot
  int tap_w[MAXTAPS];         // index of tap t in the weight tensor
  int n_clients, imgs_per_client;  // per-client weights (local steps > 1): image n uses weight set n / imgs_per_client
  long wp_client_stride, bias_client_stride;  // elements between two clients' weight planes / biases
  // Tile pixel handled by (fragment f, lane column li): perm[f * 16 + li].  Identity where the 16 pixels of a
  // fragment sit in 16 consecutive LDS slots (32- and 16-pixel-wide rows); on narrower images a fragment spans
  // several padded rows or images and consecutive pixels put two lanes of a ds_read_b128 service group on the same
  // banks (PMC: 21.6 % conflict cycles on the 8x8 layers, 44 % on 4x4).  The bank of a fragment read depends on
  // slot mod 8 only, and a read is conflict-free iff the slots of lanes li = 4..11 are distinct mod 8 and those of
  // li = 0..3, 12..15 are: the host deals the tile's pixels to fragments accordingly (dispatch()).
  unsigned short perm[256];
  unsigned long long* prof;   // PROF build: [16] cycle sums (see the kernel)
  int ablate;                 // diagnostics (BLADES_IGEMM_ABLATE): 1 no staging loads, 2 no commit, 4 no MFMAs
  int bdouble;                // two weight-tile images in LDS (where they fit): a step's successor tile is committed
                              // without waiting for the other waves, one barrier per step instead of two
  float* stats;               // forward, fp32 output: per-(client, tile, channel) sum / sum of squares of the
  int stats_B, stats_nchunk;  // output — the first pass of the per-client BatchNorm that follows (pcbn.hip's
                              // partial[(client * nchunk + chunk) * Co + ch][2]); stats_B = images per client
  int nseg;                   // reduction segments: segment g = taps [seg_begin[g], seg_begin[g + 1]) on the input view
  int seg_begin[MAXSEG + 1];  // that starts seg_goff[g] elements into the image (all views have the same size)
  long seg_goff[MAXSEG];
  int accumulate;             // out += result (the data gradient of a residual block's first convolution lands on the
                              // shortcut branch's gradient: no separate elementwise add over the tensor)
  // data gradient, fp32 tensors: first pass of the BatchNorm backward that consumes this gradient (the layer that
  // produced this convolution's input).  bstat_x = that BatchNorm's input (same shape/layout as `out`), bstat_mask =
  // its ReLU gate bitmask (or null); stats then receives per (client, chunk, channel) the raw sums
  // (sum g, sum g * x), g = the finished gradient (after `accumulate`) gated by the mask — pcbn.hip's finalize turns
  // them into (dbeta, dgamma).  stats_chunk0 = first chunk of this launch (strided layers: one launch per class).
  const float* bstat_x;
  const unsigned char* bstat_mask;
  int stats_chunk0, stats_class, stats_classes;
  int cr_taps;                // bf16 tensors (NPL = 1): > 0 = all weight tiles of a (segment, chunk) are resident in LDS
                              // ([cr_taps][64 co][32 ci]) and the taps run without barriers (see the v1 kernel)
  int Hphys, Wphys, in_rm, in_cm;  // the input may be a strided view in[:, ::rm, ::cm] of a [Hphys][Wphys] image (a
                                   // strided 1x1 convolution only ever touches those pixels: it runs as a stride-1
                                   // convolution on the view, staging a quarter of the pixels)
};

// exact for 0 <= e < 2^20, 1 <= d < 2^12: (e + 0.5) / d is at least 0.5/d away from any integer
__device__ __forceinline__ int fdiv(int e, float inv_d) { return __float2int_rz(((float)e + 0.5f) * inv_d); }

template <int NPL>
__device__ __forceinline__ void split4(f32x4 r, u32x2 (&pl)[NPL]) {
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    const bf16x4 b = __builtin_convertvector(r, bf16x4);
    pl[q] = __builtin_bit_cast(u32x2, b);
    if (q + 1 < NPL) r -= __builtin_convertvector(b, f32x4);
  }
}

// The LDS-staged weight tile kernel (rounds 2-3), kept as it was for the geometries where it is the faster one; the
// generalised kernel below adds the register-fed weight path (RF) and the diagnostic builds.
template <int NPL, int MF, bool SPLIT, bool CR = false>
__global__ __launch_bounds__(256, wgs_per_cu(MF)) void conv_igemm_v1_kernel(const IgemmParams p) {
  static_assert(SPLIT || NPL == 1, "bf16 input has a single plane");
  static_assert(!CR || (NPL == 1 && !SPLIT), "chunk-resident weights: bf16 tensors only");
  auto stamp = [](const int) {};
  constexpr int MAXPF = max_prefetch(MF);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per_xcd = gridDim.x >> 3;
  const int lid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (lid >= p.n_mtiles * p.n_ntiles) return;  // whole workgroup leaves: no barrier is skipped by a subset
  const int nt = lid % p.n_ntiles, mt = lid / p.n_ntiles;
  const int n0 = nt * BN;
  int img0, sy0;
  if (p.TI == 1) { img0 = mt / p.tiles_per_img; sy0 = (mt - img0 * p.tiles_per_img) * p.TR; }
  else { img0 = mt * p.TI; sy0 = 0; }
  const int iy_lo = sy0 * p.si + p.aymin;
  // per-client weights: a tile never straddles two clients (plan() keeps TI a divisor of imgs_per_client)
  const int client = p.n_clients > 1 ? img0 / p.imgs_per_client : 0;
  const unsigned short* wp_c = p.wp + (size_t)client * p.wp_client_stride;
  const float* bias_c = p.bias ? p.bias + (size_t)client * p.bias_client_stride : nullptr;

  const int a_plane = p.NS * 64, a_buf = NPL * a_plane;
  constexpr int b_plane = BN * 64;
  unsigned char* A_lds = smem;          // [NPL][NS slots][32 ch] bf16
  unsigned char* B_lds = smem + a_buf;  // [NPL][64 co][32 ci] bf16
  for (int i = tid * 16; i < a_buf; i += 256 * 16) *reinterpret_cast<u32x4*>(A_lds + i) = u32x4{0, 0, 0, 0};

  // Both LDS images have 64-byte rows (32 channels of one slot / one output channel).  A fragment read
  // (ds_read_b128: lane -> row li, 16-byte channel group kg) of 16 consecutive rows would put four lanes of
  // every 16-lane service group on the same banks; XOR-ing bit 1 of the channel group with bit 2 of the row
  // makes those reads conflict-free.  Stores apply the same involution.
  auto swz = [](int row) { return (row & 4) << 3; };

  // ---- this lane's output pixels: fragment pf, column li ---------------------------------------------------
  const int li = lane & 15, kg = lane >> 4;
  int a_slot[MF];
  long out_pix[MF];
#pragma unroll
  for (int pf = 0; pf < MF; ++pf) {
    const int m = p.perm[(wave * MF + pf) * 16 + li];
    const int q = fdiv(m, p.inv_ws), sx = m - q * p.Ws;
    const int il = fdiv(q, p.inv_tr), ry = q - il * p.TR;
    const bool ok = il < p.TI && sy0 + ry < p.Hs && img0 + il < p.N;
    a_slot[pf] = ok ? il * p.IS + ry * p.si * p.WS + sx * p.si + p.PL : p.PL;
    out_pix[pf] = ok ? ((long)(img0 + il) * p.Hout + (sy0 + ry) * p.so + p.poy) * p.Wout + sx * p.so + p.pox : -1;
  }

  // ---- staging roles -----------------------------------------------------------------------------------------
  // A: one pass = PIXPP region pixels x 32 channels; lane group (cg) covers CPL channels of a pixel (16 bytes)
  constexpr int CPL = SPLIT ? 4 : 8, LPP = KC / CPL, PIXPP = 256 / LPP;
  const int cg = tid % LPP, pp = tid / LPP;
  struct ADesc { long goff; int slot; };
  auto a_desc = [&](int pass, long seg_goff) {
    ADesc d;
    const int e = pass * PIXPP + pp;
    const int il = fdiv(e, p.inv_irwi), rem = e - il * p.IR * p.Wi;
    const int r = fdiv(rem, p.inv_wi), ix = rem - r * p.Wi;
    const int iy = iy_lo + r;
    const bool ok = e < p.region_px && (unsigned)iy < (unsigned)p.Hi && img0 + il < p.N;
    d.slot = il * p.IS + r * p.WS + ix + p.PL;
    d.goff = ok ? (((long)(img0 + il) * p.Hphys + (long)iy * p.in_rm) * p.Wphys + (long)ix * p.in_cm) * p.Ci + seg_goff
                : -1;
    return d;
  };
  const bool vec_in = (p.Ci % CPL) == 0;
  auto a_load = [&](const ADesc& d, int c0) {
    u32x4 v = {0, 0, 0, 0};
    const int ch = c0 + cg * CPL;
    if (d.goff >= 0 && ch < p.Ci) {
      if constexpr (SPLIT) {
        const float* src = reinterpret_cast<const float*>(p.in) + d.goff + ch;
        if (vec_in) v = *reinterpret_cast<const u32x4*>(src);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ch + j < p.Ci ? __float_as_uint(src[j]) : 0u;
        }
      } else {
        const unsigned short* src = reinterpret_cast<const unsigned short*>(p.in) + d.goff + ch;
        if (vec_in) v = *reinterpret_cast<const u32x4*>(src);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned e = ch + j < p.Ci ? src[j] : 0u;
            v[j >> 1] |= e << ((j & 1) * 16);
          }
        }
      }
    }
    return v;
  };
  auto a_store = [&](const ADesc& d, u32x4 v) {
    if (d.goff < 0) return;  // outside the image / region: the slot keeps its zero fill
    if constexpr (SPLIT) {
      unsigned char* dst = A_lds + d.slot * 64 + (((cg >> 1) * 16) ^ swz(d.slot)) + (cg & 1) * 8;
      u32x2 pl[NPL];
      split4<NPL>(__builtin_bit_cast(f32x4, v), pl);
#pragma unroll
      for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2*>(dst + q * a_plane) = pl[q];
    } else {
      *reinterpret_cast<u32x4*>(A_lds + d.slot * 64 + ((cg * 16) ^ swz(d.slot))) = v;
    }
  };
  // whole-chunk staging with every load of a batch in flight before the first conversion
  auto a_stage_sync = [&](int pass_begin, int c0, long seg) {
    for (int j0 = pass_begin; j0 < p.npass; j0 += 4) {
      ADesc d[4];
      u32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d[j].goff = -1; d[j].slot = 0;
        if (j0 + j < p.npass) d[j] = a_desc(j0 + j, seg);
        v[j] = a_load(d[j], c0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) a_store(d[j], v[j]);
    }
  };
  // B: weight tile of one step = NPL x [64 co][32 ci]; thread -> (co row tid >> 2, 8-channel group tid & 3)
  const int brow = tid >> 2, bkg = tid & 3;
  // (global planes are in fragment order, see conv_wprep_kernel: a thread's 16 bytes of plane q sit q KiB x 2 apart)
  const int nchunks_w = p.Cip / KC;
  const unsigned short* wsrc = wp_c + (size_t)nt * (2 * NPL * 2 * 64 * 8) +
                               ((((brow >> 5) * NPL) * 2 + ((brow >> 4) & 1)) * 64 + bkg * 16 + (brow & 15)) * 8;
  auto b_load = [&](int tw, int c0, int q) {
    return *reinterpret_cast<const u32x4*>(wsrc + ((size_t)tw * nchunks_w + (c0 >> 5)) * p.n_ntiles * (2 * NPL * 2 * 64 * 8) +
                                           q * (2 * 64 * 8));
  };
  const int b_dst = brow * 64 + ((bkg * 16) ^ swz(brow));
  const int b_img = p.bdouble ? NPL * b_plane : 0;  // distance between the two weight-tile images
  auto b_store = [&](u32x4 v, int q, int img) {
    *reinterpret_cast<u32x4*>(B_lds + img * b_img + q * b_plane + b_dst) = v;
  };

  f32x4 acc[MF][4];
#pragma unroll
  for (int pf = 0; pf < MF; ++pf)
#pragma unroll
    for (int cf = 0; cf < 4; ++cf) acc[pf][cf] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = p.Cip / KC, nsteps = nchunks * p.ntaps;
  __syncthreads();  // zero fill complete
  // Chunk-resident weights (bf16 tensors): a step of the loop below is 16 MFMAs per wave (256 cycles) behind a weight
  // tile commit and a barrier — 16-20 % MFMA-busy, 62 % of the bf16 round.  With a single plane the weight tiles of ALL
  // taps of a (segment, chunk) fit beside the input tile (9 x 4 KB), so they are staged together with it and the taps
  // run back to back: two barriers per chunk instead of one or two per tap.
  constexpr int CRT = 9;
  constexpr bool cr = CR;
  if (nsteps > 0) {
    a_stage_sync(0, 0, p.seg_goff[0]);
    if (cr) {
      for (int j = 0; j < p.seg_begin[1] - p.seg_begin[0]; ++j)
        *reinterpret_cast<u32x4*>(B_lds + j * b_plane + b_dst) = b_load(p.tap_w[p.seg_begin[0] + j], 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < NPL; ++q) b_store(b_load(p.tap_w[0], 0, q), q, 0);
    }
  }
  __syncthreads();

  stamp(0);  // zero fill + synchronous staging of the first chunk / weight tile
  const unsigned char* Bb0 = B_lds + li * 64 + ((kg * 16) ^ swz(li));
  // step order: segment, then 32-channel chunk, then the segment's taps; the input tile is restaged whenever the
  // (segment, chunk) pair changes
  int g = 0, c = 0, t = 0;
  long goff_cur = p.seg_goff[0];
  // The tap tables live in one register each (lane i holds entry i) and are read with v_readlane: a kernel-argument
  // load indexed by the step's tap has the latency of a cache access, shares its counter with the LDS reads, and sat
  // in front of every step's weight prefetch and fragment-read burst.
  static_assert(MAXTAPS <= 64, "one lane per tap");
  const int lane_toff = p.tap_off[lane < MAXTAPS ? lane : 0], lane_tw = p.tap_w[lane < MAXTAPS ? lane : 0];
  int seg_lo = p.seg_begin[0], seg_hi = p.seg_begin[1];  // taps of segment g (kept in registers: a kernel-argument
                                                         // load per step shows in the short steps of the bf16 kernels)
  if constexpr (CR) {
    if (nsteps > 0) {
      for (;;) {
        int cn = c + 1, gn = g;
        if (cn == nchunks) { cn = 0; gn = g + 1; }
        const bool more = gn < p.nseg;
        int nlo = seg_lo, nhi = seg_hi;
        long goff_next = goff_cur;
        if (more && gn != g) { nlo = seg_hi; nhi = p.seg_begin[gn + 1]; goff_next = p.seg_goff[gn]; }
        // 1. global loads of the next (segment, chunk): its weight tiles and its input tile
        u32x4 breg[CRT];
        u32x4 areg[MAXPF];
        if (more) {
#pragma unroll
          for (int j = 0; j < CRT; ++j) {
            breg[j] = u32x4{0, 0, 0, 0};
            if (nlo + j < nhi) breg[j] = b_load(__builtin_amdgcn_readlane(lane_tw, nlo + j), cn * KC, 0);
          }
#pragma unroll
          for (int j = 0; j < MAXPF; ++j) {
            areg[j] = u32x4{0, 0, 0, 0};
            if (j < p.npass) areg[j] = a_load(a_desc(j, goff_next), cn * KC);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // 2. the taps of (segment g, chunk c), no barrier between them
        for (int tt = seg_lo; tt < seg_hi; ++tt) {
          const int toff = __builtin_amdgcn_readlane(lane_toff, tt);
          const unsigned char* Bb = Bb0 + (tt - seg_lo) * b_plane;
          bf16x8 wf[4];
#pragma unroll
          for (int cf = 0; cf < 4; ++cf) wf[cf] = *reinterpret_cast<const bf16x8*>(Bb + cf * 1024);
#pragma unroll
          for (int pf = 0; pf < MF; ++pf) {
            const int sl = a_slot[pf] + toff;
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(A_lds + sl * 64 + ((kg * 16) ^ swz(sl)));
#pragma unroll
            for (int cf = 0; cf < 4; ++cf)
              acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cf], af, acc[pf][cf], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!more) break;
        __syncthreads();  // every wave has read this chunk's fragments
#pragma unroll
        for (int j = 0; j < CRT; ++j)
          if (nlo + j < nhi) *reinterpret_cast<u32x4*>(B_lds + j * b_plane + b_dst) = breg[j];
#pragma unroll
        for (int j = 0; j < MAXPF; ++j)
          if (j < p.npass) a_store(a_desc(j, goff_next), areg[j]);
        a_stage_sync(MAXPF, cn * KC, goff_next);
        __syncthreads();
        g = gn; c = cn; seg_lo = nlo; seg_hi = nhi; goff_cur = goff_next;
      }
    }
  }
  for (int ks = 0; ks < (CR ? 0 : nsteps); ++ks) {
    int tn = t + 1, cn = c, gn = g;
    if (tn == seg_hi) {
      tn = seg_lo; cn = c + 1;
      if (cn == nchunks) { cn = 0; gn = g + 1; tn = seg_hi; }
    }
    const bool more = ks + 1 < nsteps;
    const bool next_b = more && !(p.ablate & (1 | 8));  // bit 3 (timing only, wrong results): no weight staging and
                                                       // no barriers inside a chunk = bound of a register-fed B path
    const bool next_a = more && (cn != c || gn != g) && !(p.ablate & 1);  // this step ends (g, c): restage A after it
    long goff_next = goff_cur;
    if (gn != g && more) goff_next = p.seg_goff[gn];
    const int toff = __builtin_amdgcn_readlane(lane_toff, t);
    // 1. global loads: next step's weight tile; at a chunk's last tap also the next chunk of the input tile.
    //    They complete under this step's MFMAs (and under the MFMAs of the workgroup sharing the CU).
    u32x4 breg[NPL];
    if (next_b) {
      const int tw = __builtin_amdgcn_readlane(lane_tw, tn);
#pragma unroll
      for (int q = 0; q < NPL; ++q) breg[q] = b_load(tw, cn * KC, q);
    }
    u32x4 areg[MAXPF];
    if (next_a) {
#pragma unroll
      for (int j = 0; j < MAXPF; ++j) {
        areg[j] = u32x4{0, 0, 0, 0};
        if (j < p.npass) areg[j] = a_load(a_desc(j, goff_next), cn * KC);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(1);  // address arithmetic + issue of the global loads
    // 2. MFMAs of (chunk c, tap t)
    if (!(p.ablate & 4)) {
      bf16x8 wf[NPL][4];
      const unsigned char* Bb = Bb0 + (ks & 1) * b_img;
#pragma unroll
      for (int q = 0; q < NPL; ++q)
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) wf[q][cf] = *reinterpret_cast<const bf16x8*>(Bb + q * b_plane + cf * 1024);
#pragma unroll
      for (int pf = 0; pf < MF; ++pf) {
        const int s = a_slot[pf] + toff;
        const unsigned char* Ab = A_lds + s * 64 + ((kg * 16) ^ swz(s));
        bf16x8 af[NPL];
#pragma unroll
        for (int q = 0; q < NPL; ++q) af[q] = *reinterpret_cast<const bf16x8*>(Ab + q * a_plane);
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) {
          // (weight plane, activation plane) pairs, small terms first
          if constexpr (NPL == 3) {
            acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][cf], af[1], acc[pf][cf], 0, 0, 0);
            acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][cf], af[2], acc[pf][cf], 0, 0, 0);
            acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2][cf], af[0], acc[pf][cf], 0, 0, 0);
          }
          if constexpr (NPL >= 2) {
            acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][cf], af[1], acc[pf][cf], 0, 0, 0);
            acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][cf], af[0], acc[pf][cf], 0, 0, 0);
          }
          acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][cf], af[0], acc[pf][cf], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(2);  // fragment reads + MFMA issue
    if ((p.ablate & 8) && !next_a) { t = tn; c = cn; g = gn; goff_cur = goff_next; continue; }
    // every wave has read this step's fragments — not needed for the weight tile alone when it goes to the other image
    // (its last readers finished before the barrier that ended the previous step)
    if (!p.bdouble || next_a) __syncthreads();
    stamp(3);  // first barrier
    // 3. commit the staged data
    if (next_b && !(p.ablate & 2)) {
#pragma unroll
      for (int q = 0; q < NPL; ++q) b_store(breg[q], q, (ks + 1) & 1);
    }
    if (next_a && !(p.ablate & 2)) {
#pragma unroll
      for (int j = 0; j < MAXPF; ++j)
        if (j < p.npass) a_store(a_desc(j, goff_next), areg[j]);
      a_stage_sync(MAXPF, cn * KC, goff_next);  // geometries with more than MAXPF passes: the rest without prefetch
    }
    if (next_a) stamp(5); else stamp(4);  // commit: wait for the loads, split, LDS stores (4: weight tile only)
    __syncthreads();
    stamp(6);  // second barrier
    if (gn != g && more) { seg_lo = seg_hi; seg_hi = p.seg_begin[gn + 1]; }
    t = tn; c = cn; g = gn; goff_cur = goff_next;
  }

  // ---- epilogue: lane = (pixel li of fragment pf, channels cf*16 + kg*4 .. +3) -------------------------------
  const bool vec_out = (p.Co % 4) == 0;
  const bool want_stats = SPLIT && p.stats != nullptr;  // block-uniform
  const bool bstat = want_stats && p.bstat_x != nullptr;  // BatchNorm-backward sums instead of the forward ones
  f32x4 st_s[4], st_q[4];
#pragma unroll
  for (int cf = 0; cf < 4; ++cf) { st_s[cf] = f32x4{0.f, 0.f, 0.f, 0.f}; st_q[cf] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int pf = 0; pf < MF; ++pf) {
    if (out_pix[pf] < 0) continue;
#pragma unroll
    for (int cf = 0; cf < 4; ++cf) {
      const int co = n0 + cf * 16 + kg * 4;
      if (co >= p.Co) continue;
      f32x4 v = acc[pf][cf];
      if (bias_c != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += co + j < p.Co ? bias_c[co + j] : 0.f;
      }
      if (want_stats && !bstat) { st_s[cf] += v; st_q[cf] += v * v; }
      if constexpr (SPLIT) {
        float* dst = reinterpret_cast<float*>(p.out) + out_pix[pf] * p.Co + co;
        if (vec_out) {
          if (p.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
          *reinterpret_cast<f32x4*>(dst) = v;
          if (bstat) {  // (Co % 8 == 0 is checked on the host)
            const size_t off = (size_t)out_pix[pf] * p.Co + co;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(p.bstat_x + off);
            const unsigned mb = p.bstat_mask != nullptr ? (unsigned)p.bstat_mask[off >> 3] >> (co & 4) : 0xFu;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float g = (mb >> j) & 1u ? v[j] : 0.f;
              st_s[cf][j] += g; st_q[cf][j] += g * xv[j];
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (co + j < p.Co) dst[j] = p.accumulate ? dst[j] + v[j] : v[j];
        }
      } else {
        unsigned short* dst = reinterpret_cast<unsigned short*>(p.out) + out_pix[pf] * p.Co + co;
        if (p.accumulate) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < p.Co) v[j] += __uint_as_float((unsigned)dst[j] << 16);
        }
        const u32x2 b = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
        if (vec_out) *reinterpret_cast<u32x2*>(dst) = b;
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (co + j < p.Co) dst[j] = (unsigned short)(b[j >> 1] >> ((j & 1) * 16));
        }
      }
    }
  }
  if constexpr (SPLIT) {
    if (want_stats) {
      // BatchNorm statistics of this tile: fixed-order tree over the 16 pixel lanes, then over the four waves
      // through LDS (every wave is past its last fragment read: the K loop ends with a barrier)
#pragma unroll
      for (int cf = 0; cf < 4; ++cf)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            st_s[cf][j] += __shfl_xor(st_s[cf][j], o);
            st_q[cf][j] += __shfl_xor(st_q[cf][j], o);
          }
      float* red = reinterpret_cast<float*>(smem);  // [4 waves][64 ch][2]
      if (li == 0) {
#pragma unroll
        for (int cf = 0; cf < 4; ++cf)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ch = cf * 16 + kg * 4 + j;
            red[(wave * 64 + ch) * 2 + 0] = st_s[cf][j];
            red[(wave * 64 + ch) * 2 + 1] = st_q[cf][j];
          }
      }
      __syncthreads();
      if (tid < 128) {
        const int ch = tid >> 1, which = tid & 1;
        if (n0 + ch < p.Co) {
          float a = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) a += red[(w * 64 + ch) * 2 + which];
          const int cl = img0 / p.stats_B, in_cl = img0 - cl * p.stats_B;
          const int chunk = p.TI == 1 ? in_cl * p.tiles_per_img + (mt - img0 * p.tiles_per_img) : in_cl / p.TI;
          p.stats[(((size_t)cl * p.stats_nchunk + p.stats_chunk0 + chunk) * p.Co + n0 + ch) * 2 + which] = a;
        }
      }
    }
  }
}


// PROF (diagnostic build of one variant, BLADES_IGEMM_PROF=1): s_memtime stamps at the phase boundaries of every
// wave, summed over the launch into p.prof[16] — where a workgroup's time goes (tools/r3/igemm_prof.py)
// PROF = 2 (BLADES_IGEMM_PROF=2): a light trace instead — four wall-clock stamps per workgroup (entry, loop start,
// loop end, stores acknowledged; none inside the K loop) plus the shader-clock count of the loop and the hardware ids,
// one record per tile in p.prof: how the two workgroups of a CU interleave (tools/r3/igemm_trace.py)
// RF ("register-fed weights", the default): the four waves own (pixel half, 32-channel half) quarters of the tile
// instead of pixel quarters.  A wave then needs only ITS 32 output channels of a step's weight tile — 2 fragments x
// NPL planes — and loads them straight from global memory into registers, one step ahead, from weight planes that
// conv_wprep_kernel stores in fragment order (1 KiB per wave instruction, L1/L2 resident: all four waves of both
// workgroups of a CU and every tile of the launch read the same few KiB).  No weight tile in LDS, no LDS stores and
// no barrier per step: workgroup barriers remain only around the input restage at a chunk boundary (every `taps`
// steps).  LDS fragment reads per MFMA are unchanged (24 per 96).  RF = false keeps the LDS-staged weight tile.
template <int NPL, int MF, bool SPLIT, int PROF = 0, bool RF = false>
__global__ __launch_bounds__(256, wgs_per_cu(MF)) void conv_igemm_kernel(const IgemmParams p) {
  static_assert(SPLIT || NPL == 1, "bf16 input has a single plane");
  constexpr int NPF = RF ? 2 * MF : MF;  // pixel fragments per wave
  constexpr int NCF = RF ? 2 : 4;        // channel fragments per wave
  unsigned long long pf_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_last = 0;  // (constant indices only: scalar registers)
  auto stamp = [&](const int slot) {
    if constexpr (PROF == 1) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      pf_t[slot] += now - pf_last;
      pf_last = now;
    }
  };
  if constexpr (PROF == 1) pf_last = __builtin_amdgcn_s_memtime();
  unsigned long long tr_t0 = 0, tr_t1 = 0, tr_t2 = 0, tr_c1 = 0, tr_c2 = 0;
  if constexpr (PROF == 2) tr_t0 = __builtin_amdgcn_s_memrealtime();
  constexpr int MAXPF = max_prefetch(MF, RF);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per_xcd = gridDim.x >> 3;
  const int lid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (lid >= p.n_mtiles * p.n_ntiles) return;  // whole workgroup leaves: no barrier is skipped by a subset
  const int nt = lid % p.n_ntiles, mt = lid / p.n_ntiles;
  const int n0 = nt * BN;
  int img0, sy0;
  if (p.TI == 1) { img0 = mt / p.tiles_per_img; sy0 = (mt - img0 * p.tiles_per_img) * p.TR; }
  else { img0 = mt * p.TI; sy0 = 0; }
  const int iy_lo = sy0 * p.si + p.aymin;
  // per-client weights: a tile never straddles two clients (plan() keeps TI a divisor of imgs_per_client)
  const int client = p.n_clients > 1 ? img0 / p.imgs_per_client : 0;
  const unsigned short* wp_c = p.wp + (size_t)client * p.wp_client_stride;
  const float* bias_c = p.bias ? p.bias + (size_t)client * p.bias_client_stride : nullptr;

  const int a_plane = p.NS * 64, a_buf = NPL * a_plane;
  constexpr int b_plane = BN * 64;
  unsigned char* A_lds = smem;          // [NPL][NS slots][32 ch] bf16
  unsigned char* B_lds = smem + a_buf;  // [NPL][64 co][32 ci] bf16
  for (int i = tid * 16; i < a_buf; i += 256 * 16) *reinterpret_cast<u32x4*>(A_lds + i) = u32x4{0, 0, 0, 0};

  // Both LDS images have 64-byte rows (32 channels of one slot / one output channel).  A fragment read
  // (ds_read_b128: lane -> row li, 16-byte channel group kg) of 16 consecutive rows would put four lanes of
  // every 16-lane service group on the same banks; XOR-ing bit 1 of the channel group with bit 2 of the row
  // makes those reads conflict-free.  Stores apply the same involution.
  auto swz = [](int row) { return (row & 4) << 3; };

  // ---- this lane's output pixels: fragment pf, column li ---------------------------------------------------
  const int li = lane & 15, kg = lane >> 4;
  const int wpix = RF ? (wave & 1) : wave;      // which NPF*16-pixel group of the tile
  const int cbase = RF ? (wave >> 1) * 32 : 0;  // first output channel of the wave inside the tile
  int a_slot[NPF];
  // (the output pixel index is only needed by the epilogue and is recomputed there: nothing of it lives through the
  // K loop)
  auto pixel_of = [&](int pf, int& slot) -> long {
    const int m = p.perm[(wpix * NPF + pf) * 16 + li];
    const int q = fdiv(m, p.inv_ws), sx = m - q * p.Ws;
    const int il = fdiv(q, p.inv_tr), ry = q - il * p.TR;
    const bool ok = il < p.TI && sy0 + ry < p.Hs && img0 + il < p.N;
    slot = ok ? il * p.IS + ry * p.si * p.WS + sx * p.si + p.PL : p.PL;
    return ok ? ((long)(img0 + il) * p.Hout + (sy0 + ry) * p.so + p.poy) * p.Wout + sx * p.so + p.pox : -1;
  };
#pragma unroll
  for (int pf = 0; pf < NPF; ++pf) (void)pixel_of(pf, a_slot[pf]);

  // ---- staging roles -----------------------------------------------------------------------------------------
  // A: one pass = PIXPP region pixels x 32 channels; lane group (cg) covers CPL channels of a pixel (16 bytes)
  constexpr int CPL = SPLIT ? 4 : 8, LPP = KC / CPL, PIXPP = 256 / LPP;
  const int cg = tid % LPP, pp = tid / LPP;
  struct ADesc { long goff; int slot; };
  auto a_desc = [&](int pass, long seg_goff, int pp_in = -1) {
    ADesc d;
    const int e = pass * PIXPP + (pp_in < 0 ? pp : pp_in);
    const int il = fdiv(e, p.inv_irwi), rem = e - il * p.IR * p.Wi;
    const int r = fdiv(rem, p.inv_wi), ix = rem - r * p.Wi;
    const int iy = iy_lo + r;
    const bool ok = e < p.region_px && (unsigned)iy < (unsigned)p.Hi && img0 + il < p.N;
    d.slot = il * p.IS + r * p.WS + ix + p.PL;
    d.goff = ok ? (((long)(img0 + il) * p.Hphys + (long)iy * p.in_rm) * p.Wphys + (long)ix * p.in_cm) * p.Ci + seg_goff
                : -1;
    return d;
  };
  const bool vec_in = (p.Ci % CPL) == 0;
  auto a_load = [&](const ADesc& d, int c0) {
    u32x4 v = {0, 0, 0, 0};
    const int ch = c0 + cg * CPL;
    if (d.goff >= 0 && ch < p.Ci) {
      if constexpr (SPLIT) {
        const float* src = reinterpret_cast<const float*>(p.in) + d.goff + ch;
        if (vec_in) v = *reinterpret_cast<const u32x4*>(src);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ch + j < p.Ci ? __float_as_uint(src[j]) : 0u;
        }
      } else {
        const unsigned short* src = reinterpret_cast<const unsigned short*>(p.in) + d.goff + ch;
        if (vec_in) v = *reinterpret_cast<const u32x4*>(src);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned e = ch + j < p.Ci ? src[j] : 0u;
            v[j >> 1] |= e << ((j & 1) * 16);
          }
        }
      }
    }
    return v;
  };
  auto a_store = [&](const ADesc& d, u32x4 v) {
    if (d.goff < 0) return;  // outside the image / region: the slot keeps its zero fill
    if constexpr (SPLIT) {
      unsigned char* dst = A_lds + d.slot * 64 + (((cg >> 1) * 16) ^ swz(d.slot)) + (cg & 1) * 8;
      u32x2 pl[NPL];
      split4<NPL>(__builtin_bit_cast(f32x4, v), pl);
#pragma unroll
      for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2*>(dst + q * a_plane) = pl[q];
    } else {
      *reinterpret_cast<u32x4*>(A_lds + d.slot * 64 + ((cg * 16) ^ swz(d.slot))) = v;
    }
  };
  // whole-chunk staging with every load of a batch in flight before the first conversion
  auto a_stage_sync = [&](int pass_begin, int c0, long seg) {
    for (int j0 = pass_begin; j0 < p.npass; j0 += 4) {
      ADesc d[4];
      u32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d[j].goff = -1; d[j].slot = 0;
        if (j0 + j < p.npass) d[j] = a_desc(j0 + j, seg);
        v[j] = a_load(d[j], c0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) a_store(d[j], v[j]);
    }
  };
  // B: weight tile of one step = NPL x [64 co][32 ci]; thread -> (co row tid >> 2, 8-channel group tid & 3)
  // global weight planes are in fragment order (conv_wprep_kernel): one (tap, chunk, channel tile) = 2 halves x NPL
  // planes x 2 fragments x [64 lanes][8 bf16]
  const int brow = tid >> 2, bkg = tid & 3;
  const int nchunks = p.Cip / KC;
  constexpr size_t frag_tile = (size_t)2 * NPL * 2 * 64 * 8;
  auto wtile = [&](int tw, int cchunk) {
    return wp_c + (((size_t)tw * nchunks + cchunk) * p.n_ntiles + nt) * frag_tile;
  };
  const int b_src = ((((brow >> 5) * NPL) * 2 + ((brow >> 4) & 1)) * 64 + bkg * 16 + (brow & 15)) * 8;
  auto b_load = [&](int tw, int c0, int q) {
    return *reinterpret_cast<const u32x4*>(wtile(tw, c0 / KC) + b_src + q * (2 * 64 * 8));
  };
  const int rf_src = ((wave >> 1) * NPL * 2 * 64 + lane) * 8;
  auto rf_load = [&](bf16x8 (&w)[NPL][NCF], int tw, int cchunk) {
    const unsigned short* b = wtile(tw, cchunk) + rf_src;
#pragma unroll
    for (int q = 0; q < NPL; ++q)
#pragma unroll
      for (int cf = 0; cf < NCF; ++cf) w[q][cf] = *reinterpret_cast<const bf16x8*>(b + (q * 2 + cf) * (64 * 8));
  };
  const int b_dst = brow * 64 + ((bkg * 16) ^ swz(brow));
  const int b_img = p.bdouble ? NPL * b_plane : 0;  // distance between the two weight-tile images
  auto b_store = [&](u32x4 v, int q, int img) {
    *reinterpret_cast<u32x4*>(B_lds + img * b_img + q * b_plane + b_dst) = v;
  };

  f32x4 acc[NPF][NCF];
#pragma unroll
  for (int pf = 0; pf < NPF; ++pf)
#pragma unroll
    for (int cf = 0; cf < NCF; ++cf) acc[pf][cf] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nsteps = nchunks * p.ntaps;
  bf16x8 wfa[NPL][NCF], wfb[NPL][NCF];  // RF: this step's / the next step's weight fragments (ping-pong)
  __syncthreads();  // zero fill complete
  if (nsteps > 0) {
    if constexpr (RF) rf_load(wfa, p.tap_w[0], 0);
    a_stage_sync(0, 0, p.seg_goff[0]);
    if constexpr (!RF) {
#pragma unroll
      for (int q = 0; q < NPL; ++q) b_store(b_load(p.tap_w[0], 0, q), q, 0);
    }
  }
  __syncthreads();

  stamp(0);  // zero fill + synchronous staging of the first chunk / weight tile
  if constexpr (PROF == 2) { tr_t1 = __builtin_amdgcn_s_memrealtime(); tr_c1 = __builtin_amdgcn_s_memtime(); }
  const unsigned char* Bb0 = B_lds + li * 64 + ((kg * 16) ^ swz(li));
  // step order: segment, then 32-channel chunk, then the segment's taps; the input tile is restaged whenever the
  // (segment, chunk) pair changes
  int g = 0, c = 0, t = 0;
  long goff_cur = p.seg_goff[0];
  // The tap tables live in one register each (lane i holds entry i) and are read with v_readlane: a kernel-argument
  // load indexed by the step's tap has the latency of a cache access, shares its counter with the LDS reads, and sat
  // in front of every step's weight prefetch and fragment-read burst.
  static_assert(MAXTAPS <= 64, "one lane per tap");
  const int lane_toff = p.tap_off[lane < MAXTAPS ? lane : 0], lane_tw = p.tap_w[lane < MAXTAPS ? lane : 0];
  int seg_lo = p.seg_begin[0], seg_hi = p.seg_begin[1];  // taps of segment g (kept in registers: a kernel-argument
                                                         // load per step shows in the short steps of the bf16 kernels)
  bf16x8 af[NPL];  // activation fragment 0 of the coming step (read ahead)
  auto a_frag = [&](bf16x8 (&f)[NPL], int pf, int toff) {
    const int s = a_slot[pf] + toff;
    const unsigned char* Ab = A_lds + s * 64 + ((kg * 16) ^ swz(s));
#pragma unroll
    for (int q = 0; q < NPL; ++q) f[q] = *reinterpret_cast<const bf16x8*>(Ab + q * a_plane);
  };
#if BLADES_AF_AHEAD
  if (nsteps > 0) a_frag(af, 0, __builtin_amdgcn_readlane(lane_toff, seg_lo));
#endif
  // one K-step; RF: `wc` holds this step's weight fragments, `wn` receives the next step's (the caller alternates them)
  auto do_step = [&](bf16x8 (&wc)[NPL][NCF], bf16x8 (&wn)[NPL][NCF], const int ks) {
    int tn = t + 1, cn = c, gn = g;
    if (tn == seg_hi) {
      tn = seg_lo; cn = c + 1;
      if (cn == nchunks) { cn = 0; gn = g + 1; tn = seg_hi; }
    }
    const bool more = ks + 1 < nsteps;
    const bool next_b = more && !(p.ablate & (1 | 8));  // bit 3 (timing only, wrong results): no weight staging and
                                                       // no barriers inside a chunk = bound of a register-fed B path
    const bool next_a = more && (cn != c || gn != g) && !(p.ablate & (1 | 32));  // this step ends (g, c): restage A after it
    long goff_next = goff_cur;
    if (gn != g && more) goff_next = p.seg_goff[gn];
    const int toff = __builtin_amdgcn_readlane(lane_toff, t);
    // 1. global loads: next step's weight tile / fragments; at a chunk's last tap also the next chunk of the input
    //    tile.  They complete under this step's MFMAs (and under the MFMAs of the workgroup sharing the CU).
    u32x4 breg[NPL];
    if (next_b) {
      const int tw = __builtin_amdgcn_readlane(lane_tw, tn);
      if constexpr (RF) rf_load(wn, tw, cn);
      else {
#pragma unroll
        for (int q = 0; q < NPL; ++q) breg[q] = b_load(tw, cn * KC, q);
      }
    }
    u32x4 areg[MAXPF];
    if (next_a) {
#pragma unroll
      for (int j = 0; j < MAXPF; ++j) {
        areg[j] = u32x4{0, 0, 0, 0};
        if (j < p.npass) areg[j] = a_load(a_desc(j, goff_next), cn * KC);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(1);  // address arithmetic + issue of the global loads
    // 2. MFMAs of (chunk c, tap t)
    if (!(p.ablate & 4)) {
      bf16x8 wl[NPL][NCF];
      if constexpr (!RF) {
        const unsigned char* Bb = Bb0 + (ks & 1) * b_img;
#pragma unroll
        for (int q = 0; q < NPL; ++q)
#pragma unroll
          for (int cf = 0; cf < NCF; ++cf) wl[q][cf] = *reinterpret_cast<const bf16x8*>(Bb + q * b_plane + cf * 1024);
      }
      auto& wf = [&]() -> bf16x8 (&)[NPL][NCF] { if constexpr (RF) return wc; else return wl; }();
      // activation fragments are read one fragment ahead of the MFMAs that use them (the compiler keeps the source
      // order of LDS reads and MFMAs: read-then-multiply per fragment exposed the LDS latency eight times a step);
      // the first fragment of the NEXT step is read behind the last MFMAs of this one unless the tile is restaged
      const int toff_n = __builtin_amdgcn_readlane(lane_toff, tn);
#pragma unroll
      for (int pf = 0; pf < NPF; ++pf) {
        bf16x8 afn[NPL];
#if BLADES_AF_AHEAD
        if (pf + 1 < NPF) a_frag(afn, pf + 1, toff);
        else if (more && !next_a) a_frag(afn, 0, toff_n);
#else
        (void)toff_n;
        a_frag(af, pf, toff);
#endif
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf) {
          // (weight plane, activation plane) pairs, small terms first
          if constexpr (NPL == 3) {
            acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][cf], af[1], acc[pf][cf], 0, 0, 0);
            acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][cf], af[2], acc[pf][cf], 0, 0, 0);
            acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2][cf], af[0], acc[pf][cf], 0, 0, 0);
          }
          if constexpr (NPL >= 2) {
            acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][cf], af[1], acc[pf][cf], 0, 0, 0);
            acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][cf], af[0], acc[pf][cf], 0, 0, 0);
          }
          acc[pf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][cf], af[0], acc[pf][cf], 0, 0, 0);
        }
#if BLADES_AF_AHEAD
#pragma unroll
        for (int q = 0; q < NPL; ++q) af[q] = afn[q];
#endif
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(2);  // fragment reads + MFMA issue
    const bool sync_b = !RF && !(p.ablate & 8);  // the LDS-staged weight tile needs the workgroup every step
    if (sync_b || next_a) {
      // every wave has read this step's fragments — not needed for the weight tile alone when it goes to the other
      // image (its last readers finished before the barrier that ended the previous step)
      if (!p.bdouble || next_a || RF) __syncthreads();
      stamp(3);  // first barrier
      // 3. commit the staged data
      if constexpr (!RF) {
        if (next_b && !(p.ablate & 2)) {
#pragma unroll
          for (int q = 0; q < NPL; ++q) b_store(breg[q], q, (ks + 1) & 1);
        }
      }
      if (next_a && !(p.ablate & 2)) {
        // the descriptors are recomputed rather than kept from the load (an opaque copy of the offset stops the
        // compiler from keeping three registers per pass alive through the MFMAs)
        int pp_c = pp;
        if constexpr (RF) asm volatile("" : "+v"(pp_c));
#pragma unroll
        for (int j = 0; j < MAXPF; ++j)
          if (j < p.npass) a_store(a_desc(j, goff_next, pp_c), areg[j]);
        a_stage_sync(MAXPF, cn * KC, goff_next);  // geometries with more than MAXPF passes: the rest without prefetch
      }
      if (next_a) stamp(5); else stamp(4);  // commit: wait for the loads, split, LDS stores (4: weight tile only)
      __syncthreads();
      stamp(6);  // second barrier
#if BLADES_AF_AHEAD
      if (next_a && !(p.ablate & 4)) a_frag(af, 0, __builtin_amdgcn_readlane(lane_toff, tn));
#endif
    }
    if (gn != g && more) { seg_lo = seg_hi; seg_hi = p.seg_begin[gn + 1]; }
    t = tn; c = cn; g = gn; goff_cur = goff_next;
  };
  if constexpr (RF) {
#ifndef BLADES_RF_UNROLL
    for (int ks = 0; ks < nsteps; ++ks) {
      do_step(wfa, wfb, ks);
#pragma unroll
      for (int q = 0; q < NPL; ++q)
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf) wfa[q][cf] = wfb[q][cf];
    }
#else
    for (int ks = 0; ks < nsteps; ks += 2) {
      do_step(wfa, wfb, ks);
      if (ks + 1 < nsteps) do_step(wfb, wfa, ks + 1);
    }
#endif
  } else {
    for (int ks = 0; ks < nsteps; ++ks) do_step(wfa, wfa, ks);
  }

  if constexpr (PROF == 2) { tr_t2 = __builtin_amdgcn_s_memrealtime(); tr_c2 = __builtin_amdgcn_s_memtime(); }
  // ---- epilogue: lane = (pixel li of fragment pf, channels cf*16 + kg*4 .. +3) -------------------------------
  const bool vec_out = (p.Co % 4) == 0;
  const bool want_stats = SPLIT && p.stats != nullptr;  // block-uniform
  const bool bstat = want_stats && p.bstat_x != nullptr;  // BatchNorm-backward sums instead of the forward ones
  f32x4 st_s[NCF], st_q[NCF];
#pragma unroll
  for (int cf = 0; cf < NCF; ++cf) { st_s[cf] = f32x4{0.f, 0.f, 0.f, 0.f}; st_q[cf] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int pf = 0; pf < NPF; ++pf) {
    int slot_unused;
    const long out_pix_pf = pixel_of(pf, slot_unused);
    if (out_pix_pf < 0) continue;
#pragma unroll
    for (int cf = 0; cf < NCF; ++cf) {
      const int co = n0 + cbase + cf * 16 + kg * 4;
      if (co >= p.Co) continue;
      f32x4 v = acc[pf][cf];
      if (bias_c != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += co + j < p.Co ? bias_c[co + j] : 0.f;
      }
      if (want_stats && !bstat) { st_s[cf] += v; st_q[cf] += v * v; }
      if constexpr (SPLIT) {
        float* dst = reinterpret_cast<float*>(p.out) + out_pix_pf * p.Co + co;
        if (vec_out) {
          if (p.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
          *reinterpret_cast<f32x4*>(dst) = v;
          if (bstat) {  // (Co % 8 == 0 is checked on the host)
            const size_t off = (size_t)out_pix_pf * p.Co + co;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(p.bstat_x + off);
            const unsigned mb = p.bstat_mask != nullptr ? (unsigned)p.bstat_mask[off >> 3] >> (co & 4) : 0xFu;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float g = (mb >> j) & 1u ? v[j] : 0.f;
              st_s[cf][j] += g; st_q[cf][j] += g * xv[j];
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (co + j < p.Co) dst[j] = p.accumulate ? dst[j] + v[j] : v[j];
        }
      } else {
        unsigned short* dst = reinterpret_cast<unsigned short*>(p.out) + out_pix_pf * p.Co + co;
        if (p.accumulate) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < p.Co) v[j] += __uint_as_float((unsigned)dst[j] << 16);
        }
        const u32x2 b = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
        if (vec_out) *reinterpret_cast<u32x2*>(dst) = b;
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (co + j < p.Co) dst[j] = (unsigned short)(b[j >> 1] >> ((j & 1) * 16));
        }
      }
    }
  }
  if constexpr (SPLIT) {
    if (want_stats) {
      // BatchNorm statistics of this tile: fixed-order tree over the 16 pixel lanes, then over the waves that hold
      // the channel (all four; RF: the two pixel halves) through LDS
#pragma unroll
      for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            st_s[cf][j] += __shfl_xor(st_s[cf][j], o);
            st_q[cf][j] += __shfl_xor(st_q[cf][j], o);
          }
      __syncthreads();  // every wave is past its last fragment read (RF has no barrier at the end of the K loop)
      float* red = reinterpret_cast<float*>(smem);  // [pixel groups][64 ch][2]
      if (li == 0) {
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ch = cbase + cf * 16 + kg * 4 + j;
            red[(wpix * 64 + ch) * 2 + 0] = st_s[cf][j];
            red[(wpix * 64 + ch) * 2 + 1] = st_q[cf][j];
          }
      }
      __syncthreads();
      if (tid < 128) {
        const int ch = tid >> 1, which = tid & 1;
        if (n0 + ch < p.Co) {
          float a = 0.f;
#pragma unroll
          for (int w = 0; w < (RF ? 2 : 4); ++w) a += red[(w * 64 + ch) * 2 + which];
          const int cl = img0 / p.stats_B, in_cl = img0 - cl * p.stats_B;
          const int chunk = p.TI == 1 ? in_cl * p.tiles_per_img + (mt - img0 * p.tiles_per_img) : in_cl / p.TI;
          p.stats[(((size_t)cl * p.stats_nchunk + p.stats_chunk0 + chunk) * p.Co + n0 + ch) * 2 + which] = a;
        }
      }
    }
  }
  if constexpr (PROF == 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tr_t3 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && p.prof != nullptr && lid < (1 << 20)) {
      unsigned long long* r = p.prof + 16 + (size_t)lid * 8;
      r[0] = tr_t0; r[1] = tr_t1; r[2] = tr_t2; r[3] = tr_t3; r[4] = tr_c2 - tr_c1;
      r[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
      r[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
      r[7] = blockIdx.x;
    }
  }
  if constexpr (PROF == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(7);  // epilogue, stores acknowledged
    if (lane == 0 && p.prof != nullptr) {
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(p.prof + i, pf_t[i]);
      atomicAdd(p.prof + 8, 1ull);
    }
  }
}

// Weight planes.  w is the PyTorch tensor [Co][Ci][taps] (fp32, contiguous).  With rows = produced channels and
// k = reduced channels (forward: rows = Co, k = Ci; transpose = 1, data gradient: rows = Ci, k = Co), padded to
// multiples of 64 / 32 with zeros, the planes are stored in the order the kernels consume them:
//   [tap][32-wide k chunk][64-row tile][32-row half][plane][16-row fragment][lane = (k % 32 / 8) * 16 + row % 16][k % 8]
// i.e. one (tap, chunk, tile) is 2 x NPL x 2 MFMA A-operand fragments of 1 KiB, each read by one wave instruction.
__global__ __launch_bounds__(256) void conv_wprep_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp,
                                                         int Co, int Ci, int taps, int rows, int rows_pad, int k,
                                                         int k_pad, int transpose, int npl, long w_client_stride) {
  // one thread = one lane's 8 consecutive k of a fragment (16 bytes per plane): with per-client weights (local steps
  // > 1) this pass runs per layer, client and step, so it stores 16 bytes at a time instead of 2
  const size_t total = (size_t)taps * rows_pad * k_pad;
  w += (size_t)blockIdx.y * w_client_stride;          // one weight set per client (grid.y), planes [client][..]
  wp += (size_t)blockIdx.y * npl * total;
  const int nchunks = k_pad / KC, ntiles = rows_pad / BN;
  const size_t n8 = total >> 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    // i enumerates the destination order without the plane index: (t, chunk, tile, half, frag, lane)
    const int ln = (int)(i & 63), frag = (int)((i >> 6) & 1), half = (int)((i >> 7) & 1);
    size_t r = i >> 8;
    const int tile = (int)(r % ntiles); r /= ntiles;
    const int chunk = (int)(r % nchunks);
    const int t = (int)(r / nchunks);
    const int row = tile * BN + half * 32 + frag * 16 + (ln & 15), kk0 = chunk * KC + (ln >> 4) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kk = kk0 + e;
      v[e] = 0.f;
      if (row < rows && kk < k) {
        const int co = transpose ? kk : row, ci = transpose ? row : kk;
        v[e] = w[((size_t)co * Ci + ci) * taps + t];
      }
    }
    const size_t grp = i >> 7;  // (t, chunk, tile, half)
    for (int q = 0; q < npl; ++q) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const __bf16 b0 = (__bf16)v[e], b1 = (__bf16)v[e + 1];
        o[e >> 1] = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
        v[e] -= (float)b0; v[e + 1] -= (float)b1;
      }
      *reinterpret_cast<u32x4*>(wp + ((grp * npl + q) * 2 + frag) * 512 + (size_t)ln * 8) = o;
    }
  }
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// BLADES_IGEMM_RF: 1 = register-fed weights everywhere, 0 = nowhere; default: where the LDS that the weight tile no
// longer takes buys a larger pixel tile (the 4x4 / 512-channel layers: 256 instead of 128 pixels, measured -12 %;
// at equal tile size the two paths measured within 1-2 % of each other on the stride-1 layers and the LDS-staged
// one faster on the short chunks of the strided segments: profiles/r3_conv_igemm_rf_ab_same_box.log)
inline int rf_mode() {  // (read per call: the tests switch it inside one process)
  const char* e = getenv("BLADES_IGEMM_RF");
  return e ? (atoi(e) != 0 ? 1 : 0) : -1;
}

// tile choice over both weight paths; sets `rf`
int plan(IgemmParams& p, int npl, const int* ay, const int* ax, bool rf);
int choose(IgemmParams& p, int npl, const int* ay, const int* ax, bool& rf) {
  IgemmParams a = p, b = p;
  const int mf_lds = rf_mode() == 1 ? 0 : plan(a, npl, ay, ax, false);
  const int mf_rf = rf_mode() == 0 ? 0 : plan(b, npl, ay, ax, true);
  rf = mf_rf > mf_lds;
  p = rf ? b : a;
  return rf ? mf_rf : mf_lds;
}

inline bool no_bdouble() {
  static const bool off = [] { const char* e = getenv("BLADES_IGEMM_BDOUBLE"); return e && atoi(e) == 0; }();
  return off;
}

// Tile choice + geometry for one launch.  Returns the number of pixel fragments per wave (4, 2, 1), 0 = no fit.
int plan(IgemmParams& p, int npl, const int* ay, const int* ax, bool rf) {
  int aymin = 0, aymax = 0, axmin = 0, axmax = 0;
  for (int t = 0; t < p.ntaps; ++t) {
    if (t == 0 || ay[t] < aymin) aymin = ay[t];
    if (t == 0 || ay[t] > aymax) aymax = ay[t];
    if (t == 0 || ax[t] < axmin) axmin = ax[t];
    if (t == 0 || ax[t] > axmax) axmax = ax[t];
  }
  p.aymin = aymin;
  p.PL = axmin < 0 ? -axmin : 0;
  int overflow = (p.Ws - 1) * p.si + axmax - (p.Wi - 1);
  if (overflow < 0) overflow = 0;
  const int hpad = p.PL > overflow ? p.PL : overflow;
  p.WS = p.Wi + hpad;
  int padT = aymin < 0 ? -aymin : 0;
  int padB = (p.Hs - 1) * p.si + aymax - (p.Hi - 1);
  if (padB < 0) padB = 0;
  const int share = padT < padB ? padT : padB;
  int best_mf = 0;
  double best_cost = 0;
  IgemmParams best = p;
  // BLADES_IGEMM_MF (tests, A/B runs): this tile size where it fits the geometry, else the planner's own choice
  const char* force = getenv("BLADES_IGEMM_MF");
  for (int pass = force ? 0 : 1; pass < 2 && best_mf == 0; ++pass)
  for (int mf = 4; mf >= 1; mf >>= 1) {
    const int BM = 64 * mf;
    if (p.Ws > BM) continue;
    if (pass == 0 && atoi(force) != mf) continue;
    // candidates: TI whole images per tile (largest first), then row bands of one image (TR rows, fewest first
    // that fits the 160 KB of LDS: the staged receptive field grows with stride and kernel size)
    int ti_max = p.Hs * p.Ws * 2 <= BM ? BM / (p.Hs * p.Ws) : 1;
    if (p.n_clients > 1)
      while (ti_max > 1 && p.imgs_per_client % ti_max) --ti_max;
    int tr_max = BM / p.Ws;
    if (tr_max > p.Hs) tr_max = p.Hs;
    for (int ti = ti_max, tr = tr_max; ti >= 1 && tr >= 1;) {
      IgemmParams q = p;
      if (ti > 1) {
        q.TI = ti; q.TR = p.Hs; q.tiles_per_img = 1;
        q.n_mtiles = (p.N + q.TI - 1) / q.TI;
      } else {
        q.TI = 1;
        q.tiles_per_img = (p.Hs + tr - 1) / tr;
        q.TR = (p.Hs + q.tiles_per_img - 1) / q.tiles_per_img;
        q.n_mtiles = p.N * q.tiles_per_img;
      }
      q.IR = (q.TR - 1) * p.si + (aymax - aymin) + 1;
      const int rows = q.TI > 1 ? (q.TI - 1) * (q.IR - share) + q.IR : q.IR;
      q.IS = q.TI > 1 ? (q.IR - share) * p.WS : q.IR * p.WS;
      q.NS = round_up(rows * p.WS + hpad + p.PL + 1, 4);
      long lds = (long)npl * q.NS * 64 + (rf ? 0L : 2L * npl * BN * 64);
      q.bdouble = rf ? 0 : 1;
      if (!rf && (lds > LDS_TOTAL / wgs_per_cu(mf) || no_bdouble())) { lds -= (long)npl * BN * 64; q.bdouble = 0; }
      if (!rf && p.cr_taps > 0) { lds = (long)q.NS * 64 + (long)p.cr_taps * BN * 64; q.bdouble = 0; }  // (npl = 1)
      q.region_px = q.TI * q.IR * p.Wi;
      const bool fits = lds <= LDS_TOTAL / wgs_per_cu(mf) && q.region_px < (1 << 20) && p.Wi < 4096 && q.IR * p.Wi < 4096 * 64;
      if (fits) {
        // cost = rounds of resident workgroups x tile size x per-MFMA overhead of the smaller tiles.  A launch that
        // does not fill the chip once (the Linear layers of the small models, run as 1x1 convolutions over [M, K, 1, 1])
        // takes as long as one tile: smaller tiles then win.
        const long tiles = (long)q.n_mtiles * (p.Cop / BN);
        const long slots = 256L * wgs_per_cu(mf);
        const double rounds = tiles > slots ? (double)tiles / slots : 1.0;
        const double cost = rounds * BM * (mf == 4 ? 1.0 : mf == 2 ? 1.12 : 1.35);
        if (best_mf == 0 || cost < best_cost) { best_mf = mf; best_cost = cost; best = q; }
        break;
      }
      if (ti > 1) {
        ti >>= 1;
        if (p.n_clients > 1)
          while (ti > 1 && p.imgs_per_client % ti) --ti;
      } else --tr;
    }
  }
  if (best_mf) p = best;
  return best_mf;
}

// tiles per client of a planned forward launch (= BatchNorm partial chunks); 0: a tile would straddle two clients
int stats_chunks(const IgemmParams& p, int B) {
  if (B <= 0 || p.N % B != 0) return 0;
  if (p.TI > 1) return B % p.TI == 0 ? B / p.TI : 0;
  return B * p.tiles_per_img;
}

// BLADES_IGEMM_PROF=1: device buffer of the phase-cycle sums (16 x u64), read and cleared by blades_conv_igemm_prof
int prof_mode() {
  static const int m = [] { const char* e = getenv("BLADES_IGEMM_PROF"); return e ? atoi(e) : 0; }();
  return m == 1 || m == 2 ? m : 0;
}
constexpr size_t TRACE_TILES = 1 << 20;  // mode 2: one 8-word record per tile behind the 16 sums
unsigned long long* prof_buffer() {
  static unsigned long long* buf = [] {
    unsigned long long* b = nullptr;
    const size_t words = 16 + (prof_mode() == 2 ? TRACE_TILES * 8 : 0);
    if (prof_mode() != 0 && hipMalloc(&b, words * sizeof(unsigned long long)) == hipSuccess)
      (void)hipMemset(b, 0, words * sizeof(unsigned long long));
    return b;
  }();
  return buf;
}

template <int NPL, int MF, bool SPLIT, int PROF, bool RF>
void launch1(const IgemmParams& p, int grid, size_t lds, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<NPL, MF, SPLIT, PROF, RF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL / wgs_per_cu(MF));
    attr = true;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<NPL, MF, SPLIT, PROF, RF>), dim3(grid), dim3(256), lds, st, p);
}

template <int NPL, int MF, bool SPLIT>
int launch(const IgemmParams& p, hipStream_t st, bool rf) {
  size_t lds = (size_t)NPL * p.NS * 64 + (rf ? 0 : (size_t)(p.bdouble ? 2 : 1) * NPL * BN * 64);
  if (!rf && p.cr_taps > 0) lds = (size_t)p.NS * 64 + (size_t)p.cr_taps * BN * 64;
  const int blocks = p.n_mtiles * p.n_ntiles;
  const int grid = (blocks + 7) / 8 * 8;
  if constexpr (NPL == 3 && MF >= 2 && SPLIT) {
    if (prof_buffer() != nullptr) {
      IgemmParams q = p;
      q.prof = prof_buffer();
      if (prof_mode() == 2) rf ? launch1<3, MF, true, 2, true>(q, grid, lds, st) : launch1<3, MF, true, 2, false>(q, grid, lds, st);
      else rf ? launch1<3, MF, true, 1, true>(q, grid, lds, st) : launch1<3, MF, true, 1, false>(q, grid, lds, st);
      return 0;
    }
  }
  if (rf) { launch1<NPL, MF, SPLIT, 0, true>(p, grid, lds, st); return 0; }
  if constexpr (NPL == 1 && !SPLIT) {
    if (p.cr_taps > 0) {
      static bool attr_cr = false;
      if (!attr_cr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_v1_kernel<1, MF, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL / wgs_per_cu(MF));
        attr_cr = true;
      }
      hipLaunchKernelGGL((conv_igemm_v1_kernel<1, MF, false, true>), dim3(grid), dim3(256), lds, st, p);
      return 0;
    }
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_v1_kernel<NPL, MF, SPLIT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL / wgs_per_cu(MF));
    attr = true;
  }
  hipLaunchKernelGGL((conv_igemm_v1_kernel<NPL, MF, SPLIT>), dim3(grid), dim3(256), lds, st, p);
  return 0;
}

// pixel -> (fragment, lane) assignment of a tile, see IgemmParams::perm
void build_perm(IgemmParams& p, int mf) {
  const int BM = 64 * mf, nfrag = BM / 16;
  for (int i = 0; i < 256; ++i) p.perm[i] = (unsigned short)(i < BM ? i : 0);
  { const char* e = getenv("BLADES_IGEMM_PERM"); if (e && atoi(e) == 0) return; }
  // LDS slot of tile pixel m (the kernel's own formula; rows beyond the image keep the formula's slot: they only
  // occur in a tile's last row band)
  int slot[256];
  for (int m = 0; m < BM; ++m) {
    const int q = m / p.Ws, sx = m - q * p.Ws;
    const int il = q / p.TR, ry = q - il * p.TR;
    slot[m] = il < p.TI ? il * p.IS + ry * p.si * p.WS + sx * p.si + p.PL : p.PL;
  }
  auto clean = [&](const unsigned short* f) {  // conflict-free fragment?
    unsigned a = 0, b = 0;
    for (int li = 0; li < 16; ++li) {
      const unsigned bit = 1u << (slot[f[li]] & 7);
      unsigned& set = (li >= 4 && li < 12) ? a : b;
      if (set & bit) return false;
      set |= bit;
    }
    return true;
  };
  bool all_clean = true;
  for (int f = 0; f < nfrag; ++f) all_clean = all_clean && clean(p.perm + f * 16);
  if (all_clean) return;
  // deal the pixels by residue: every fragment needs two pixels of each residue class (one per lane set)
  int bucket[8][256], cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int m = 0; m < BM; ++m) { const int r = slot[m] & 7; bucket[r][cnt[r]++] = m; }
  for (int r = 0; r < 8; ++r)
    if (cnt[r] != 2 * nfrag) return;  // unbalanced geometry: keep the identity
  static const int lanes_a[8] = {4, 5, 6, 7, 8, 9, 10, 11}, lanes_b[8] = {0, 1, 2, 3, 12, 13, 14, 15};
  for (int f = 0; f < nfrag; ++f)
    for (int r = 0; r < 8; ++r) {
      p.perm[f * 16 + lanes_a[r]] = (unsigned short)bucket[r][2 * f];
      p.perm[f * 16 + lanes_b[r]] = (unsigned short)bucket[r][2 * f + 1];
    }
}

int dispatch(IgemmParams& p, int npl, const int* ay, const int* ax, hipStream_t st) {
  bool rf = false;
  p.cr_taps = 0;
  if (npl == 1) {  // bf16 tensors: chunk-resident weights where a segment has at most 9 taps (BLADES_IGEMM_CR=0: off)
    int tmax = p.nseg == 0 ? p.ntaps : 0;
    for (int g = 0; g < p.nseg; ++g) tmax = std::max(tmax, p.seg_begin[g + 1] - p.seg_begin[g]);
    const char* e = getenv("BLADES_IGEMM_CR");
    if (tmax >= 1 && tmax <= 9 && !(e && atoi(e) == 0)) p.cr_taps = tmax;
  }
  int mf = choose(p, npl, ay, ax, rf);
  if (mf == 0 && p.cr_taps > 0) { p.cr_taps = 0; mf = choose(p, npl, ay, ax, rf); }
  if (mf == 0) return 3;
  if (rf) p.cr_taps = 0;
  const int PIXPP = npl == 1 ? 64 : 32;
  p.npass = (p.region_px + PIXPP - 1) / PIXPP;
  p.inv_ws = 1.0f / p.Ws; p.inv_tr = 1.0f / p.TR; p.inv_wi = 1.0f / p.Wi; p.inv_irwi = 1.0f / (p.IR * p.Wi);
  for (int t = 0; t < p.ntaps; ++t) p.tap_off[t] = (ay[t] - p.aymin) * p.WS + ax[t];
  p.n_ntiles = p.Cop / BN;
  build_perm(p, mf);
  if (p.stats != nullptr) {
    const int per = stats_chunks(p, p.stats_B);
    if (per == 0 || npl == 1) return 6;
    const int classes = p.stats_classes > 0 ? p.stats_classes : 1;
    p.stats_nchunk = per * classes;
    p.stats_chunk0 = p.stats_class * per;
  }
  if (p.nseg == 0) { p.nseg = 1; p.seg_begin[0] = 0; p.seg_begin[1] = p.ntaps; p.seg_goff[0] = 0; }
  { const char* e = getenv("BLADES_IGEMM_ABLATE"); p.ablate = e ? atoi(e) : 0; }
#define GO(NPL, SPLIT)                                           \
  do {                                                           \
    if (mf == 4) return launch<NPL, 4, SPLIT>(p, st, rf);        \
    if (mf == 2) return launch<NPL, 2, SPLIT>(p, st, rf);        \
    return launch<NPL, 1, SPLIT>(p, st, rf);                     \
  } while (0)
  if (npl == 3) GO(3, true);
  if (npl == 2) GO(2, true);
  if (npl == 1) GO(1, false);
#undef GO
  return 4;
}

// geometry and tap table of a forward convolution (p.N and the channel paddings are set by the caller)
void fill_forward(IgemmParams& p, int* ay, int* ax, int H, int W, int Ci, int Ho, int Wo, int Co, int kh, int kw,
                  int stride, int pad) {
  const int taps = kh * kw;
  p.in_rm = p.in_cm = 1;
  p.Hi = H; p.Wi = W; p.Ci = Ci; p.Hout = Ho; p.Wout = Wo; p.Co = Co;
  p.Hs = Ho; p.Ws = Wo; p.so = 1; p.poy = 0; p.pox = 0; p.si = stride;
  p.Hphys = H; p.Wphys = W;
  if (taps == 1 && pad == 0 && stride > 1) {  // strided 1x1: stride-1 on the view in[:, ::stride, ::stride]
    p.Hi = Ho; p.Wi = Wo; p.si = 1; p.in_rm = p.in_cm = stride;
  }
  p.ntaps = taps;
  static const bool seg_off = [] { const char* e = getenv("BLADES_IGEMM_SEGMENTS"); return e && atoi(e) == 0; }();
  if (taps > 1 && stride > 1 && H % stride == 0 && W % stride == 0 && stride * stride <= MAXSEG && !seg_off) {
    // one reduction segment per input parity class: stride-1 taps on the view in[:, py::stride, px::stride]
    auto fl = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
    p.Hi = H / stride; p.Wi = W / stride; p.si = 1; p.in_rm = p.in_cm = stride;
    int n = 0;
    p.nseg = 0;
    for (int py = 0; py < stride; ++py)
      for (int px = 0; px < stride; ++px) {
        const int n0 = n;
        for (int r = 0; r < kh; ++r) {
          const int a = fl(r - pad, stride);
          if (r - pad - a * stride != py) continue;
          for (int s = 0; s < kw; ++s) {
            const int b = fl(s - pad, stride);
            if (s - pad - b * stride != px) continue;
            ay[n] = a; ax[n] = b; p.tap_w[n] = r * kw + s;
            ++n;
          }
        }
        if (n == n0) continue;
        p.seg_begin[p.nseg] = n0;
        p.seg_goff[p.nseg] = ((long)py * W + px) * Ci;
        ++p.nseg;
      }
    p.seg_begin[p.nseg] = n;
    return;
  }
  for (int r = 0; r < kh; ++r)
    for (int s = 0; s < kw; ++s) { ay[r * kw + s] = r - pad; ax[r * kw + s] = s - pad; p.tap_w[r * kw + s] = r * kw + s; }
}

}  // namespace

// bytes of weight-plane workspace a call needs
BLADES_API long blades_conv_igemm_ws(int Co, int Ci, int kh, int kw, int mode, int npl, int n_clients) {
  const int rows = mode == 0 ? Co : Ci, k = mode == 0 ? Ci : Co;
  return (long)(n_clients < 1 ? 1 : n_clients) * npl * kh * kw * round_up(rows, BN) * round_up(k, KC) * 2;
}

// Weight planes of one layer: `ws` (blades_conv_igemm_ws bytes) <- split / transposed / padded copy of `w`.  The
// FedSGD engine calls this once per layer and round (the weights are constant while the clients train) and hands
// the planes to every chunk's blades_conv_igemm2(..., flags & 1).
BLADES_API int blades_conv_wprep(const float* w, void* ws, int Co, int Ci, int kh, int kw, int mode, int npl,
                                 int n_clients, long w_client_stride, hipStream_t st) {
  if (npl < 1 || npl > 3 || kh * kw > MAXTAPS) return 1;
  if (n_clients < 1) n_clients = 1;
  if (n_clients > 65535) return 5;
  const int taps = kh * kw;
  const int rows = mode == 0 ? Co : Ci, k = mode == 0 ? Ci : Co;
  const int rows_pad = round_up(rows, BN), k_pad = round_up(k, KC);
  const size_t total = (size_t)taps * rows_pad * k_pad;
  int gx = cdiv((long)(total / 8), 256);
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(conv_wprep_kernel, dim3(gx, n_clients), dim3(256), 0, st, w, (unsigned short*)ws, Co, Ci, taps,
                     rows, rows_pad, k, k_pad, mode, npl, w_client_stride);
  return 0;
}

// mode 0: forward    in = x  [N,H,W,Ci]   -> out = y  [N,Ho,Wo,Co]   (bias optional)
// mode 1: data grad  in = gy [N,Ho,Wo,Co] -> out = dx [N,H,W,Ci]
// w: [Co][Ci][kh][kw] fp32 contiguous.  npl 3 / 2: fp32 tensors; npl 1: bf16 tensors.
// n_clients > 1: per-client weights — image n uses the weight set n / (N / n_clients), found at
// w + client * w_client_stride (bias + client * bias_client_stride); the FedSGD fast path passes n_clients = 1.
// stats != null (forward, fp32 tensors): also writes the per-client BatchNorm partial sums of the output,
// [N / stats_B clients][blades_conv_igemm_stats_nchunk(...) chunks][Co][2] floats.
// flags: 1 = `ws` already holds this layer's planes (blades_conv_wprep; `w` is not read); 2 = out += result.
// returns 0 ok; nonzero = unsupported (caller falls back)
// bstat_x / bstat_mask (mode 1, fp32 tensors, stats != null): instead of the forward sums, `stats` receives the raw
// BatchNorm-backward sums of the finished data gradient against bstat_x (see IgemmParams), in
// blades_conv_dgrad_stats_nchunk(...) chunks per client.
BLADES_API int blades_conv_igemm3(const void* in, const float* w, const float* bias, void* out, void* ws, int N, int H,
                                  int W, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad, int mode,
                                  int npl, int n_clients, long w_client_stride, long bias_client_stride,
                                  float* stats, int stats_B, int flags, const float* bstat_x,
                                  const unsigned char* bstat_mask, hipStream_t st) {
  if (npl < 1 || npl > 3 || kh * kw > MAXTAPS || stride < 1 || N <= 0) return 1;
  if (n_clients < 1) n_clients = 1;
  if (N % n_clients || n_clients > 65535) return 5;
  if (Ho != (H + 2 * pad - kh) / stride + 1 || Wo != (W + 2 * pad - kw) / stride + 1) return 2;
  const int taps = kh * kw;
  const int rows = mode == 0 ? Co : Ci, k = mode == 0 ? Ci : Co;
  const int rows_pad = round_up(rows, BN), k_pad = round_up(k, KC);
  if (!(flags & 1)) {
    const int rc = blades_conv_wprep(w, ws, Co, Ci, kh, kw, mode, npl, n_clients, w_client_stride, st);
    if (rc != 0) return rc;
  }
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.wp = (const unsigned short*)ws; p.out = out; p.bias = mode == 0 ? bias : nullptr;
  p.N = N; p.ntw = taps; p.Cop = rows_pad; p.Cip = k_pad;
  p.n_clients = n_clients; p.imgs_per_client = N / n_clients;
  p.wp_client_stride = (long)npl * taps * rows_pad * k_pad; p.bias_client_stride = bias_client_stride;
  p.accumulate = (flags & 2) ? 1 : 0;
  int ay[MAXTAPS], ax[MAXTAPS];
  p.in_rm = p.in_cm = 1;
  if (mode == 0) {
    p.stats = stats; p.stats_B = stats_B;
    fill_forward(p, ay, ax, H, W, Ci, Ho, Wo, Co, kh, kw, stride, pad);
    return dispatch(p, npl, ay, ax, st);
  }
  if (stats != nullptr) {
    if (bstat_x == nullptr || npl == 1 || Ci % 8 != 0) return 6;
    // every output parity class must be one launch of the same size (the chunk ranges of the classes are laid side
    // by side), and must have a tap (a class without one is skipped in accumulate mode)
    if (H % stride != 0 || W % stride != 0 || kh < stride || kw < stride) return 6;
    p.stats = stats; p.stats_B = stats_B; p.bstat_x = bstat_x; p.bstat_mask = bstat_mask;
    p.stats_classes = stride * stride;
  }
  p.Hi = Ho; p.Wi = Wo; p.Ci = Co; p.Hout = H; p.Wout = W; p.Co = Ci;
  p.Hphys = Ho; p.Wphys = Wo;
  p.so = stride; p.si = 1;
  for (int ph = 0; ph < stride; ++ph)
    for (int pw = 0; pw < stride; ++pw) {
      if (ph >= H || pw >= W) continue;
      IgemmParams q = p;
      q.poy = ph; q.pox = pw;
      q.stats_class = ph * stride + pw;
      q.Hs = (H - ph + stride - 1) / stride; q.Ws = (W - pw + stride - 1) / stride;
      int n = 0;
      for (int r = 0; r < kh; ++r) {
        if ((ph + pad - r) % stride != 0) continue;
        for (int s = 0; s < kw; ++s) {
          if ((pw + pad - s) % stride != 0) continue;
          ay[n] = (ph + pad - r) / stride; ax[n] = (pw + pad - s) / stride; q.tap_w[n] = r * kw + s;
          ++n;
        }
      }
      q.ntaps = n;
      if (n == 0 && q.accumulate) continue;  // no tap reaches this parity class: nothing to add
      const int rc = dispatch(q, npl, ay, ax, st);
      if (rc != 0) return rc;
    }
  return 0;
}

// diagnostics: copies the 16 phase-cycle sums of the PROF kernel to `host` and clears them; 1 = profiling is off
// mode 2: copies the first `tiles` trace records (8 words each) of the last launch to `host`
BLADES_API int blades_conv_igemm_trace(unsigned long long* host, long tiles) {
  unsigned long long* b = prof_buffer();
  if (b == nullptr || prof_mode() != 2 || tiles > (long)TRACE_TILES) return 1;
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  if (hipMemcpy(host, b + 16, (size_t)tiles * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  return 0;
}

BLADES_API int blades_conv_igemm_prof(unsigned long long* host) {
  unsigned long long* b = prof_buffer();
  if (b == nullptr) return 1;
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  if (hipMemcpy(host, b, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  (void)hipMemset(b, 0, 16 * sizeof(unsigned long long));
  return 0;
}

BLADES_API int blades_conv_igemm2(const void* in, const float* w, const float* bias, void* out, void* ws, int N, int H,
                                  int W, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad, int mode,
                                  int npl, int n_clients, long w_client_stride, long bias_client_stride,
                                  float* stats, int stats_B, int flags, hipStream_t st) {
  return blades_conv_igemm3(in, w, bias, out, ws, N, H, W, Ci, Ho, Wo, Co, kh, kw, stride, pad, mode, npl, n_clients,
                            w_client_stride, bias_client_stride, stats, stats_B, flags, nullptr, nullptr, st);
}

BLADES_API int blades_conv_igemm(const void* in, const float* w, const float* bias, void* out, void* ws, int N, int H,
                                 int W, int Ci, int Ho, int Wo, int Co, int kh, int kw, int stride, int pad, int mode,
                                 int npl, int n_clients, long w_client_stride, long bias_client_stride,
                                 float* stats, int stats_B, hipStream_t st) {
  return blades_conv_igemm2(in, w, bias, out, ws, N, H, W, Ci, Ho, Wo, Co, kh, kw, stride, pad, mode, npl, n_clients,
                            w_client_stride, bias_client_stride, stats, stats_B, 0, st);
}

// Chunks per client of the BatchNorm partial sums a forward call with `stats` writes; 0 = not available for this
// geometry (a tile would straddle two clients, or no tile shape fits).
BLADES_API int blades_conv_igemm_stats_nchunk(int N, int H, int W, int Ci, int Ho, int Wo, int Co, int kh, int kw,
                                              int stride, int pad, int npl, int n_clients, int stats_B) {
  if (npl < 2 || npl > 3 || kh * kw > MAXTAPS || stride < 1 || N <= 0 || stats_B <= 0) return 0;
  if (n_clients < 1) n_clients = 1;
  if (N % n_clients) return 0;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.N = N; p.ntw = kh * kw; p.Cop = round_up(Co, BN); p.Cip = round_up(Ci, KC);
  p.n_clients = n_clients; p.imgs_per_client = N / n_clients;
  int ay[MAXTAPS], ax[MAXTAPS];
  fill_forward(p, ay, ax, H, W, Ci, Ho, Wo, Co, kh, kw, stride, pad);
  bool rf = false;
  if (choose(p, npl, ay, ax, rf) == 0) return 0;
  return stats_chunks(p, stats_B);
}

// Chunks per client of the BatchNorm-backward sums a data-gradient call with bstat_x writes; 0 = not available.
BLADES_API int blades_conv_dgrad_stats_nchunk(int N, int H, int W, int Ci, int Ho, int Wo, int Co, int kh, int kw,
                                              int stride, int pad, int npl, int n_clients, int stats_B) {
  if (npl < 2 || npl > 3 || kh * kw > MAXTAPS || stride < 1 || N <= 0 || stats_B <= 0 || Ci % 8 != 0) return 0;
  if (H % stride != 0 || W % stride != 0 || kh < stride || kw < stride) return 0;
  if (n_clients < 1) n_clients = 1;
  if (N % n_clients) return 0;
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.N = N; p.ntw = kh * kw; p.Cop = round_up(Ci, BN); p.Cip = round_up(Co, KC);
  p.n_clients = n_clients; p.imgs_per_client = N / n_clients;
  p.in_rm = p.in_cm = 1;
  p.Hi = Ho; p.Wi = Wo; p.Ci = Co; p.Hout = H; p.Wout = W; p.Co = Ci;
  p.Hphys = Ho; p.Wphys = Wo; p.so = stride; p.si = 1;
  int per = -1;
  for (int ph = 0; ph < stride; ++ph)
    for (int pw = 0; pw < stride; ++pw) {
      IgemmParams q = p;
      int ay[MAXTAPS], ax[MAXTAPS];
      q.poy = ph; q.pox = pw;
      q.Hs = (H - ph + stride - 1) / stride; q.Ws = (W - pw + stride - 1) / stride;
      int n = 0;
      for (int r = 0; r < kh; ++r) {
        if ((ph + pad - r) % stride != 0) continue;
        for (int s = 0; s < kw; ++s) {
          if ((pw + pad - s) % stride != 0) continue;
          ay[n] = (ph + pad - r) / stride; ax[n] = (pw + pad - s) / stride; ++n;
        }
      }
      q.ntaps = n;
      if (n == 0) return 0;
      bool rf = false;
      if (choose(q, npl, ay, ax, rf) == 0) return 0;
      const int c = stats_chunks(q, stats_B);
      if (c == 0 || (per >= 0 && c != per)) return 0;
      per = c;
    }
  return per * stride * stride;
}
