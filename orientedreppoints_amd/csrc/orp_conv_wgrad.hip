// orp_conv_wgrad.hip -- weight gradient of the head's 256 -> 256 3x3 tower / FPN convolutions, all FPN levels in one launch
// (gfx950), on the 16-bit matrix pipe with the fp16-pieces arithmetic of orp_dcn_split.hip.
//
//   dW[o][c][tap] = sum over images and positions p of  G[b][o][p] * X[b][c][p + shift(tap)]        (zero outside the map)
//
// (torch.nn.functional.conv2d's backward for the weight, what autograd runs behind ConvModule.conv,
// mmdet/ops/conv_module.py:130-140, for the layers of mmdet/models/anchor_heads/orientedreppoints_head.py:91-113 and
// mmdet/models/necks/fpn.py:150-153).  The contraction runs over POSITIONS, and in NCHW -- the layout both tensors have in
// training -- positions are the contiguous axis of every channel row: a lane's eight k-values of an MFMA operand are eight
// consecutive floats of one row, no transposition anywhere (the forward kernel contracts over channels and reads
// channels-last for the same reason).  The library's kernel for these shapes (igemm_wrw_gtcx35_nhwc_fp32, exact-fp32 MFMA,
// after transposing both tensors) took 0.64 ms per layer at 2 x 1024^2, 6 ms of a 33 ms training step.
//
// Workgroup = one tap x one slice of the position chunks, 8 waves, output tile 256 (o) x 256 (c) in registers (wave: 64 x 128 =
// 2 x 4 accumulators).  K runs in steps of 32 positions of one image of one level: the 256 G rows and the 256 (shifted) X rows of
// the step are fetched once by the workgroup (8 consecutive floats per item, 4 items per thread), scaled by the tensors'
// power-of-two range factors, split into two fp16 pieces (hi = fp16(v), lo = fp16(v - hi): 2^-22 relative) and written to LDS
// as [operand][piece][row][32 positions]; every wave then reads its fragments with one 16-byte LDS read per lane and issues
// hi*hi + hi*lo + lo*hi.  LDS is double buffered (2 x 80 KB): the next step's rows are converted and written in the shadow of
// this step's MFMAs, the step after that is in flight from memory; one barrier per step.
// Partial tiles per (slice, tap) go to the workspace and are summed in slice order by a second launch: deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/orp_hip.h"
#include "orp_prof.hpp"
#include "orp_launch.hpp"

#ifndef ORP_WG_ALIGNED
#define ORP_WG_ALIGNED 1  // X rows: aligned 16-byte loads + one neighbour element instead of 4-byte-aligned 16-byte loads of the shifted octet
#endif
#ifndef ORP_WG_PAIRS
#define ORP_WG_PAIRS 1    // the X fragments of a chunk in two pairs of column blocks, the second pair's LDS reads under the first pair's MFMAs
#endif
#ifndef ORP_WG_DBG
#define ORP_WG_DBG 0      // dev aid (timing only, wrong results): 1 = no fetches after the first, 2 = no MFMA, 4 = no conversion / LDS writes
#endif

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int kMaxLv = 8;
constexpr int CH = 256;                 // Cin = Cout = 256
constexpr int KS = 32;                  // positions per K step
constexpr int RS = 40;                  // LDS row stride in halfs (80 B: 64 B of payload + pad)
constexpr int kThreadsW = 512;
constexpr int kTapsMaxW = 9;

struct WLevel {
  const float* x;                       // [B][CH][H][W]
  const float* g;                       // [B][CH][H][W] (stride 1, 'same' padding: the output has the input's size)
  int H, W;
  int chunk0;                           // first K step of this level; steps of one image are consecutive
  int cpi;                              // steps per image = ceil(H*W / KS)
};
struct WParams {
  WLevel lv[kMaxLv];
  int nlev, B;
  int kh, kw, ph, pw, dh, dw;
  int total_chunks, nsplit;
  const unsigned* amax_x;               // float bits of (a bound of) max |x| / max |g| over all levels (device scalars)
  const unsigned* amax_g;
  float* partial;                       // [nsplit][taps][CH (o)][CH (c)]
};

struct Item { float v[8]; };
struct __attribute__((packed, aligned(4))) F4u { float v[4]; };     // a 16-byte load at 4-byte alignment

// FAST (chosen by the host when every level has W % 8 == 0, H * W % 32 == 0 and the taps' column shifts are -1 / 0 / +1 -- every
// BASELINE shape): all fetches are branch-free (a fixed number of loads per item, so the compiler's s_waitcnt vmcnt counts are exact)
// and the software pipeline is one step deeper: the two items a 16-position chunk has just converted are re-armed with the rows of
// the step after next RIGHT THERE, behind the chunk's MFMAs -- a whole step for the loads to land.  In the general form the rows of
// step t + 2 are requested at the end of step t and converted at the start of step t + 1: one barrier of latency hiding.
template <bool FAST>
__global__ void __launch_bounds__(kThreadsW)
conv_wgrad_split_kernel(const WParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* sT = reinterpret_cast<_Float16*>(smem);        // [operand: 0 = G, 1 = X][piece][CH rows][RS]
  constexpr int PL = CH * RS;                               // halfs per (operand, piece) plane
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (Measured and dropped: all `taps` workgroups of a slice on one XCD so that they share its L2 -- 285 us with the 24 slices
  //  that fill the XCDs evenly against 257 us for the plain grid of 28 slices; the rows are L2 / MALL resident either way.)
  const int taps = P.kh * P.kw, tap = blockIdx.y, slice = blockIdx.x;
  const int ki = tap / P.kw, kj = tap - ki * P.kw;
  const int sh_h = ki * P.dh - P.ph, sh_w = kj * P.dw - P.pw;           // the tap's shift of the input position
  const int per = (P.total_chunks + P.nsplit - 1) / P.nsplit;
  const int c_begin = slice * per;
  const int c_end = min(c_begin + per, P.total_chunks);

  // range factors (powers of two): the tensor's largest magnitude lands in [2^14, 2^15)
  auto scale_of = [](unsigned am) {
    int k = am == 0u ? 0 : 14 - ((int)((am >> 23) & 0xffu) - 127);
    k = k < -100 ? -100 : k > 100 ? 100 : k;
    return __uint_as_float((unsigned)(127 + k) << 23);
  };
  const float sx = scale_of(*P.amax_x), sg = scale_of(*P.amax_g);

  // this thread's four items of a step: item u = (row r = (tid >> 2) + 128 * u of the 512 rows [G 0..255 | X 256..511],
  // octet q = tid & 3): eight consecutive positions of one channel row
  const int q = tid & 3;
  auto fetch = [&](int chunk, Item (&it)[4], int u0, int u1) {
    int l = 0;
#pragma unroll 1
    for (int i = 1; i < P.nlev; i++) if (chunk >= P.lv[i].chunk0) l = i;
    const WLevel& L = P.lv[l];
    const int id = chunk - L.chunk0;
    const int b = id / L.cpi, p0 = (id - b * L.cpi) * KS + q * 8;
    const int HW = L.H * L.W;
    const int h0 = p0 / L.W, w0 = p0 - h0 * L.W;             // one division per step; the eight positions walk on from here
    const bool vec = (HW & 3) == 0 && p0 + 8 <= HW;          // G rows: two aligned 16-byte loads
    // (16-byte loads wherever possible: position-by-position loads of the shifted rows cost 427 us for the launch against 257;
    //  eight lanes per row with one 16-byte load each -- half the cache lines per load instruction -- measured the same, 269)
#pragma unroll
    for (int u = u0; u < u1; u++) {
      const int r = (tid >> 2) + 128 * u;
      const bool is_x = r >= CH;                             // (u < 2: a G row, u >= 2: an X row -- uniform per u)
      const float* row = (is_x ? L.x : L.g) + ((size_t)b * CH + (r & (CH - 1))) * HW;
      if (!is_x) {
        if (vec) {
          const float4 a = *reinterpret_cast<const float4*>(row + p0), c = *reinterpret_cast<const float4*>(row + p0 + 4);
          it[u].v[0] = a.x; it[u].v[1] = a.y; it[u].v[2] = a.z; it[u].v[3] = a.w;
          it[u].v[4] = c.x; it[u].v[5] = c.y; it[u].v[6] = c.z; it[u].v[7] = c.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; e++) it[u].v[e] = (p0 + e < HW) ? row[p0 + e] : 0.f;
        }
      } else {
        // the shifted run: eight consecutive floats of the same input row whenever the octet neither wraps nor touches the
        // border (two 16-byte loads, 4-byte aligned); else position by position (row starts / ends, map borders, the tail)
        const int hs0 = h0 + sh_h, ws0 = w0 + sh_w;
        const bool inrow = p0 + 8 <= HW && w0 + 8 <= L.W;              // the octet lies inside one image row
        if (inrow && (hs0 < 0 || hs0 >= L.H)) {                         // ... of a row above / below the map: zeros, no loads
#pragma unroll
          for (int e = 0; e < 8; e++) it[u].v[e] = 0.f;
        } else if (ORP_WG_ALIGNED && inrow && (L.W & 3) == 0 && sh_w >= -1 && sh_w <= 1) {
          // 16-byte ALIGNED loads of the unshifted octet plus the one neighbour the tap's column shift brings in (zero at the row's
          // end); the shift itself is a renaming under a workgroup-uniform condition
          const float* src = row + hs0 * L.W + w0;
          const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
          float nb = 0.f;
          if (sh_w < 0) { if (w0 > 0) nb = src[-1]; }
          else if (sh_w > 0) { if (w0 + 8 < L.W) nb = src[8]; }
          const float t[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
          for (int e = 0; e < 8; e++)
            it[u].v[e] = sh_w == 0 ? t[e] : sh_w < 0 ? (e > 0 ? t[e > 0 ? e - 1 : 0] : nb) : (e < 7 ? t[e < 7 ? e + 1 : 7] : nb);
        } else if (inrow && ws0 >= -1 && ws0 + 8 <= L.W + 1) {
          // inside the row, or hanging over its left / right end by ONE position (the +-1 column shift of a 3 x 3 tap -- half
          // of all steps have such octets, and a wave runs the position-by-position path below as soon as one lane needs it):
          // the eight floats from the clamped start, moved by one lane-private select per element, zero at the border
          const int wsc = min(max(ws0, 0), L.W - 8), d = wsc - ws0;     // d = +1: left end, -1: right end, 0: inside
          const F4u a = *reinterpret_cast<const F4u*>(row + hs0 * L.W + wsc), c = *reinterpret_cast<const F4u*>(row + hs0 * L.W + wsc + 4);
          const float t[8] = {a.v[0], a.v[1], a.v[2], a.v[3], c.v[0], c.v[1], c.v[2], c.v[3]};
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const float lft = e > 0 ? t[e > 0 ? e - 1 : 0] : 0.f, rgt = e < 7 ? t[e < 7 ? e + 1 : 7] : 0.f;
            it[u].v[e] = d == 0 ? t[e] : d > 0 ? lft : rgt;
          }
        } else {
          int h = h0, w = w0;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const int hs = h + sh_h, ws = w + sh_w;
            const bool ok = p0 + e < HW && hs >= 0 && hs < L.H && ws >= 0 && ws < L.W;
            it[u].v[e] = ok ? row[hs * L.W + ws] : 0.f;
            if (++w == L.W) { w = 0; h++; }
          }
        }
      }
    }
  };
  auto fetch_fast = [&](int chunk, Item (&it)[4], int u0, int u1) {
    int l = 0;
#pragma unroll 1
    for (int i = 1; i < P.nlev; i++) if (chunk >= P.lv[i].chunk0) l = i;
    const WLevel& L = P.lv[l];
    const int id = chunk - L.chunk0;
    const int b = id / L.cpi, p0 = (id - b * L.cpi) * KS + q * 8;
    const int HW = L.H * L.W;
    const int h0 = p0 / L.W, w0 = p0 - h0 * L.W;             // (w0 + 8 <= W: W is a multiple of 8)
#pragma unroll
    for (int u = u0; u < u1; u++) {
      const int r = (tid >> 2) + 128 * u;
      const bool is_x = r >= CH;
      const float* row = (is_x ? L.x : L.g) + ((size_t)b * CH + (r & (CH - 1))) * HW;
      if (!is_x) {
        const float4 a = *reinterpret_cast<const float4*>(row + p0), c = *reinterpret_cast<const float4*>(row + p0 + 4);
        it[u].v[0] = a.x; it[u].v[1] = a.y; it[u].v[2] = a.z; it[u].v[3] = a.w;
        it[u].v[4] = c.x; it[u].v[5] = c.y; it[u].v[6] = c.z; it[u].v[7] = c.w;
      } else {
        // the aligned octet of the (clamped) shifted row and the one neighbour element the column shift brings in; validity as selects
        const int hs0 = h0 + sh_h;
        const bool rv = hs0 >= 0 && hs0 < L.H;
        const float* src = row + min(max(hs0, 0), L.H - 1) * L.W + w0;
        const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
        const bool nv = sh_w < 0 ? w0 > 0 : w0 + 8 < L.W;
        const float nl = src[sh_w < 0 ? (nv ? -1 : 0) : (nv ? 8 : 7)];
        const float nb = nv ? nl : 0.f;
        const float t[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float v = sh_w == 0 ? t[e] : sh_w < 0 ? (e > 0 ? t[e > 0 ? e - 1 : 0] : nb) : (e < 7 ? t[e < 7 ? e + 1 : 7] : nb);
          it[u].v[e] = rv ? v : 0.f;
        }
      }
    }
  };
  constexpr int BUF = 4 * PL;                               // halfs per LDS buffer (G hi | G lo | X hi | X lo)
  auto stash = [&](const Item (&it)[4], int buf, int u0, int u1) {
#pragma unroll
    for (int u = u0; u < u1; u++) {
      const int r = (tid >> 2) + 128 * u;
      const bool is_x = r >= CH;
      const float sc = is_x ? sx : sg;
      h8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float sv = it[u].v[e] * sc;                                 // exact
        hi[e] = (_Float16)sv;
        lo[e] = (_Float16)(sv - (float)hi[e]);                            // the residual is exact in fp32
      }
      _Float16* dst = sT + (size_t)buf * BUF + (size_t)(is_x ? 2 : 0) * PL + (size_t)(r & (CH - 1)) * RS + q * 8;
      *reinterpret_cast<h8*>(dst) = hi;
      *reinterpret_cast<h8*>(dst + PL) = lo;
    }
  };

  floatx16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int c = 0; c < 4; c++) acc[a][c] = floatx16{0};
  const int m = lane & 31, kg = lane >> 5;
  const int wo = wave >> 1, wc = wave & 1;                  // o rows [64 wo, +64), c columns [128 wc, +128)

  if (c_begin < c_end) {
    // LDS double buffered: while the waves contract step t out of buffer t & 1, the rows of step t + 1 (fetched one step
    // earlier, in registers) are converted and written to the other buffer in the shadow of the MFMAs (one MFMA : four VALU),
    // then the rows of step t + 2 are requested; ONE barrier per step
    Item it[4];
    if (FAST) fetch_fast(c_begin, it, 0, 4); else fetch(c_begin, it, 0, 4);
    stash(it, 0, 0, 4);
    if (FAST) fetch_fast(min(c_begin + 1, c_end - 1), it, 0, 4);
    else if (c_begin + 1 < c_end) fetch(c_begin + 1, it, 0, 4);
    __syncthreads();
#pragma unroll 1
    for (int chunk = c_begin; chunk < c_end; chunk++) {
      const int cur = (chunk - c_begin) & 1;
      const bool more = chunk + 1 < c_end, more2 = chunk + 2 < c_end;
      const _Float16* sB = sT + (size_t)cur * BUF;
#if ORP_WG_PAIRS
      // The X fragments in two PAIRS of column blocks: the second pair's LDS reads fly under the first pair's twelve MFMAs, and the first
      // pair of the step's second chunk under the second pair's (its registers are free by then; the G fragments follow when theirs
      // are).  Exposed LDS reads per step and wave: 12 KB instead of 24 -- all eight waves read at once, 1 536 cycles of the CU's LDS
      // per step against 3 072 of matrix work.  Per accumulator the order of the three products is unchanged: the same bits.
      h8 ga[2][2], xb[4][2];
      auto ld_ga = [&](int j) {
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int pl = 0; pl < 2; pl++)
            ga[a][pl] = *reinterpret_cast<const h8*>(sB + (size_t)pl * PL + (size_t)(wo * 64 + a * 32 + m) * RS + j * 16 + kg * 8);
      };
      auto ld_xb = [&](int j, int c0) {
#pragma unroll
        for (int c = c0; c < c0 + 2; c++)
#pragma unroll
          for (int pl = 0; pl < 2; pl++)
            xb[c][pl] = *reinterpret_cast<const h8*>(sB + (size_t)(2 + pl) * PL + (size_t)(wc * 128 + c * 32 + m) * RS + j * 16 + kg * 8);
      };
      auto mfma_pair = [&](int c0) {
#pragma unroll
        for (int pr = 0; pr < 3; pr++)
#pragma unroll
          for (int a = 0; a < 2; a++)
#pragma unroll
            for (int c = c0; c < c0 + 2; c++)
              if (ORP_WG_DBG & 2) acc[a][c][0] += (float)ga[a][pr == 0 ? 1 : 0][0] * (float)xb[c][pr == 1 ? 1 : 0][0];
              else acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[a][pr == 0 ? 1 : 0], xb[c][pr == 1 ? 1 : 0], acc[a][c], 0, 0, 0);
      };
      ld_ga(0); ld_xb(0, 0);
#pragma unroll
      for (int j = 0; j < KS / 16; j++) {
        ld_xb(j, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (more && !(ORP_WG_DBG & 4)) stash(it, cur ^ 1, 2 * j, 2 * j + 2);
        if (FAST && !(ORP_WG_DBG & 1)) fetch_fast(min(chunk + 2, c_end - 1), it, 2 * j, 2 * j + 2);
        mfma_pair(0);
#pragma unroll
        for (int i = 0; i < 12; i++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < KS / 16) ld_xb(j + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(2);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < KS / 16) ld_ga(j + 1);
      }
#else
#pragma unroll
      for (int j = 0; j < KS / 16; j++) {
        h8 ga[2][2], xb[4][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int pl = 0; pl < 2; pl++)
            ga[a][pl] = *reinterpret_cast<const h8*>(sB + (size_t)pl * PL + (size_t)(wo * 64 + a * 32 + m) * RS + j * 16 + kg * 8);
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
          for (int pl = 0; pl < 2; pl++)
            xb[c][pl] = *reinterpret_cast<const h8*>(sB + (size_t)(2 + pl) * PL + (size_t)(wc * 128 + c * 32 + m) * RS + j * 16 + kg * 8);
        __builtin_amdgcn_sched_barrier(0);
        // half of the next step's rows per 16-position chunk, converted in the shadow of the chunk's MFMAs (1 MFMA : 4 VALU).
        // (Converting first and re-arming the registers with the step after next right away -- a whole step for the loads to
        //  land instead of a barrier -- measured slower: 281 us against 257.)
        if (more && !(ORP_WG_DBG & 4)) stash(it, cur ^ 1, 2 * j, 2 * j + 2);
        // FAST: the two items just converted take the rows of the step after next (past the end: the last step again, dropped)
        if (FAST && !(ORP_WG_DBG & 1)) fetch_fast(min(chunk + 2, c_end - 1), it, 2 * j, 2 * j + 2);
        // smallest products first; eight independent accumulators between two MFMAs into the same one
#pragma unroll
        for (int pr = 0; pr < 3; pr++)
#pragma unroll
          for (int a = 0; a < 2; a++)
#pragma unroll
            for (int c = 0; c < 4; c++)
              if (ORP_WG_DBG & 2) acc[a][c][0] += (float)ga[a][pr == 0 ? 1 : 0][0] * (float)xb[c][pr == 1 ? 1 : 0][0];
              else acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[a][pr == 0 ? 1 : 0], xb[c][pr == 1 ? 1 : 0], acc[a][c], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 24; i++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      if (!FAST && more2 && !(ORP_WG_DBG & 1)) fetch(chunk + 2, it, 0, 4);  // lands during the next step
      __syncthreads();
    }
  }

  const float osc = 1.f / (sx * sg);
  float* outp = P.partial + ((size_t)slice * taps + tap) * CH * CH;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int o = wo * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        outp[(size_t)o * CH + wc * 128 + c * 32 + m] = acc[a][c][r] * osc;
      }
}

// dW[o][c][tap] = sum over slices, in slice order, of partial[slice][tap][o][c]: one thread per (tap, o, c), consecutive
// threads read consecutive floats of every partial image
__global__ void __launch_bounds__(256)
conv_wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, int taps, float* __restrict__ dw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= taps * CH * CH) return;
  const int tap = i / (CH * CH), oc = i - tap * (CH * CH);
  float s = 0.f;
  for (int z = 0; z < nsplit; z++) s += partial[((size_t)z * taps + tap) * CH * CH + oc];
  dw[(size_t)oc * taps + tap] = s;
}

// max |x| over the levels of x (slot 0) and of g (slot 1), float bits
struct WAbs { const float* p[2 * kMaxLv]; size_t n[2 * kMaxLv]; int slot[2 * kMaxLv]; int bx0[2 * kMaxLv + 1]; int count; };
__global__ void __launch_bounds__(256)
wgrad_absmax_kernel(const WAbs A, unsigned* __restrict__ out) {
  __shared__ unsigned red[4];
  int t = 0;
#pragma unroll 1
  for (int i = 1; i < A.count; i++) if ((int)blockIdx.x >= A.bx0[i]) t = i;
  const float* x = A.p[t];
  const size_t n = A.n[t];
  const int nb = A.bx0[t + 1] - A.bx0[t], b = (int)blockIdx.x - A.bx0[t];
  unsigned mx = 0u;
  for (size_t i = (size_t)b * 256 + threadIdx.x; i < n; i += (size_t)nb * 256) mx = max(mx, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned v = max(max(red[0], red[1]), max(red[2], red[3]));
    if (v > __atomic_load_n(out + A.slot[t], __ATOMIC_RELAXED)) atomicMax(out + A.slot[t], v);
  }
}

inline int pick_nsplit(int total_chunks, int taps) {
  int ns = 252 / taps;                                        // one workgroup per CU, one round (9 taps: 28 slices)
  if (ns > total_chunks) ns = total_chunks;
  return ns < 1 ? 1 : ns;
}
inline size_t align256w(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int orp_conv_wgrad_split_ok(int c_in, int c_out, int kh, int kw) { return (c_in == CH && c_out == CH && kh * kw <= kTapsMaxW && kh > 0 && kw > 0) ? 1 : 0; }

size_t orp_conv_wgrad_split_workspace_bytes(const orp_wgrad_level* levels_host, int nlevels, int batch, int kh, int kw) {
  if (!levels_host || nlevels <= 0 || nlevels > kMaxLv || batch <= 0) return 0;
  long chunks = 0;
  for (int i = 0; i < nlevels; i++) chunks += (long)batch * (((long)levels_host[i].height * levels_host[i].width + KS - 1) / KS);
  const int ns = pick_nsplit((int)chunks, kh * kw);
  return 256 + align256w(sizeof(float) * (size_t)ns * kh * kw * CH * CH);
}

int orp_conv_wgrad_split(const orp_wgrad_level* levels_host, int nlevels, int batch, int c_in, int c_out, int kh, int kw,
                         int pad_h, int pad_w, int dil_h, int dil_w, const uint32_t* amax_x, const uint32_t* amax_g,
                         float* grad_weight, void* workspace, size_t workspace_bytes, void* stream) {
  if (!levels_host || nlevels <= 0 || nlevels > kMaxLv || batch <= 0 || !grad_weight) return ORP_EINVAL;
  if (!orp_conv_wgrad_split_ok(c_in, c_out, kh, kw) || dil_h <= 0 || dil_w <= 0) return ORP_EINVAL;
  if (2 * pad_h != dil_h * (kh - 1) || 2 * pad_w != dil_w * (kw - 1)) return ORP_EINVAL;      // 'same' convolutions, stride 1
  const size_t need = orp_conv_wgrad_split_workspace_bytes(levels_host, nlevels, batch, kh, kw);
  if (!workspace || workspace_bytes < need) return ORP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  WParams P;
  P.nlev = nlevels; P.B = batch; P.kh = kh; P.kw = kw; P.ph = pad_h; P.pw = pad_w; P.dh = dil_h; P.dw = dil_w;
  long chunks = 0;
  for (int i = 0; i < nlevels; i++) {
    const orp_wgrad_level& lv = levels_host[i];
    if (!lv.input || !lv.grad_output || lv.height <= 0 || lv.width <= 0) return ORP_EINVAL;
    if ((long)lv.height * lv.width >= (1L << 30)) return ORP_ETOOBIG;
    WLevel& L = P.lv[i];
    L.x = lv.input; L.g = lv.grad_output; L.H = lv.height; L.W = lv.width;
    L.cpi = (lv.height * lv.width + KS - 1) / KS; L.chunk0 = (int)chunks;
    chunks += (long)batch * L.cpi;
    if (chunks >= (1L << 30)) return ORP_ETOOBIG;
  }
  for (int i = nlevels; i < kMaxLv; i++) { P.lv[i] = P.lv[0]; P.lv[i].chunk0 = 0x7fffffff; }
  P.total_chunks = (int)chunks;
  P.nsplit = pick_nsplit(P.total_chunks, kh * kw);
  unsigned* amax = reinterpret_cast<unsigned*>(workspace);
  P.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + 256);
  OrpProfScope prof(ORP_PROF_CONV_WGRAD, st);
  P.amax_x = amax_x; P.amax_g = amax_g;
  if (!amax_x || !amax_g) {                                   // no producer left (both) ranges: take them here
    hipError_t me = orp::fill_async(amax, 0, 2 * sizeof(unsigned), st);
    if (me != hipSuccess) return (int)me;
    WAbs A;
    int bx = 0, cnt = 0;
    for (int s = 0; s < 2; s++)
      for (int i = 0; i < nlevels; i++) {
        A.p[cnt] = s ? levels_host[i].grad_output : levels_host[i].input;
        A.n[cnt] = (size_t)batch * CH * levels_host[i].height * levels_host[i].width;
        A.slot[cnt] = s; A.bx0[cnt] = bx;
        long nb = (long)((A.n[cnt] + 256 * 16 - 1) / (256 * 16)); if (nb < 1) nb = 1; if (nb > 512) nb = 512;
        bx += (int)nb; cnt++;
      }
    for (int i = cnt; i <= 2 * kMaxLv; i++) A.bx0[i] = bx;
    for (int i = cnt; i < 2 * kMaxLv; i++) { A.p[i] = A.p[0]; A.n[i] = 0; A.slot[i] = 0; }
    A.count = cnt;
    hipLaunchKernelGGL(wgrad_absmax_kernel, dim3(bx), dim3(256), 0, st, A, amax);
    P.amax_x = amax; P.amax_g = amax + 1;
  }
  const size_t smem = sizeof(_Float16) * 2 * 4 * CH * RS;     // two buffers of 80 KB: all of a CU's LDS
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_split_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (ae == hipSuccess) ae = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_split_kernel<true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (ae != hipSuccess) return (int)ae;
    attr_set = true;
  }
  // the branch-free, deeper-pipelined form where every octet is a piece of one image row and a tap moves it by at most one column
  static const int fast_env = getenv("ORP_WGRAD_FAST") ? atoi(getenv("ORP_WGRAD_FAST")) : 1;   // 0: dev aid (A/B timing)
  bool fast = fast_env != 0;
  for (int i = 0; i < nlevels; i++)
    fast = fast && (levels_host[i].width % 8 == 0) && ((long)levels_host[i].height * levels_host[i].width % KS == 0);
  for (int kj = 0; kj < kw; kj++) fast = fast && (kj * dil_w - pad_w >= -1) && (kj * dil_w - pad_w <= 1);
  if (fast) hipLaunchKernelGGL(conv_wgrad_split_kernel<true>, dim3(P.nsplit, kh * kw), dim3(kThreadsW), smem, st, P);
  else hipLaunchKernelGGL(conv_wgrad_split_kernel<false>, dim3(P.nsplit, kh * kw), dim3(kThreadsW), smem, st, P);
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(kh * kw * CH * CH / 256), dim3(256), 0, st, P.partial, P.nsplit, kh * kw, grad_weight);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

}  // extern "C"
