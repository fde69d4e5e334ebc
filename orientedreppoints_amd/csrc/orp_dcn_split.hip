// orp_dcn_split.hip -- fp32 deformable convolution forward on the bf16 matrix pipe, for gfx950 (MI355X).
//
// Same operator as orp_dcn.hip (deform_conv_forward_cuda / modulated_deform_conv_cuda_forward,
// mmdet/ops/dcn/src/deform_conv_cuda.cpp:152-260, 490-567; kernels deform_conv_cuda_kernel.cu:190-243, 570-633): fp32 tensors
// in, fp32 tensors out, fp32 accumulation.  What changes is the instruction the contraction is issued on.  gfx950 has no
// fast fp32 matrix path: v_mfma_f32_32x32x2_f32 runs at the VECTOR rate (64 FLOP/clk/SIMD, 157 TF/s), 1/16 of
// v_mfma_f32_32x32x16_bf16.  Two rounds of scheduling work left the exact-fp32 kernel at 0.68 of that peak.  Here every fp32
// operand is split EXACTLY into three bf16 pieces by truncation,
//
//      v = hi + mid + lo,   hi = v & 0xffff0000,   mid = (v - hi) & 0xffff0000,   lo = (v - hi) - mid
//
// (8 + 8 + 8 significant bits: the subtractions are exact in fp32 and lo has no bits left below its bf16 mantissa), once per
// weight at pack time and once per bilinear sample when the A tile is produced, and the product  v * w  becomes partial
// products of bf16 pieces -- each EXACT in the fp32 accumulator the MFMA adds them into:
//      nprod = 9: all of them: the sum of products is formed with no representation error at all, only the accumulator's
//                 roundings remain (9 per 16 channels, where the fp32 MFMA chain has 8);
//      nprod = 6: without lo*lo, lo*mid, mid*lo (each <= 2^-24 |v w|, below one fp32 rounding of the product itself).
// Small terms are issued first.  Matrix time per 16 channels of a 32 x 32 tile: 8 fp32 MFMAs x 64 clk = 512 clk before,
// 9 (6) bf16 MFMAs x 32 clk = 288 (192) clk now.
//
// Kernel organisation (one workgroup = 8 waves = MT*32 positions x 256 output channels, all FPN levels in one launch, one
// or two layers over the same offsets as grid halves: XCDs 0-3 layer 0, XCDs 4-7 layer 1):
//   * K runs in phases of 64 input channels of one tap.  The A tile (bilinear samples, never in HBM) lives in LDS as three
//     bf16 planes [MT*32][64 + 8], DOUBLE buffered: while the waves contract phase p out of buffer p & 1, every wave
//     gathers its 4*MT rows of phase p + 1 (fp32 NHWC rows: 16 lanes x float4 = one 256 B row piece, four rows per
//     instruction), combines them with the four bilinear weights in fp32, splits, and writes the other buffer.  One
//     barrier per phase.
//   * weights: three bf16 planes per layer packed [plane][tap][Cin/16][2][Cout][8] (one 16 B load = the 8 k-values of an
//     MFMA lane), streamed from L2 into a register ring of one phase (4 chunks x 3 planes), refilled in place for the next
//     phase right after use.
//   * output NCHW (operands swapped: lanes along positions) or NHWC; bias / ReLU in the epilogue.
// Bit-level behaviour: deterministic (fixed order), not bit-identical to the exact-fp32 kernel (different summation
// grouping); tests/test_gpu_dcn_split.py holds both against the fp64-accumulated oracle and prints both errors.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "orp_dcn_split.hpp"
#include "orp_launch.hpp"

#ifndef ORP_DCNS_DBG
#define ORP_DCNS_DBG 0     // dev aid, compile-time (timing only, wrong results): 1 = no gathers, 2 = no weight refills, 4 = no MFMA, 8 = no combine / split / LDS write, 16 = no A-fragment LDS reads, 32 = no per-phase barrier
#endif

#ifndef ORP_DCNS_COMBINE_IN_LAST
#define ORP_DCNS_COMBINE_IN_LAST 1   // 0: combine + split as a VALU-only tail after the phase's last MFMA (measured: the matrix pipe idles through it)
#endif
#ifndef ORP_DCNS_REFILL_LAG
#define ORP_DCNS_REFILL_LAG 0
#endif
#ifndef ORP_DCNS_FENCE
// 1 (default): a scheduling fence between the MFMAs of a chunk and the in-place refill of the chunk's weight registers -- the
// MFMAs stay together, the refills go out behind them (measured +3 % on the pair launch against the scheduler's own mix).
// History: round 4 had an inline-asm v_mov of one accumulator element here ("accumulator drain"), on the hypothesis that the
// refill's VMEM return could overtake queued MFMAs.  tests/checks/mfma_war.hip settles it: 6.5e9 in-place refills right behind
// their MFMAs, four waves per SIMD, every accumulator exact -- an issued MFMA has read its A / B operands --, and the v_mov
// waited for nothing (it read a value two MFMAs old).  The wrong rows it seemed to cure were packed-fp32 VALU instructions of
// the coefficient-table code miscomputing next to a second workgroup's MFMA loop; the v_mov only shifted that workgroup's timing.
#define ORP_DCNS_FENCE 1
#endif
#ifndef ORP_DCNS_OWN_SIMD
#define ORP_DCNS_OWN_SIMD 1          // 0: dev aid (the instantiations of tile height 1 / 2 then share their SIMDs with other waves)
#endif
#ifndef ORP_DCNS_TABLE_SELECTS
#define ORP_DCNS_TABLE_SELECTS 0     // dev aid: 1 = the coefficient table's border conditions as selects (round 4; wrong rows under co-residency)
#endif
#ifndef ORP_DCNS_TRACE
#define ORP_DCNS_TRACE 0             // dev aid (tests/checks/split_trace.py): 1 = every workgroup dumps its coefficient table, 2 = also a hash of every A-tile row of every phase, into the orp_debug_amax_log buffer
#endif
#ifndef ORP_DCNS_SIDE_ACC
#define ORP_DCNS_SIDE_ACC 1          // PLAIN instantiation: second accumulator set for the small partial products (see Products)
#endif
#ifndef ORP_DCNS_CC
#define ORP_DCNS_CC 0                // dev aid: number of trailing chunks that carry the combine (0: the rule in the kernel)
#endif
#ifndef ORP_DCNS_AHEAD2
#define ORP_DCNS_AHEAD2 1            // PLAIN instantiation: the rows of phase p + 2 are gathered during phase p (two register sets, the loop unrolled by two phases)
#endif
#ifndef ORP_DCNS_EARLYBAR
#define ORP_DCNS_EARLYBAR 0          // with AHEAD2: the phase's barrier in front of the LAST chunk's MFMAs, the next phase's first A fragments read behind it (measured: 200.4 - 202.5 vs 202.1 - 206.8 us, within the noise: off)
#endif
#ifndef ORP_DCNS_PRIO
#define ORP_DCNS_PRIO 0             // dev aid: bit 0 = s_setprio 2 while the phase's gathers are issued, bit 1 = while a chunk carries the combine
#endif
#ifndef ORP_DCNS_INTERLEAVE
#define ORP_DCNS_INTERLEAVE 4        // VALU instructions of the combine pinned behind every MFMA of the chunk that carries it (0: scheduler's choice)
#endif

namespace orp_split {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));     // one MFMA operand: 8 k-values (of either 16-bit format: see F16)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int kTapsMax = 9;
constexpr int CBS = 64;            // input channels per phase
constexpr int ASTRS = CBS + 8;     // A row stride in bf16 elements (36 dwords: conflict-free ds_read_b128 over 16 rows)
constexpr int NCH = CBS / 16;      // MFMA chunks (16 channels) per phase
constexpr int kThreadsS = 512;

struct LevelK {
  const float* x[2];
  const float* off;
  const float* mask;
  float* out[2];
  int H, W, Ho, Wo;
  int tile0;
  int tpi;                        // 0: the level's positions of all images tiled back to back; > 0: tiles per IMAGE (no tile spans two images: GroupNorm statistics)
  const uint16_t* planes;         // this level's own layer (or nullptr: FwdS::planes / bias / wscale)
  const float* bias;
  const float* wscale;
};
struct FwdS {
  LevelK lv[kMaxLevels];
  int nlev, B, Cin, Cout;
  int kh, kw, sh, sw, ph, pw, dh, dw;
  const uint16_t* planes[2];
  const float* bias[2];
  int relu, nconv;
  size_t plane_stride;            // elements between two planes of a layer
  const float* wscale[2];         // F16: the power of two the layer's weights were multiplied by at pack time (device scalar)
  const unsigned* amax;           // F16: bits of (bounds of) max |x| over the inputs of layer cv: the maximum of the amax_count words at amax[cv * amax_stride]
  int amax_stride, amax_count;
  // PLAIN, GroupNorm fused around the convolution (orp_conv_split_multi_gn): the INPUT tensors are read as relu?(x * a[c] + b[c]) with
  // the (a, b) of the previous layer's normalisation, coef_in [layer][level][image][Cin] float2; the OUTPUT tiles leave their per-group
  // (mean, M2 around it, max |y|, count) in gn_part [layer][tile][group] for orp_conv_split_gn_finish
  const float2* coef_in;
  int relu_in;
  float4* gn_part;
  int G;
  unsigned* dbg;                  // dev aid (orp_debug_amax_log): the first tile of layer cv leaves [cv] = the range word it READ, [2 + cv] = its weight scale
};

// w [o][c][tap] fp32 -> three bf16 planes [pl][tap][c/16][kg][o][8]  (kg = (c % 16) / 8, e = c % 8), exact truncation split
__global__ void pack_planes_kernel(const float* __restrict__ w, int cout, int cin, int taps, uint16_t* __restrict__ planes) {
  const long total = (long)cout * cin * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7);
    long r = i >> 3;
    const int o = (int)(r % cout); r /= cout;
    const int kg = (int)(r & 1); r >>= 1;
    const int cblk = (int)(r % (cin / 16)), tap = (int)(r / (cin / 16));
    const int c = cblk * 16 + kg * 8 + e;
    const float v = w[((long)o * cin + c) * taps + tap];
    const float hi = __uint_as_float(__float_as_uint(v) & 0xffff0000u);
    const float r1 = v - hi;
    const float mid = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float lo = r1 - mid;
    planes[i] = (uint16_t)(__float_as_uint(hi) >> 16);
    planes[total + i] = (uint16_t)(__float_as_uint(mid) >> 16);
    planes[2 * total + i] = (uint16_t)(__float_as_uint(lo) >> 16);
  }
}

// max |x| over up to kAbsMaxT tensors, as float bits (monotonic for non-negative values), into out[slot of the tensor]
constexpr int kAbsMaxT = 2 * kMaxLevels;
struct AbsMaxArgs {
  const float* x[kAbsMaxT];
  size_t n[kAbsMaxT];
  int slot[kAbsMaxT];
  int bx0[kAbsMaxT + 1];
  int count;
};
__global__ void __launch_bounds__(256)
absmax_kernel(const AbsMaxArgs A, unsigned* __restrict__ out) {
  __shared__ unsigned red[4];
  int t = 0;
#pragma unroll 1
  for (int i = 1; i < A.count; i++) if ((int)blockIdx.x >= A.bx0[i]) t = i;
  const float* x = A.x[t];
  const size_t n = A.n[t];
  const int nb = A.bx0[t + 1] - A.bx0[t], b = (int)blockIdx.x - A.bx0[t];
  unsigned m = 0u;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)b * 256 + threadIdx.x; i < n4; i += (size_t)nb * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu,
            max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
  }
  if (b == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(x[(n4 << 2) + threadIdx.x]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {                                  // (the atomic only where it would change the value: one address)
    const unsigned mx = max(max(red[0], red[1]), max(red[2], red[3]));
    if (mx > __atomic_load_n(out + A.slot[t], __ATOMIC_RELAXED)) atomicMax(out + A.slot[t], mx);
  }
}

// w [o][c][tap] fp32 -> two fp16 planes [pl][tap][c/16][kg][o][8] of w * 2^k, k from amax = max |w| (float bits) so that the
// largest magnitude lands in [2^14, 2^15); wscale[0] = 2^k
__global__ void pack_planes16_kernel(const float* __restrict__ w, int cout, int cin, int taps, const unsigned* __restrict__ amax,
                                     uint16_t* __restrict__ planes, float* __restrict__ wscale) {
  const unsigned am = *amax;
  int k = am == 0u ? 0 : 14 - ((int)((am >> 23) & 0xffu) - 127);
  k = k < -100 ? -100 : k > 100 ? 100 : k;
  const float sc = __uint_as_float((unsigned)(127 + k) << 23);
  if (blockIdx.x == 0 && threadIdx.x == 0) wscale[0] = sc;
  const long total = (long)cout * cin * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7);
    long r = i >> 3;
    const int o = (int)(r % cout); r /= cout;
    const int kg = (int)(r & 1); r >>= 1;
    const int cblk = (int)(r % (cin / 16)), tap = (int)(r / (cin / 16));
    const int c = cblk * 16 + kg * 8 + e;
    const float v = w[((long)o * cin + c) * taps + tap] * sc;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    planes[i] = __builtin_bit_cast(uint16_t, hi);
    planes[total + i] = __builtin_bit_cast(uint16_t, lo);
  }
}

// two fp32 values whose low 16 bits are zero (or may be dropped) -> one dword of two bf16: (a >> 16) | (b & 0xffff0000)
__device__ __forceinline__ unsigned pack_hi16(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}

// partial products (A plane, W plane), smallest first: 0 = hi, 1 = mid, 2 = lo.  Base-3 digit strings, so that the unrolled
// loop indexes registers with compile-time constants (a constexpr array would be materialised in scratch memory)
__device__ __forceinline__ constexpr int prod_a(int t) { const int tab[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}; return tab[t]; }
__device__ __forceinline__ constexpr int prod_b(int t) { const int tab[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}; return tab[t]; }
// F16 (two fp16 pieces, 0 = hi, 1 = lo): lo * hi, hi * lo, hi * hi
#ifndef ORP_DCNS_PROD_ORDER16
#define ORP_DCNS_PROD_ORDER16 0      // 1: lo*hi, hi*hi, hi*lo -- with SIDE the two chains (side, main) alternate; the bits do not change (each chain keeps its own order)
#endif
#if ORP_DCNS_PROD_ORDER16
__device__ __forceinline__ constexpr int prod_a16(int t) { const int tab[3] = {1, 0, 0}; return tab[t]; }
__device__ __forceinline__ constexpr int prod_b16(int t) { const int tab[3] = {0, 0, 1}; return tab[t]; }
#else
__device__ __forceinline__ constexpr int prod_a16(int t) { const int tab[3] = {1, 0, 0}; return tab[t]; }
__device__ __forceinline__ constexpr int prod_b16(int t) { const int tab[3] = {0, 1, 0}; return tab[t]; }
#endif

// SIDE: the small partial products (everything but hi * hi) go to a second accumulator set that is added once in the
// epilogue -- the main chain then rounds once per 16 channels at the output's magnitude instead of 3 (6, 9) times, and the
// roundings of the side chain happen 2^-8 (2^-11) further down (PLAIN instantiation: the registers are there)
template <int T, int TEND, int MT, bool OUT_NCHW, bool SIDE, bool F16>
struct Products {
  static __device__ __forceinline__ void run(floatx16 (&acc)[MT], floatx16 (&side)[SIDE ? MT : 1], const bf8 (&a)[MT][3],
                                             const bf8 (&b)[3]) {
    constexpr int pa = F16 ? prod_a16(T) : prod_a(T), pb = F16 ? prod_b16(T) : prod_b(T);
    constexpr bool to_side = SIDE && !(pa == 0 && pb == 0);          // everything but hi * hi
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      if (ORP_DCNS_DBG & 4) { acc[mt][0] += (float)a[mt][pa][0] * (float)b[pb][0]; continue; }
      floatx16& d = to_side ? side[mt] : acc[mt];
      if (F16) {
        const h8 av = __builtin_bit_cast(h8, a[mt][pa]), bv = __builtin_bit_cast(h8, b[pb]);
        if (OUT_NCHW) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, av, d, 0, 0, 0);
        else          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, d, 0, 0, 0);
      } else {
        if (OUT_NCHW) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[pb], a[mt][pa], d, 0, 0, 0);   // D[channel][position]
        else          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][pa], b[pb], d, 0, 0, 0);   // D[position][channel]
      }
    }
    Products<T + 1, TEND, MT, OUT_NCHW, SIDE, F16>::run(acc, side, a, b);
  }
};
template <int TEND, int MT, bool OUT_NCHW, bool SIDE, bool F16>
struct Products<TEND, TEND, MT, OUT_NCHW, SIDE, F16> {
  static __device__ __forceinline__ void run(floatx16 (&)[MT], floatx16 (&)[SIDE ? MT : 1], const bf8 (&)[MT][3], const bf8 (&)[3]) {}
};

// PLAIN: no offsets -- the ordinary convolution (sample = the tap-shifted pixel itself, zero outside the map): one row fetch
// per sample instead of four, no bilinear combine; everything else (split, planes, MFMA schedule, epilogue) is shared
template <int MT, int NPROD, bool OUT_NCHW, bool PLAIN>
__global__ void __launch_bounds__(kThreadsS)
dcn_fwd_split_kernel(const FwdS P, int total_tiles) {
  constexpr int BMS = 32 * MT;
  constexpr int PLANE = BMS * ASTRS;                                          // elements of one plane of one buffer
  // NPROD = 3: TWO fp16 pieces per operand (11 + 11 significant bits: |v - (hi + lo)| <= 2^-22 |v|), products hi*hi, hi*lo,
  // lo*hi, each exact in the fp32 accumulator; the dropped lo*lo <= 2^-22 |v w|.  fp16 has 5 exponent bits, so both operands
  // are first multiplied by a power of two that puts their tensor's largest magnitude at 2^14..2^15 (exact; the weights at
  // pack time, the samples here from amax = max |x| over the launch's inputs) and the sum is scaled back in the epilogue.
  // Measured representation error 8e-8 of the output scale against 5e-7 .. 9e-7 of the fp32 accumulation itself.
  constexpr bool F16 = NPROD == 3;
  constexpr int NPL = F16 ? 2 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* sA = reinterpret_cast<uint16_t*>(smem);                           // [2 buffers][NPL planes][BMS][ASTRS]
  float4* sCw = reinterpret_cast<float4*>(sA + 2 * NPL * PLANE);              // [BMS * taps] bilinear weights
  int4* sCi = reinterpret_cast<int4*>(sCw + BMS * kTapsMax);                  // [BMS * taps] pixel indices

#if ORP_DCNS_OWN_SIMD
  // the kernel claims the whole register budget of its waves (256 VGPRs: two waves fill a SIMD's file), whatever the tile height
  // needs: no wave of another workgroup -- of this kernel or of any other stream's -- runs on a SIMD beside a wave that is in the
  // MFMA loop (what such neighbours suffered)
  asm volatile("" ::: "v255");
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int taps = P.kh * P.kw;
  int tile, conv;
  {   // XCD-aware map: an XCD takes a contiguous slab of ONE layer's tiles (the layer's weights stay in that XCD's L2)
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int nx = P.nconv == 2 ? 4 : 8;
    conv = P.nconv == 2 ? (xcd >> 2) : 0;
    const int xl = P.nconv == 2 ? (xcd & 3) : xcd;
    const int per = (total_tiles + nx - 1) / nx;
    tile = xl * per + slot;
    if (slot >= per || tile >= total_tiles) return;
  }
  int lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if (tile >= P.lv[i].tile0) lvl = i;
  const LevelK L = P.lv[lvl];
  const int HoWo = L.Ho * L.Wo;
  const long npos = (long)P.B * HoWo;
  long p0, plim;                                                              // the tile's positions [p0, plim) of the level's B * HoWo
  int img = 0;
  if (L.tpi > 0) {                                                            // per-image tiles: the tile ends with its image
    const int t_in = tile - L.tile0;
    img = t_in / L.tpi;
    const int pin = (t_in - img * L.tpi) * BMS;
    p0 = (long)img * HoWo + pin;
    plim = p0 + (HoWo - pin < BMS ? HoWo - pin : BMS);
  } else {
    p0 = (long)(tile - L.tile0) * BMS;
    plim = p0 + BMS < npos ? p0 + BMS : npos;
  }
  const float* xin = conv ? L.x[1] : L.x[0];
  float sx = 1.f, osc = 1.f;                                                  // F16: sample scale 2^k, output scale 1 / (sx * sw)
  float* sAB = reinterpret_cast<float*>(sCi + BMS * kTapsMax);                 // [2][Cin] the input normalisation's (a[c]) then (b[c]) (coef_in only)
  if (F16) {
    unsigned am = P.amax[conv * P.amax_stride];
    if (P.amax_count > 1) {                                                   // (block-uniform) the producer left one bound per (tensor, image, group)
      unsigned* red = reinterpret_cast<unsigned*>(sCw);                         // (the table is built after this; no static LDS in front of the dynamic region)
      unsigned m_ = 0u;
      for (int i = tid; i < P.amax_count; i += kThreadsS) m_ = max(m_, P.amax[conv * P.amax_stride + i]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m_ = max(m_, (unsigned)__shfl_xor((int)m_, o, 64));
      if (lane == 0) red[wave] = m_;
      __syncthreads();
      am = red[0];
#pragma unroll
      for (int i = 1; i < kThreadsS / 64; i++) am = max(am, red[i]);
      __syncthreads();                                                          // (before the table build overwrites the scratch)
    }
    int k = am == 0u ? 0 : 14 - ((int)((am >> 23) & 0xffu) - 127);
    k = k < -100 ? -100 : k > 100 ? 100 : k;
    sx = __uint_as_float((unsigned)(127 + k) << 23);
    const float sw = *(L.planes ? L.wscale : conv ? P.wscale[1] : P.wscale[0]);
    osc = 1.f / (sx * sw);
    if (P.dbg && tile == 0 && blockIdx.y == 0 && tid == 0) { P.dbg[conv] = am; P.dbg[2 + conv] = __float_as_uint(sw); }
  }

  // ---- bilinear coefficient table, one entry per (position, tap): deformable_im2col_bilinear (:84-115) hoisted out of the
  //      channel loop; a sample outside (-1, H) x (-1, W) has weight 0 (:229); DCNv2 folds the modulation scalar in (:620) ----
  for (int e = tid; e < BMS * taps; e += kThreadsS) {
    const int m = e / taps, tap = e - m * taps;
    const long p = p0 + m;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 ix = make_int4(0, 0, 0, 0);
    if (p < plim) {
      const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
      const int ho = hw / L.Wo, wo = hw - ho * L.Wo;
      const int ki = tap / P.kw, kj = tap - ki * P.kw;
      if (PLAIN) {
        const int hi = ho * P.sh - P.ph + ki * P.dh, wi = wo * P.sw - P.pw + kj * P.dw;
        if (hi >= 0 && hi < L.H && wi >= 0 && wi < L.W) { w.x = 1.f; ix.x = (b * L.H + hi) * L.W + wi; }
        sCw[e] = w; sCi[e] = ix;
        continue;
      }
      const float* ob = L.off + ((size_t)b * 2 * taps + 2 * tap) * HoWo + hw;
      const float h_im = (float)(ho * P.sh - P.ph + ki * P.dh) + ob[0];
      const float w_im = (float)(wo * P.sw - P.pw + kj * P.dw) + ob[HoWo];
      if (h_im > -1.f && w_im > -1.f && h_im < (float)L.H && w_im < (float)L.W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        const float hh = 1.f - lh, hw_ = 1.f - lw;
#if ORP_DCNS_TABLE_SELECTS
        // (the first formulation, kept as a dev aid: its lane-mask code is where the wrong rows of round 4 came from -- see below)
        const int h_high = h_low + 1, w_high = w_low + 1;
        const bool t_ok = h_low >= 0, b_ok = h_high <= L.H - 1, l_ok = w_low >= 0, r_ok = w_high <= L.W - 1;
        const int hl = t_ok ? h_low : 0, hhg = b_ok ? h_high : L.H - 1, wl = l_ok ? w_low : 0, whg = r_ok ? w_high : L.W - 1;
        w.x = (t_ok && l_ok) ? hh * hw_ : 0.f;
        w.y = (t_ok && r_ok) ? hh * lw : 0.f;
        w.z = (b_ok && l_ok) ? lh * hw_ : 0.f;
        w.w = (b_ok && r_ok) ? lh * lw : 0.f;
#else
        // The four border conditions as 0 / 1 FACTORS and min / max clamps -- no lane masks.  h_low is in [-1, H - 1] here, so
        // t_ok = (h_low >= 0) = min(h_low + 1, 1), b_ok = (h_low + 1 <= H - 1) = min(H - 1 - h_low, 1), likewise l_ok / r_ok; a
        // product times 1.f is the product, times 0.f is +0 (the products are >= 0): the same bits as the selects of the
        // reference (deform_conv_cuda_kernel.cu:84-115).  Why: the compiler turned the selects into v_cmp_*_e64 -> s_and_b64 ->
        // v_cndmask chains between packed-fp32 instructions, and with a second workgroup of this kernel in its K loop on the
        // same CU the mask of w.z arrived with its last lane quarter (lanes 48..63) stale -- w.z = 0 in 16 consecutive table
        // entries, i.e. three wrong A rows in all channels (tests/checks/split_trace.py, sgpr_mask_probe.hip).
        const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(L.H - 1 - h_low, 1);
        const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(L.W - 1 - w_low, 1);
        const int hl = max(h_low, 0), hhg = min(h_low + 1, L.H - 1), wl = max(w_low, 0), whg = min(w_low + 1, L.W - 1);
        w.x = (hh * hw_) * (t_ok * l_ok);
        w.y = (hh * lw) * (t_ok * r_ok);
        w.z = (lh * hw_) * (b_ok * l_ok);
        w.w = (lh * lw) * (b_ok * r_ok);
#endif
        const int base = b * L.H;
        ix.x = (base + hl) * L.W + wl;
        ix.y = (base + hl) * L.W + whg;
        ix.z = (base + hhg) * L.W + wl;
        ix.w = (base + hhg) * L.W + whg;
        if (L.mask) {
          const float mm = L.mask[((size_t)b * taps + tap) * HoWo + hw];
          w.x *= mm; w.y *= mm; w.z *= mm; w.w *= mm;
        }
      }
    }
    sCw[e] = w; sCi[e] = ix;
  }
  const bool has_coef = PLAIN && P.coef_in != nullptr;                        // (block-uniform)
  if (has_coef) {
    const float2* cf = P.coef_in + ((size_t)(conv * P.nlev + lvl) * P.B + img) * P.Cin;
    for (int c = tid; c < P.Cin; c += kThreadsS) { const float2 ab = cf[c]; sAB[c] = ab.x; sAB[P.Cin + c] = ab.y; }
  }
  __syncthreads();
#if ORP_DCNS_TRACE
  const int trace_wg = conv * total_tiles + tile;
  if (P.dbg && !PLAIN)                                    // (the DeformConv instantiation only: a PLAIN neighbour stream does not write)
    for (int e = tid; e < BMS * taps; e += kThreadsS) {
      unsigned* t = P.dbg + 16 + ((size_t)trace_wg * BMS * kTapsMax + e) * 8;
      const float4 w = sCw[e]; const int4 ix = sCi[e];
      t[0] = __float_as_uint(w.x); t[1] = __float_as_uint(w.y); t[2] = __float_as_uint(w.z); t[3] = __float_as_uint(w.w);
      t[4] = ix.x; t[5] = ix.y; t[6] = ix.z; t[7] = ix.w;
    }
#endif

  const int ncb = P.Cin / CBS;
  const int nphase = taps * ncb;
  // A rows: one row piece = 64 channels fp32 = 16 lanes x float4; a wave fetches one neighbour of FOUR rows per instruction
  const int q4 = lane >> 4, c4 = (lane & 15) * 4;
  auto row_of = [&](int g) { return g * 32 + wave * 4 + q4; };
  auto gather_issue = [&](int tap, int cb, int g, float4 (&v)[4]) {
    const int4 ix = sCi[row_of(g) * taps + tap];
    const float* base = xin + cb * CBS + c4;
    if (ORP_DCNS_DBG & 1) { v[0] = v[1] = v[2] = v[3] = make_float4((float)ix.x, (float)ix.y, (float)ix.z, (float)ix.w); return; }
    v[0] = *reinterpret_cast<const float4*>(base + (size_t)ix.x * P.Cin);
    if (PLAIN) return;
    v[1] = *reinterpret_cast<const float4*>(base + (size_t)ix.y * P.Cin);
    v[2] = *reinterpret_cast<const float4*>(base + (size_t)ix.z * P.Cin);
    v[3] = *reinterpret_cast<const float4*>(base + (size_t)ix.w * P.Cin);
  };
  auto combine_store = [&](int tap, int cbk, int g, const float4 (&v)[4], int buf) {
    const int m = row_of(g);
    if (ORP_DCNS_DBG & 8) return;
    const float4 cw = sCw[m * taps + tap];
    // the reference's own float expression, w1*v1 + w2*v2 + w3*v3 + w4*v4 evaluated left to right WITHOUT contraction
    // (deform_conv_cuda_kernel.cu:111-113): the samples are the reference's bits (DCNv1), and the value does not depend on
    // which fused / packed forms the compiler would pick in one instantiation or another
    auto bil = [&](float a, float b, float c, float d) {
      return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(cw.x, a), __fmul_rn(cw.y, b)), __fmul_rn(cw.z, c)), __fmul_rn(cw.w, d));
    };
    float s[4];
    if (PLAIN) {
      const bool in = cw.x != 0.f;
      float4 x = v[0];
      if (has_coef) {                                                         // the previous layer's GroupNorm (+ ReLU) on the way in
        const float4 ca = *reinterpret_cast<const float4*>(sAB + cbk * CBS + c4);
        const float4 cb_ = *reinterpret_cast<const float4*>(sAB + P.Cin + cbk * CBS + c4);
        x.x = fmaf(x.x, ca.x, cb_.x); x.y = fmaf(x.y, ca.y, cb_.y); x.z = fmaf(x.z, ca.z, cb_.z); x.w = fmaf(x.w, ca.w, cb_.w);
        if (P.relu_in) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
      }
      s[0] = in ? x.x : 0.f; s[1] = in ? x.y : 0.f; s[2] = in ? x.z : 0.f; s[3] = in ? x.w : 0.f;   // (the padding of the NORMALISED tensor is 0)
    } else {
      s[0] = bil(v[0].x, v[1].x, v[2].x, v[3].x);
      s[1] = bil(v[0].y, v[1].y, v[2].y, v[3].y);
      s[2] = bil(v[0].z, v[1].z, v[2].z, v[3].z);
      s[3] = bil(v[0].w, v[1].w, v[2].w, v[3].w);
    }
    uint16_t* dst = sA + (size_t)buf * NPL * PLANE + (size_t)m * ASTRS + c4;
    if (F16) {
      _Float16 h[4], l[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float sv = s[i] * sx;                                             // exact (power of two)
        h[i] = (_Float16)sv;                                                    // round to nearest
        l[i] = (_Float16)(sv - (float)h[i]);                                    // the residual is exact in fp32
      }
      const h2 h01 = {h[0], h[1]}, h23 = {h[2], h[3]}, l01 = {l[0], l[1]}, l23 = {l[2], l[3]};
      *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
      *reinterpret_cast<uint2*>(dst + PLANE) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
      return;
    }
    float hi[4], mid[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      hi[i] = __uint_as_float(__float_as_uint(s[i]) & 0xffff0000u);
      const float r1 = s[i] - hi[i];                                           // exact
      mid[i] = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
      lo[i] = r1 - mid[i];                                                     // exact, <= 8 significant bits
    }
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi16(hi[0], hi[1]), pack_hi16(hi[2], hi[3]));
    *reinterpret_cast<uint2*>(dst + PLANE) = make_uint2(pack_hi16(mid[0], mid[1]), pack_hi16(mid[2], mid[3]));
    *reinterpret_cast<uint2*>(dst + 2 * PLANE) = make_uint2(pack_hi16(lo[0], lo[1]), pack_hi16(lo[2], lo[3]));
  };
  // weight fragments of (tap, cb, chunk j): lane (n = lane & 31, kg = lane >> 5) -> 8 k-values of each plane, 16 B loads
  const int n_wave = blockIdx.y * 256 + wave * 32;
  const int mrow = lane & 31, kg = lane >> 5;
  const bool live = n_wave < P.Cout;                      // c_out % 64 == 0 -> ... % 32 == 0: a wave is live or idle as a whole
  const uint16_t* wp = (L.planes ? L.planes : conv ? P.planes[1] : P.planes[0]) + ((size_t)kg * P.Cout + (live ? n_wave : 0) + mrow) * 8;
  const size_t wblk = (size_t)2 * P.Cout * 8;             // elements per 16-channel block
  auto load_b = [&](int tap, int cb, int j, bf8 (&b)[3]) {
    const uint16_t* a = wp + ((size_t)tap * (P.Cin / 16) + cb * NCH + j) * wblk;
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) b[pl] = *reinterpret_cast<const bf8*>(a + (size_t)pl * P.plane_stride);
  };

  // ---- prologue: weight ring and A tile of phase 0 -------------------------------------------------------------------------
  bf8 bq[NCH][3];
#pragma unroll
  for (int j = 0; j < NCH; j++) load_b(0, 0, j, bq[j]);
  {
    float4 g[MT][4];
#pragma unroll
    for (int r = 0; r < MT; r++) gather_issue(0, 0, r, g[r]);
#pragma unroll
    for (int r = 0; r < MT; r++) combine_store(0, 0, r, g[r], 0);
  }
  __syncthreads();

  constexpr bool SIDE = PLAIN && ORP_DCNS_SIDE_ACC;
  floatx16 acc[MT], side[SIDE ? MT : 1];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) acc[mt] = floatx16{0};
#pragma unroll
  for (int mt = 0; mt < (SIDE ? MT : 1); mt++) side[mt] = floatx16{0};

  // (tap, cb) of the NEXT phase, advanced incrementally; past the end it stays on the last phase: the loads of the loop
  // body are UNCONDITIONAL (the final iteration re-fetches the last phase's rows and weights and drops them), so that the
  // compiler's s_waitcnt vmcnt counts are exact -- with the loads under `if (next_phase)` it has to assume the shortest
  // path and made the first MFMA of every phase wait for this phase's own gathers
  // PLAIN (a row group is one float4 per lane): the gathers run TWO phases ahead -- the rows of phase p + 2 go out at the start of
  // phase p into the register set phase p - 1 emptied, and are split into the other LDS buffer during phase p + 1.  With one phase of
  // lead the first row group was consumed one chunk (~300 cycles of this wave's matrix work) after its loads went out and the wave sat
  // in s_waitcnt: the timing variants without gathers / without their consumer both ran 40 us of 218 faster (profiles/r05_anatomy.log).
  constexpr bool AHEAD2 = PLAIN && F16 && ORP_DCNS_AHEAD2;          // (the three-plane modes have no registers left at tile height 3)
  int tap_n = 0, cb_n = 0, tap_n2 = 0, cb_n2 = 0;                             // phase + 1, phase + 2 (clamped to the last phase)
  auto step = [&](int& t, int& c, int ph) {
    if (ph + 1 < nphase) { if (++c == ncb) { c = 0; t++; } }
  };
  step(tap_n, cb_n, 0);
  tap_n2 = tap_n; cb_n2 = cb_n;
  step(tap_n2, cb_n2, 1);
  float4 gA[MT][4], gB[AHEAD2 ? MT : 1][4];
  if (AHEAD2) {
#pragma unroll
    for (int r = 0; r < MT; r++) gather_issue(tap_n, cb_n, r, gA[r]);         // the rows of phase 1
  }
  auto load_a = [&](const uint16_t* abase, int j, bf8 (&a)[MT][3]) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int pl = 0; pl < NPL; pl++)
        if (!(ORP_DCNS_DBG & 16)) a[mt][pl] = *reinterpret_cast<const bf8*>(abase + (size_t)pl * PLANE + (size_t)mt * 32 * ASTRS + j * 16);
  };

  // EARLYBAR (with AHEAD2: the rows are already in registers when the phase starts): the row groups are split into the other buffer
  // in the FIRST chunks, the phase's one barrier sits in front of the last chunk's MFMAs, and the first A fragments of the next phase
  // are read right behind it -- under the last chunk's matrix work instead of in front of the next phase's first MFMA.  Safe: at the
  // barrier every wave has completed its reads of this phase's buffer (the last chunk's fragments are in registers) and its writes of
  // the other one; nobody writes this phase's buffer before the next phase's first chunk.
  constexpr bool EARLYBAR = AHEAD2 && ORP_DCNS_EARLYBAR && !(ORP_DCNS_DBG & 16);
  bf8 a[2][MT][3];
  if (EARLYBAR) {
    const uint16_t* ab0 = sA + (size_t)mrow * ASTRS + 8 * kg;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int pl = 0; pl < NPL; pl++) a[0][mt][pl] = *reinterpret_cast<const bf8*>(ab0 + (size_t)pl * PLANE + (size_t)mt * 32 * ASTRS);
  }
  auto phase_body = [&](int phase, float4 (&g)[MT][4], float4 (&gf)[AHEAD2 ? MT : 1][4]) __attribute__((always_inline)) {
    const int cur = phase & 1;
#if ORP_DCNS_TRACE >= 2
    if (P.dbg && !PLAIN) {   // hash of every row of the buffer this phase reads, as the readers see it (16 threads per row)
      const int r = tid >> 4, sub = tid & 15;
      if (r < BMS) {
        unsigned h = 0;
        for (int pl = 0; pl < NPL; pl++) {
          const uint16_t* src = sA + (size_t)cur * NPL * PLANE + (size_t)pl * PLANE + (size_t)r * ASTRS + sub * 4;
          for (int i = 0; i < 4; i++) h = h * 0x9E3779B1u + src[i] + 1u;
        }
        for (int o = 8; o > 0; o >>= 1) h = h * 31u + (unsigned)__shfl_down((int)h, o, 16);
        if (sub == 0) P.dbg[16 + (size_t)(1 << 20) + ((size_t)trace_wg * 128 + phase) * BMS + r] = h;
      }
    }
#endif
    // (1) the gathers of the next phase's rows go out first: a whole phase of matrix work to land  (AHEAD2: of the phase after it)
#if ORP_DCNS_PRIO & 1
    __builtin_amdgcn_s_setprio(2);
#endif
    if (AHEAD2) {
#pragma unroll
      for (int r = 0; r < MT; r++) gather_issue(tap_n2, cb_n2, r, gf[r]);
    } else {
#pragma unroll
      for (int r = 0; r < MT; r++) gather_issue(tap_n, cb_n, r, g[r]);
    }
    // (the scheduler would otherwise SINK these loads down to their use to save registers -- measured in the ISA: the
    //  gathers ended up between the last MFMAs with s_waitcnt vmcnt(0) right behind them; the barriers pin the pipeline)
    __builtin_amdgcn_sched_barrier(0);
#if ORP_DCNS_PRIO & 1
    __builtin_amdgcn_s_setprio(0);
#endif
    const uint16_t* abase = sA + (size_t)cur * NPL * PLANE + (size_t)mrow * ASTRS + 8 * kg;
    // (2) the phase: the A fragments of chunk j + 1 are read from LDS BEFORE the MFMAs of chunk j are issued (a second
    //     register set), the weight registers of chunk j are refilled for the next phase right after use
    if (ORP_DCNS_DBG & 16) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
          for (int pl = 0; pl < 3; pl++) a[i][mt][pl] = bq[(i + mt + pl) & 3][pl];
    }
    if (!EARLYBAR) load_a(abase, 0, a[0]);
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      if (j + 1 < NCH) load_a(abase, j + 1, a[(j + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      // (3) the last chunk's scheduling region also holds the combine + split of the gathered rows into the OTHER buffer
      //     (nobody reads it before the barrier below; everybody finished reading it before the barrier that ended the
      //     previous phase): ~70 VALU per row group that the scheduler can place in the shadow of the region's MFMAs
      //     (a 32-cycle MFMA leaves ~5 issue slots) instead of a VALU-only tail during which the matrix pipe idles
      //     The row groups ride in the LAST CC chunks, one each (CC = MT).  (All of them in the last chunk, so that the
      //     gathers have three chunks to land instead of one, measured no faster with fp16 pieces: 211 vs 200 us.)
      constexpr int CC = (ORP_DCNS_CC > 0) ? (ORP_DCNS_CC < MT ? ORP_DCNS_CC : MT) : MT;
      const bool with_combine = ORP_DCNS_COMBINE_IN_LAST && (EARLYBAR ? j < CC : j >= NCH - CC);
      const int jc = EARLYBAR ? j : j - (NCH - CC);
#if ORP_DCNS_PRIO & 2
      if (with_combine) __builtin_amdgcn_s_setprio(2);                          // dev aid: the chunks that carry the combine issue ahead of the SIMD's other wave
      else __builtin_amdgcn_s_setprio(0);
#endif
      if (with_combine) {
#pragma unroll
        for (int r = 0; r < MT; r++)
          if (r * CC / MT == jc) combine_store(tap_n, cb_n, r, g[r], cur ^ 1);
      }
      if (EARLYBAR && j == NCH - 1) {
        if (!(ORP_DCNS_DBG & 32)) __syncthreads();
        load_a(sA + (size_t)(cur ^ 1) * NPL * PLANE + (size_t)mrow * ASTRS + 8 * kg, 0, a[0]);
        __builtin_amdgcn_sched_barrier(0);
      }
      Products<F16 ? 0 : 9 - NPROD, F16 ? 3 : 9, MT, OUT_NCHW, SIDE, F16>::run(acc, side, a[j & 1], bq[j]);
#if ORP_DCNS_FENCE
      __builtin_amdgcn_sched_barrier(0);                  // the chunk's MFMAs stay together, its refills behind them
#endif
#if ORP_DCNS_REFILL_LAG
      // the registers of chunk j - 1 are refilled one chunk LATER, behind the MFMAs of chunk j (chunk NCH - 1: after the loop)
      if (!(ORP_DCNS_DBG & 2) && j > 0) load_b(tap_n, cb_n, j - 1, bq[j - 1]);
#else
      if (!(ORP_DCNS_DBG & 2)) load_b(tap_n, cb_n, j, bq[j]);
#endif
      if (with_combine && ORP_DCNS_INTERLEAVE > 0) {
        // pin the interleave: one MFMA, then the chunk's share of the combine's VALU in its 32-cycle shadow, ...
        constexpr int kGroups = (MT + CC - 1) / CC;
#pragma unroll
        for (int i = 0; i < NPROD * MT; i++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, ORP_DCNS_INTERLEAVE * kGroups, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!ORP_DCNS_COMBINE_IN_LAST) {
#pragma unroll
      for (int r = 0; r < MT; r++) combine_store(tap_n, cb_n, r, g[r], cur ^ 1);
    }
#if ORP_DCNS_REFILL_LAG
    if (!(ORP_DCNS_DBG & 2)) load_b(tap_n, cb_n, NCH - 1, bq[NCH - 1]);
#endif
    step(tap_n, cb_n, phase + 1);
    step(tap_n2, cb_n2, phase + 2);
    if (!EARLYBAR && !(ORP_DCNS_DBG & 32)) __syncthreads();
  };
  if constexpr (AHEAD2) {
#pragma unroll 1
    for (int phase = 0; phase < nphase; phase += 2) {
      phase_body(phase, gA, gB);                                               // splits gA (rows of phase + 1), fills gB (phase + 2)
      if (phase + 1 < nphase) phase_body(phase + 1, gB, gA);
    }
  } else {
#pragma unroll 1
    for (int phase = 0; phase < nphase; phase++) phase_body(phase, gA, gB);
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------------------------
  if (!live) return;
  if (SIDE) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] += side[mt];
  }
  const float* bias = L.planes ? L.bias : conv ? P.bias[1] : P.bias[0];
  float* outp = conv ? L.out[1] : L.out[0];
  bool scaled = false;
  if (PLAIN && !OUT_NCHW && P.gn_part) {
    // GroupNorm statistics of the tile while it is in registers: lane = channel (n_wave + lane % 32), its 16 * MT values = the
    // rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) + 32 mt; a group = Cout / G consecutive channels = consecutive lanes of both
    // half-waves.  Two passes (mean, then M2 around it), fixed shuffle order; orp_conv_split_gn_finish merges the tiles of an
    // image (Chan et al.) in tile order.
    if (F16) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++) acc[mt] *= osc;
      scaled = true;
    }
    const int cg = P.Cout / P.G, nrow = (int)(plim - p0);
    auto row_ok = [&](int mt, int r) { return mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) < nrow; };
    auto group_sum = [&](float v) {
      for (int o = 1; o < cg; o <<= 1) v += __shfl_xor(v, o, 64);
      return v + __shfl_xor(v, 32, 64);
    };
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int r = 0; r < 16; r++) sum += row_ok(mt, r) ? acc[mt][r] : 0.f;
    const float cnt = (float)(nrow * cg);
    const float mean = group_sum(sum) / cnt;
    float m2 = 0.f, mx = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float d = acc[mt][r] - mean;
        m2 += row_ok(mt, r) ? d * d : 0.f;
        mx = fmaxf(mx, row_ok(mt, r) ? fabsf(acc[mt][r]) : 0.f);
      }
    m2 = group_sum(m2);
    for (int o = 1; o < cg; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lane < 32 && (lane & (cg - 1)) == 0)
      P.gn_part[((size_t)conv * total_tiles + tile) * P.G + (n_wave + lane) / cg] = make_float4(mean, m2, mx, cnt);
  }
  auto finish = [&](float v, int ch) { if (F16 && !scaled) v *= osc; if (bias) v += bias[ch]; return P.relu ? fmaxf(v, 0.f) : v; };
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    if (OUT_NCHW) {
      const long p = p0 + mt * 32 + (lane & 31);
      if (p < plim) {
        const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
        float* ob = outp + (size_t)b * P.Cout * HoWo + hw;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int ch = n_wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          ob[(size_t)ch * HoWo] = finish(acc[mt][r], ch);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const long p = p0 + mt * 32 + m;
        if (p < plim) outp[(size_t)p * P.Cout + n_wave + (lane & 31)] = finish(acc[mt][r], n_wave + (lane & 31));
      }
    }
  }
}

// ================================================================================================================================
// Round 6: the same tile, WAVE-SPECIALISED (fp16-pieces mode only; ORP_DCNS_WS=1 -- built, measured, NOT the default).
//
// The kernel above gives every wave every job: gather rows, combine / split them into LDS, read A fragments, stream its 32 output
// channels' weights, issue MFMAs -- and the two waves of a SIMD do the same job at the same time.  Here the 8 waves split into 4
// CONSUMERS (waves 0-3, one per SIMD: A fragments from LDS, weights from L2, MFMAs, epilogue; each owns the whole tile height x 64
// output channels = MT x 2 accumulator blocks) and 4 PRODUCERS (waves 4-7, the other wave of each SIMD: row gathers two phases
// ahead, bilinear combine / GroupNorm-on-the-way-in, fp16 split, LDS writes).  Every A fragment is read by 4 waves instead of 8 (LDS
// read traffic per phase 196 -> 98 KB), the weight stream is unchanged.  The two roles are two separate loops, so the kernel needs
// max(consumer, producer) registers; A fragments and weights are refilled in place behind the MFMAs that read them.
//
// What the measurements say (profiles/r06_ws_anatomy.log; pair launches at 1024^2, us incl. the range pre-pass and the host's launch
// gap, symmetric kernel 242 DeformConv / 185 convolution):
//   * as first written 315 / 232: the producers' VALU starves beside the consumers' MFMAs (issue arbitration is priority, then age).
//     s_setprio 3 on the producers: 244 / 175.  The DeformConv launch gains nothing -- its producers (349 VALU + 24 row fetches per
//     wave and phase) take as long as a phase's 72 MFMAs --, the convolution gains 5 %.
//   * consumers alone 178 / 166, consumers without weight refills 158 / 148, consumers issuing nothing but MFMAs 156 / 146: the tile
//     structure's floor (2 tiles per CU, 2 592 MFMAs per SIMD and tile at the ~1.4 GHz the part sustains under dense MFMAs, plus
//     prologue / epilogue) is ~146 us -- the symmetric kernel's 180 us of the bench is within 20 % of it.
//   * L2: 93 % hits, 132 cycles average read latency, 11 TB/s of 34 (profiles/r06_l2_counters.log): not the bound.
// Why it is not the default: the convolution instantiation of the symmetric kernel keeps the two small partial products in a second
// accumulator set (SIDE); a consumer would need 2 x 96 accumulator registers + fragments > 256 (the attempt spills inside the MFMA
// loop), and without SIDE the error against float64 is 1.28e-6 where tests/test_gpu_conv_split.py admits 1.20e-6 (1.5 x the library's
// own).  The DeformConv instantiation is bit-identical to the symmetric kernel's (same products, same order, no SIDE there).
#ifndef ORP_WS_PRIO
#define ORP_WS_PRIO 0                // dev aid: s_setprio of the consumer waves (0: none)
#endif
#ifndef ORP_WS_SIDE
#define ORP_WS_SIDE 0                // PLAIN: second accumulator set for the small partial products
#endif
#ifndef ORP_WS_PPRIO
#define ORP_WS_PPRIO 3               // s_setprio of the producer waves: without it their VALU starves beside the consumers' MFMAs (pair launch 314 us, with it 243)
#endif
#ifndef ORP_WS_ROT
#define ORP_WS_ROT 0
#endif
#ifndef ORP_WS_DBG
#define ORP_WS_DBG 0                 // dev aid (timing only, wrong results): 1 = producers idle, 2 = consumers issue no MFMA, 4 = no weight refills, 8 = no A-fragment reads, 16 = producers: no gathers, 32 = producers: gathers only (no combine / split / LDS write)
#endif

template <int MT, bool OUT_NCHW, bool PLAIN>
__global__ void __launch_bounds__(kThreadsS)
dcn_fwd_split_ws_kernel(const FwdS P, int total_tiles) {
  constexpr int BMS = 32 * MT;
  constexpr int PLANE = BMS * ASTRS;
  constexpr int NPL = 2, NT = 2;
  constexpr int RG = 2 * MT;                                                  // row groups (4 rows each) of one producer wave per phase
  constexpr int NB = PLAIN ? 1 : 4;                                           // neighbours fetched per sample
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* sA = reinterpret_cast<uint16_t*>(smem);                           // [2 buffers][2 planes][BMS][ASTRS]
  float4* sCw = reinterpret_cast<float4*>(sA + 2 * NPL * PLANE);
  int4* sCi = reinterpret_cast<int4*>(sCw + BMS * kTapsMax);
  float* sAB = reinterpret_cast<float*>(sCi + BMS * kTapsMax);
#if ORP_DCNS_OWN_SIMD
  asm volatile("" ::: "v255");
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the role branch is an s_cbranch)
  const int taps = P.kh * P.kw;
  int tile, conv;
  {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int nx = P.nconv == 2 ? 4 : 8;
    conv = P.nconv == 2 ? (xcd >> 2) : 0;
    const int xl = P.nconv == 2 ? (xcd & 3) : xcd;
    const int per = (total_tiles + nx - 1) / nx;
    tile = xl * per + slot;
    if (slot >= per || tile >= total_tiles) return;
  }
  int lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if (tile >= P.lv[i].tile0) lvl = i;
  const LevelK L = P.lv[lvl];
  const int HoWo = L.Ho * L.Wo;
  const long npos = (long)P.B * HoWo;
  long p0, plim;
  int img = 0;
  if (L.tpi > 0) {
    const int t_in = tile - L.tile0;
    img = t_in / L.tpi;
    const int pin = (t_in - img * L.tpi) * BMS;
    p0 = (long)img * HoWo + pin;
    plim = p0 + (HoWo - pin < BMS ? HoWo - pin : BMS);
  } else {
    p0 = (long)(tile - L.tile0) * BMS;
    plim = p0 + BMS < npos ? p0 + BMS : npos;
  }
  const float* xin = conv ? L.x[1] : L.x[0];
  float sx, osc;
  {
    unsigned am = P.amax[conv * P.amax_stride];
    if (P.amax_count > 1) {
      unsigned* red = reinterpret_cast<unsigned*>(sCw);
      unsigned m_ = 0u;
      for (int i = tid; i < P.amax_count; i += kThreadsS) m_ = max(m_, P.amax[conv * P.amax_stride + i]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m_ = max(m_, (unsigned)__shfl_xor((int)m_, o, 64));
      if (lane == 0) red[wave] = m_;
      __syncthreads();
      am = red[0];
#pragma unroll
      for (int i = 1; i < kThreadsS / 64; i++) am = max(am, red[i]);
      __syncthreads();
    }
    int k = am == 0u ? 0 : 14 - ((int)((am >> 23) & 0xffu) - 127);
    k = k < -100 ? -100 : k > 100 ? 100 : k;
    sx = __uint_as_float((unsigned)(127 + k) << 23);
    const float sw = *(L.planes ? L.wscale : conv ? P.wscale[1] : P.wscale[0]);
    osc = 1.f / (sx * sw);
    if (P.dbg && tile == 0 && blockIdx.y == 0 && tid == 0) { P.dbg[conv] = am; P.dbg[2 + conv] = __float_as_uint(sw); }
  }

  // ---- coefficient table (same as the symmetric kernel: see there for the border-factor formulation) ----
  for (int e = tid; e < BMS * taps; e += kThreadsS) {
    const int m = e / taps, tap = e - m * taps;
    const long p = p0 + m;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 ix = make_int4(0, 0, 0, 0);
    if (p < plim) {
      const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
      const int ho = hw / L.Wo, wo = hw - ho * L.Wo;
      const int ki = tap / P.kw, kj = tap - ki * P.kw;
      if (PLAIN) {
        const int hi = ho * P.sh - P.ph + ki * P.dh, wi = wo * P.sw - P.pw + kj * P.dw;
        if (hi >= 0 && hi < L.H && wi >= 0 && wi < L.W) { w.x = 1.f; ix.x = (b * L.H + hi) * L.W + wi; }
        sCw[e] = w; sCi[e] = ix;
        continue;
      }
      const float* ob = L.off + ((size_t)b * 2 * taps + 2 * tap) * HoWo + hw;
      const float h_im = (float)(ho * P.sh - P.ph + ki * P.dh) + ob[0];
      const float w_im = (float)(wo * P.sw - P.pw + kj * P.dw) + ob[HoWo];
      if (h_im > -1.f && w_im > -1.f && h_im < (float)L.H && w_im < (float)L.W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        const float hh = 1.f - lh, hw_ = 1.f - lw;
        const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(L.H - 1 - h_low, 1);
        const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(L.W - 1 - w_low, 1);
        const int hl = max(h_low, 0), hhg = min(h_low + 1, L.H - 1), wl = max(w_low, 0), whg = min(w_low + 1, L.W - 1);
        w.x = (hh * hw_) * (t_ok * l_ok);
        w.y = (hh * lw) * (t_ok * r_ok);
        w.z = (lh * hw_) * (b_ok * l_ok);
        w.w = (lh * lw) * (b_ok * r_ok);
        const int base = b * L.H;
        ix.x = (base + hl) * L.W + wl;
        ix.y = (base + hl) * L.W + whg;
        ix.z = (base + hhg) * L.W + wl;
        ix.w = (base + hhg) * L.W + whg;
        if (L.mask) {
          const float mm = L.mask[((size_t)b * taps + tap) * HoWo + hw];
          w.x *= mm; w.y *= mm; w.z *= mm; w.w *= mm;
        }
      }
    }
    sCw[e] = w; sCi[e] = ix;
  }
  const bool has_coef = PLAIN && P.coef_in != nullptr;
  if (has_coef) {
    const float2* cf = P.coef_in + ((size_t)(conv * P.nlev + lvl) * P.B + img) * P.Cin;
    for (int c = tid; c < P.Cin; c += kThreadsS) { const float2 ab = cf[c]; sAB[c] = ab.x; sAB[P.Cin + c] = ab.y; }
  }
  __syncthreads();

  const int ncb = P.Cin / CBS;
  const int nphase = taps * ncb;
#if ORP_WS_ROT
  // dev aid (hypothesis test): the workgroups walk the channel blocks in rotated orders, so that at any moment the CUs of an XCD read
  // different 256-byte columns of the 1 KB input rows (L2 channel = address bits above the 256-byte piece?)
  const int rot = (tile * ORP_WS_ROT) % ncb;
  auto phys = [&](int cb) { const int c = cb + rot; return c >= ncb ? c - ncb : c; };
#else
  auto phys = [&](int cb) { return cb; };
#endif
  const bool consumer = wave < 4;                                              // (waves w and w + 4 share a SIMD: one of each role per SIMD)
  const int wq = wave & 3;

  // ---- producer side ----------------------------------------------------------------------------------------------------------
  const int q4 = lane >> 4, c4 = (lane & 15) * 4;
  auto row_of = [&](int g) { return g * 16 + wq * 4 + q4; };
  auto gather_issue = [&](int tap, int cb, int g, float4 (&v)[NB]) {
    const int4 ix = sCi[row_of(g) * taps + tap];
    const float* base = xin + phys(cb) * CBS + c4;
    v[0] = *reinterpret_cast<const float4*>(base + (size_t)ix.x * P.Cin);
    if constexpr (!PLAIN) {
      v[1] = *reinterpret_cast<const float4*>(base + (size_t)ix.y * P.Cin);
      v[2] = *reinterpret_cast<const float4*>(base + (size_t)ix.z * P.Cin);
      v[3] = *reinterpret_cast<const float4*>(base + (size_t)ix.w * P.Cin);
    }
  };
  auto combine_store = [&](int tap, int cbk, int g, const float4 (&v)[NB], int buf) {
    const int m = row_of(g);
    const float4 cw = sCw[m * taps + tap];
    float s[4];
    if constexpr (PLAIN) {
      const bool in = cw.x != 0.f;
      float4 x = v[0];
      if (has_coef) {
        const float4 ca = *reinterpret_cast<const float4*>(sAB + phys(cbk) * CBS + c4);
        const float4 cb_ = *reinterpret_cast<const float4*>(sAB + P.Cin + phys(cbk) * CBS + c4);
        x.x = fmaf(x.x, ca.x, cb_.x); x.y = fmaf(x.y, ca.y, cb_.y); x.z = fmaf(x.z, ca.z, cb_.z); x.w = fmaf(x.w, ca.w, cb_.w);
        if (P.relu_in) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
      }
      s[0] = in ? x.x : 0.f; s[1] = in ? x.y : 0.f; s[2] = in ? x.z : 0.f; s[3] = in ? x.w : 0.f;
    } else {
      auto bil = [&](float a, float b, float c, float d) {     // deform_conv_cuda_kernel.cu:111-113, left to right, unfused
        return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(cw.x, a), __fmul_rn(cw.y, b)), __fmul_rn(cw.z, c)), __fmul_rn(cw.w, d));
      };
      s[0] = bil(v[0].x, v[1].x, v[2].x, v[3].x);
      s[1] = bil(v[0].y, v[1].y, v[2].y, v[3].y);
      s[2] = bil(v[0].z, v[1].z, v[2].z, v[3].z);
      s[3] = bil(v[0].w, v[1].w, v[2].w, v[3].w);
    }
    uint16_t* dst = sA + (size_t)buf * NPL * PLANE + (size_t)m * ASTRS + c4;
    _Float16 h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float sv = s[i] * sx;
      h[i] = (_Float16)sv;
      l[i] = (_Float16)(sv - (float)h[i]);
    }
    const h2 h01 = {h[0], h[1]}, h23 = {h[2], h[3]}, l01 = {l[0], l[1]}, l23 = {l[2], l[3]};
    *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    *reinterpret_cast<uint2*>(dst + PLANE) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
  };

  // ---- consumer side ----------------------------------------------------------------------------------------------------------
  const int n_wave = blockIdx.y * 256 + wq * 64;
  const int mrow = lane & 31, kg = lane >> 5;
  const bool live = n_wave < P.Cout;                                           // Cout % 64 == 0
  const uint16_t* wp = (L.planes ? L.planes : conv ? P.planes[1] : P.planes[0]) + ((size_t)kg * P.Cout + (live ? n_wave : 0) + mrow) * 8;
  const size_t wblk = (size_t)2 * P.Cout * 8;
  auto load_b = [&](int tap, int cb, int j, bf8 (&b)[NT][NPL]) {
    const uint16_t* a = wp + ((size_t)tap * (P.Cin / 16) + phys(cb) * NCH + j) * wblk;
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int pl = 0; pl < NPL; pl++) b[nt][pl] = *reinterpret_cast<const bf8*>(a + (size_t)pl * P.plane_stride + nt * 32 * 8);
  };

  int tap_n = 0, cb_n = 0, tap_n2 = 0, cb_n2 = 0;                             // (tap, channel block) of phase + 1 / phase + 2, clamped to the last phase
  auto step = [&](int& t, int& c, int ph) {
    if (ph + 1 < nphase) { if (++c == ncb) { c = 0; t++; } }
  };
  step(tap_n, cb_n, 0);
  tap_n2 = tap_n; cb_n2 = cb_n;
  step(tap_n2, cb_n2, 1);

  // The two roles are two separate loops (not one loop with a branch inside): their register sets are then disjoint live ranges and
  // the kernel needs max(consumer, producer) registers, not the sum.  Both execute exactly nphase + 1 barriers.
  if (!consumer) {
    float4 gA[RG][NB], gB[RG][NB];
#pragma unroll
    for (int r = 0; r < RG; r++) gather_issue(0, 0, r, gB[r]);
#pragma unroll
    for (int r = 0; r < RG; r++) gather_issue(tap_n, cb_n, r, gA[r]);          // the rows of phase 1
#pragma unroll
    for (int r = 0; r < RG; r++) combine_store(0, 0, r, gB[r], 0);
    __syncthreads();
#if ORP_WS_PPRIO
    __builtin_amdgcn_s_setprio(ORP_WS_PPRIO);
#endif
    auto produce = [&](int phase, float4 (&g)[RG][NB], float4 (&gf)[RG][NB]) __attribute__((always_inline)) {
      if (!(ORP_WS_DBG & 1)) {
        if (!(ORP_WS_DBG & 16)) {
#pragma unroll
          for (int r = 0; r < RG; r++) gather_issue(tap_n2, cb_n2, r, gf[r]);  // the rows of phase + 2: a whole phase to land
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(ORP_WS_DBG & 32)) {
#pragma unroll
          for (int r = 0; r < RG; r++) combine_store(tap_n, cb_n, r, g[r], (phase & 1) ^ 1);
        } else {
#pragma unroll
          for (int r = 0; r < RG; r++) asm volatile("" :: "v"(g[r][0].x), "v"(g[r][NB - 1].w));
        }
      }
      step(tap_n, cb_n, phase + 1);
      step(tap_n2, cb_n2, phase + 2);
      __syncthreads();
    };
#pragma unroll 1
    for (int phase = 0; phase < nphase; phase += 2) {
      produce(phase, gA, gB);
      if (phase + 1 < nphase) produce(phase + 1, gB, gA);
    }
    return;
  }

  // Register plan of a consumer (256 per lane): accumulators MT x NT x 16 = 96; PLAIN: a second set for the two small partial
  // products (SIDE, as in the symmetric kernel: the main chain then rounds once per 16 channels at the output's magnitude instead of
  // three times -- the accuracy gate of tests/test_gpu_conv_split.py) = 192.  What is left holds ONE set of A fragments (24) and a weight
  // ring of BR chunks (16 each).  Both are refilled IN PLACE right behind the MFMAs that read them (an issued MFMA has read its A / B
  // operands: tests/checks/mfma_war.hip): a chunk's MFMAs run tile row by tile row (mt outer), the A fragments of row mt are re-read
  // for the next chunk as soon as its 6 MFMAs are out -- 12 MFMAs = 384 cycles before their next use --, the weights of chunk j are
  // replaced by those of chunk j + BR behind the chunk's last MFMA.
  constexpr bool SIDE = PLAIN && ORP_WS_SIDE;
  constexpr int BR = SIDE ? 2 : NCH;                                           // weight ring depth in chunks
  bf8 bq[BR][NT][NPL];
  bf8 af[MT][NPL];
  floatx16 acc[MT][NT], side[SIDE ? MT : 1][SIDE ? NT : 1];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) { acc[mt][nt] = floatx16{0}; if (SIDE) side[SIDE ? mt : 0][SIDE ? nt : 0] = floatx16{0}; }
#pragma unroll
  for (int j = 0; j < BR; j++) load_b(0, 0, j, bq[j]);
  __syncthreads();
#if ORP_WS_PRIO
  __builtin_amdgcn_s_setprio(ORP_WS_PRIO);
#endif
  int tap_c = 0, cb_c = 0;                                                     // (tap, channel block) of the current phase
  auto load_a_row = [&](const uint16_t* abase, int j, int mt) {
#pragma unroll
    for (int pl = 0; pl < NPL; pl++)
      if (!(ORP_WS_DBG & 8)) af[mt][pl] = *reinterpret_cast<const bf8*>(abase + (size_t)pl * PLANE + (size_t)mt * 32 * ASTRS + j * 16);
  };
  if (ORP_WS_DBG & 8) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int pl = 0; pl < NPL; pl++) af[mt][pl] = bq[mt & (BR - 1)][pl][pl];
  }
#pragma unroll 1
  for (int phase = 0; phase < nphase; phase++) {
    const uint16_t* abase = sA + (size_t)(phase & 1) * NPL * PLANE + (size_t)mrow * ASTRS + 8 * kg;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) load_a_row(abase, 0, mt);
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        if (!(ORP_WS_DBG & 2)) {
#pragma unroll
          for (int t = 0; t < 3; t++) {                                        // lo*hi, hi*lo, hi*hi (smallest first)
            const int pa = t == 0 ? 1 : 0, pb = t == 1 ? 1 : 0;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
              floatx16& d = (SIDE && t < 2) ? side[SIDE ? mt : 0][SIDE ? nt : 0] : acc[mt][nt];
              const h8 av = __builtin_bit_cast(h8, af[mt][pa]), bv = __builtin_bit_cast(h8, bq[j % BR][nt][pb]);
              if (OUT_NCHW) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, av, d, 0, 0, 0);
              else          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, d, 0, 0, 0);
            }
          }
        } else {
          acc[0][0][0] += (float)af[mt][0][0] * (float)bq[j % BR][0][0][0];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < NCH) load_a_row(abase, j + 1, mt);                          // in place, for the next chunk
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(ORP_WS_DBG & 4)) {                                                 // the ring slot of chunk j takes chunk j + BR (of the next phase when past this one's end)
        if (j + BR < NCH) load_b(tap_c, cb_c, j + BR, bq[j % BR]);
        else              load_b(tap_n, cb_n, j + BR - NCH, bq[j % BR]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    tap_c = tap_n; cb_c = cb_n;
    step(tap_n, cb_n, phase + 1);
    __syncthreads();
  }
  if (SIDE) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) acc[mt][nt] += side[SIDE ? mt : 0][SIDE ? nt : 0];
  }

  // ---- epilogue (consumers) ------------------------------------------------------------------------------------------------------
  if (!live) return;
#if ORP_WS_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  const float* bias = L.planes ? L.bias : conv ? P.bias[1] : P.bias[0];
  float* outp = conv ? L.out[1] : L.out[0];
  bool scaled = false;
  if (PLAIN && !OUT_NCHW && P.gn_part) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++) acc[mt][nt] *= osc;
    scaled = true;
    const int cg = P.Cout / P.G, nrow = (int)(plim - p0);
    auto row_ok = [&](int mt, int r) { return mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) < nrow; };
    auto group_sum = [&](float v) {
      for (int o = 1; o < cg; o <<= 1) v += __shfl_xor(v, o, 64);
      return v + __shfl_xor(v, 32, 64);
    };
    const float cnt = (float)(nrow * cg);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      float sum = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) sum += row_ok(mt, r) ? acc[mt][nt][r] : 0.f;
      const float mean = group_sum(sum) / cnt;
      float m2 = 0.f, mx = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float d = acc[mt][nt][r] - mean;
          m2 += row_ok(mt, r) ? d * d : 0.f;
          mx = fmaxf(mx, row_ok(mt, r) ? fabsf(acc[mt][nt][r]) : 0.f);
        }
      m2 = group_sum(m2);
      for (int o = 1; o < cg; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (lane < 32 && (lane & (cg - 1)) == 0)
        P.gn_part[((size_t)conv * total_tiles + tile) * P.G + (n_wave + nt * 32 + lane) / cg] = make_float4(mean, m2, mx, cnt);
    }
  }
  auto finish = [&](float v, int ch) { if (!scaled) v *= osc; if (bias) v += bias[ch]; return P.relu ? fmaxf(v, 0.f) : v; };
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int nb = n_wave + nt * 32;
      if (OUT_NCHW) {
        const long p = p0 + mt * 32 + (lane & 31);
        if (p < plim) {
          const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
          float* ob = outp + (size_t)b * P.Cout * HoWo + hw;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int ch = nb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            ob[(size_t)ch * HoWo] = finish(acc[mt][nt][r], ch);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const long p = p0 + mt * 32 + m;
          if (p < plim) outp[(size_t)p * P.Cout + nb + (lane & 31)] = finish(acc[mt][nt][r], nb + (lane & 31));
        }
      }
    }
}

constexpr int kCoefCinMax = 512;                                              // orp_conv_split_multi_gn: input channels of a layer that normalises on the way in
template <int MT, int NPL>
constexpr size_t split_smem() {
  return (size_t)2 * NPL * 32 * MT * ASTRS * 2 + (sizeof(float4) + sizeof(int4)) * 32 * MT * kTapsMax + sizeof(float) * 2 * kCoefCinMax;
}

static const int g_ws = getenv("ORP_DCNS_WS") ? atoi(getenv("ORP_DCNS_WS")) : 0;   // 1: the wave-specialised kernel in the fp16-pieces mode (measured, not the default: see its header)

template <int MT, int NPROD, bool OUT_NCHW, bool PLAIN>
hipError_t launch_one(const FwdS& P, int tiles, int nblk_n, hipStream_t st) {
  constexpr size_t smem = split_smem<MT, NPROD == 3 ? 2 : 3>();
  if constexpr (NPROD == 3) {
    if (g_ws) {
      struct TagW {};
      hipError_t e = orp::set_max_dynamic_lds_once<TagW>(reinterpret_cast<const void*>(&dcn_fwd_split_ws_kernel<MT, OUT_NCHW, PLAIN>), smem);
      if (e != hipSuccess) return e;
      const int nx = P.nconv == 2 ? 4 : 8;
      const int per = (tiles + nx - 1) / nx;
      hipLaunchKernelGGL((dcn_fwd_split_ws_kernel<MT, OUT_NCHW, PLAIN>), dim3(per * 8, nblk_n), dim3(kThreadsS), smem, st, P, tiles);
      return hipGetLastError();
    }
  }
  struct Tag {};
  hipError_t e = orp::set_max_dynamic_lds_once<Tag>(reinterpret_cast<const void*>(&dcn_fwd_split_kernel<MT, NPROD, OUT_NCHW, PLAIN>), smem);
  if (e != hipSuccess) return e;
  const int nx = P.nconv == 2 ? 4 : 8;
  const int per = (tiles + nx - 1) / nx;
  hipLaunchKernelGGL((dcn_fwd_split_kernel<MT, NPROD, OUT_NCHW, PLAIN>), dim3(per * 8, nblk_n), dim3(kThreadsS), smem, st, P, tiles);
  return hipGetLastError();
}

template <int MT, int NPROD, bool PLAIN>
hipError_t launch_l(const FwdS& P, int tiles, int nblk_n, bool nchw, hipStream_t st) {
  return nchw ? launch_one<MT, NPROD, true, PLAIN>(P, tiles, nblk_n, st) : launch_one<MT, NPROD, false, PLAIN>(P, tiles, nblk_n, st);
}
template <int NPROD, bool PLAIN>
hipError_t launch_m(int MT, const FwdS& P, int tiles, int nblk_n, bool nchw, hipStream_t st) {
  return MT == 1 ? launch_l<1, NPROD, PLAIN>(P, tiles, nblk_n, nchw, st)
       : MT == 2 ? launch_l<2, NPROD, PLAIN>(P, tiles, nblk_n, nchw, st) : launch_l<3, NPROD, PLAIN>(P, tiles, nblk_n, nchw, st);
}

inline int out_dim(int in, int pad, int dil, int k, int stride) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }

}  // namespace

bool shape_ok(int c_in, int c_out, int kh, int kw) {
  return kh * kw <= kTapsMax && c_in % CBS == 0 && c_in >= CBS && c_out % 64 == 0 && c_out >= 64;
}

// one layer's planes in uint16_t elements: three bf16 planes, then two fp16 planes, then 8 elements that hold the fp16
// planes' scale (a float) and the weights' max |w| (float bits): [3N bf16][2N fp16][scale, amax, pad]
size_t plane_elems(int c_out, int c_in, int taps) { return (size_t)5 * c_out * c_in * taps + 8; }

const uint16_t* planes_of(const float* packed, int c_out, int c_in, int taps, int nprod) {
  const size_t n = (size_t)c_out * c_in * taps;
  const uint16_t* base = reinterpret_cast<const uint16_t*>(packed + 2 * n);
  return nprod == 3 ? base + 3 * n : base;
}
const float* wscale_of(const float* packed, int c_out, int c_in, int taps) {
  const size_t n = (size_t)c_out * c_in * taps;
  return reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(packed + 2 * n) + 5 * n);
}

hipError_t pack_planes(const float* weight, int c_out, int c_in, int taps, uint16_t* planes, hipStream_t st) {
  const long total = (long)c_out * c_in * taps;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_planes_kernel, dim3(blocks), dim3(256), 0, st, weight, c_out, c_in, taps, planes);
  // the fp16 planes: max |w| -> scale -> two planes
  float* tail = reinterpret_cast<float*>(planes + 5 * total);              // [0] scale, [1] max |w| bits
  unsigned* amax = reinterpret_cast<unsigned*>(tail + 1);
  hipError_t e = orp::fill_async(amax, 0, sizeof(unsigned), st);
  if (e != hipSuccess) return e;
  AbsMaxArgs M;
  long nb = (total / 4 + 256 * 8 - 1) / (256 * 8); if (nb < 1) nb = 1; if (nb > 512) nb = 512;
  for (int i = 0; i < kAbsMaxT; i++) { M.x[i] = weight; M.n[i] = i == 0 ? (size_t)total : 0; M.slot[i] = 0; M.bx0[i] = i == 0 ? 0 : (int)nb; }
  M.bx0[kAbsMaxT] = (int)nb; M.count = 1;
  hipLaunchKernelGGL(absmax_kernel, dim3((int)nb), dim3(256), 0, st, M, amax);
  hipLaunchKernelGGL(pack_planes16_kernel, dim3(blocks), dim3(256), 0, st, weight, c_out, c_in, taps, amax, planes + 3 * total, tail);
  return hipGetLastError();
}

Plan plan(const Args& a) {
  Plan pl;
  long npos_all = 0;
  for (int i = 0; i < a.nlev; i++) npos_all += (long)a.B * a.lv[i].Ho * a.lv[i].Wo;
  // tile height: rounds x height on 256 CUs (one layer) / 128 CUs per layer (pair: the grid halves run side by side), times
  // what a 32-position unit costs at that height -- a shorter tile streams the layer's weight planes more often per position
  // (measured per unit and round, pair launches at 1024^2 x 2 and 1536^2: MT = 1 1.30, MT = 2 1.09 of MT = 3;
  // tests/checks/time_towers.py, time_dcn_pair.py with ORP_DCNS_MT)
  const int cus = a.nconv == 2 ? 128 : 256;
  int MT = 1;
  long best = -1;
  for (int mt = 1; mt <= 3; mt++) {
    const long t = (npos_all + 32 * mt - 1) / (32 * mt) + a.nlev * (a.per_image ? a.B : 1);
    const long cost = ((t + cus - 1) / cus) * mt * (mt == 1 ? 130 : mt == 2 ? 110 : 100);
    if (best < 0 || cost < best) { best = cost; MT = mt; }
  }
  static const int force_mt = getenv("ORP_DCNS_MT") ? atoi(getenv("ORP_DCNS_MT")) : 0;
  if (force_mt >= 1 && force_mt <= 3) MT = force_mt;
  int tiles = 0;
  for (int i = 0; i < kMaxLevels; i++) { pl.tile0[i] = 0x7fffffff; pl.tpi[i] = 0; }
  for (int i = 0; i < a.nlev; i++) {
    const long hw = (long)a.lv[i].Ho * a.lv[i].Wo;
    pl.tile0[i] = tiles;
    if (a.per_image) {
      pl.tpi[i] = (int)((hw + 32 * MT - 1) / (32 * MT));
      tiles += a.B * pl.tpi[i];
    } else {
      tiles += (int)((a.B * hw + 32 * MT - 1) / (32 * MT));
    }
  }
  pl.MT = MT; pl.tiles = tiles;
  return pl;
}

// dev aid: a log of what the fp16-pieces launches read as their range words (4 words per launch, in launch order; baked into a
// captured graph's kernel nodes like every other argument, so a replay writes the slots its capture was given)
static unsigned* g_amax_log = nullptr;
static int g_amax_log_cap = 0, g_amax_log_next = 0;
int set_amax_log(unsigned* log, int capacity_launches) {
  const int used = g_amax_log_next;
  g_amax_log = log; g_amax_log_cap = log ? capacity_launches : 0; g_amax_log_next = 0;
  return used;
}

hipError_t launch(const Args& a, hipStream_t st) {
  FwdS P;
  P.dbg = nullptr;
#if ORP_DCNS_TRACE
  P.dbg = g_amax_log;                                     // the trace buffer (>= 16 + 2^20 + 2^20 words), every launch from its start
#else
  if (a.nprod == 3 && g_amax_log && g_amax_log_next < g_amax_log_cap) P.dbg = g_amax_log + 4 * (g_amax_log_next++);
#endif
  P.nlev = a.nlev; P.B = a.B; P.Cin = a.Cin; P.Cout = a.Cout;
  P.kh = a.kh; P.kw = a.kw; P.sh = a.sh; P.sw = a.sw; P.ph = a.ph; P.pw = a.pw; P.dh = a.dh; P.dw = a.dw;
  P.planes[0] = a.planes[0]; P.planes[1] = a.planes[1]; P.bias[0] = a.bias[0]; P.bias[1] = a.bias[1];
  P.relu = a.relu; P.nconv = a.nconv;
  P.plane_stride = (size_t)a.Cout * a.Cin * a.kh * a.kw;
  P.coef_in = reinterpret_cast<const float2*>(a.coef_in); P.relu_in = a.relu_in;
  P.gn_part = reinterpret_cast<float4*>(a.gn_part); P.G = a.groups > 0 ? a.groups : 1;
  P.amax_count = a.amax_count > 1 ? a.amax_count : 1;
  if ((a.coef_in || a.gn_part) && (!a.per_image || a.lv[0].off != nullptr)) return hipErrorInvalidValue;
  if (a.coef_in && (a.Cin > kCoefCinMax || (a.nprod == 3 && !a.amax_in))) return hipErrorInvalidValue;   // (a pre-pass would see the raw inputs)
  if (a.gn_part && (a.out_nchw || a.groups <= 0 || a.Cout % a.groups != 0 || (32 % (a.Cout / a.groups)) != 0 || a.Cout / a.groups < 1 ||
                    a.relu || a.bias[0] || a.bias[1]))
    return hipErrorInvalidValue;
  const Plan pl = plan(a);
  const int MT = pl.MT, tiles = pl.tiles;
  for (int i = 0; i < a.nlev; i++) {
    LevelK& D = P.lv[i];
    D.x[0] = a.lv[i].x[0]; D.x[1] = a.lv[i].x[1]; D.off = a.lv[i].off; D.mask = a.lv[i].mask;
    D.out[0] = a.lv[i].out[0]; D.out[1] = a.lv[i].out[1];
    D.H = a.lv[i].H; D.W = a.lv[i].W; D.Ho = a.lv[i].Ho; D.Wo = a.lv[i].Wo;
    D.planes = a.nconv == 1 ? a.lv[i].planes : nullptr; D.bias = a.lv[i].bias; D.wscale = a.lv[i].wscale;
    D.tile0 = pl.tile0[i]; D.tpi = pl.tpi[i];
  }
  for (int i = a.nlev; i < kMaxLevels; i++) { P.lv[i] = P.lv[0]; P.lv[i].tile0 = 0x7fffffff; }
  const int nblk_n = (a.Cout + 255) / 256;
  bool plain = a.lv[0].off == nullptr;                   // no offsets anywhere: the ordinary convolution
  for (int i = 0; i < a.nlev; i++)
    if ((a.lv[i].off == nullptr) != plain || (plain && a.lv[i].mask)) return hipErrorInvalidValue;
  if (a.nprod == 3) {
    // fp16 pieces: max |x| of the launch's inputs first (one slot per layer; a pair launch reading the same tensors
    // twice gets the same value in both), into the caller's scratch
    if ((!a.scratch && !a.amax_in) || !a.wscale[0] || (a.nconv == 2 && !a.wscale[1])) return hipErrorInvalidValue;
    for (int i = 0; i < a.nlev; i++) if (a.lv[i].mask) return hipErrorInvalidValue;     // a modulation scalar has no known range
    if (a.amax_in) {                                       // the producer of the inputs left the bound: no pre-pass
      P.amax = a.amax_in; P.amax_stride = a.amax_stride;
    } else {
      P.amax_count = 1;
      AbsMaxArgs M;
      int bx = 0, cnt = 0;
      for (int cv = 0; cv < a.nconv; cv++)
        for (int i = 0; i < a.nlev; i++) {
          M.x[cnt] = a.lv[i].x[cv]; M.n[cnt] = (size_t)a.B * a.lv[i].H * a.lv[i].W * a.Cin; M.slot[cnt] = cv; M.bx0[cnt] = bx;
          long nb = (long)((M.n[cnt] / 4 + 256 * 8 - 1) / (256 * 8)); if (nb < 1) nb = 1; if (nb > 512) nb = 512;
          bx += (int)nb; cnt++;
        }
      for (int i = cnt; i <= kAbsMaxT; i++) M.bx0[i] = bx;
      for (int i = cnt; i < kAbsMaxT; i++) { M.x[i] = M.x[0]; M.n[i] = 0; M.slot[i] = 0; }
      M.count = cnt;
      hipError_t e = orp::fill_async(a.scratch, 0, sizeof(unsigned) * (size_t)(2), st);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(absmax_kernel, dim3(bx), dim3(256), 0, st, M, a.scratch);
      P.amax = a.scratch; P.amax_stride = 1;
    }
    P.wscale[0] = a.wscale[0]; P.wscale[1] = a.nconv == 2 ? a.wscale[1] : a.wscale[0];
    return plain ? launch_m<3, true>(MT, P, tiles, nblk_n, a.out_nchw != 0, st)
                 : launch_m<3, false>(MT, P, tiles, nblk_n, a.out_nchw != 0, st);
  }
  P.amax = nullptr; P.amax_stride = 0; P.wscale[0] = P.wscale[1] = nullptr;
  if (plain)
    return a.nprod == 9 ? launch_m<9, true>(MT, P, tiles, nblk_n, a.out_nchw != 0, st)
                        : launch_m<6, true>(MT, P, tiles, nblk_n, a.out_nchw != 0, st);
  return a.nprod == 9 ? launch_m<9, false>(MT, P, tiles, nblk_n, a.out_nchw != 0, st)
                      : launch_m<6, false>(MT, P, tiles, nblk_n, a.out_nchw != 0, st);
}

}  // namespace orp_split
