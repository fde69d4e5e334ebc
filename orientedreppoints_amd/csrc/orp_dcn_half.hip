// orp_dcn_half.hip -- deformable convolution forward (DCNv1 / DCNv2) in fp16 / bf16 for gfx950 (MI355X).
//
// The reference dispatches its DeformConv kernels over float AND half (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:259,353,451,781,813) and runs the im2col in half, the GEMM in the library's
// half path.  Here (BASELINE configs[4]: "DCNv2 MFMA path + fp16"):
//   * same implicit GEMM as orp_dcn.hip (A = bilinear samples, never written to HBM; all FPN levels in one launch;
//     NHWC inputs so that a wave fetches one neighbour of one position as a coalesced 512 B row of 256 channels),
//     on v_mfma_f32_32x32x16_{f16,bf16}: 16 k-values per instruction, 16x the fp32 matrix rate, fp32 accumulation;
//   * the bilinear combine is done in fp32 on the four gathered neighbours and rounded ONCE to the storage type (the
//     reference rounds every partial product in half) -- closer to the fp32 result than the reference's own half path;
//   * at this matrix rate one kernel tap of MFMA work (16 chunks x 3 instructions x 32 clk) is SHORTER than an L2 round
//     trip, so the pipeline is organised per TAP, not per chunk: at the top of a tap a wave issues ALL gathers of the
//     next tap's A rows (12 rows x 4 neighbours, 96 VGPRs in flight), the weight fragment of chunk j is reloaded for
//     the next tap right after chunk j has used it (in-place register ring, prefetch distance = one whole tap), and the
//     gathered rows are combined and written to the single LDS A tile between two barriers at the end of the tap.
// Round 3: a wave gathers with 16-byte loads (one neighbour of TWO rows per instruction) and combines with packed arithmetic
// (v_pk_fma_f16; bf16 widened to v_pk_fma_f32): 74 -> 55 us per layer at 1024^2 (21 824 positions).  Decomposition with the
// ORP_DCNH_DBG switches: without the gathers 42.7 us, without the per-tap weight loads 48.9 us, without both 34.1 us -- of
// which the matrix work is 11.5 us at the instruction rate; the rest of that floor is the A tile's LDS reads (every one
// of the 8 waves reads the whole tile: 128 B/clk/CU at full matrix rate, half the LDS's peak), so the next step is a wave
// tile of 64 output channels (half the LDS reads per MFMA), not more prefetch.  Measured without gain: a second A buffer
// with one barrier per tap (55.4 us), the NCHW output staged through LDS into 16-byte stores (56.3 us).
// Tolerance (tests): |out - fp32 oracle on the same rounded inputs| <= 2e-3 (fp16) / 1.6e-2 (bf16) of the output scale.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/orp_hip.h"
#include "orp_launch.hpp"
#include "orp_prof.hpp"

#ifndef ORP_DCNH_DBG
#define ORP_DCNH_DBG 0     // dev aid, compile-time (timing only, wrong results): 1 = no gathers, 2 = no per-tap weight loads, 4 = no MFMA, 8 = no combine
#endif

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

constexpr int MAX_TAPS = 9;
constexpr int MAX_LEVELS = 8;
constexpr int CBH = 256;          // input channels per tap phase
constexpr int ASTRH = CBH + 8;    // padded A row stride in ELEMENTS (132 dwords: conflict-free ds_read_b128 / ds_write_b64)
constexpr int KCH = 16;           // input channels per MFMA
constexpr int kThreadsH = 512;    // 8 waves: wave w owns output channels [32w, 32w+32)

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

// element type traits: storage <-> float, MFMA, and the bilinear combine of 8 channels (one 16-byte piece of a row per
// neighbour).  fp16 combines with packed half arithmetic (v_pk_mul_f16 / v_pk_fma_f16: two channels per instruction, 16
// instructions per 8 channels -- the reference's own half kernels also multiply and add in half); bf16 has no packed ALU
// form on gfx950: its pairs are widened with one shift / mask each and combined with v_pk_fma_f32, then rounded once.
template <typename T> struct Elem;
template <> struct Elem<_Float16> {
  typedef half8 v8;
  static __device__ __forceinline__ uint4 combine8(const uint4 (&g)[4], float4 w) {
    const half2v w0 = {(_Float16)w.x, (_Float16)w.x}, w1 = {(_Float16)w.y, (_Float16)w.y};
    const half2v w2 = {(_Float16)w.z, (_Float16)w.z}, w3 = {(_Float16)w.w, (_Float16)w.w};
    union U { uint4 u; half2v h[4]; } a, b, c, d, o;
    a.u = g[0]; b.u = g[1]; c.u = g[2]; d.u = g[3];
#pragma unroll
    for (int q = 0; q < 4; q++) o.h[q] = w3 * d.h[q] + (w2 * c.h[q] + (w1 * b.h[q] + w0 * a.h[q]));
    return o.u;
  }
  static __device__ __forceinline__ float to_f(_Float16 x) { return (float)x; }
  static __device__ __forceinline__ _Float16 from_f(float x) { return (_Float16)x; }
  static __device__ __forceinline__ floatx16 mfma(v8 a, v8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Elem<__bf16> {
  typedef bf8 v8;
  static __device__ __forceinline__ uint4 combine8(const uint4 (&g)[4], float4 w) {
    const float2v w0 = {w.x, w.x}, w1 = {w.y, w.y}, w2 = {w.z, w.z}, w3 = {w.w, w.w};
    const unsigned* a = reinterpret_cast<const unsigned*>(&g[0]);
    const unsigned* b = reinterpret_cast<const unsigned*>(&g[1]);
    const unsigned* c = reinterpret_cast<const unsigned*>(&g[2]);
    const unsigned* d = reinterpret_cast<const unsigned*>(&g[3]);
    auto widen = [](unsigned u) { float2v r = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; return r; };
    uint4 o;
    unsigned* op = reinterpret_cast<unsigned*>(&o);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float2v v = w3 * widen(d[q]) + (w2 * widen(c[q]) + (w1 * widen(b[q]) + w0 * widen(a[q])));
      union { __bf16 h[2]; unsigned u; } r;
      r.h[0] = (__bf16)v.x; r.h[1] = (__bf16)v.y;                        // one rounding to bf16
      op[q] = r.u;
    }
    return o;
  }
  static __device__ __forceinline__ float to_f(__bf16 x) { return (float)x; }
  static __device__ __forceinline__ __bf16 from_f(float x) { return (__bf16)x; }
  static __device__ __forceinline__ floatx16 mfma(v8 a, v8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

struct LevelH {
  const void* x;       // NHWC [B, H, W, Cin]
  const void* off;     // NCHW [B, 2*taps, Ho, Wo]
  const void* mask;    // DCNv2 modulation or nullptr
  void* out;           // NCHW or NHWC
  int H, W, Ho, Wo;
  int tile0;
};
struct FwdH {
  LevelH lv[MAX_LEVELS];
  int nlev, B, Cin, Cout;
  int kh, kw, sh, sw, ph, pw, dh, dw;
  const void* wp;      // packed [tap][Cin/16][2][Cout][8]
  const void* bias;    // [Cout] or nullptr
  int relu;
};

// w [o][c][tap] -> wp [tap][c/16][kg][o][8]   (kg = (c % 16) / 8, e = c % 8): lane (o, kg) of a wave reads its 8 k-values of a
// 16-channel chunk as ONE 16 B load, lanes 0-31 / 32-63 two contiguous 512 B segments
template <typename T>
__global__ void pack_weight_h_kernel(const T* __restrict__ w, int cout, int cin, int taps, T* __restrict__ wp) {
  const long total = (long)cout * cin * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7);
    long r = i >> 3;
    const int o = (int)(r % cout); r /= cout;
    const int kg = (int)(r & 1); r >>= 1;
    const int cblk = (int)(r % (cin / 16)), tap = (int)(r / (cin / 16));
    const int c = cblk * 16 + kg * 8 + e;
    wp[i] = w[((long)o * cin + c) * taps + tap];
  }
}

// [B][C][HW] -> [B][HW][C] for 2-byte elements, all levels in one launch
struct TransposeH {
  const void* in[MAX_LEVELS];
  void* out[MAX_LEVELS];
  int hw[MAX_LEVELS];
  int bx0[MAX_LEVELS + 1];
  int nlev;
};
__global__ void nchw_to_nhwc_h_kernel(const TransposeH T, int C) {
  __shared__ unsigned short tile[32][33];
  int l = 0;
#pragma unroll
  for (int i = 1; i < MAX_LEVELS; i++) l = (i < T.nlev && (int)blockIdx.x >= T.bx0[i]) ? i : l;
  const int HW = T.hw[l];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, p0 = ((int)blockIdx.x - T.bx0[l]) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const unsigned short* src = reinterpret_cast<const unsigned short*>(T.in[l]) + (size_t)b * C * HW;
  unsigned short* dst = reinterpret_cast<unsigned short*>(T.out[l]) + (size_t)b * C * HW;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    tile[r][tx] = (c < C && p < HW) ? src[(size_t)c * HW + p] : (unsigned short)0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    if (p < HW && c < C) dst[(size_t)p * C + c] = tile[tx][r];
  }
}

template <typename T, int MT, bool OUT_NCHW>
__global__ void __launch_bounds__(kThreadsH)
dcn_fwd_half_kernel(const FwdH P, int total_tiles) {
  typedef typename Elem<T>::v8 v8;
  constexpr int BMH = 32 * MT;
  constexpr int ROWS = BMH / 8;                                              // A rows produced per wave per tap
  constexpr int NCHUNK = CBH / KCH;                                          // 16 MFMA steps per tap
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sA = reinterpret_cast<T*>(smem);                                        // [BMH][ASTRH]
  float4* sCw = reinterpret_cast<float4*>(sA + BMH * ASTRH);                 // [BMH * taps] bilinear weights (fp32)
  int4* sCi = reinterpret_cast<int4*>(sCw + BMH * MAX_TAPS);                 // [BMH * taps] pixel indices

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int taps = P.kh * P.kw;
  int tile;
  {                                                                          // XCD-aware remap: XCD x takes a contiguous slab
    const int b = blockIdx.x, per = (total_tiles + 7) >> 3;
    tile = (b & 7) * per + (b >> 3);
    if (tile >= total_tiles) return;
  }
  int lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if (tile >= P.lv[i].tile0) lvl = i;
  const LevelH L = P.lv[lvl];
  const int HoWo = L.Ho * L.Wo;
  const long npos = (long)P.B * HoWo;
  const long p0 = (long)(tile - L.tile0) * BMH;
  const int nb = blockIdx.y;
  const T* xin = reinterpret_cast<const T*>(L.x);
  const T* offp = reinterpret_cast<const T*>(L.off);
  const T* maskp = reinterpret_cast<const T*>(L.mask);

  // ---- bilinear coefficient table (fp32), one entry per (position, tap) --------------------------------------------------
  for (int e = tid; e < BMH * taps; e += kThreadsH) {
    const int m = e / taps, tap = e - m * taps;
    const long p = p0 + m;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 ix = make_int4(0, 0, 0, 0);
    if (p < npos) {
      const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
      const int ho = hw / L.Wo, wo = hw - ho * L.Wo;
      const int ki = tap / P.kw, kj = tap - ki * P.kw;
      const T* ob = offp + ((size_t)b * 2 * taps + 2 * tap) * HoWo + hw;
      const float off_h = Elem<T>::to_f(ob[0]), off_w = Elem<T>::to_f(ob[HoWo]);
      const float h_im = (float)(ho * P.sh - P.ph + ki * P.dh) + off_h;
      const float w_im = (float)(wo * P.sw - P.pw + kj * P.dw) + off_w;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)L.H && w_im < (float)L.W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        const float hh = 1.f - lh, hw_ = 1.f - lw;
        const bool t_ok = h_low >= 0, b_ok = h_high <= L.H - 1, l_ok = w_low >= 0, r_ok = w_high <= L.W - 1;
        const int hl = t_ok ? h_low : 0, hhg = b_ok ? h_high : L.H - 1, wl = l_ok ? w_low : 0, whg = r_ok ? w_high : L.W - 1;
        w.x = (t_ok && l_ok) ? hh * hw_ : 0.f;
        w.y = (t_ok && r_ok) ? hh * lw : 0.f;
        w.z = (b_ok && l_ok) ? lh * hw_ : 0.f;
        w.w = (b_ok && r_ok) ? lh * lw : 0.f;
        const int base = b * L.H;
        ix.x = (base + hl) * L.W + wl;
        ix.y = (base + hl) * L.W + whg;
        ix.z = (base + hhg) * L.W + wl;
        ix.w = (base + hhg) * L.W + whg;
        if (maskp) {                                      // DCNv2: the sample is scaled by its modulation scalar
          const float mm = Elem<T>::to_f(maskp[((size_t)b * taps + tap) * HoWo + hw]);
          w.x *= mm; w.y *= mm; w.z *= mm; w.w *= mm;
        }
      }
    }
    sCw[e] = w; sCi[e] = ix;
  }
  __syncthreads();

  const int ncb = P.Cin / CBH;                       // 256-channel blocks per tap
  const int nphase = taps * ncb;
  // one A row = 256 channels = 32 lanes x 8 elements (16 B): a wave fetches the same neighbour of TWO rows per
  // instruction (lanes 0-31 row 2q, lanes 32-63 row 2q + 1), each a coalesced 512 B piece -- 16-byte accesses move twice
  // the bytes per L1 cycle of the 8-byte ones this kernel used before
  const int half_id = lane >> 5, l8 = (lane & 31) * 8;
  auto gather_issue = [&](int phase, int m2, uint4 (&g)[4]) {          // m2: even row of the pair
    const int tap = phase / ncb, cb = phase - tap * ncb;
    const int4 ix = sCi[(m2 + half_id) * taps + tap];
    const T* base = xin + cb * CBH + l8;
    if (ORP_DCNH_DBG & 1) { g[0] = g[1] = g[2] = g[3] = make_uint4(ix.x, ix.y, ix.z, ix.w); return; }
    g[0] = *reinterpret_cast<const uint4*>(base + (size_t)ix.x * P.Cin);
    g[1] = *reinterpret_cast<const uint4*>(base + (size_t)ix.y * P.Cin);
    g[2] = *reinterpret_cast<const uint4*>(base + (size_t)ix.z * P.Cin);
    g[3] = *reinterpret_cast<const uint4*>(base + (size_t)ix.w * P.Cin);
  };
  auto combine_store = [&](int phase, int m2, const uint4 (&g)[4]) {
    const int tap = phase / ncb;
    const int m = m2 + half_id;
    if (ORP_DCNH_DBG & 8) { *reinterpret_cast<uint4*>(sA + (size_t)m * ASTRH + l8) = g[0]; return; }
    *reinterpret_cast<uint4*>(sA + (size_t)m * ASTRH + l8) = Elem<T>::combine8(g, sCw[m * taps + tap]);
  };
  // weight fragment of (phase, chunk j): lane (n = lane & 31, kg = lane >> 5) -> 8 k-values, one 16 B load
  const int n_wave = nb * 256 + wave * 32;
  const int mrow = lane & 31, kg = lane >> 5;
  const bool live = n_wave < P.Cout;                      // c_out % 64 == 0: a wave is live or idle as a whole
  const T* wp = reinterpret_cast<const T*>(P.wp);
  auto load_b = [&](int phase, int j) -> v8 {
    const int tap = phase / ncb, cb = phase - tap * ncb;
    const size_t blk = (size_t)tap * (P.Cin / 16) + cb * (CBH / 16) + j;
    return *reinterpret_cast<const v8*>(wp + ((blk * 2 + kg) * P.Cout + (live ? n_wave : 0) + mrow) * 8);
  };

  // ---- prologue: A tile of phase 0, all weight fragments of phase 0 ----------------------------------------------------------
  v8 bq[NCHUNK];
#pragma unroll
  for (int j = 0; j < NCHUNK; j++) bq[j] = load_b(0, j);
  // wave w produces the row pairs (2 * (q * 8 + w), +1), q = 0 .. ROWS / 2 - 1
  {
    uint4 g[ROWS / 2][4];
#pragma unroll
    for (int r = 0; r < ROWS / 2; r++) gather_issue(0, 2 * (r * 8 + wave), g[r]);
#pragma unroll
    for (int r = 0; r < ROWS / 2; r++) combine_store(0, 2 * (r * 8 + wave), g[r]);
  }
  __syncthreads();

  floatx16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) acc[mt] = floatx16{0};

#pragma unroll 1
  for (int phase = 0; phase < nphase; phase++) {
    const bool next_phase = phase + 1 < nphase;
    // (1) every gather of the NEXT tap's A rows goes out now: a whole tap of MFMA work to land
    uint4 g[ROWS / 2][4];
    if (next_phase) {
#pragma unroll
      for (int r = 0; r < ROWS / 2; r++) gather_issue(phase + 1, 2 * (r * 8 + wave), g[r]);
    }
    // (2) the tap: one MFMA per (chunk, row block); the weight register of chunk j is refilled for the next tap at once
    const T* arow = sA + (size_t)mrow * ASTRH + 8 * kg;
    // the A fragments of chunk j + 1 are read from LDS before the MFMAs of chunk j are issued: with 32-cycle MFMAs a
    // chunk is only ~100 cycles of matrix work, less than the LDS round trip it would otherwise wait for
    v8 apre[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) apre[mt] = *reinterpret_cast<const v8*>(arow + (size_t)mt * 32 * ASTRH);
#pragma unroll
    for (int j = 0; j < NCHUNK; j++) {
      v8 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; mt++) a[mt] = apre[mt];
      if (j + 1 < NCHUNK) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) apre[mt] = *reinterpret_cast<const v8*>(arow + (size_t)mt * 32 * ASTRH + (j + 1) * KCH);
      }
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        if (ORP_DCNH_DBG & 4) { acc[mt][0] += (float)a[mt][0] * (float)bq[j][0]; continue; }
        if (OUT_NCHW) acc[mt] = Elem<T>::mfma(bq[j], a[mt], acc[mt]);     // D[channel][position]
        else          acc[mt] = Elem<T>::mfma(a[mt], bq[j], acc[mt]);     // D[position][channel]
      }
      if (next_phase && !(ORP_DCNH_DBG & 2)) bq[j] = load_b(phase + 1, j);
    }
    // (3) two barriers per tap: every wave is past its last read of the A tile -> overwrite it with the next tap's rows
    //     (a second A buffer with ONE barrier per tap was measured in round 3: 55.4 vs 54.9 us -- the barriers are not what
    //     the tap waits for -- and is not kept: it doubles the tile's LDS)
    if (next_phase) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < ROWS / 2; r++) combine_store(phase + 1, 2 * (r * 8 + wave), g[r]);
      __syncthreads();
    }
  }

  // ---- epilogue: fp32 accumulators (+ bias, ReLU) rounded once to the storage type -------------------------------------------
  const T* biasp = reinterpret_cast<const T*>(P.bias);
  T* outp = reinterpret_cast<T*>(L.out);
  auto finish = [&](float v, int ch) { if (biasp) v += Elem<T>::to_f(biasp[ch]); return Elem<T>::from_f(P.relu ? fmaxf(v, 0.f) : v); };
  if (!live) return;
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    if (OUT_NCHW) {
      const long p = p0 + mt * 32 + (lane & 31);
      if (p < npos) {
        const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
        T* ob = outp + (size_t)b * P.Cout * HoWo + hw;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int ch = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          ob[(size_t)(n_wave + ch) * HoWo] = finish(acc[mt][r], n_wave + ch);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const long p = p0 + mt * 32 + m;
        if (p < npos) outp[(size_t)p * P.Cout + n_wave + (lane & 31)] = finish(acc[mt][r], n_wave + (lane & 31));
      }
    }
  }
}

// ---- round 6: the same operator, wave-specialised ---------------------------------------------------------------------------------
// The kernel above runs every tap as  gathers out -> 16 chunks of MFMAs -> barrier -> combine + LDS writes -> barrier: the matrix pipe
// idles through the combine, and every one of the 8 waves reads the whole A tile (its own analysis, top of this file: "the next step is
// a wave tile of 64 output channels").  Here, as in csrc/orp_dcn_split.hip's wave-specialised kernel: 4 CONSUMER waves (one per SIMD;
// the whole tile height x 64 output channels each = MT x 2 accumulator blocks; A fragments from a DOUBLE-buffered LDS tile, weights L2
// -> an in-place-refilled register ring of one phase) and 4 PRODUCER waves at s_setprio 3 (the rows of phase + 2 gathered during phase
// p, packed bilinear combine, one 16-byte LDS write per row piece); phases of 128 input channels of one tap, ONE barrier per phase.
// Same products in the same order per accumulator as the kernel above (one MFMA per 16 channels, channels ascending, taps outer):
// bit-identical results.
constexpr int CBW = 128;            // input channels per phase
constexpr int ASTRW = CBW + 8;      // A row stride in elements (68 dwords: 16 rows x 4 dwords = all 64 banks per ds_read_b128 group)
constexpr int NCHW8 = CBW / KCH;    // 8 MFMA steps per phase

template <typename T, int MT, bool OUT_NCHW>
__global__ void __launch_bounds__(kThreadsH)
dcn_fwd_half_ws_kernel(const FwdH P, int total_tiles) {
  typedef typename Elem<T>::v8 v8;
  constexpr int BMH = 32 * MT;
  constexpr int RG = 2 * MT;                                                 // row groups (4 rows each) of one producer wave per phase
  constexpr int NT = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sA = reinterpret_cast<T*>(smem);                                        // [2][BMH][ASTRW]
  float4* sCw = reinterpret_cast<float4*>(sA + 2 * BMH * ASTRW);
  int4* sCi = reinterpret_cast<int4*>(sCw + BMH * MAX_TAPS);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int taps = P.kh * P.kw;
  int tile;
  {
    const int b = blockIdx.x, per = (total_tiles + 7) >> 3;
    tile = (b & 7) * per + (b >> 3);
    if (tile >= total_tiles) return;
  }
  int lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if (tile >= P.lv[i].tile0) lvl = i;
  const LevelH L = P.lv[lvl];
  const int HoWo = L.Ho * L.Wo;
  const long npos = (long)P.B * HoWo;
  const long p0 = (long)(tile - L.tile0) * BMH;
  const int nb = blockIdx.y;
  const T* xin = reinterpret_cast<const T*>(L.x);
  const T* offp = reinterpret_cast<const T*>(L.off);
  const T* maskp = reinterpret_cast<const T*>(L.mask);

  // ---- bilinear coefficient table (as above) ----
  for (int e = tid; e < BMH * taps; e += kThreadsH) {
    const int m = e / taps, tap = e - m * taps;
    const long p = p0 + m;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 ix = make_int4(0, 0, 0, 0);
    if (p < npos) {
      const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
      const int ho = hw / L.Wo, wo = hw - ho * L.Wo;
      const int ki = tap / P.kw, kj = tap - ki * P.kw;
      const T* ob = offp + ((size_t)b * 2 * taps + 2 * tap) * HoWo + hw;
      const float off_h = Elem<T>::to_f(ob[0]), off_w = Elem<T>::to_f(ob[HoWo]);
      const float h_im = (float)(ho * P.sh - P.ph + ki * P.dh) + off_h;
      const float w_im = (float)(wo * P.sw - P.pw + kj * P.dw) + off_w;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)L.H && w_im < (float)L.W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        const float hh = 1.f - lh, hw_ = 1.f - lw;
        const bool t_ok = h_low >= 0, b_ok = h_high <= L.H - 1, l_ok = w_low >= 0, r_ok = w_high <= L.W - 1;
        const int hl = t_ok ? h_low : 0, hhg = b_ok ? h_high : L.H - 1, wl = l_ok ? w_low : 0, whg = r_ok ? w_high : L.W - 1;
        w.x = (t_ok && l_ok) ? hh * hw_ : 0.f;
        w.y = (t_ok && r_ok) ? hh * lw : 0.f;
        w.z = (b_ok && l_ok) ? lh * hw_ : 0.f;
        w.w = (b_ok && r_ok) ? lh * lw : 0.f;
        const int base = b * L.H;
        ix.x = (base + hl) * L.W + wl;
        ix.y = (base + hl) * L.W + whg;
        ix.z = (base + hhg) * L.W + wl;
        ix.w = (base + hhg) * L.W + whg;
        if (maskp) {
          const float mm = Elem<T>::to_f(maskp[((size_t)b * taps + tap) * HoWo + hw]);
          w.x *= mm; w.y *= mm; w.z *= mm; w.w *= mm;
        }
      }
    }
    sCw[e] = w; sCi[e] = ix;
  }
  __syncthreads();

  const int ncb = P.Cin / CBW;
  const int nphase = taps * ncb;
  const bool consumer = wave < 4;
  const int wq = wave & 3;
  int tap_n = 0, cb_n = 0, tap_n2 = 0, cb_n2 = 0;                             // (tap, channel block) of phase + 1 / + 2, clamped to the last
  auto step = [&](int& t, int& c, int ph) {
    if (ph + 1 < nphase) { if (++c == ncb) { c = 0; t++; } }
  };
  step(tap_n, cb_n, 0);
  tap_n2 = tap_n; cb_n2 = cb_n;
  step(tap_n2, cb_n2, 1);

  if (!consumer) {
    // one A row piece = 128 channels = 16 lanes x 8 elements (16 B): a wave fetches the same neighbour of FOUR rows per instruction
    const int q4 = lane >> 4, l8 = (lane & 15) * 8;
    auto row_of = [&](int g) { return g * 16 + wq * 4 + q4; };
    auto gather_issue = [&](int tap, int cb, int g, uint4 (&v)[4]) {
      const int4 ix = sCi[row_of(g) * taps + tap];
      const T* base = xin + cb * CBW + l8;
      v[0] = *reinterpret_cast<const uint4*>(base + (size_t)ix.x * P.Cin);
      v[1] = *reinterpret_cast<const uint4*>(base + (size_t)ix.y * P.Cin);
      v[2] = *reinterpret_cast<const uint4*>(base + (size_t)ix.z * P.Cin);
      v[3] = *reinterpret_cast<const uint4*>(base + (size_t)ix.w * P.Cin);
    };
    auto combine_store = [&](int tap, int g, const uint4 (&v)[4], int buf) {
      const int m = row_of(g);
      *reinterpret_cast<uint4*>(sA + (size_t)buf * BMH * ASTRW + (size_t)m * ASTRW + l8) = Elem<T>::combine8(v, sCw[m * taps + tap]);
    };
    uint4 gA[RG][4], gB[RG][4];
#pragma unroll
    for (int r = 0; r < RG; r++) gather_issue(0, 0, r, gB[r]);
#pragma unroll
    for (int r = 0; r < RG; r++) gather_issue(tap_n, cb_n, r, gA[r]);         // the rows of phase 1
#pragma unroll
    for (int r = 0; r < RG; r++) combine_store(0, r, gB[r], 0);
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);                                           // (a VALU / VMEM wave starves beside an MFMA wave otherwise: docs/notebook/round6.md 2)
    auto produce = [&](int phase, uint4 (&g)[RG][4], uint4 (&gf)[RG][4]) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < RG; r++) gather_issue(tap_n2, cb_n2, r, gf[r]);    // the rows of phase + 2: a whole phase to land
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < RG; r++) combine_store(tap_n, r, g[r], (phase & 1) ^ 1);
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

  // ---- consumers ----
  const int n_wave = nb * 256 + wq * 64;
  const int mrow = lane & 31, kg = lane >> 5;
  const bool live = n_wave < P.Cout;                                          // Cout % 64 == 0
  const T* wp = reinterpret_cast<const T*>(P.wp);
  auto load_b = [&](int tap, int cb, int j, v8 (&b)[NT]) {
    const size_t blk = (size_t)tap * (P.Cin / 16) + cb * (CBW / 16) + j;
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
      b[nt] = *reinterpret_cast<const v8*>(wp + ((blk * 2 + kg) * P.Cout + (live ? n_wave : 0) + nt * 32 + mrow) * 8);
  };
  v8 bq[NCHW8][NT];
  v8 af[2][MT];
  floatx16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) acc[mt][nt] = floatx16{0};
#pragma unroll
  for (int j = 0; j < NCHW8; j++) load_b(0, 0, j, bq[j]);
  __syncthreads();
#pragma unroll 1
  for (int phase = 0; phase < nphase; phase++) {
    const T* arow = sA + (size_t)(phase & 1) * BMH * ASTRW + (size_t)mrow * ASTRW + 8 * kg;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) af[0][mt] = *reinterpret_cast<const v8*>(arow + (size_t)mt * 32 * ASTRW);
#pragma unroll
    for (int j = 0; j < NCHW8; j++) {
      if (j + 1 < NCHW8) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) af[(j + 1) & 1][mt] = *reinterpret_cast<const v8*>(arow + (size_t)mt * 32 * ASTRW + (j + 1) * KCH);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          if (OUT_NCHW) acc[mt][nt] = Elem<T>::mfma(bq[j][nt], af[j & 1][mt], acc[mt][nt]);     // D[channel][position]
          else          acc[mt][nt] = Elem<T>::mfma(af[j & 1][mt], bq[j][nt], acc[mt][nt]);     // D[position][channel]
        }
      __builtin_amdgcn_sched_barrier(0);
      load_b(tap_n, cb_n, j, bq[j]);                                          // refilled in place for the next phase
      __builtin_amdgcn_sched_barrier(0);
    }
    step(tap_n, cb_n, phase + 1);
    __syncthreads();
  }

  // ---- epilogue ----
  if (!live) return;
  const T* biasp = reinterpret_cast<const T*>(P.bias);
  T* outp = reinterpret_cast<T*>(L.out);
  auto finish = [&](float v, int ch) { if (biasp) v += Elem<T>::to_f(biasp[ch]); return Elem<T>::from_f(P.relu ? fmaxf(v, 0.f) : v); };
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int nbase = n_wave + nt * 32;
      if (OUT_NCHW) {
        const long p = p0 + mt * 32 + (lane & 31);
        if (p < npos) {
          const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
          T* ob = outp + (size_t)b * P.Cout * HoWo + hw;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int ch = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            ob[(size_t)(nbase + ch) * HoWo] = finish(acc[mt][nt][r], nbase + ch);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const long p = p0 + mt * 32 + m;
          if (p < npos) outp[(size_t)p * P.Cout + nbase + (lane & 31)] = finish(acc[mt][nt][r], nbase + (lane & 31));
        }
      }
    }
}

template <int MT>
size_t half_ws_smem() { return 2 * ((size_t)2 * 32 * MT * ASTRW) + (sizeof(float4) + sizeof(int4)) * 32 * MT * MAX_TAPS; }

static const int g_half_ws = getenv("ORP_DCNH_WS") ? atoi(getenv("ORP_DCNH_WS")) : 1;   // 0: the symmetric kernel above (A/B timing)

template <int MT>
size_t half_smem() { return 2 * ((size_t)32 * MT * ASTRH) + (sizeof(float4) + sizeof(int4)) * 32 * MT * MAX_TAPS; }

template <typename T, int MT, bool OUT_NCHW>
hipError_t launch_half(const FwdH& P, int tiles, int nblk_n, hipStream_t st) {
  if (g_half_ws && P.Cin % CBW == 0) {
    const size_t smem_w = half_ws_smem<MT>();
    struct TagW {};
    hipError_t ew = orp::set_max_dynamic_lds_once<TagW>(reinterpret_cast<const void*>(&dcn_fwd_half_ws_kernel<T, MT, OUT_NCHW>), smem_w);
    if (ew != hipSuccess) return ew;
    const int per_w = (tiles + 7) >> 3;
    hipLaunchKernelGGL((dcn_fwd_half_ws_kernel<T, MT, OUT_NCHW>), dim3(per_w * 8, nblk_n), dim3(kThreadsH), smem_w, st, P, tiles);
    return hipGetLastError();
  }
  const size_t smem = half_smem<MT>();
  struct Tag {};
  hipError_t e = orp::set_max_dynamic_lds_once<Tag>(reinterpret_cast<const void*>(&dcn_fwd_half_kernel<T, MT, OUT_NCHW>), smem);
  if (e != hipSuccess) return e;
  const int per = (tiles + 7) >> 3;
  hipLaunchKernelGGL((dcn_fwd_half_kernel<T, MT, OUT_NCHW>), dim3(per * 8, nblk_n), dim3(kThreadsH), smem, st, P, tiles);
  return hipGetLastError();
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int out_dim(int in, int pad, int dil, int k, int stride) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }

}  // namespace

extern "C" {

int orp_dcn_half_path_ok(int c_in, int c_out, int kh, int kw, int groups, int deformable_groups) {
  return (groups == 1 && deformable_groups == 1 && kh * kw <= MAX_TAPS && c_in % CBH == 0 && c_out % 64 == 0 && c_out >= 64) ? 1 : 0;
}

int orp_dcn_pack_weight_h(const void* weight, int c_out, int c_in, int kh, int kw, void* packed, int dtype, void* stream) {
  if (!weight || !packed || c_out <= 0 || c_in <= 0 || c_in % 16 || kh <= 0 || kw <= 0 || (dtype != 1 && dtype != 2)) return ORP_EINVAL;
  const long total = (long)c_out * c_in * kh * kw;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  if (dtype == 1)
    hipLaunchKernelGGL(pack_weight_h_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)weight, c_out, c_in, kh * kw, (_Float16*)packed);
  else
    hipLaunchKernelGGL(pack_weight_h_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)weight,
                       c_out, c_in, kh * kw, (__bf16*)packed);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

size_t orp_dcn_forward_h_workspace_bytes(const orp_dcn_level_h* levels_host, int nlevels, int batch, int c_in, int in_layout) {
  if (in_layout == 1 || !levels_host) return 256;
  size_t tot = 0;
  for (int i = 0; i < nlevels; i++) tot += align256((size_t)2 * batch * c_in * levels_host[i].height * levels_host[i].width);
  return tot + 256;
}

int orp_dcn_forward_multi_h(const orp_dcn_level_h* levels_host, const void* const* masks_host, int nlevels, int batch, int c_in,
                            int c_out, const void* weight_packed, const void* bias, int relu, int kh, int kw, int stride_h,
                            int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int in_layout, int out_layout, int dtype,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!levels_host || nlevels <= 0 || nlevels > MAX_LEVELS || batch <= 0 || !weight_packed) return ORP_EINVAL;
  if (!orp_dcn_half_path_ok(c_in, c_out, kh, kw, 1, 1) || (dtype != 1 && dtype != 2)) return ORP_EINVAL;
  if ((in_layout != 0 && in_layout != 1) || (out_layout != 0 && out_layout != 1)) return ORP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (in_layout == 0 && workspace_bytes < orp_dcn_forward_h_workspace_bytes(levels_host, nlevels, batch, c_in, 0)) return ORP_EWORKSPACE;
  char* wsp = reinterpret_cast<char*>(workspace);
  long npos_all = 0;
  for (int i = 0; i < nlevels; i++)
    npos_all += (long)batch * out_dim(levels_host[i].height, pad_h, dil_h, kh, stride_h) *
                out_dim(levels_host[i].width, pad_w, dil_w, kw, stride_w);
  int MT = 1;
  long best = -1;
  for (int mt = 1; mt <= 3; mt++) {
    const long t = (npos_all + 32 * mt - 1) / (32 * mt) + nlevels;
    const long cost = ((t + 255) / 256) * mt * 100 + (mt == 1 ? 40 : mt == 2 ? 10 : 0);
    if (best < 0 || cost < best) { best = cost; MT = mt; }
  }
  FwdH P;
  P.nlev = nlevels; P.B = batch; P.Cin = c_in; P.Cout = c_out;
  P.kh = kh; P.kw = kw; P.sh = stride_h; P.sw = stride_w; P.ph = pad_h; P.pw = pad_w; P.dh = dil_h; P.dw = dil_w;
  P.wp = weight_packed; P.bias = bias; P.relu = relu ? 1 : 0;
  TransposeH TL;
  int tbx = 0, tiles = 0;
  const int bm = 32 * MT;
  for (int i = 0; i < nlevels; i++) {
    const orp_dcn_level_h& lv = levels_host[i];
    if (!lv.input || !lv.offset || !lv.output || lv.height <= 0 || lv.width <= 0) return ORP_EINVAL;
    LevelH& D = P.lv[i];
    D.H = lv.height; D.W = lv.width;
    D.Ho = out_dim(lv.height, pad_h, dil_h, kh, stride_h);
    D.Wo = out_dim(lv.width, pad_w, dil_w, kw, stride_w);
    if (D.Ho <= 0 || D.Wo <= 0) return ORP_EINVAL;
    if ((long)batch * lv.height * lv.width >= (1L << 31)) return ORP_ETOOBIG;
    D.off = lv.offset; D.out = lv.output;
    D.mask = masks_host ? masks_host[i] : nullptr;
    if (in_layout == 0) {
      const int HW = lv.height * lv.width;
      TL.in[i] = lv.input; TL.out[i] = wsp; TL.hw[i] = HW; TL.bx0[i] = tbx;
      tbx += (HW + 31) / 32;
      D.x = wsp;
      wsp += align256((size_t)2 * batch * c_in * HW);
    } else {
      D.x = lv.input;
    }
    D.tile0 = tiles;
    tiles += (int)(((long)batch * D.Ho * D.Wo + bm - 1) / bm);
  }
  for (int i = nlevels; i < MAX_LEVELS; i++) { P.lv[i] = P.lv[0]; P.lv[i].tile0 = 0x7fffffff; }
  if (in_layout == 0) {
    TL.nlev = nlevels;
    for (int i = nlevels; i <= MAX_LEVELS; i++) TL.bx0[i] = tbx;
    for (int i = nlevels; i < MAX_LEVELS; i++) { TL.in[i] = TL.in[0]; TL.out[i] = TL.out[0]; TL.hw[i] = 0; }
    hipLaunchKernelGGL(nchw_to_nhwc_h_kernel, dim3(tbx, (c_in + 31) / 32, batch), dim3(256), 0, st, TL, c_in);
  }
  OrpProfScope prof(ORP_PROF_DCN_FWD, st);
  const int nblk_n = (c_out + 255) / 256;
  const bool nchw = out_layout == 0;
  hipError_t e;
#define ORP_LAUNCH_H(TT) \
  (MT == 1 ? (nchw ? launch_half<TT, 1, true>(P, tiles, nblk_n, st) : launch_half<TT, 1, false>(P, tiles, nblk_n, st)) \
   : MT == 2 ? (nchw ? launch_half<TT, 2, true>(P, tiles, nblk_n, st) : launch_half<TT, 2, false>(P, tiles, nblk_n, st)) \
             : (nchw ? launch_half<TT, 3, true>(P, tiles, nblk_n, st) : launch_half<TT, 3, false>(P, tiles, nblk_n, st)))
  e = dtype == 1 ? ORP_LAUNCH_H(_Float16) : ORP_LAUNCH_H(__bf16);
#undef ORP_LAUNCH_H
  return e == hipSuccess ? ORP_OK : (int)e;
}

}  // extern "C"
