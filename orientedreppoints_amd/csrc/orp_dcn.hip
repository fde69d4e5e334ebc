// orp_dcn.hip -- deformable convolution (DCNv1 / DCNv2) forward for gfx950 (MI355X).
//
// Replaces deform_conv_forward_cuda / modulated_deform_conv_cuda_forward
//   (mmdet/ops/dcn/src/deform_conv_cuda.cpp:152-260, 490-590; deform_conv_cuda_kernel.cu:84-277, 570-700).
// The reference materialises the im2col "columns" buffer in HBM ([C*9, B*H*W] fp32 = 151 MB at the 128x128 level)
// and then calls a library GEMM per im2col_step.  Here the contraction is an IMPLICIT GEMM on the matrix cores:
//
//   out[p, o] = sum_{tap, c}  A[p, (tap, c)] * W2[(tap, c), o],      A[p, (tap, c)] = bilinear(x[b, :, :, c], p + tap + d)
//
//   * A is never written to HBM: each workgroup (4 waves) owns BM = 32 output positions x up to 256 output
//     channels and produces the A rows of one kernel tap at a time straight into LDS.  The input is read in NHWC
//     so that one wave fetches ONE bilinear neighbour of ONE position as a fully coalesced 1 KB row (64 lanes x
//     float4 = 256 channels); the four bilinear weights of a (position, tap) are wave-uniform and are computed
//     once per tile, not once per channel as in the reference kernel.
//   * weights are pre-packed to W2[tap][c][o] (o contiguous), streamed L2 -> LDS in 32-row chunks, double buffered.
//   * MFMA: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain = the reference's fp32 precision class), wave tile
//     32 positions x 64 channels.  K order inside a 32-channel chunk is permuted (step (t,i), half kh -> channel
//     8t + 4kh + i) so that the A fragment is one conflict-free ds_read_b128 per four MFMA steps (row stride padded
//     to 260 floats).
//   * all FPN levels go in ONE launch (tile table in the kernel arguments): 21 824 positions -> 682 tiles, instead
//     of five launches whose small levels cannot fill 256 CUs.
//   * output is written NCHW (operands swapped so lanes run along positions) or NHWC.
// A straightforward direct kernel covers every configuration the fast path does not (groups > 1,
// deformable_groups > 1, odd channel counts, DCNv2 modulation + bias).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/orp_hip.h"
#include "orp_launch.hpp"
#include "orp_prof.hpp"
#include "orp_dcn_split.hpp"

#ifndef ORP_DCN_APF_PIN
#define ORP_DCN_APF_PIN 0    // 1: sched_barrier behind the prefetch reads (measured: 6 spills, 487 vs 483 us)
#endif
#ifndef ORP_DCN_IPF
#define ORP_DCN_IPF 0      // gather pixel indices read from LDS one chunk ahead (measured: no gain on top of APF)
#endif
#ifndef ORP_DCN_APF_ALL
#define ORP_DCN_APF_ALL 1  // ... also in the two-layer instantiation (254 VGPRs, no spill; 494 -> 482 us per pair launch)
#endif
#ifndef ORP_DCN_APF
#define ORP_DCN_APF 1      // A-fragment LDS prefetch one k-step ahead in the single-layer second-generation kernel
#endif
#ifndef ORP_DCN_GDIST
#define ORP_DCN_GDIST 1    // gather-to-combine distance (chunks) of the single-layer second-generation kernel: 1 or 2
#endif
#ifndef ORP_DCN_WDIST
#define ORP_DCN_WDIST 1    // weight prefetch distance (chunks) of the single-layer second-generation kernel: 1 or 2 (2: measured, no gain)
#endif
#ifndef ORP_DCN_KS_DBG
#define ORP_DCN_KS_DBG 0   // dev aid for the tap-granular split (timing only): 1 = no hand-over of the cut tiles at all, 4 = coefficient table built once only
#endif
#ifndef ORP_DCN_DBG
#define ORP_DCN_DBG 0      // dev aid, compile-time (timing only, wrong results): 1 = no A gather, 2 = no weight loads, 4 = no per-tap barriers / LDS refill, 8 = no MFMA
#endif

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 32;          // output positions per workgroup
constexpr int BN = 256;         // output channels per workgroup (4 waves x 64)
constexpr int CB = 256;         // input channels per K phase (one tap)
constexpr int KC = 32;          // input channels per weight chunk
constexpr int ASTR = CB + 4;    // padded A row stride (floats): conflict-free ds_read_b128 / ds_write_b128
constexpr int MAX_TAPS = 9;
constexpr int MAX_LEVELS = 8;
constexpr int kThreads = 256;

struct LevelDesc {
  const float* x;      // NHWC [B, H, W, Cin]
  const float* off;    // NCHW [B, 2*taps, Ho, Wo]
  const float* mask;   // DCNv2 modulation, NCHW [B, taps, Ho, Wo], or nullptr (DCNv1)
  float* out;          // NCHW [B, Cout, Ho, Wo] or NHWC [B, Ho, Wo, Cout]
  const float* x2;     // second layer of a pair launch (same offsets / mask): its input ...
  float* out2;         // ... and output
  float* hout;         // fused 1x1 head of the first layer: [B, head_k[0], Ho, Wo] (HEADS instantiation)
  float* hout2;        // ... of the second layer
  const float* hres2;  // residual added to the second head's output ([B, head_k[1], Ho, Wo]) or nullptr
  int H, W, Ho, Wo;
  int tile0;           // first tile of this level
};
struct FwdParams {
  LevelDesc lv[MAX_LEVELS];
  int nlev, B, Cin, Cout;
  int kh, kw, sh, sw, ph, pw, dh, dw;
  const float* w2;     // packed [taps][Cin][Cout]
  const float* w3;     // packed [taps][Cin/4][Cout][4] (second-generation kernel: B fragments straight from L2)
  const float* bias;   // [Cout] or nullptr (DCNv2 bias)
  int relu;            // fuse max(., 0) into the epilogue (the head applies ReLU right after both DeformConvs)
  int nconv;           // 1, or 2: a second layer (w3b, bias2, LevelDesc::x2 / out2) over the same offsets in the same launch
  const float* w3b;
  const float* bias2;
  const float* head_w[2];   // fused 1x1 heads: packed [256][KH] (zero padded), bias [k], output channels k <= KH
  const float* head_b[2];
  int head_k[2];
  // tap-granular split of the launch's (tile, layer, tap) sequence over ks_nwg workgroups (KSPLIT instantiation)
  float* ks_scratch;        // [ks_nwg] accumulator images of the cut (tile, layer) pairs, register layout [MT][8][16][64]
  int* ks_flags;            // [ks_nwg] 0 -> 1 when the image of slot i is complete (zeroed before every launch)
  int ks_nwg, ks_total;     // workgroups (all layers), tiles * taps
};
constexpr int KH = 20;           // packed output-channel count of a fused 1x1 head
inline size_t align256_(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- helpers -------------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ w, int cout, int cin, int taps, float* __restrict__ w2) {
  // w [o][c][tap] -> w2 [tap][c][o]
  const long total = (long)cout * cin * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int o = (int)(i % cout);
    const long r = i / cout;
    const int c = (int)(r % cin), tap = (int)(r / cin);
    const float v = w[((long)o * cin + c) * taps + tap];
    w2[i] = v;
    if ((cin & 3) == 0) w2[total + (((long)tap * (cin >> 2) + (c >> 2)) * cout + o) * 4 + (c & 3)] = v;
  }
}

// head weight [k][256] -> [256][KH], zero padded
__global__ void pack_head_kernel(const float* __restrict__ w, int k, float* __restrict__ packed) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 256 * KH; i += gridDim.x * blockDim.x) {
    const int kk = i % KH, c = i / KH;
    packed[i] = kk < k ? w[(size_t)kk * 256 + c] : 0.f;
  }
}

// [B][C][HW] -> [B][HW][C] through a 32x33 LDS tile, for every level of one launch: blockIdx.x walks the levels'
// position tiles back to back
struct TransposeLevels {               // up to MAX_LEVELS levels of up to two layers
  const float* in[2 * MAX_LEVELS];
  float* out[2 * MAX_LEVELS];
  int hw[2 * MAX_LEVELS];
  int bx0[2 * MAX_LEVELS + 1];        // first blockIdx.x of each tensor; bx0[nlev] = gridDim.x
  int nlev;
};
__global__ void nchw_to_nhwc_multi_kernel(const TransposeLevels T, int C) {
  __shared__ float tile[32][33];
  int l = 0;
#pragma unroll
  for (int i = 1; i < 2 * MAX_LEVELS; i++) l = (i < T.nlev && (int)blockIdx.x >= T.bx0[i]) ? i : l;
  const int HW = T.hw[l];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, p0 = ((int)blockIdx.x - T.bx0[l]) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
  const float* src = T.in[l] + (size_t)b * C * HW;
  float* dst = T.out[l] + (size_t)b * C * HW;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    tile[r][tx] = (c < C && p < HW) ? src[(size_t)c * HW + p] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    if (p < HW && c < C) dst[(size_t)p * C + c] = tile[tx][r];
  }
}

// ---- the MFMA implicit-GEMM kernel -------------------------------------------------------------------------------
template <bool OUT_NCHW>
__global__ void __launch_bounds__(kThreads)
dcn_fwd_mfma_kernel(const FwdParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sA = reinterpret_cast<float*>(smem);                                // [2][BM][ASTR]
  float* sB = sA + 2 * BM * ASTR;                                            // [2][KC][BN]
  float4* sCw = reinterpret_cast<float4*>(sB + 2 * KC * BN);                 // [BM * taps] bilinear weights
  int4* sCi = reinterpret_cast<int4*>(sCw + BM * MAX_TAPS);                  // [BM * taps] pixel indices

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int taps = P.kh * P.kw;
  // which level does this tile belong to?
  int lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if ((int)blockIdx.x >= P.lv[i].tile0) lvl = i;
  const LevelDesc L = P.lv[lvl];
  const int HoWo = L.Ho * L.Wo;
  const long npos = (long)P.B * HoWo;
  const long p0 = (long)(blockIdx.x - L.tile0) * BM;
  const int nb = blockIdx.y;                                                  // 256-channel output block

  // ---- bilinear coefficient table: one entry per (position, tap), shared by all channels ---------------------
  for (int e = tid; e < BM * taps; e += kThreads) {
    const int m = e / taps, tap = e - m * taps;
    const long p = p0 + m;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 ix = make_int4(0, 0, 0, 0);
    if (p < npos) {
      const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
      const int ho = hw / L.Wo, wo = hw - ho * L.Wo;
      const int ki = tap / P.kw, kj = tap - ki * P.kw;
      const float* ob = L.off + ((size_t)b * 2 * taps + 2 * tap) * HoWo + hw;
      const float off_h = ob[0], off_w = ob[HoWo];
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
        if (L.mask) {                                     // DCNv2: the sample is scaled by its modulation scalar
          const float mm = L.mask[((size_t)b * taps + tap) * HoWo + hw];
          w.x *= mm; w.y *= mm; w.z *= mm; w.w *= mm;
        }
      }
    }
    sCw[e] = w; sCi[e] = ix;
  }
  __syncthreads();

  const int ncb = (P.Cin + CB - 1) / CB;            // channel blocks per tap
  const int nphase = taps * ncb;

  // A-row producer: this wave builds rows m = wave, wave+4, ... (8 rows per phase) of the phase's A tile
  auto gather_row = [&](int phase, int m, float4 (&g)[4], float4& wgt, bool& live) {
    const int tap = phase / ncb, cb = phase - tap * ncb;
    const int c = cb * CB + lane * 4;
    live = (c < P.Cin);
    wgt = sCw[m * taps + tap];
    const int4 ix = sCi[m * taps + tap];
    if (live) {
      g[0] = *reinterpret_cast<const float4*>(L.x + (size_t)ix.x * P.Cin + c);
      g[1] = *reinterpret_cast<const float4*>(L.x + (size_t)ix.y * P.Cin + c);
      g[2] = *reinterpret_cast<const float4*>(L.x + (size_t)ix.z * P.Cin + c);
      g[3] = *reinterpret_cast<const float4*>(L.x + (size_t)ix.w * P.Cin + c);
    }
  };
  auto store_row = [&](int buf, int m, const float4 (&g)[4], const float4 wgt, bool live) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
      v.x = wgt.x * g[0].x + wgt.y * g[1].x + wgt.z * g[2].x + wgt.w * g[3].x;
      v.y = wgt.x * g[0].y + wgt.y * g[1].y + wgt.z * g[2].y + wgt.w * g[3].y;
      v.z = wgt.x * g[0].z + wgt.y * g[1].z + wgt.z * g[2].z + wgt.w * g[3].z;
      v.w = wgt.x * g[0].w + wgt.y * g[1].w + wgt.z * g[2].w + wgt.w * g[3].w;
    }
    *reinterpret_cast<float4*>(sA + ((size_t)buf * BM + m) * ASTR + lane * 4) = v;
  };
  // weight chunk loader: 32 rows x 256 cols, 8 float4 per thread
  auto load_b = [&](int phase, int j, float4 (&r)[8]) {
    const int tap = phase / ncb, cb = phase - tap * ncb;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int f = tid + kThreads * q;
      const int row = f >> 6, col = (f & 63) * 4;
      const int c = cb * CB + j * KC + row, n = nb * BN + col;
      r[q] = (c < P.Cin && n < P.Cout)
                 ? *reinterpret_cast<const float4*>(P.w2 + ((size_t)tap * P.Cin + c) * P.Cout + n)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_b = [&](int buf, const float4 (&r)[8]) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int f = tid + kThreads * q;
      const int row = f >> 6, col = (f & 63) * 4;
      *reinterpret_cast<float4*>(sB + ((size_t)buf * KC + row) * BN + col) = r[q];
    }
  };

  // ---- prologue: A tile of phase 0, weight chunk 0 -------------------------------------------------------------
  {
    for (int rr = 0; rr < BM / 4; rr++) {
      const int m = rr * 4 + wave;
      float4 g[4], wgt; bool live;
      gather_row(0, m, g, wgt, live);
      store_row(0, m, g, wgt, live);
    }
    float4 r[8];
    load_b(0, 0, r);
    store_b(0, r);
  }
  __syncthreads();

  floatx16 acc0 = {0}, acc1 = {0};
  const int n_wave = nb * BN + wave * 64;                 // first output channel of this wave
  const bool wave_live = n_wave < P.Cout;
  const int mrow = lane & 31, kh = lane >> 5;
  int bbuf = 0;

  for (int phase = 0; phase < nphase; phase++) {
    const int cb = phase % ncb;
    const int cbeff = min(CB, P.Cin - cb * CB);
    const int nchunk = (cbeff + KC - 1) / KC;
    const int rows_per_chunk = (BM / 4 + nchunk - 1) / nchunk;      // A rows of the NEXT phase built per chunk
    const float* a_cur = sA + (size_t)(phase & 1) * BM * ASTR;
    for (int j = 0; j < nchunk; j++) {
      // (1) issue the global loads of the next weight chunk and of the next phase's A rows
      const bool last_chunk = (j + 1 == nchunk);
      const bool have_next_b = !(last_chunk && phase + 1 == nphase);
      float4 rb[8];
      if (have_next_b) load_b(last_chunk ? phase + 1 : phase, last_chunk ? 0 : j + 1, rb);
      float4 g[4], wgt = make_float4(0.f, 0.f, 0.f, 0.f); bool live = false; int m_next = -1;
      // (rows_per_chunk is 1 for 256-channel phases; the loop below handles the general case one row at a time)
      // (2) MFMA over the current chunk
      for (int rr = 0; rr < rows_per_chunk; rr++) {
        const int ridx = j * rows_per_chunk + rr;
        const bool do_row = (phase + 1 < nphase) && ridx < BM / 4;
        if (do_row) { m_next = ridx * 4 + wave; gather_row(phase + 1, m_next, g, wgt, live); }
        if (rr == 0 && wave_live) {
          const float* bcur = sB + (size_t)bbuf * KC * BN + wave * 64 + mrow;
          const float* arow = a_cur + (size_t)mrow * ASTR + j * KC + 4 * kh;
#pragma unroll
          for (int t = 0; t < KC / 8; t++) {
            const float4 a4 = *reinterpret_cast<const float4*>(arow + 8 * t);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
              const int kk = 8 * t + 4 * kh + i;
              const float b0 = bcur[(size_t)kk * BN];
              const float b1 = bcur[(size_t)kk * BN + 32];
              if (OUT_NCHW) {   // D[channel][position]
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, av[i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, av[i], acc1, 0, 0, 0);
              } else {          // D[position][channel]
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b1, acc1, 0, 0, 0);
              }
            }
          }
        }
        if (do_row) store_row((phase + 1) & 1, m_next, g, wgt, live);
      }
      // (3) park the prefetched weight chunk in the other buffer
      if (have_next_b) store_b(bbuf ^ 1, rb);
      __syncthreads();
      bbuf ^= 1;
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------
  if (!wave_live) return;
  auto finish = [&](float v, int ch) { if (P.bias) v += P.bias[ch]; return P.relu ? fmaxf(v, 0.f) : v; };
  if (OUT_NCHW) {
    // D rows = channels, cols = positions: lane&31 -> position, reg -> channel
    const long p = p0 + (lane & 31);
    if (p < npos) {
      const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
      float* ob = L.out + (size_t)b * P.Cout * HoWo + hw;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ch = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (n_wave + ch < P.Cout) ob[(size_t)(n_wave + ch) * HoWo] = finish(acc0[r], n_wave + ch);
        if (n_wave + 32 + ch < P.Cout) ob[(size_t)(n_wave + 32 + ch) * HoWo] = finish(acc1[r], n_wave + 32 + ch);
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const long p = p0 + m;
      if (p < npos) {
        float* ob = L.out + (size_t)p * P.Cout + n_wave + (lane & 31);
        if (n_wave + (lane & 31) < P.Cout) ob[0] = finish(acc0[r], n_wave + (lane & 31));
        if (n_wave + 32 + (lane & 31) < P.Cout) ob[32] = finish(acc1[r], n_wave + 32 + (lane & 31));
      }
    }
  }
}

// ---- MFMA implicit GEMM, second generation: MT x 32 positions per workgroup, 8 waves (2 per SIMD) -----------------
// Same contraction and the same exact-fp32 MFMA as above, re-tiled after the first rocprofv3 PMC pass (round 1:
// SQ_VALU_MFMA_BUSY 35 %, SQ_WAIT_ANY 37 % of wave cycles, 2.36 MB of packed weights streamed L2 -> LDS per 32-position
// tile = 16 B/clk/CU):
//   * a workgroup owns MT*32 positions (MT = 3 -> the 21 824 positions of a 1024x1024 image are 228 tiles, ONE round
//     on 256 CUs; the weight stream per position drops 3x);
//   * 8 waves = 2 per SIMD: wave w owns the 32 output channels [32w, 32w+32) for all MT position sub-tiles (MT
//     accumulators), so while one wave of a SIMD waits on LDS / the barrier the other one feeds the matrix pipe;
//   * the A tile (one kernel tap, 256 channels) is single-buffered in LDS: the rows of the NEXT tap are gathered
//     (coalesced 1 KB NHWC rows), bilinearly combined and parked in registers while the current tap is contracted, and
//     written after its last chunk (two barriers per tap, none per chunk); the weight fragments of a wave go L2 ->
//     registers directly (packing [tap][c/4][o][4]: one float4 per lane per four k-steps), prefetched one chunk ahead;
//   * blockIdx -> tile is XCD-aware: the 8 XCDs each take a contiguous slab of tiles, so a feature-map row is pulled
//     into ONE XCD's L2 instead of all eight.
constexpr int KC2 = 16;          // input channels per weight chunk
// the two-layer MT = 3 instantiation needs 10 registers more than the 256 a wave can have at two waves per SIMD: four of
// the twelve next-tap rows a wave parks per tap go to a private LDS slot (32 KB) instead of registers
template <int MT, int NCONV> constexpr int mfma2_stage_rows() { return (MT == 3 && NCONV == 2) ? 4 : 0; }
constexpr int kThreads2 = 512;

// C256: Cin = Cout = 256 as compile-time constants (the head): the strides become immediates, which is what keeps the
// two-layer instantiation free of register spills (254 VGPRs + 52 B of scratch per lane otherwise: 17 MB of extra writes)
// HEADS: the 1x1 convolution that follows each DeformConv + ReLU in the head (reppoints_cls_out / reppoints_pts_refine_out,
// orientedreppoints_head.py:166-170) is applied in the epilogue -- the 256-channel DeformConv output never goes to HBM.
template <int MT, bool OUT_NCHW, int NCONV, bool C256, bool HEADS = false, bool KSPLIT = false>
__global__ void __launch_bounds__(kThreads2)
dcn_fwd_mfma2_kernel(const FwdParams P, int total_tiles) {
  const int Cin = C256 ? 256 : P.Cin, Cout = C256 ? 256 : P.Cout;
  constexpr int BM2 = 32 * MT;
  constexpr int ROWS = BM2 / 8;                                              // A rows produced per wave per tap
  constexpr int STAGE = mfma2_stage_rows<MT, NCONV>();                       // of them, parked in LDS instead of registers
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sA = reinterpret_cast<float*>(smem);                                // [BM2][ASTR]
  float4* sCw = reinterpret_cast<float4*>(sA + BM2 * ASTR);                  // [BM2 * taps] bilinear weights
  int4* sCi = reinterpret_cast<int4*>(sCw + BM2 * MAX_TAPS);                 // [BM2 * taps] pixel indices
  float4* sStage = reinterpret_cast<float4*>(sCi + BM2 * MAX_TAPS);          // [8 waves][STAGE][64 lanes]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int taps = P.kh * P.kw;
  // XCD-aware remap: hardware places block b on XCD b % 8; give XCD x a contiguous slab of the work
  //   plain:  workgroup = tile, XCD x takes the tiles [x*per, (x+1)*per)
  //   KSPLIT: whole tiles leave the last round of workgroups partly filled (228 tiles on 256 CUs: 18 tap steps on 228 CUs,
  //           none on 28).  Here an XCD owns a contiguous slab of ONE layer's tiles (pair launch: XCDs 0-3 the first layer,
  //           4-7 the second -- an XCD's L2 then only ever holds one layer's 2.36 MB of packed weights); its W workgroups
  //           take the slab's tiles round by round, workgroup i tile r * W + i, exactly as whole-tile scheduling would --
  //           neighbouring tiles run at the same time and share their halo rows in the L2, every workgroup is at the same
  //           tap -- and only the LAST, partial round is split: its `rem` tiles form a sequence of rem * 9 tap steps cut
  //           into W equal ranges.  A range is shorter than a tile, so a workgroup gets the tail (or a middle part) of one
  //           tile and possibly the head of the next; the parts of a tile hand their accumulators down the chain head ->
  //           middle -> tail through scratch images (see the epilogue).  57 tiles on 32 workgroups: 9 + 7.03 tap steps.
  //           Measured (round 3): 2 x 1024^2 (456 tiles) 933 us against 962 us with whole tiles, FETCH_SIZE 316 MB against
  //           220 MB per launch; ONE image (228 tiles) 529 against 516 us -- a third segment per workgroup (coefficient
  //           table, exposed first gather, hand-over) costs what 1.5 tap steps save, so one-round launches keep whole
  //           tiles.  First version: one contiguous range of the layer's (tile, tap) sequence per workgroup -- same step
  //           count, but concurrently running workgroups were 1.8 tiles apart and at different taps: 906 - 925 us at
  //           2 x 1024^2 with 1.92 GB fetched per launch (the whole weight stream and every halo row from the Infinity Cache).
  //           Cutting inside a tap: slower, a partial tap still gathers and stages its whole A tile.
  int tile = 0, wg = 0, seq = 0, seq_end = NCONV * taps, ks_conv = 0;
  int ks_stage = 0, ks_round = 0, ks_nfull = 0, ks_tb = 0, ks_i = 0, ks_per = 1, ks_rs = 0, ks_re = 0, ks_rbase = 0;
  if (KSPLIT) {
    const int b = blockIdx.x, xcd = b & 7;
    ks_per = P.ks_nwg >> 3;                                         // workgroups per XCD
    ks_i = b >> 3;
    int xq, nx;
    if (NCONV == 2) { ks_conv = xcd >> 2; xq = xcd & 3; nx = 4; } else { xq = xcd; nx = 8; }
    wg = (ks_conv * nx + xq) * ks_per + ks_i;                        // scratch slot; predecessor in the same XCD = wg - 1
    ks_tb = (int)(((long)xq * total_tiles) / nx);
    const int te = (int)(((long)(xq + 1) * total_tiles) / nx), slab = te - ks_tb;
    ks_nfull = slab / ks_per;
    const int rem = slab - ks_nfull * ks_per;
    ks_rbase = ks_tb + ks_nfull * ks_per;                            // first tile of the partial round
    ks_rs = (int)(((long)ks_i * rem * taps) / ks_per);
    ks_re = (int)(((long)(ks_i + 1) * rem * taps) / ks_per);
  } else {
    const int b = blockIdx.x, per = (total_tiles + 7) >> 3;
    tile = (b & 7) * per + (b >> 3);
    if (tile >= total_tiles) return;                                         // whole workgroup leaves together
  }
  const int nb = blockIdx.y;
  LevelDesc L = P.lv[0];
  int HoWo = 0, cur_tile = -1;
  long npos = 0, p0 = 0;
  bool first_seg = true;

  // a pair launch runs its two layers one after the other on the SAME coefficient table (same offsets / masks)
  // (NCONV == 1 with gridDim.z == 2: "pair as grid" -- the second layer is a second workgroup of the same tile)
#pragma unroll 1
  for (;;) {
  int conv, t0, t1;
  if (KSPLIT) {
    conv = ks_conv;
    if (ks_stage == 0) {                                    // whole tiles, round by round
      if (ks_round < ks_nfull) { tile = ks_tb + ks_round * ks_per + ks_i; ks_round++; t0 = 0; t1 = taps; }
      else { ks_stage = 1; continue; }
    } else if (ks_stage == 1) {                             // the HEAD of the second tile of this range, if it reaches one
      ks_stage = 2;
      const int a = ks_rs / taps;
      if (ks_re <= (a + 1) * taps) continue;
      tile = ks_rbase + a + 1; t0 = 0; t1 = ks_re - (a + 1) * taps;
    } else if (ks_stage == 2) {                             // the part of the first tile: [rs, min(re, end of that tile))
      ks_stage = 3;
      if (ks_re <= ks_rs) continue;
      const int a = ks_rs / taps;
      tile = ks_rbase + a; t0 = ks_rs - a * taps; t1 = min(ks_re - a * taps, taps);
    } else {
      break;
    }
  } else {
    if (seq >= seq_end) break;
    conv = NCONV == 1 ? (int)blockIdx.z : seq / taps; t0 = 0; t1 = taps;
    seq += taps;
  }
  if (!first_seg) __syncthreads();                          // every wave is past its last read of the previous segment's A tile / table
  first_seg = false;
  if (tile != cur_tile && !((ORP_DCN_KS_DBG & 4) && cur_tile >= 0)) {
  cur_tile = tile;
  int lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if (tile >= P.lv[i].tile0) lvl = i;
  L = P.lv[lvl];
  HoWo = L.Ho * L.Wo;
  npos = (long)P.B * HoWo;
  p0 = (long)(tile - L.tile0) * BM2;

  for (int e = tid; e < BM2 * taps; e += kThreads2) {
    const int m = e / taps, tap = e - m * taps;
    const long p = p0 + m;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 ix = make_int4(0, 0, 0, 0);
    if (p < npos) {
      const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
      const int ho = hw / L.Wo, wo = hw - ho * L.Wo;
      const int ki = tap / P.kw, kj = tap - ki * P.kw;
      const float* ob = L.off + ((size_t)b * 2 * taps + 2 * tap) * HoWo + hw;
      const float off_h = ob[0], off_w = ob[HoWo];
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
        if (L.mask) {                                     // DCNv2: the sample is scaled by its modulation scalar
          const float mm = L.mask[((size_t)b * taps + tap) * HoWo + hw];
          w.x *= mm; w.y *= mm; w.z *= mm; w.w *= mm;
        }
      }
    }
    sCw[e] = w; sCi[e] = ix;
  }
  __syncthreads();
  }   // coefficient table of `tile`

  const float* xin = conv ? L.x2 : L.x;
  const float* w3 = conv ? P.w3b : P.w3;
  const float* bias = conv ? P.bias2 : P.bias;
  float* outp = conv ? L.out2 : L.out;
  const int ncb = Cin / CB;                        // 256-channel blocks per tap (Cin % 256 == 0 on this path)
  const int nphase = taps * ncb;
  const int ph0 = t0 * ncb, ph1 = t1 * ncb;        // this segment's phases
  constexpr int NCHUNK = CB / KC2;                   // 16 chunks per phase

  // one A row = 256 channels = 64 lanes x float4: four coalesced 1 KB neighbour rows, combined with wave-uniform weights
  auto gather_issue = [&](int phase, int m, float4 (&g)[4]) {
    const int tap = phase / ncb, cb = phase - tap * ncb;
    const int4 ix = sCi[m * taps + tap];
    const float* base = xin + cb * CB + lane * 4;
#if ORP_DCN_DBG & 1
    g[0] = g[1] = g[2] = g[3] = make_float4(1.f, 1.f, 1.f, 1.f); return;
#endif
    g[0] = *reinterpret_cast<const float4*>(base + (size_t)ix.x * Cin);
    g[1] = *reinterpret_cast<const float4*>(base + (size_t)ix.y * Cin);
    g[2] = *reinterpret_cast<const float4*>(base + (size_t)ix.z * Cin);
    g[3] = *reinterpret_cast<const float4*>(base + (size_t)ix.w * Cin);
  };
  auto gather_issue_ix = [&](int phase, const int4 ix, float4 (&g)[4]) {      // ... with the pixel indices already in registers
    const int tap = phase / ncb, cb = phase - tap * ncb;
    const float* base = xin + cb * CB + lane * 4;
    g[0] = *reinterpret_cast<const float4*>(base + (size_t)ix.x * Cin);
    g[1] = *reinterpret_cast<const float4*>(base + (size_t)ix.y * Cin);
    g[2] = *reinterpret_cast<const float4*>(base + (size_t)ix.z * Cin);
    g[3] = *reinterpret_cast<const float4*>(base + (size_t)ix.w * Cin);
  };
  auto combine = [&](int phase, int m, const float4 (&g)[4]) {
    const int tap = phase / ncb;
    const float4 wgt = sCw[m * taps + tap];
    float4 v;                                             // explicit fma chain: every instantiation rounds identically
    v.x = __builtin_fmaf(wgt.w, g[3].x, __builtin_fmaf(wgt.z, g[2].x, __builtin_fmaf(wgt.y, g[1].x, wgt.x * g[0].x)));
    v.y = __builtin_fmaf(wgt.w, g[3].y, __builtin_fmaf(wgt.z, g[2].y, __builtin_fmaf(wgt.y, g[1].y, wgt.x * g[0].y)));
    v.z = __builtin_fmaf(wgt.w, g[3].z, __builtin_fmaf(wgt.z, g[2].z, __builtin_fmaf(wgt.y, g[1].z, wgt.x * g[0].z)));
    v.w = __builtin_fmaf(wgt.w, g[3].w, __builtin_fmaf(wgt.z, g[2].w, __builtin_fmaf(wgt.y, g[1].w, wgt.x * g[0].w)));
    return v;
  };
  // B fragments never touch LDS: wave w only ever needs its own 32 output channels, and with the [tap][c/4][o][4]
  // packing the four k-steps (t, i = 0..3) of lane (n, kh) are ONE float4 (channels 8t + 4kh + i of output n):
  // lanes 0-31 / 32-63 read two contiguous 512 B segments.  No shared weight buffer -> no per-chunk barrier.
  const int n_wave = nb * BN + wave * 32;                 // first output channel of this wave
  const int mrow = lane & 31, kh = lane >> 5;
  const bool n_ok = (n_wave + mrow) < Cout;
  auto load_bq = [&](int phase, int j, float4 (&r)[2]) {
    const int tap = phase / ncb, cb = phase - tap * ncb;
    const size_t c4 = (size_t)(tap * Cin + cb * CB + j * KC2 + 4 * kh) >> 2;
    const float* base = w3 + (c4 * Cout + n_wave + mrow) * 4;
#if ORP_DCN_DBG & 2
    r[0] = r[1] = make_float4(1.f, 1.f, 1.f, 1.f); return;
#endif
    if (n_ok) {
      r[0] = *reinterpret_cast<const float4*>(base);
      r[1] = *reinterpret_cast<const float4*>(base + (size_t)8 * Cout);     // channels + 8 -> c4 + 2
    } else {
      r[0] = r[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  // ---- prologue: A tile of phase 0, weight fragments of chunk 0 ------------------------------------------------
  // the weight fragments are fetched WDIST chunks ahead of their use (the single-layer instantiation has the registers for
  // two; measured round 2: 505 us vs 503 us per pair -- the weight latency is not what the matrix pipe waits for)
  constexpr int WDIST = (NCONV == 1) ? ORP_DCN_WDIST : 1;
  // a gathered row is combined GDIST chunks after its loads were issued (the single-layer instantiation has the 16
  // registers for a second row in flight; ROWS + GDIST - 1 <= NCHUNK)
  constexpr int GDIST = (NCONV == 1 && STAGE == 0 && ROWS + 1 <= CB / KC2) ? ORP_DCN_GDIST : 1;
  constexpr bool APF = (NCONV == 1 || ORP_DCN_APF_ALL) && ORP_DCN_APF;
  constexpr bool IPF = APF && ORP_DCN_IPF;
  auto load_lin = [&](int phase, int j, float4 (&r)[2]) {            // chunk (phase, j) with j possibly >= NCHUNK
    const int ph = phase + j / NCHUNK, jj = j % NCHUNK;
    if (ph < nphase) load_bq(ph, jj, r);
    else r[0] = r[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  float4 bq[2], bq1[2];
  {
    load_bq(ph0, 0, bq);
    if (WDIST == 2) load_lin(ph0, 1, bq1);
    // four rows in flight per wave (16 outstanding 1 KB loads) so the first tap's gather latency is paid ROWS/4 times
#pragma unroll 1
    for (int r0 = 0; r0 < ROWS; r0 += 4) {
      float4 g[4][4];
#pragma unroll
      for (int u = 0; u < 4; u++) gather_issue(ph0, (r0 + u) * 8 + wave, g[u]);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int m = (r0 + u) * 8 + wave;
        *reinterpret_cast<float4*>(sA + (size_t)m * ASTR + lane * 4) = combine(ph0, m, g[u]);
      }
    }
  }
  __syncthreads();

  floatx16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) acc[mt] = floatx16{0};

#pragma unroll 1
  for (int phase = ph0; phase < ph1; phase++) {
    const bool next_phase = phase + 1 < ph1;
    float4 hold[ROWS];
    float4 gq[2][4];                                         // gathered rows in flight (GDIST = 2: two)
    int4 ixn = make_int4(0, 0, 0, 0);                        // IPF: pixel indices of the row gathered in the next chunk
    if (IPF && next_phase) ixn = sCi[wave * taps + (phase + 1) / ncb];
    float4 apre[MT];                                         // APF: the A fragments of the next k-step
    if (APF) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++) apre[mt] = *reinterpret_cast<const float4*>(sA + (size_t)(mt * 32 + mrow) * ASTR + 4 * kh);
    }
#pragma unroll
    for (int j = 0; j < NCHUNK; j++) {
      // (1) issue the global loads of the next chunk's weight fragments and of one row of the next phase's A tile
      float4 bn[2];
      load_lin(phase, j + WDIST, bn);
      float4 (&g)[4] = gq[GDIST == 2 ? (j & 1) : 0];
      const bool do_row = next_phase && (j < ROWS);
      if (do_row) { if (IPF) gather_issue_ix(phase + 1, ixn, g); else gather_issue(phase + 1, j * 8 + wave, g); }
      // (2) MFMA over the current chunk
      if (APF) {
        // the A fragments of the NEXT k-step are read from LDS before the MFMAs of this one are issued (the wave's own
        // LDS latency is then hidden behind its own matrix work; the tile only changes behind the tap's barriers, so the
        // chain stops at the last k-step of a tap)
        const float* arow = sA + (size_t)mrow * ASTR + j * KC2 + 4 * kh;
#pragma unroll
        for (int t = 0; t < KC2 / 8; t++) {
          float4 a4[MT];
#pragma unroll
          for (int mt = 0; mt < MT; mt++) a4[mt] = apre[mt];
          const bool more = !(j + 1 == NCHUNK && t + 1 == KC2 / 8);
          if (more) {
            const float* nrow = (t + 1 < KC2 / 8) ? arow + 8 * (t + 1) : arow + KC2;      // next k-step: same chunk or chunk j + 1
#pragma unroll
            for (int mt = 0; mt < MT; mt++) apre[mt] = *reinterpret_cast<const float4*>(nrow + (size_t)mt * 32 * ASTR);
          }
#if ORP_DCN_APF_PIN
          __builtin_amdgcn_sched_barrier(0);                 // keep the reads in FRONT of this k-step's MFMAs (the scheduler sinks them)
#endif
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float b0 = (i == 0) ? bq[t].x : (i == 1) ? bq[t].y : (i == 2) ? bq[t].z : bq[t].w;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
              const float av = (i == 0) ? a4[mt].x : (i == 1) ? a4[mt].y : (i == 2) ? a4[mt].z : a4[mt].w;
              if (OUT_NCHW) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, av, acc[mt], 0, 0, 0);
              else          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[mt], 0, 0, 0);
            }
          }
        }
      } else {
        const float* arow = sA + (size_t)mrow * ASTR + j * KC2 + 4 * kh;
#pragma unroll
        for (int t = 0; t < KC2 / 8; t++) {
          float4 a4[MT];
#pragma unroll
          for (int mt = 0; mt < MT; mt++) a4[mt] = *reinterpret_cast<const float4*>(arow + (size_t)mt * 32 * ASTR + 8 * t);
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float b0 = (i == 0) ? bq[t].x : (i == 1) ? bq[t].y : (i == 2) ? bq[t].z : bq[t].w;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
              const float av = (i == 0) ? a4[mt].x : (i == 1) ? a4[mt].y : (i == 2) ? a4[mt].z : a4[mt].w;
#if ORP_DCN_DBG & 8
              acc[mt][0] += av * b0; continue;
#endif
              if (OUT_NCHW) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, av, acc[mt], 0, 0, 0);   // D[channel][position]
              else          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[mt], 0, 0, 0);   // D[position][channel]
            }
          }
        }
      }
      if (IPF && next_phase && j + 1 < ROWS)                 // pixel indices of the NEXT row: read now, needed one chunk later
        ixn = sCi[((j + 1) * 8 + wave) * taps + (phase + 1) / ncb];
      if (GDIST == 1) {
        if (do_row) {
          const float4 v = combine(phase + 1, j * 8 + wave, g);
          if (j < STAGE) sStage[(wave * (STAGE ? STAGE : 1) + (j < STAGE ? j : 0)) * 64 + lane] = v;   // own slot: no sync needed
          else hold[j < ROWS ? j : 0] = v;
        }
      } else if (next_phase && j >= 1 && j - 1 < ROWS) {      // GDIST = 2: the row gathered one chunk ago is due now
        hold[j - 1 < ROWS ? j - 1 : 0] = combine(phase + 1, (j - 1) * 8 + wave, gq[(j - 1) & 1]);
      }
      if (WDIST == 2) { bq[0] = bq1[0]; bq[1] = bq1[1]; bq1[0] = bn[0]; bq1[1] = bn[1]; }
      else { bq[0] = bn[0]; bq[1] = bn[1]; }
    }
    // two barriers per tap: every wave is past its last read of this tap's A tile -> overwrite it with the next tap's rows
    if (next_phase && !(ORP_DCN_DBG & 4)) {
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < ROWS; rr++)
        *reinterpret_cast<float4*>(sA + (size_t)(rr * 8 + wave) * ASTR + lane * 4) =
            rr < STAGE ? sStage[(wave * (STAGE ? STAGE : 1) + (rr < STAGE ? rr : 0)) * 64 + lane] : hold[rr];
      __syncthreads();
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------
  auto finish = [&](float v, int ch) { if (bias) v += bias[ch]; return P.relu ? fmaxf(v, 0.f) : v; };
  bool store_out = true;
  if (KSPLIT && !(t0 == 0 && t1 == taps)) {
    // a tile of the partial round is cut into two or three parts owned by consecutive workgroups of the XCD.  The part
    // that starts at tap 0 (HEAD) leaves its accumulators in its owner's scratch image (register layout) and raises the
    // owner's flag; a part that starts later waits for its predecessor's image, adds it -- own + (predecessor's), a fixed
    // order: reproducible -- and either publishes the sum in turn (MIDDLE) or finishes the tile (the part that ends at
    // tap 9).  A workgroup computes its head BEFORE the part that waits, waits only on a lower-numbered workgroup (those
    // are dispatched first) and publishes at most once, so the waits cannot deadlock.
    const bool waits = t0 > 0, publishes = t1 < taps;
    constexpr size_t kImage = (size_t)MT * 8 * 16 * 64;
    const size_t lane_off = (size_t)wave * 16 * 64 + lane;
#if !(ORP_DCN_KS_DBG & 1)
    // Every access to the scratch images and flags is an agent-scope atomic (write-through / cache-bypassing per
    // INSTRUCTION): an agent-scope FENCE instead would write back and invalidate the whole L2 of the XCD -- the packed
    // weights every other workgroup streams from it (measured: 575 us with fences vs 520 us without the split).
    if (waits) {
      const float* src = P.ks_scratch + (size_t)(wg - 1) * kImage + lane_off;
      const int* flag = P.ks_flags + (wg - 1);
      if (tid == 0) {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1) __builtin_amdgcn_s_sleep(4);
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
          acc[mt][r] += __hip_atomic_load(src + (size_t)(mt * 8 * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (publishes) {
      float* dst = P.ks_scratch + (size_t)wg * kImage + lane_off;
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
          __hip_atomic_store(dst + (size_t)(mt * 8 * 16 + r) * 64, acc[mt][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // Drain THIS wave's scratch stores before the barrier: a workgroup-scope release fence emits no s_waitcnt vmcnt(0)
      // on gfx950 (the ISA showed the last `global_store_dword ... sc1` directly followed by s_barrier), so without the
      // explicit wait a store could still be in flight to another L2 channel when the flag becomes visible and the
      // consumer would add whatever the previous launch left in the (reused) scratch image.  The stores are sc1
      // write-through, so once vmcnt reaches 0 they are visible at agent scope; no L2 write-back is needed.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();                                          // ... and every other wave has drained its own
      if (tid == 0) __hip_atomic_store(P.ks_flags + wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
    if (publishes) store_out = false;
  }
  if (HEADS) {
    // out1[p, k] = sum_c relu(dcn[p, c]) * W1[k, c] + b1[k] (+ residual): every wave holds 32 of the 256 channels of its
    // positions -> per-wave partial sums, added over the two half-waves (shuffle) and the eight waves (LDS, fixed order).
    // The A tile is dead: its memory holds the head's weights and the partial sums.
    float* sW = sA;                                        // [256][KH]
    float* sRed = sA + 256 * KH;                           // [8 waves][BM2][KH]
    const float* hw_ = P.head_w[conv];
    const int K = P.head_k[conv];
    __syncthreads();                                       // every wave is past its last read of the A tile
    for (int i = tid; i < 256 * KH; i += kThreads2) sW[i] = hw_[i];
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll 1
      for (int q = 0; q < KH / 4; q++) {                     // four output channels at a time keeps the register need small
        float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int ch = n_wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const float v = finish(acc[mt][r], ch);
          const float4 w4 = *reinterpret_cast<const float4*>(sW + ch * KH + 4 * q);
          part.x = __builtin_fmaf(v, w4.x, part.x);
          part.y = __builtin_fmaf(v, w4.y, part.y);
          part.z = __builtin_fmaf(v, w4.z, part.z);
          part.w = __builtin_fmaf(v, w4.w, part.w);
        }
        part.x += __shfl_xor(part.x, 32, 64);
        part.y += __shfl_xor(part.y, 32, 64);
        part.z += __shfl_xor(part.z, 32, 64);
        part.w += __shfl_xor(part.w, 32, 64);
        if (lane < 32) *reinterpret_cast<float4*>(sRed + ((size_t)wave * BM2 + mt * 32 + lane) * KH + 4 * q) = part;
      }
    }
    __syncthreads();
    float* ho = conv ? L.hout2 : L.hout;
    const float* hres = conv ? L.hres2 : nullptr;
    const float* hb = P.head_b[conv];
    for (int idx = tid; idx < BM2 * K; idx += kThreads2) {
      const int k = idx / BM2, m = idx - k * BM2;
      const long p = p0 + m;
      if (p >= npos) continue;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; w++) v += sRed[((size_t)w * BM2 + m) * KH + k];
      const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
      const size_t o = ((size_t)b * K + k) * HoWo + hw;
      if (hb) v += hb[k];
      if (hres) v += hres[o];
      ho[o] = v;
    }
    store_out = false;                                     // (the next layer's prologue is behind its own barrier)
  }
  if (n_wave >= Cout) store_out = false;                 // idle wave: it still meets the other waves at every barrier above
  if (store_out) {
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    if (OUT_NCHW) {
      const long p = p0 + mt * 32 + (lane & 31);
      if (p < npos) {
        const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
        float* ob = outp + (size_t)b * Cout * HoWo + hw;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int ch = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (n_wave + ch < Cout) ob[(size_t)(n_wave + ch) * HoWo] = finish(acc[mt][r], n_wave + ch);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const long p = p0 + mt * 32 + m;
        if (p < npos && n_wave + (lane & 31) < Cout) outp[(size_t)p * Cout + n_wave + (lane & 31)] = finish(acc[mt][r], n_wave + (lane & 31));
      }
    }
  }
  }
  }
}

template <int MT, int NCONV>
size_t mfma2_smem() {
  return sizeof(float) * ((size_t)32 * MT * ASTR) + (sizeof(float4) + sizeof(int4)) * 32 * MT * MAX_TAPS +
         sizeof(float4) * 8 * 64 * mfma2_stage_rows<MT, NCONV>();
}

template <int MT, bool OUT_NCHW, int NCONV, bool C256>
hipError_t launch_mfma2_nc(const FwdParams& P, int tiles, int nblk_n, hipStream_t st) {
  const size_t smem = mfma2_smem<MT, NCONV>();
  // once per kernel instantiation (not per launch: the call is not allowed while a stream is being captured)
  struct Tag {};
  hipError_t e = orp::set_max_dynamic_lds_once<Tag>(reinterpret_cast<const void*>(&dcn_fwd_mfma2_kernel<MT, OUT_NCHW, NCONV, C256>), smem);
  if (e != hipSuccess) return e;
  const int per = (tiles + 7) >> 3;
  hipLaunchKernelGGL((dcn_fwd_mfma2_kernel<MT, OUT_NCHW, NCONV, C256>), dim3(per * 8, nblk_n, (NCONV == 1 && P.nconv == 2) ? 2 : 1),
                     dim3(kThreads2), smem, st, P, tiles);
  return hipGetLastError();
}
template <int MT>
hipError_t launch_mfma2_heads(const FwdParams& P, int tiles, hipStream_t st) {
  const size_t smem = mfma2_smem<MT, 2>();
  struct TagH { int unused; };
  hipError_t e = orp::set_max_dynamic_lds_once<TagH>(reinterpret_cast<const void*>(&dcn_fwd_mfma2_kernel<MT, true, 2, true, true>), smem);
  if (e != hipSuccess) return e;
  const int per = (tiles + 7) >> 3;
  hipLaunchKernelGGL((dcn_fwd_mfma2_kernel<MT, true, 2, true, true>), dim3(per * 8, 1, 1), dim3(kThreads2), smem, st, P, tiles);
  return hipGetLastError();
}
// tap-granular split (MT = 3, Cin = Cout = 256): ks_nwg workgroups, one per CU
template <bool OUT_NCHW, int NCONV>
hipError_t launch_mfma2_ksplit(const FwdParams& P, int tiles, hipStream_t st) {
  const size_t smem = mfma2_smem<3, NCONV>();
  struct TagK {};
  hipError_t e = orp::set_max_dynamic_lds_once<TagK>(reinterpret_cast<const void*>(&dcn_fwd_mfma2_kernel<3, OUT_NCHW, NCONV, true, false, true>), smem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((dcn_fwd_mfma2_kernel<3, OUT_NCHW, NCONV, true, false, true>), dim3(P.ks_nwg, 1, 1), dim3(kThreads2), smem, st, P, tiles);
  return hipGetLastError();
}
constexpr size_t kKsSlotBytes = sizeof(float) * 3 * 8 * 16 * 64;       // one accumulator image (MT = 3)
inline int ks_workgroups() {                                            // one workgroup per CU, a multiple of the 8 XCDs
  static const int n = [] {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    const int cu = prop.multiProcessorCount & ~7;
    return cu >= 8 ? (cu > 1024 ? 1024 : cu) : 8;
  }();
  return n;
}
inline size_t ks_bytes() { return align256_((size_t)ks_workgroups() * kKsSlotBytes) + align256_(sizeof(int) * 1024); }

template <int MT, bool OUT_NCHW, int NCONV>
hipError_t launch_mfma2_n(const FwdParams& P, int tiles, int nblk_n, hipStream_t st) {
  return (P.Cin == 256 && P.Cout == 256) ? launch_mfma2_nc<MT, OUT_NCHW, NCONV, true>(P, tiles, nblk_n, st)
                                         : launch_mfma2_nc<MT, OUT_NCHW, NCONV, false>(P, tiles, nblk_n, st);
}
template <int MT, bool OUT_NCHW>
hipError_t launch_mfma2(const FwdParams& P, int tiles, int nblk_n, hipStream_t st) {
  static const bool pair_as_grid = getenv("ORP_DCN_PAIR_GRID") && atoi(getenv("ORP_DCN_PAIR_GRID")) == 1;   // dev aid
  return (P.nconv == 2 && !pair_as_grid) ? launch_mfma2_n<MT, OUT_NCHW, 2>(P, tiles, nblk_n, st)
                                         : launch_mfma2_n<MT, OUT_NCHW, 1>(P, tiles, nblk_n, st);
}

// ---- direct kernel: every configuration (groups, deformable groups, DCNv2 mask + bias), NCHW in / out -----------
__global__ void dcn_fwd_direct_kernel(const float* __restrict__ x, const float* __restrict__ off,
                                      const float* __restrict__ mask, const float* __restrict__ w,
                                      const float* __restrict__ bias, float* __restrict__ out, int B, int Cin, int H,
                                      int W, int Cout, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw,
                                      int dh, int dw, int groups, int dg) {
  const long total = (long)B * Cout * Ho * Wo;
  const int taps = kh * kw, cpg = Cin / groups, opg = Cout / groups, cpdg = Cin / dg;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int wo = (int)(idx % Wo);
    const int ho = (int)((idx / Wo) % Ho);
    const int o = (int)((idx / ((long)Wo * Ho)) % Cout);
    const int b = (int)(idx / ((long)Wo * Ho * Cout));
    const int g = o / opg;
    float acc = 0.f;
    for (int cc = 0; cc < cpg; cc++) {
      const int c = g * cpg + cc;
      const int dgi = c / cpdg;
      const float* xp = x + ((size_t)b * Cin + c) * H * W;
      const float* op = off + ((size_t)b * dg + dgi) * 2 * taps * Ho * Wo;
      const float* mp = mask ? mask + ((size_t)b * dg + dgi) * taps * Ho * Wo : nullptr;
      for (int tap = 0; tap < taps; tap++) {
        const int ki = tap / kw, kj = tap - ki * kw;
        const float off_h = op[((size_t)(2 * tap) * Ho + ho) * Wo + wo];
        const float off_w = op[((size_t)(2 * tap + 1) * Ho + ho) * Wo + wo];
        const float h_im = (float)(ho * sh - ph + ki * dh) + off_h;
        const float w_im = (float)(wo * sw - pw + kj * dw) + off_w;
        float val = 0.f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const int h_high = h_low + 1, w_high = w_low + 1;
          const float lh = h_im - h_low, lw = w_im - w_low, hh = 1.f - lh, hw_ = 1.f - lw;
          const float v1 = (h_low >= 0 && w_low >= 0) ? xp[h_low * W + w_low] : 0.f;
          const float v2 = (h_low >= 0 && w_high <= W - 1) ? xp[h_low * W + w_high] : 0.f;
          const float v3 = (h_high <= H - 1 && w_low >= 0) ? xp[h_high * W + w_low] : 0.f;
          const float v4 = (h_high <= H - 1 && w_high <= W - 1) ? xp[h_high * W + w_high] : 0.f;
          val = hh * hw_ * v1 + hh * lw * v2 + lh * hw_ * v3 + lh * lw * v4;
        }
        if (mp) val *= mp[((size_t)tap * Ho + ho) * Wo + wo];
        acc += w[((size_t)o * cpg + cc) * taps + tap] * val;
      }
    }
    if (bias) acc += bias[o];
    out[idx] = acc;
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int out_dim(int in, int pad, int dil, int k, int stride) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }

}  // namespace

extern "C" {

int orp_dcn_pack_weight(const float* weight, int c_out, int c_in, int kh, int kw, float* packed, void* stream) {
  if (!weight || !packed || c_out <= 0 || c_in <= 0 || kh <= 0 || kw <= 0) return ORP_EINVAL;
  const long total = (long)c_out * c_in * kh * kw;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, c_out, c_in, kh * kw,
                     packed);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && orp_split::shape_ok(c_in, c_out, kh, kw))        // the three bf16 planes of the split path
    e = orp_split::pack_planes(weight, c_out, c_in, kh * kw, reinterpret_cast<uint16_t*>(packed + 2 * total), (hipStream_t)stream);
  return e == hipSuccess ? ORP_OK : (int)e;
}

size_t orp_dcn_packed_weight_floats(int c_out, int c_in, int kh, int kw) {
  // [tap][c][o] followed by [tap][c/4][o][4], followed (shapes the split path takes) by the weights split exactly into
  // three bf16 planes [3][tap][c/16][2][o][8] = 6 bytes per weight
  const size_t n = (size_t)c_out * c_in * kh * kw;
  return 2 * n + (orp_split::shape_ok(c_in, c_out, kh, kw) ? (orp_split::plane_elems(c_out, c_in, kh * kw) + 1) / 2 : 0);
}

// fp32-by-splitting path (orp_dcn_split.hip): 0 = off (exact fp32 MFMA); 6 / 9 = partial products of three bf16 pieces per
// operand pair (operands represented exactly); 3 = two fp16 pieces per operand (11 + 11 bits: 2^-22 relative, after an exact
// power-of-two range scaling from max |x| of the launch's inputs), products hi*hi, hi*lo, lo*hi -- needs scratch in the
// workspace and no modulation mask, else the call runs as mode 6.
// Default 3 (round 5): at least as accurate as mode 6 on every test shape (three accumulator roundings per 16 channels instead of
// eight; error against the fp64-accumulated oracle below the exact-fp32 path's own, tests/test_gpu_dcn_split.py prints all four
// modes) and half the matrix work: the head's pair launch 483 us (mode 0) -> 296 (mode 6) -> 192 (mode 3) inside the step.  Round 4
// had to leave it opt-in: graph replays after an eager call lost their detections.  That was the range words being zeroed by
// hipMemsetAsync NODES, which replayed with a wrong pattern (0x80808080 read back by the kernel behind them); every fill of this
// library is a kernel node now (orp_launch.hpp fill_async; tests/test_gpu_conv_split.py::test_fp16_pieces_mode_graph_replay_...).
// The environment overrides the default (ORP_DCN_SPLIT = 0 | 3 | 6 (or 1) | 9), orp_dcn_set_split_mode() overrides both.
static int g_split_mode = -1;
static int split_mode() {
  if (g_split_mode < 0) {
    const char* e = getenv("ORP_DCN_SPLIT");
    const int v = e ? atoi(e) : 3;
    g_split_mode = v == 9 ? 9 : v == 3 ? 3 : (v == 1 || v == 6) ? 6 : 0;
  }
  return g_split_mode;
}
int orp_dcn_set_split_mode(int mode) {
  if (mode != 0 && mode != 3 && mode != 6 && mode != 9 && mode != -1) return ORP_EINVAL;
  g_split_mode = mode;                                   // -1: back to the environment's choice
  return ORP_OK;
}
int orp_dcn_get_split_mode(void) { return split_mode(); }

int orp_dcn_fast_path_ok(int c_in, int c_out, int kh, int kw, int groups, int deformable_groups) {
  return (groups == 1 && deformable_groups == 1 && kh * kw <= MAX_TAPS && c_in % 32 == 0 && c_in >= 32 &&
          c_out % 64 == 0 && c_out >= 64) ? 1 : 0;
}

size_t orp_dcn_forward_workspace_bytes(const orp_dcn_level* levels_host, int nlevels, int batch, int c_in,
                                       int in_layout) {
  // (+ the scratch of the tap-granular split: accumulator images of the cut tiles and their flags)
  if (in_layout == 1 || !levels_host) return 256 + ks_bytes();
  size_t tot = 0;
  for (int i = 0; i < nlevels; i++) tot += align256(sizeof(float) * (size_t)batch * c_in * levels_host[i].height * levels_host[i].width);
  return tot + 256 + ks_bytes();
}

int orp_dcn_forward_multi(const orp_dcn_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                          const float* weight_packed, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w,
                          int dil_h, int dil_w, int in_layout, int out_layout, void* workspace, size_t workspace_bytes,
                          void* stream) {
  return orp_dcn_forward_multi_ex(levels_host, nullptr, nlevels, batch, c_in, c_out, weight_packed, nullptr, 0, kh, kw,
                                  stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, in_layout, out_layout, workspace,
                                  workspace_bytes, stream);
}

// One or two DeformConv layers (nconv) over the same levels / offsets / masks.  levels2_host, weight2, bias2: the second
// layer (nconv == 2), else ignored.
static int dcn_forward_impl(const orp_dcn_level* levels_host, const orp_dcn_level* levels2_host,
                            const float* const* masks_host, int nlevels, int batch, int c_in, int c_out,
                            const float* weight_packed, const float* weight2_packed, const float* bias, const float* bias2,
                            int relu, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h,
                            int dil_w, int in_layout, int out_layout, void* workspace, size_t workspace_bytes,
                            void* stream, const orp_dcn_heads* heads = nullptr, const uint32_t* amax_in = nullptr,
                            int amax_stride = 0) {
  const int nconv = levels2_host ? 2 : 1;
  if (heads && (nconv != 2 || c_in != 256 || c_out != 256 || out_layout != 0 || !heads->weight_a_packed ||
                !heads->weight_b_packed || !heads->levels || heads->k_a <= 0 || heads->k_a > KH || heads->k_b <= 0 ||
                heads->k_b > KH))
    return ORP_EINVAL;
  if (!levels_host || nlevels <= 0 || nlevels > MAX_LEVELS || batch <= 0 || !weight_packed) return ORP_EINVAL;
  if (nconv == 2 && !weight2_packed) return ORP_EINVAL;
  if (!orp_dcn_fast_path_ok(c_in, c_out, kh, kw, 1, 1)) return ORP_EINVAL;
  if ((in_layout != 0 && in_layout != 1) || (out_layout != 0 && out_layout != 1)) return ORP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const size_t w3_off = (size_t)kh * kw * c_in * c_out;
  if (in_layout == 0 && workspace_bytes < (size_t)nconv * (orp_dcn_forward_workspace_bytes(levels_host, nlevels, batch, c_in, 0) - ks_bytes()))
    return ORP_EWORKSPACE;
  char* wsp = reinterpret_cast<char*>(workspace);
  char* const ws_end = wsp + workspace_bytes;
  // kernel generation: 2 (MT*32-position tiles, 256-channel phases, one or two layers per launch) when Cin is a multiple
  // of 256, else 1.  ORP_DCN_MT: dev aid.
  static const int force_mt = getenv("ORP_DCN_MT") ? atoi(getenv("ORP_DCN_MT")) : -1;   // 0 = first-generation kernel
  int gen = (c_in % CB == 0) ? 2 : 1;
  if (nconv == 2 && gen != 2) return ORP_EINVAL;
  // tile height: MT*32 positions per workgroup, chosen to minimise rounds x tile height on 256 CUs (B=1, 1024x1024:
  // MT = 3 -> 228 tiles, one round)
  int MT = 0;
  if (gen >= 2) {
    long npos_all = 0;
    for (int i = 0; i < nlevels; i++)
      npos_all += (long)batch * out_dim(levels_host[i].height, pad_h, dil_h, kh, stride_h) *
                  out_dim(levels_host[i].width, pad_w, dil_w, kw, stride_w);
    long best = -1;
    for (int mt = 1; mt <= 3; mt++) {
      const long t = (npos_all + 32 * mt - 1) / (32 * mt) + nlevels;     // upper bound incl. per-level remainders
      const long cost = ((t + 255) / 256) * mt * 100 + (mt == 1 ? 40 : mt == 2 ? 10 : 0);   // small bias to taller tiles
      if (best < 0 || cost < best) { best = cost; MT = mt; }
    }
    if (force_mt >= 1 && force_mt <= 3) MT = force_mt;
    if (heads && MT < 2) MT = 2;
    if (force_mt == 0 && nconv == 1) { gen = 1; MT = 0; }
  }
  const int bm = MT > 0 ? 32 * MT : BM;
  int tiles = 0;
  FwdParams P;
  P.nlev = nlevels; P.B = batch; P.Cin = c_in; P.Cout = c_out;
  P.kh = kh; P.kw = kw; P.sh = stride_h; P.sw = stride_w; P.ph = pad_h; P.pw = pad_w; P.dh = dil_h; P.dw = dil_w;
  P.w2 = weight_packed; P.w3 = weight_packed + w3_off; P.bias = bias; P.relu = relu ? 1 : 0;
  P.nconv = nconv;
  P.ks_scratch = nullptr; P.ks_flags = nullptr; P.ks_nwg = 0; P.ks_total = 0;
  P.w3b = weight2_packed ? weight2_packed + w3_off : P.w3;
  P.bias2 = bias2;
  TransposeLevels TL;
  int tbx = 0, ntl = 0;
  for (int i = 0; i < nlevels; i++) {
    const orp_dcn_level& lv = levels_host[i];
    if (!lv.input || !lv.offset || (!lv.output && !heads) || lv.height <= 0 || lv.width <= 0) return ORP_EINVAL;
    if (heads && (!heads->levels[i].output_a || !heads->levels[i].output_b)) return ORP_EINVAL;
    if (nconv == 2 && (!levels2_host[i].input || (!levels2_host[i].output && !heads) || levels2_host[i].height != lv.height ||
                       levels2_host[i].width != lv.width))
      return ORP_EINVAL;
    LevelDesc& D = P.lv[i];
    D.H = lv.height; D.W = lv.width;
    D.Ho = out_dim(lv.height, pad_h, dil_h, kh, stride_h);
    D.Wo = out_dim(lv.width, pad_w, dil_w, kw, stride_w);
    if (D.Ho <= 0 || D.Wo <= 0) return ORP_EINVAL;
    if ((long)batch * lv.height * lv.width >= (1L << 31)) return ORP_ETOOBIG;
    D.off = lv.offset; D.out = lv.output;
    D.mask = masks_host ? masks_host[i] : nullptr;
    D.out2 = nconv == 2 ? levels2_host[i].output : lv.output;
    D.hout = heads ? heads->levels[i].output_a : nullptr;
    D.hout2 = heads ? heads->levels[i].output_b : nullptr;
    D.hres2 = heads ? heads->levels[i].residual_b : nullptr;
    for (int cv = 0; cv < nconv; cv++) {
      const float* src = cv ? levels2_host[i].input : lv.input;
      if (in_layout == 0) {
        if (ntl >= 2 * MAX_LEVELS) return ORP_EINVAL;
        float* nhwc = reinterpret_cast<float*>(wsp);
        const int HW = lv.height * lv.width;
        wsp += align256(sizeof(float) * (size_t)batch * c_in * HW);
        TL.in[ntl] = src; TL.out[ntl] = nhwc; TL.hw[ntl] = HW; TL.bx0[ntl] = tbx;
        tbx += (HW + 31) / 32;
        ntl++;
        src = nhwc;
      }
      if (cv == 0) { D.x = src; D.x2 = src; } else { D.x2 = src; }
    }
    D.tile0 = tiles;
    tiles += (int)(((long)batch * D.Ho * D.Wo + bm - 1) / bm);
  }
  for (int i = nlevels; i < MAX_LEVELS; i++) { P.lv[i] = P.lv[0]; P.lv[i].tile0 = 0x7fffffff; }
  if (in_layout == 0) {                                  // NCHW inputs: one transposition launch for all levels / layers
    TL.nlev = ntl;
    for (int i = ntl; i <= 2 * MAX_LEVELS; i++) TL.bx0[i] = tbx;
    for (int i = ntl; i < 2 * MAX_LEVELS; i++) { TL.in[i] = TL.in[0]; TL.out[i] = TL.out[0]; TL.hw[i] = 0; }
    hipLaunchKernelGGL(nchw_to_nhwc_multi_kernel, dim3(tbx, (c_in + 31) / 32, batch), dim3(256), 0, st, TL, c_in);
  }
  // opt-in: the contraction on the bf16 matrix pipe with every fp32 operand split exactly into three bf16 pieces
  if (split_mode() != 0 && !heads && orp_split::shape_ok(c_in, c_out, kh, kw)) {
    orp_split::Args A;
    A.nlev = nlevels; A.B = batch; A.Cin = c_in; A.Cout = c_out;
    A.kh = kh; A.kw = kw; A.sh = stride_h; A.sw = stride_w; A.ph = pad_h; A.pw = pad_w; A.dh = dil_h; A.dw = dil_w;
    int mode = split_mode();
    char* scratch = reinterpret_cast<char*>(align256_(reinterpret_cast<size_t>(wsp)));
    // (amax_in: the producer of the channels-last inputs left an upper bound of max |x| -- no pre-pass, no scratch; it only
    //  describes the tensors as handed over, so not with in_layout 0 where this call transposes copies of its own)
    const bool have_amax = amax_in && in_layout == 1;
    if (mode == 3 && (masks_host || (!have_amax && (!workspace || scratch + 256 > ws_end)))) mode = 6;
    const int taps_ = kh * kw;
    A.planes[0] = orp_split::planes_of(weight_packed, c_out, c_in, taps_, mode);
    A.planes[1] = weight2_packed ? orp_split::planes_of(weight2_packed, c_out, c_in, taps_, mode) : A.planes[0];
    A.wscale[0] = orp_split::wscale_of(weight_packed, c_out, c_in, taps_);
    A.wscale[1] = weight2_packed ? orp_split::wscale_of(weight2_packed, c_out, c_in, taps_) : A.wscale[0];
    A.scratch = (mode == 3 && !have_amax) ? reinterpret_cast<unsigned*>(scratch) : nullptr;
    A.amax_in = (mode == 3 && have_amax) ? amax_in : nullptr; A.amax_stride = amax_stride;
    A.bias[0] = bias; A.bias[1] = bias2;
    A.relu = relu ? 1 : 0; A.nconv = nconv; A.out_nchw = out_layout == 0 ? 1 : 0; A.nprod = mode;
    for (int i = 0; i < nlevels; i++) {
      const LevelDesc& D = P.lv[i];
      orp_split::Level& S = A.lv[i];
      S.x[0] = D.x; S.x[1] = D.x2; S.off = D.off; S.mask = D.mask; S.out[0] = D.out; S.out[1] = D.out2;
      S.H = D.H; S.W = D.W; S.Ho = D.Ho; S.Wo = D.Wo;
      S.planes = nullptr; S.bias = nullptr; S.wscale = nullptr;
    }
    OrpProfScope prof(ORP_PROF_DCN_FWD, st);
    const hipError_t se = orp_split::launch(A, st);
    return se == hipSuccess ? ORP_OK : (int)se;
  }
  // tap-granular split of the last round of tiles: for launches of MORE tiles than CUs (456 tiles on 256 CUs: 36 tap steps
  // on the busiest CU with whole tiles, 32.06 on average; 933 against 962 us at 2 x 1024^2).  One-round launches (228 tiles,
  // one 1024^2 image) keep whole tiles: measured 529 against 516 us (see the kernel).  ORP_DCN_KSPLIT=0 / 1: dev aid (off /
  // whenever the ranges are at least three taps long)
  static const int ks_env = getenv("ORP_DCN_KSPLIT") ? atoi(getenv("ORP_DCN_KSPLIT")) : -1;
  bool use_ks = false;
  if (gen == 2 && MT == 3 && !heads && c_in == 256 && c_out == 256 && ks_env != 0 && workspace &&
      !(getenv("ORP_DCN_PAIR_GRID") && atoi(getenv("ORP_DCN_PAIR_GRID")) == 1)) {
    const int nwg = ks_workgroups(), taps = kh * kw, per = nwg >> 3, nx = 8 / nconv;      // workgroups per XCD, XCDs per layer
    // steps of the busiest workgroup: whole tiles (both layers in one workgroup) vs whole-tile rounds + the split last round
    const long per_old = (long)((tiles + nwg - 1) / nwg) * nconv * taps;
    long per_new = 0, min_range = taps;
    for (int x = 0; x < nx; x++) {
      const int slab = (int)(((long)(x + 1) * tiles) / nx) - (int)(((long)x * tiles) / nx);
      const int nfull = slab / per, rem = slab - nfull * per;
      const long steps = (long)nfull * taps + ((long)rem * taps + per - 1) / per;
      if (steps > per_new) per_new = steps;
      if (rem > 0 && (long)rem * taps / per < min_range) min_range = (long)rem * taps / per;
    }
    wsp = reinterpret_cast<char*>(align256_(reinterpret_cast<size_t>(wsp)));
    // (ranges of fewer than 3 taps would cut a tile into long hand-over chains: whole tiles then)
    if (wsp + ks_bytes() <= ws_end && (long)tiles * taps < (1L << 30) && per >= 1 && min_range >= 3 &&
        (ks_env == 1 || (tiles > nwg && per_new * 100 <= per_old * 95))) {
      P.ks_scratch = reinterpret_cast<float*>(wsp);
      P.ks_flags = reinterpret_cast<int*>(wsp + align256_((size_t)nwg * kKsSlotBytes));
      P.ks_nwg = nwg; P.ks_total = tiles * taps;
      const hipError_t me = orp::fill_async(P.ks_flags, 0, sizeof(int) * nwg, st);
      if (me != hipSuccess) return (int)me;
      use_ks = true;
    }
  }
  hipError_t e;
  OrpProfScope prof(ORP_PROF_DCN_FWD, st);
  const int nblk_n = (c_out + BN - 1) / BN;
  const bool nchw = out_layout == 0;
  P.head_w[0] = heads ? heads->weight_a_packed : nullptr; P.head_w[1] = heads ? heads->weight_b_packed : nullptr;
  P.head_b[0] = heads ? heads->bias_a : nullptr; P.head_b[1] = heads ? heads->bias_b : nullptr;
  P.head_k[0] = heads ? heads->k_a : 0; P.head_k[1] = heads ? heads->k_b : 0;
  if (heads) {                                                 // (MT >= 2: the partial sums need the A tile's LDS)
    if (gen != 2) return ORP_EINVAL;
    e = MT == 2 ? launch_mfma2_heads<2>(P, tiles, st) : launch_mfma2_heads<3>(P, tiles, st);
    return e == hipSuccess ? ORP_OK : (int)e;
  }
  if (use_ks) {
    e = nconv == 2 ? (nchw ? launch_mfma2_ksplit<true, 2>(P, tiles, st) : launch_mfma2_ksplit<false, 2>(P, tiles, st))
                   : (nchw ? launch_mfma2_ksplit<true, 1>(P, tiles, st) : launch_mfma2_ksplit<false, 1>(P, tiles, st));
    return e == hipSuccess ? ORP_OK : (int)e;
  }
  if (gen == 2) {
    if (MT == 1) e = nchw ? launch_mfma2<1, true>(P, tiles, nblk_n, st) : launch_mfma2<1, false>(P, tiles, nblk_n, st);
    else if (MT == 2) e = nchw ? launch_mfma2<2, true>(P, tiles, nblk_n, st) : launch_mfma2<2, false>(P, tiles, nblk_n, st);
    else e = nchw ? launch_mfma2<3, true>(P, tiles, nblk_n, st) : launch_mfma2<3, false>(P, tiles, nblk_n, st);
    return e == hipSuccess ? ORP_OK : (int)e;
  }
  const size_t smem = sizeof(float) * (2 * BM * ASTR + 2 * KC * BN) + (sizeof(float4) + sizeof(int4)) * BM * MAX_TAPS;
  dim3 grid(tiles, nblk_n);
  if (out_layout == 0) {
    struct TagA {};
    const hipError_t attr = orp::set_max_dynamic_lds_once<TagA>(reinterpret_cast<const void*>(&dcn_fwd_mfma_kernel<true>), smem);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(dcn_fwd_mfma_kernel<true>, grid, dim3(kThreads), smem, st, P);
  } else {
    struct TagB {};
    const hipError_t attr = orp::set_max_dynamic_lds_once<TagB>(reinterpret_cast<const void*>(&dcn_fwd_mfma_kernel<false>), smem);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(dcn_fwd_mfma_kernel<false>, grid, dim3(kThreads), smem, st, P);
  }
  e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

int orp_dcn_forward_multi_ex(const orp_dcn_level* levels_host, const float* const* masks_host, int nlevels, int batch,
                             int c_in, int c_out, const float* weight_packed, const float* bias, int relu, int kh, int kw,
                             int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int in_layout,
                             int out_layout, void* workspace, size_t workspace_bytes, void* stream) {
  return dcn_forward_impl(levels_host, nullptr, masks_host, nlevels, batch, c_in, c_out, weight_packed, nullptr, bias,
                          nullptr, relu, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, in_layout, out_layout,
                          workspace, workspace_bytes, stream);
}

int orp_dcn_forward_pair(const orp_dcn_level* levels_a, const orp_dcn_level* levels_b, const float* const* masks_host,
                         int nlevels, int batch, int c_in, int c_out, const float* weight_a_packed,
                         const float* weight_b_packed, const float* bias_a, const float* bias_b, int relu, int kh, int kw,
                         int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int in_layout,
                         int out_layout, void* workspace, size_t workspace_bytes, void* stream) {
  if (!levels_b) return ORP_EINVAL;
  if (c_in % CB != 0) return ORP_EINVAL;
  return dcn_forward_impl(levels_a, levels_b, masks_host, nlevels, batch, c_in, c_out, weight_a_packed, weight_b_packed,
                          bias_a, bias_b, relu, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, in_layout,
                          out_layout, workspace, workspace_bytes, stream);
}

int orp_dcn_forward_pair_amax(const orp_dcn_level* levels_a, const orp_dcn_level* levels_b, const float* const* masks_host,
                              int nlevels, int batch, int c_in, int c_out, const float* weight_a_packed,
                              const float* weight_b_packed, const float* bias_a, const float* bias_b, int relu, int kh, int kw,
                              int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int in_layout,
                              int out_layout, void* workspace, size_t workspace_bytes, const uint32_t* amax_in, int amax_stride,
                              void* stream) {
  if (!levels_b) return ORP_EINVAL;
  if (c_in % CB != 0 || (amax_in && amax_stride != 0 && amax_stride != 1)) return ORP_EINVAL;
  return dcn_forward_impl(levels_a, levels_b, masks_host, nlevels, batch, c_in, c_out, weight_a_packed, weight_b_packed,
                          bias_a, bias_b, relu, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, in_layout,
                          out_layout, workspace, workspace_bytes, stream, nullptr, amax_in, amax_stride);
}

size_t orp_dcn_head_packed_floats(void) { return (size_t)256 * KH; }

int orp_dcn_pack_head_weight(const float* weight, int k, float* packed, void* stream) {
  if (!weight || !packed || k <= 0 || k > KH) return ORP_EINVAL;
  hipLaunchKernelGGL(pack_head_kernel, dim3(20), dim3(256), 0, (hipStream_t)stream, weight, k, packed);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

int orp_dcn_forward_pair_heads(const orp_dcn_level* levels_a, const orp_dcn_level* levels_b, int nlevels, int batch,
                               const float* weight_a_packed, const float* weight_b_packed, const orp_dcn_heads* heads,
                               int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                               int in_layout, void* workspace, size_t workspace_bytes, void* stream) {
  if (!levels_b || !heads) return ORP_EINVAL;
  return dcn_forward_impl(levels_a, levels_b, nullptr, nlevels, batch, 256, 256, weight_a_packed, weight_b_packed, nullptr,
                          nullptr, 1, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, in_layout, 0, workspace,
                          workspace_bytes, stream, heads);
}

int orp_dcn_forward_direct(const float* input, const float* offset, const float* mask, const float* weight,
                           const float* bias, float* output, int batch, int c_in, int height, int width, int c_out,
                           int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                           int groups, int deformable_groups, void* stream) {
  if (!input || !offset || !weight || !output || batch <= 0 || c_in <= 0 || c_out <= 0 || groups <= 0 ||
      deformable_groups <= 0 || c_in % groups || c_out % groups || c_in % deformable_groups)
    return ORP_EINVAL;
  const int Ho = out_dim(height, pad_h, dil_h, kh, stride_h), Wo = out_dim(width, pad_w, dil_w, kw, stride_w);
  if (Ho <= 0 || Wo <= 0) return ORP_EINVAL;
  const long total = (long)batch * c_out * Ho * Wo;
  long blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(dcn_fwd_direct_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, input, offset, mask,
                     weight, bias, output, batch, c_in, height, width, c_out, Ho, Wo, kh, kw, stride_h, stride_w, pad_h,
                     pad_w, dil_h, dil_w, groups, deformable_groups);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}

}  // extern "C"
