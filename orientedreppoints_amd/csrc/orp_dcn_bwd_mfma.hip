// orp_dcn_bwd_mfma.hip -- deformable convolution (DCNv1) backward for gfx950 as two MFMA implicit GEMMs.
//
// Replaces deform_conv_backward_input_cuda + deform_conv_backward_parameters_cuda (mmdet/ops/dcn/src/deform_conv_cuda.cpp:
// 262-488) and their kernels deformable_im2col / deformable_col2im / deformable_col2im_coord
// (deform_conv_cuda_kernel.cu:190-465) for the head's configuration (Cin = Cout = 256, groups = deformable_groups = 1).
// The reference (and orp_dcn_bwd.hip's column formulation) writes the [Cin*9, B*Ho*Wo] column buffers to HBM twice
// (grad columns = W^T . grad_out, and im2col for grad_W) around two library GEMMs; here neither buffer exists:
//
//   kernel A  (grad_input, grad_offset)   per tile of MT*32 positions and per kernel tap t:
//       G_t[p, c] = sum_o  go[p, o] * W[o, c, t]                      (v_mfma_f32_32x32x2_f32, K = 256 output channels)
//     the grad_out tile stays in LDS for all nine taps, wave w owns input channels [32w, 32w+32) and streams its W^T
//     fragments L2 -> registers; the accumulator of a tap is consumed in place: the two coordinate derivatives
//     (get_coordinate_weight, deform_conv_cuda_kernel.cu:145-188), reduced over the 32 lanes with DPP and over the 8 waves
//     in a fixed order through LDS, and the row G_t[p, :] is stored (1 KB, coalesced) for kernel A2.
//   kernel A2 (grad_input, round 3: NO atomics)   one workgroup owns an 8 x 8-pixel region of grad_input for all 256
//     channels as a 64 KB LDS accumulator.  A pre-pass files every (position, tap) sample under the <= 4 regions its
//     bilinear footprint touches (fixed slots + a STABLE device radix sort by region: every region's list is in ascending
//     sample order); the region's workgroup (thread = channel, so no two threads ever touch the same accumulator) walks
//     its list, adds  weight x G  for the corners inside the region with plain LDS read-modify-writes, and writes its 64
//     pixels once, coalesced.  Every pixel is written by exactly one workgroup (no zero fill, no atomics), the summation
//     order is fixed: bitwise reproducible.  Round 2 scattered with 4 x 9 x 256 fp32 atomics per position: 402 M
//     lane-atomics per launch left the L2 as 1.555 GB of single transactions (profiles/r02_pmc.json) for 45 MB of
//     gradient, 1.43 ms; that path is still selectable (ORP_DCN_BWD_ATOMIC=1, dev aid) for comparison.
//   kernel B  (grad_weight)   workgroup (split s, tap t):
//       gW_t[o, c] = sum_p  go[p, o] * col_t[p, c],    col_t[p, c] = bilinear(x[:, c], p + t + offset)
//     K = positions, walked in 32-position chunks: the col tile is gathered exactly like the forward's A tile (coalesced
//     1 KB NHWC rows, wave-uniform weights) into LDS next to the grad_out rows; 64 accumulator tiles (256 x 256) live in
//     the 8 waves' registers.  Splits write partial sums, a second kernel adds them in FIXED order (deterministic).
//
// Inputs arrive NCHW (autograd); x and grad_out are transposed to NHWC in the workspace, grad_input is accumulated NHWC
// and transposed back.
//
// Row sparsity: a detection head's regression branch only receives gradient at its positive points (a few hundred of the
// 43 648 positions of two 1024^2 images), so most grad_out rows are exactly zero.  The transposition pass records which
// 32-position chunks hold any non-zero value, an ordered compaction turns the flags into the list of ACTIVE chunks, and
// both kernels walk that list only (dense gradients: the list is the identity).  Deterministic: the list is in chunk order.
//
// Measured and rejected for kernel A (round 2, tests/checks/atomic_rate.hip and the ORP_BWD_DBG switches): pre-accumulating
// grad_input in LDS rows found through a per-tile hash table.  An fp32 global atomic costs one L2-channel clock per LANE
// (313 G lane-atomics/s, agent and workgroup scope alike) -- but ds_add_f32 ran at the same ~310 G lanes/s on this part,
// and the 112 KB of rows cut the occupancy from three workgroups per CU to one: 2.56 ms vs 1.40 ms without.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/orp_hip.h"
#include "orp_launch.hpp"
#include "orp_prof.hpp"

#ifndef ORP_BWD_SPLITACC
#define ORP_BWD_SPLITACC 0   // kernel A: even / odd k-steps into two independent accumulator tiles (measured: 792 vs 791 us, +16 VGPRs: off)
#endif
#ifndef ORP_BWD_WDIST
#define ORP_BWD_WDIST 3      // kernel A: weight fragments fetched this many chunks ahead (ring of WDIST + 1 slots; 1, 3 or 7)
#endif
#ifndef ORP_BWD_WDIST16
#define ORP_BWD_WDIST16 3    // ... of the fp16-pieces contraction (a chunk is 3 MFMAs = 96 cycles there; the other waves of the SIMD cover the rest)
#endif
#ifndef ORP_BWD_LANEPOS
#define ORP_BWD_LANEPOS 1    // kernel A, dense path: accumulator as D[channel][position] (lane = position) and the in-lane derivative sums; 0 = the round-3 lane = channel epilogue
#endif
#ifndef ORP_BWD_DBG
#define ORP_BWD_DBG 0      // dev aid, compile-time (timing only, wrong results): 1 = no grad_input atomics, 2 = no x loads / derivative reduction, 4 = no G store, 8 = no epilogue at all, 16 = scatter kernel without G row reads, 32 = scatter kernel without accumulator updates
#endif

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));     // one operand of v_mfma_f32_32x32x16_f16: 8 k-values
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int CH = 256;          // Cin = Cout on this path
constexpr int MAXL = 8;
constexpr int MAXT = 9;
constexpr int ASTR = CH + 4;     // kernel A: grad_out tile row stride (conflict-free ds_read_b128 of the permuted K order)
constexpr int ASTRH = CH + 8;    // kernel A, fp16 pieces: row stride of a plane in halves (132 dwords: conflict-free ds_read_b128 over 16 rows)
constexpr int RS = CH + 32;      // kernel B: row stride with RS % 64 == 32 (two half-waves read rows k, k+1 conflict-free)
constexpr int kThreads = 512;

struct BLevel {
  const float* x;      // NHWC [B, H, W, 256]
  const float* go;     // NHWC [B, Ho, Wo, 256]
  const float* off;    // NCHW [B, 2*taps, Ho, Wo]
  float* gx;           // NHWC [B, H, W, 256], zeroed
  float* goff;         // NCHW [B, 2*taps, Ho, Wo]
  const float* mask;   // DCNv2: NCHW [B, taps, Ho, Wo] modulation, or nullptr (DCNv1)
  float* gmask;        // DCNv2: its gradient
  int H, W, Ho, Wo;
  int tile0;           // kernel A: first tile of this level
  int chunk0;          // kernel B: first 32-position chunk of this level
  int reg0, RH, RW;    // kernel A2: first 8 x 8-pixel region of this level, regions per image column / row
};
struct BwdParams {
  BLevel lv[MAXL];
  int nlev, B;
  int kh, kw, sh, sw, ph, pw, dh, dw;
  const float* wT;     // [tap][o/4][c][4]
  // kernel A on the 16-bit matrix pipe (round 5): W as two fp16 planes [plane][tap][o/16][o%16/8][c][8] of w * wscale[0] (a power
  // of two), go_amax[0] = bits of max |grad_out| over the launch (transpose_set_kernel); nullptr: the exact-fp32 contraction
  const uint16_t* wT16;
  size_t w16_plane;    // elements per plane
  const float* wscale;
  const unsigned* go_amax;
  float* partial;      // [nsplit][tap][o][c]
  int nsplit, total_chunks;
  const int* active;   // [total_chunks] chunk indices with a non-zero grad_out row, ascending; active[total_chunks] = count
  float* G;            // kernel A -> A2: G[(position * taps + tap)][256], positions numbered chunk0 * 32 + p (NULL: atomics)
  const int* flags;    // [total_chunks] chunk holds a non-zero grad_out row
  int nregions;
  unsigned* keys;      // [4 * total_chunks * 32 * taps] region of slot (sample, k); nregions = unused slot
  unsigned* vals;      //   the sample index of the slot
  int* rcount;         // [nregions + 1] first sorted slot of region r (region_bounds_kernel)
  const unsigned* sorted_vals;
};

inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
inline int out_dim(int in, int pad, int dil, int k, int stride) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }

// ---- batched [B][R][S] -> [B][S][R] for up to 16 tensors in one launch -------------------------------------------
struct TransposeSet {
  const void* in[2 * MAXL];
  void* out[2 * MAXL];
  int in_code, out_code;         // element type of the inputs / outputs: 0 fp32, 1 fp16, 2 bf16 (the other side is fp32)
  int R[2 * MAXL], S[2 * MAXL];
  int t0[2 * MAXL + 1];
  int chunk0[2 * MAXL];          // >= 0: tensor i is a grad_out [256][HoWo]; flags[chunk0 + (b*S + s) / 32] = 1 where non-zero
  int n;
  int* flags;
  unsigned* amax;                // (or nullptr) [1] max |v| over the grad_out tensors as float bits, raised with atomicMax (zeroed by the caller)
  unsigned* amax_x;              // (or nullptr) the same over the tensors that are not a grad_out (the inputs x; kernel B on the 16-bit pipe)
};
__global__ void transpose_set_kernel(const TransposeSet T) {
  __shared__ float tile[32][33];
  int i = 0;
#pragma unroll
  for (int k = 1; k < 2 * MAXL; k++) i = (k < T.n && (int)blockIdx.x >= T.t0[k]) ? k : i;
  const int R = T.R[i], S = T.S[i];
  const int ts_n = (S + 31) >> 5;
  const int t = (int)blockIdx.x - T.t0[i];
  const int r0 = (t / ts_n) * 32, s0 = (t % ts_n) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t plane = (size_t)blockIdx.y * R * S;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, s = s0 + tx;
    float v = 0.f;
    if (r < R && s < S) {
      const size_t at = plane + (size_t)r * S + s;
      v = T.in_code == 0 ? reinterpret_cast<const float*>(T.in[i])[at]
        : T.in_code == 1 ? (float)reinterpret_cast<const _Float16*>(T.in[i])[at]
                         : (float)reinterpret_cast<const __bf16*>(T.in[i])[at];
    }
    tile[k][tx] = v;
  }
  __syncthreads();
  const int c0 = T.chunk0[i];
  unsigned vmax = 0u;
  for (int k = ty; k < 32; k += 8) {
    const int s = s0 + k, r = r0 + tx;
    const float v = tile[tx][k];
    vmax = max(vmax, __float_as_uint(v) & 0x7fffffffu);
    if (s < S && r < R) {
      const size_t at = plane + (size_t)s * R + r;
      if (T.out_code == 0) reinterpret_cast<float*>(T.out[i])[at] = v;
      else if (T.out_code == 1) reinterpret_cast<_Float16*>(T.out[i])[at] = (_Float16)v;
      else reinterpret_cast<__bf16*>(T.out[i])[at] = (__bf16)v;
    }
    if (c0 >= 0) {                                                  // lanes 0-31 / 32-63 of a wave = 32 channels of ONE position
      const unsigned long long nz = __ballot(v != 0.f);
      const bool mine = (threadIdx.x & 32) ? (nz >> 32) != 0 : (nz & 0xffffffffull) != 0;
      if (mine && tx == 0 && s < S) T.flags[c0 + (int)(((long)blockIdx.y * S + s) >> 5)] = 1;
    }
  }
  if (c0 >= 0 && T.amax) {                                          // (block-uniform) range of grad_out for the fp16-pieces contraction
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = max(vmax, (unsigned)__shfl_xor((int)vmax, o, 64));
    if ((threadIdx.x & 63) == 0 && vmax > __atomic_load_n(T.amax, __ATOMIC_RELAXED)) atomicMax(T.amax, vmax);
  }
  if (c0 < 0 && T.amax_x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = max(vmax, (unsigned)__shfl_xor((int)vmax, o, 64));
    if ((threadIdx.x & 63) == 0 && vmax > __atomic_load_n(T.amax_x, __ATOMIC_RELAXED)) atomicMax(T.amax_x, vmax);
  }
}

// flags[n] -> ascending list of the set indices, list[n] = count (one workgroup; n is a few thousand)
__global__ void __launch_bounds__(1024) compact_flags_kernel(const int* __restrict__ flags, int n, int* __restrict__ list) {
  __shared__ int wsum[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const bool f = i < n && flags[i] != 0;
    const unsigned long long m = __ballot(f);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; w++) off += wsum[w];
    if (f) list[off + before] = i;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < 16; w++) t += wsum[w]; base += t; }
    __syncthreads();
  }
  if (tid == 0) list[n] = base;
}

// w [o][c][tap] -> wT [tap][o/4][c][4]
__global__ void pack_wT_kernel(const float* __restrict__ w, int taps, float* __restrict__ wT) {
  const int total = CH * CH * taps;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int q = i & 3, c = (i >> 2) % CH, og = ((i >> 2) / CH) % (CH / 4), tap = (i >> 2) / (CH * (CH / 4));
    wT[i] = w[((size_t)(og * 4 + q) * CH + c) * taps + tap];
  }
}

// max |w| as float bits (out zeroed by the caller)
__global__ void __launch_bounds__(256) absmax_w_kernel(const float* __restrict__ w, int n, unsigned* __restrict__ out) {
  unsigned m = 0u;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, __float_as_uint(w[i]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, m);
}
// the power of two that puts a tensor's largest magnitude (float bits `am`) into [2^14, 2^15): fp16 pieces then neither overflow
// nor lose their low piece to the subnormal range (same rule as csrc/orp_dcn_split.hip)
__device__ __forceinline__ float range_scale(unsigned am) {
  int k = am == 0u ? 0 : 14 - ((int)((am >> 23) & 0xffu) - 127);
  k = k < -100 ? -100 : k > 100 ? 100 : k;
  return __uint_as_float((unsigned)(127 + k) << 23);
}
// w [o][c][tap] -> two fp16 planes [pl][tap][o/16][kh][c][8] of w * 2^k (hi = nearest fp16, lo = the residual: exact in fp32, then rounded):
// lane (c, kh) of kernel A reads the 8 k-values (output channels) of its MFMA operand as one 16-byte load
__global__ void pack_wT16_kernel(const float* __restrict__ w, int taps, const unsigned* __restrict__ amax, uint16_t* __restrict__ planes,
                                 float* __restrict__ wscale) {
  const float sc = range_scale(*amax);
  if (blockIdx.x == 0 && threadIdx.x == 0) wscale[0] = sc;
  const int total = CH * CH * taps;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int e = i & 7, c = (i >> 3) % CH;
    int r = (i >> 3) / CH;
    const int khh = r & 1; r >>= 1;
    const int ob = r % (CH / 16), tap = r / (CH / 16);
    const int o = ob * 16 + khh * 8 + e;
    const float v = w[((size_t)o * CH + c) * taps + tap] * sc;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    planes[i] = __builtin_bit_cast(uint16_t, hi);
    planes[total + i] = __builtin_bit_cast(uint16_t, lo);
  }
}

// sum_s partial[s][tap][o][c] -> gw[o][c][tap], s in ascending order
__global__ void reduce_partial_kernel(const float* __restrict__ partial, int nsplit, int taps, float* __restrict__ gw) {
  const int total = taps * CH * CH;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int s = 0; s < nsplit; s++) acc += partial[(size_t)s * total + i];
    const int c = i % CH, o = (i / CH) % CH, tap = i / (CH * CH);
    gw[((size_t)o * CH + c) * taps + tap] = acc;
  }
}

// the sampling point of (position p of level L, kernel tap): fractional parts and the four pixel indices, -1 = that
// corner contributes nothing (outside the map); everything -1 when the point is outside (-1, H) x (-1, W)
__device__ inline void sample_point(const BwdParams& P, const BLevel& L, long p, int tap, int taps, int HoWo, int4& ix,
                                    float2& frac) {
  ix = make_int4(-1, -1, -1, -1);
  frac = make_float2(0.f, 0.f);
  const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
  const int ho = hw / L.Wo, wo = hw - ho * L.Wo;
  const int ki = tap / P.kw, kj = tap - ki * P.kw;
  const float* ob = L.off + ((size_t)b * 2 * taps + 2 * tap) * HoWo + hw;
  const float h_im = (float)(ho * P.sh - P.ph + ki * P.dh) + ob[0];
  const float w_im = (float)(wo * P.sw - P.pw + kj * P.dw) + ob[HoWo];
  if (h_im > -1.f && w_im > -1.f && h_im < (float)L.H && w_im < (float)L.W) {
    const int hl = (int)floorf(h_im), wl = (int)floorf(w_im), hh = hl + 1, wh = wl + 1;
    frac.x = h_im - (float)hl; frac.y = w_im - (float)wl;
    const bool t_ok = hl >= 0, b_ok = hh <= L.H - 1, l_ok = wl >= 0, r_ok = wh <= L.W - 1;
    const int base = b * L.H;
    if (t_ok && l_ok) ix.x = (base + hl) * L.W + wl;
    if (t_ok && r_ok) ix.y = (base + hl) * L.W + wh;
    if (b_ok && l_ok) ix.z = (base + hh) * L.W + wl;
    if (b_ok && r_ok) ix.w = (base + hh) * L.W + wh;
  }
}

// DCNv2 modulation scalar of (position p of level L, tap); 1 for DCNv1
__device__ inline float sample_mask(const BLevel& L, long p, int tap, int taps, int HoWo) {
  if (!L.mask) return 1.f;
  const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
  return L.mask[((size_t)b * taps + tap) * HoWo + hw];
}

template <int CTRL, int ROW_MASK>
__device__ inline float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 32 lanes of each half-wave; valid in lanes 16..31 and 48..63
__device__ inline float half_wave_sum(float v) {
  v = dpp_add<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);    // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);    // row_mirror
  v = dpp_add<0x142, 0xa>(v);    // row_bcast15 into rows 1 and 3
  return v;
}

// ---- kernel A: grad_input + grad_offset ---------------------------------------------------------------------------
// One tile = one 32-position chunk (MT = 1: two to three workgroups per CU hide the epilogue's load / atomic latency; two
// sub-tiles per workgroup measured 5 % slower in round 2, and again in round 3 for the G-row variant: 64 positions per
// workgroup halve the L2 -> CU weight stream (3.2 GB per launch) but leave one workgroup per CU (140 KB of LDS); even with
// the x values of a tap's coordinate derivatives loaded one group of rows ahead of their use: 1 356 vs 1 231 us for
// grad_input + grad_offset at 2 x 21 824 positions).  Also measured on the one-chunk kernel, dense path: the epilogue's x
// loads issued one group of four rows ahead of their use (1 203 - 1 215 vs 1 217 - 1 219 us: nothing; +40 VGPRs).  PMC:
// ~1 400 non-MFMA VALU instructions per wave and tap next to 128 MFMAs -- the per-row address / select / DPP work of the
// epilogue, replicated over the 64 lanes, is what the matrix pipe waits for.
// Round 5, F16: the contraction on the 16-bit matrix pipe, as the forward's (csrc/orp_dcn_split.hip): grad_out and W each as TWO fp16
// pieces after an exact power-of-two range scaling (|v - (hi + lo)| <= 2^-22 |v|), products lo*hi, hi*lo into a side accumulator and
// hi*hi into the main one, fp32 accumulation, scaled back once per tap.  48 MFMAs of 32 cycles per tap and wave instead of 128 of 64.
template <int MT>
constexpr size_t input_tile_bytes() {
  return sizeof(float) * 32 * MT * ASTR > (size_t)2 * 2 * 32 * MT * ASTRH ? sizeof(float) * 32 * MT * ASTR : (size_t)2 * 2 * 32 * MT * ASTRH;
}
template <int MT, bool STORE_G, bool F16>
__global__ void __launch_bounds__(kThreads)
dcn_bwd_input_kernel(const BwdParams P) {
  constexpr int BM2 = 32 * MT;
  static_assert(MT == 1, "the active-chunk list is in units of 32 positions");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sG = reinterpret_cast<float*>(smem);                       // [BM2][ASTR] grad_out rows  (F16: two planes [BM2][ASTRH] of halves)
  uint16_t* sGh = reinterpret_cast<uint16_t*>(smem);
  int4* sCi = reinterpret_cast<int4*>(smem + input_tile_bytes<MT>());   // [BM2 * taps]
  float4* sCl = reinterpret_cast<float4*>(sCi + BM2 * MAXT);        // [BM2 * taps] (lh, lw, modulation, -)
  float* sGO = reinterpret_cast<float*>(sCl + BM2 * MAXT);          // [8 waves][BM2][taps][3] grad_offset (+ grad_mask) partials
  int* sNZ = reinterpret_cast<int*>(sGO + 8 * BM2 * MAXT * 3);      // [BM2] row has a non-zero grad_out value

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int taps = P.kh * P.kw;
  int tile;
  {
    const int n_active = P.active[P.total_chunks];
    const int b = blockIdx.x, per = (n_active + 7) >> 3;            // XCD x takes the contiguous active tiles [x*per, (x+1)*per)
    const int idx = (b & 7) * per + (b >> 3);
    if ((b >> 3) >= per || idx >= n_active) return;
    tile = P.active[idx];
  }
  int lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if (tile >= P.lv[i].tile0) lvl = i;
  const BLevel L = P.lv[lvl];
  const int HoWo = L.Ho * L.Wo;
  const long npos = (long)P.B * HoWo;
  const long p0 = (long)(tile - L.tile0) * BM2;

  for (int e = tid; e < BM2 * taps; e += kThreads) {
    const int m = e / taps, tap = e - m * taps;
    int4 ix = make_int4(-1, -1, -1, -1);
    float2 fr = make_float2(0.f, 0.f);
    float mm = 1.f;
    if (p0 + m < npos) { sample_point(P, L, p0 + m, tap, taps, HoWo, ix, fr); mm = sample_mask(L, p0 + m, tap, taps, HoWo); }
    sCi[e] = ix; sCl[e] = make_float4(fr.x, fr.y, mm, 0.f);
  }
  for (int e = tid; e < 8 * BM2 * MAXT * 3; e += kThreads) sGO[e] = 0.f;
  float sx = 1.f, osc = 1.f;                                        // F16: grad_out scale 2^k, accumulator scale 1 / (sx * wscale)
  if (F16) { sx = range_scale(P.go_amax[0]); osc = 1.f / (sx * P.wscale[0]); }
  for (int r = wave; r < BM2; r += 8) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p0 + r < npos) v = *reinterpret_cast<const float4*>(L.go + (size_t)(p0 + r) * CH + lane * 4);
    if (F16) {
      const float sv[4] = {v.x * sx, v.y * sx, v.z * sx, v.w * sx};   // exact (power of two)
      _Float16 h[4], l[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { h[i] = (_Float16)sv[i]; l[i] = (_Float16)(sv[i] - (float)h[i]); }
      const h2 h01 = {h[0], h[1]}, h23 = {h[2], h[3]}, l01 = {l[0], l[1]}, l23 = {l[2], l[3]};
      uint16_t* dst = sGh + (size_t)r * ASTRH + lane * 4;
      *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
      *reinterpret_cast<uint2*>(dst + BM2 * ASTRH) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
    } else {
      *reinterpret_cast<float4*>(sG + (size_t)r * ASTR + lane * 4) = v;
    }
    const unsigned long long nz = __ballot((v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f));
    if (lane == 0) sNZ[r] = nz != 0;
  }
  __syncthreads();

  const int mrow = lane & 31, kh = lane >> 5;
  const int c = wave * 32 + mrow;                                   // this lane's input channel
  auto load_bq = [&](int tap, int j, float4 (&r)[2]) {
    if (F16) {                                                       // r[0] / r[1] = the lane's 8 k-values of the hi / lo plane
      const uint16_t* base = P.wT16 + ((((size_t)tap * (CH / 16) + j) * 2 + kh) * CH + c) * 8;
      r[0] = *reinterpret_cast<const float4*>(base);
      r[1] = *reinterpret_cast<const float4*>(base + P.w16_plane);
      return;
    }
    const float* base = P.wT + (((size_t)tap * (CH / 4) + j * 4 + kh) * CH + c) * 4;
    r[0] = *reinterpret_cast<const float4*>(base);
    r[1] = *reinterpret_cast<const float4*>(base + (size_t)2 * CH * 4);
  };
  // the weight fragments of chunk (tap, j) are fetched WD chunks ahead of their use: one chunk is 8 MFMAs = 512 cycles
  // of matrix work per wave, an L2 hit takes longer than that (measured: 1 221 vs 1 245 us at WD = 3 vs 1; 7 costs occupancy)
  constexpr int WD = F16 ? ORP_BWD_WDIST16 : ORP_BWD_WDIST;
  float4 bqr[WD + 1][2];
#pragma unroll
  for (int d = 0; d < WD; d++) load_bq(d / (CH / 16), d % (CH / 16), bqr[d]);

#pragma unroll 1
  for (int tap = 0; tap < taps; tap++) {
    floatx16 acc[MT], side[MT];                                     // (side: F16 only -- the small partial products)
#pragma unroll
    for (int mt = 0; mt < MT; mt++) { acc[mt] = floatx16{0}; side[mt] = floatx16{0}; }
#if ORP_BWD_SPLITACC
    floatx16 acc_odd[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc_odd[mt] = floatx16{0};
#endif
#pragma unroll
    for (int j = 0; j < CH / 16; j++) {
      // (16 chunks per tap and WD + 1 ring slots: the slot of chunk (tap, j) is j % (WD + 1), a compile-time index because
      // the ring length divides 16)
      static_assert((CH / 16) % (WD + 1) == 0, "ring length must divide the chunks per tap");
      float4 (&bq)[2] = bqr[j % (WD + 1)];
      {
        const int jn = j + WD;                                       // the chunk WD ahead: this tap's or the next one's
        float4 (&bn)[2] = bqr[(j + WD) % (WD + 1)];
        if (jn < CH / 16) load_bq(tap, jn, bn);
        else if (tap + 1 < taps) load_bq(tap + 1, jn - CH / 16, bn);
      }
      if (F16) {
        const uint16_t* ar = sGh + (size_t)mrow * ASTRH + j * 16 + 8 * kh;
        const h8 w_hi = __builtin_bit_cast(h8, bq[0]), w_lo = __builtin_bit_cast(h8, bq[1]);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          const h8 g_hi = *reinterpret_cast<const h8*>(ar + (size_t)mt * 32 * ASTRH);
          const h8 g_lo = *reinterpret_cast<const h8*>(ar + (size_t)mt * 32 * ASTRH + BM2 * ASTRH);
          if (STORE_G && ORP_BWD_LANEPOS) {                            // D[channel][position]: lane = position
            side[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w_lo, g_hi, side[mt], 0, 0, 0);
            side[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w_hi, g_lo, side[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w_hi, g_hi, acc[mt], 0, 0, 0);
          } else {                                                     // D[position][channel]: lane = channel
            side[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(g_hi, w_lo, side[mt], 0, 0, 0);
            side[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(g_lo, w_hi, side[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(g_hi, w_hi, acc[mt], 0, 0, 0);
          }
        }
        continue;
      }
      const float* arow = sG + (size_t)mrow * ASTR + j * 16 + 4 * kh;
#pragma unroll
      for (int t = 0; t < 2; t++) {
        float4 a4[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) a4[mt] = *reinterpret_cast<const float4*>(arow + (size_t)mt * 32 * ASTR + 8 * t);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float b0 = (i == 0) ? bq[t].x : (i == 1) ? bq[t].y : (i == 2) ? bq[t].z : bq[t].w;
#pragma unroll
          for (int mt = 0; mt < MT; mt++) {
            const float av = (i == 0) ? a4[mt].x : (i == 1) ? a4[mt].y : (i == 2) ? a4[mt].z : a4[mt].w;
#if ORP_BWD_SPLITACC
            if (i & 1) acc_odd[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc_odd[mt], 0, 0, 0);
            else
#endif
            if (STORE_G && ORP_BWD_LANEPOS)
              acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, av, acc[mt], 0, 0, 0);     // D[channel][position]: lane = position
            else
              acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[mt], 0, 0, 0);     // D[position][channel]: lane = channel
          }
        }
      }
    }
#if ORP_BWD_SPLITACC
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] += acc_odd[mt];
#endif
    if (F16) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++) acc[mt] = (acc[mt] + side[mt]) * osc;
    }
    // ---- consume G_t: scatter into grad_input, coordinate derivatives into the tile's grad_offset ---------------
    if (STORE_G && ORP_BWD_LANEPOS) {
      // Dense gradients, round 4: the accumulator holds D[channel][position] -- lane = position m (lane & 31), register r =
      // channel n_wave + (r & 3) + 8 (r >> 2) + 4 kh.  The sums over channels the coordinate derivatives need
      //     S_k = sum_c G[m, c] * x[corner_k, c],  k = 1 .. 4
      // are then IN-LANE multiply-adds over the 16 registers (x arrives as four 16-byte loads per corner: the lane's four
      // channel quadruples of its wave's 128-byte row piece), one cross-half exchange adds the other 16 channels, and the
      // bilinear factors are applied ONCE to the four sums:
      //     dh = mm (uw (S3 - S1) + lw (S4 - S2)),  dw = mm (uh (S2 - S1) + lh (S4 - S3)),  dm = uh uw S1 + uh lw S2 + lh uw S3 + lh lw S4
      // (get_coordinate_weight, deform_conv_cuda_kernel.cu:145-188, regrouped).  Before: lane = channel, every one of the 16
      // rows did its own address arithmetic, validity selects, expanded formulas and two to three 5-step DPP reductions,
      // replicated over 64 lanes: ~1 400 VALU instructions per wave and tap beside 128 MFMAs (profiles/r03_pmc.json).
      const int m = mrow;                                              // this lane's position of the tile
      const int e = m * taps + tap;
      const bool row_ok = p0 + m < npos;
      // (a) the row G_t[m, :] for kernel A2: 16 channels per lane as four 16-byte stores (the two half-waves interleave
      //     into whole 32-byte sectors; the row's 1 KB is completed by the 8 waves)
      if (!(ORP_BWD_DBG & 4) && row_ok) {
        float* grow = P.G + ((size_t)((long)tile * BM2 + m) * taps + tap) * CH + wave * 32 + 4 * kh;
#pragma unroll
        for (int q = 0; q < 4; q++)
          *reinterpret_cast<float4*>(grow + 8 * q) = make_float4(acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]);
      }
      // (b) the derivative sums
      const bool live = !(ORP_BWD_DBG & 8) && sNZ[m] != 0;             // a zero grad_out row gives G = 0: nothing to add
      float S[4] = {0.f, 0.f, 0.f, 0.f};
      if (__ballot(live) != 0) {
        const int4 ix = sCi[e];
        const int ixs[4] = {ix.x, ix.y, ix.z, ix.w};
        const float* xb = L.x + wave * 32 + 4 * kh;
#if !(ORP_BWD_DBG & 2)
        float4 v[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float* xr = xb + (size_t)(live && ixs[k] >= 0 ? ixs[k] : 0) * CH;   // an invalid corner reads pixel 0 and is dropped below
#pragma unroll
          for (int q = 0; q < 4; q++) v[k][q] = *reinterpret_cast<const float4*>(xr + 8 * q);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float sk = 0.f;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            sk = __builtin_fmaf(acc[0][4 * q], v[k][q].x, sk);
            sk = __builtin_fmaf(acc[0][4 * q + 1], v[k][q].y, sk);
            sk = __builtin_fmaf(acc[0][4 * q + 2], v[k][q].z, sk);
            sk = __builtin_fmaf(acc[0][4 * q + 3], v[k][q].w, sk);
          }
          S[k] = (live && ixs[k] >= 0) ? sk : 0.f;
        }
#endif
        // the other half-wave holds the other 16 channels of the same position: lanes m and m + 32
#pragma unroll
        for (int k = 0; k < 4; k++) S[k] += __shfl_xor(S[k], 32, 64);
        if (kh == 0) {
          const float4 fr = sCl[e];
          const float lh = fr.x, lw = fr.y, uh = 1.f - lh, uw = 1.f - lw, mm = fr.z;
          float* slot = sGO + (size_t)(wave * BM2 * MAXT + e) * 3;
          slot[0] = mm * (uw * (S[2] - S[0]) + lw * (S[3] - S[1]));
          slot[1] = mm * (uh * (S[1] - S[0]) + lh * (S[3] - S[2]));
          slot[2] = L.gmask ? (uh * uw * S[0] + uh * lw * S[1] + lh * uw * S[2] + lh * lw * S[3]) : 0.f;
        }
      }
      continue;
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll 4
      for (int r = 0; r < 16; r++) {
        const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int e = m * taps + tap;
        const bool live = (ORP_BWD_DBG & 8) ? false : (sNZ[m] != 0);   // a zero grad_out row gives G = 0: nothing to add
        if (STORE_G && !(ORP_BWD_DBG & 4)) {                        // the row of (position, tap) for kernel A2 (zeros included)
          if (p0 + m < npos) P.G[((size_t)((long)tile * BM2 + m) * taps + tap) * CH + c] = acc[mt][r];
        }
        if (__ballot(live) == 0) continue;
        const int4 ix = live ? sCi[e] : make_int4(-1, -1, -1, -1);
        const float4 fr = sCl[e];
        const float g = acc[mt][r];
        const float lh = fr.x, lw = fr.y, uh = 1.f - lh, uw = 1.f - lw, mm = fr.z;
        const size_t o1 = (size_t)(ix.x < 0 ? 0 : ix.x) * CH + c, o2 = (size_t)(ix.y < 0 ? 0 : ix.y) * CH + c;
        const size_t o3 = (size_t)(ix.z < 0 ? 0 : ix.z) * CH + c, o4 = (size_t)(ix.w < 0 ? 0 : ix.w) * CH + c;
#if ORP_BWD_DBG & 2
        const float v1 = uh, v2 = uw, v3 = lh, v4 = lw;
#else
        const float v1 = ix.x >= 0 ? L.x[o1] : 0.f, v2 = ix.y >= 0 ? L.x[o2] : 0.f;
        const float v3 = ix.z >= 0 ? L.x[o3] : 0.f, v4 = ix.w >= 0 ? L.x[o4] : 0.f;
#endif
#if !(ORP_BWD_DBG & 1)
        if (!STORE_G) {
          const float gm = g * mm;
          if (ix.x >= 0) atomicAdd(L.gx + o1, uh * uw * gm);
          if (ix.y >= 0) atomicAdd(L.gx + o2, uh * lw * gm);
          if (ix.z >= 0) atomicAdd(L.gx + o3, lh * uw * gm);
          if (ix.w >= 0) atomicAdd(L.gx + o4, lh * lw * gm);
        }
#endif
        float dh = g * mm * (-uw * v1 - lw * v2 + uw * v3 + lw * v4);
        float dw = g * mm * (-uh * v1 + uh * v2 - lh * v3 + lh * v4);
        dh = half_wave_sum(dh);
        dw = half_wave_sum(dw);
        float dm = 0.f;
        if (L.gmask) {                                               // DCNv2: d loss / d modulation = G . sampled value
          dm = g * (uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4);
          dm = half_wave_sum(dm);
        }
        // this wave's 32-channel partial of (position, tap): one writer per slot, summed over the waves in order below
        if (mrow == 31) {
          float* slot = sGO + (size_t)(wave * BM2 * MAXT + e) * 3;
          slot[0] = dh; slot[1] = dw; slot[2] = dm;
        }
      }
    }
  }
  __syncthreads();
  // grad_offset [B, 2*taps, Ho, Wo]: plane-major so that consecutive lanes write consecutive positions
  for (int e2 = tid; e2 < BM2 * taps * 2; e2 += kThreads) {
    const int plane = e2 / BM2, m = e2 - plane * BM2;
    const long p = p0 + m;
    if (p < npos) {
      const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
      const int tap = plane >> 1, comp = plane & 1;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; w++) v += sGO[(size_t)(w * BM2 * MAXT + m * taps + tap) * 3 + comp];   // fixed order: reproducible
      L.goff[((size_t)b * 2 * taps + plane) * HoWo + hw] = v;
    }
  }
  if (L.gmask) {
    for (int e2 = tid; e2 < BM2 * taps; e2 += kThreads) {
      const int tap = e2 / BM2, m = e2 - tap * BM2;
      const long p = p0 + m;
      if (p < npos) {
        const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; w++) v += sGO[(size_t)(w * BM2 * MAXT + m * taps + tap) * 3 + 2];
        L.gmask[((size_t)b * taps + tap) * HoWo + hw] = v;
      }
    }
  }
}

template <int MT>
size_t input_smem() {
  return input_tile_bytes<MT>() + (sizeof(int4) + sizeof(float4) + 8 * 3 * sizeof(float)) * 32 * MT * MAXT + sizeof(int) * 32 * MT;
}

// ---- kernel A2: grad_input without atomics ------------------------------------------------------------------------------
// region of the pixel with linear index q = (b * H + h) * W + w of level L
__device__ inline int region_of_pixel(const BLevel& L, int q) {
  const int w = q % L.W, bh = q / L.W;
  const int h = bh % L.H, b = bh / L.H;
  return L.reg0 + (b * L.RH + (h >> 3)) * L.RW + (w >> 3);
}
__device__ inline int level_of_chunk(const BwdParams& P, int chunk) {
  int lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if (chunk >= P.lv[i].chunk0) lvl = i;
  return lvl;
}

// one thread per (position, tap) sample e = (chunk * 32 + m) * taps + tap: slots 4e .. 4e+3 receive the distinct regions
// the sample's four corners fall into (unused slots: key = nregions, sorted to the end)
__global__ void bin_samples_kernel(const BwdParams P) {
  const int taps = P.kh * P.kw;
  const long E = (long)P.total_chunks * 32 * taps;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const long pg = e / taps;
  const int tap = (int)(e - pg * taps);
  const int chunk = (int)(pg >> 5);
  unsigned k[4] = {(unsigned)P.nregions, (unsigned)P.nregions, (unsigned)P.nregions, (unsigned)P.nregions};
  const BLevel& L = P.lv[level_of_chunk(P, chunk)];
  const int HoWo = L.Ho * L.Wo;
  const long p = pg - (long)L.chunk0 * 32;
  if (P.flags[chunk] == 0) {
    // kernel A skips this chunk: its grad_offset is zero (written here instead of one memset per level)
    if (p < (long)P.B * HoWo) {
      const int b = (int)(p / HoWo), hw = (int)(p - (long)b * HoWo);
      float* o = L.goff + ((size_t)b * 2 * taps + 2 * tap) * HoWo + hw;
      o[0] = 0.f; o[HoWo] = 0.f;
      if (L.gmask) L.gmask[((size_t)b * taps + tap) * HoWo + hw] = 0.f;
    }
  } else {
    if (p < (long)P.B * HoWo) {
      int4 ix; float2 fr;
      sample_point(P, L, p, tap, taps, HoWo, ix, fr);
      const int r0 = ix.x >= 0 ? region_of_pixel(L, ix.x) : -1, r1 = ix.y >= 0 ? region_of_pixel(L, ix.y) : -1;
      const int r2 = ix.z >= 0 ? region_of_pixel(L, ix.z) : -1, r3 = ix.w >= 0 ? region_of_pixel(L, ix.w) : -1;
      int n = 0;
      if (r0 >= 0) k[n++] = (unsigned)r0;
      if (r1 >= 0 && r1 != r0) k[n++] = (unsigned)r1;
      if (r2 >= 0 && r2 != r0 && r2 != r1) k[n++] = (unsigned)r2;
      if (r3 >= 0 && r3 != r0 && r3 != r1 && r3 != r2) k[n++] = (unsigned)r3;
    }
  }
  *reinterpret_cast<uint4*>(P.keys + 4 * e) = make_uint4(k[0], k[1], k[2], k[3]);
  *reinterpret_cast<uint4*>(P.vals + 4 * e) = make_uint4((unsigned)e, (unsigned)e, (unsigned)e, (unsigned)e);
}

// first[r] = index of the first sorted slot with key >= r, for r in [0, nregions]: region r's list is
// [first[r], first[r + 1]).  Every boundary between two different keys fills the entries in between (empty regions too).
__global__ void region_bounds_kernel(const unsigned* __restrict__ keys_sorted, long nslots, int nregions, int* __restrict__ first) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nslots) return;
  const long k = (long)keys_sorted[i];
  const long kprev = i > 0 ? (long)keys_sorted[i - 1] : -1;
  for (long r = kprev + 1; r <= k && r <= nregions; r++) first[r] = (int)i;
  if (i == nslots - 1) for (long r = k + 1; r <= nregions; r++) first[r] = (int)nslots;
}

// One workgroup (256 threads, thread = channel) per 8 x 8-pixel region.  LDS: acc[64][256] | sample meta of 64 list entries.
constexpr int kScatterThreads = 256;
constexpr int kAccRows = 65;                                         // 64 pixels + one dummy row for corners of other regions
constexpr size_t scatter_smem() { return sizeof(float) * kAccRows * CH; }
// One descriptor per list entry, built once per call right after the sort: the G row and, per corner, the accumulator row
// offset (in floats; 64 * CH = the dummy row) and the bilinear weight (0 for the dummy).  The region's waves then read
// their list with one coalesced 48-byte load per lane and no arithmetic (sample_point and the pixel -> region-row
// mapping cost ~10 integer divisions per entry; with them inside the region kernel its empty shell took 145 us).
struct __attribute__((aligned(16))) SampleDesc { int q[4]; float w[4]; unsigned row; int pad0, pad1, pad2; };

__global__ void build_desc_kernel(const BwdParams P, const unsigned* __restrict__ keys_sorted, SampleDesc* __restrict__ desc) {
  const int taps = P.kh * P.kw;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)P.rcount[P.nregions]) return;                       // slots past the last region's list are unused
  const int reg = (int)keys_sorted[i];
  const unsigned e = P.sorted_vals[i];
  int lvl = 0;
#pragma unroll 1
  for (int k = 1; k < P.nlev; k++) if (reg >= P.lv[k].reg0) lvl = k;
  const BLevel& L = P.lv[lvl];
  const int rl = reg - L.reg0;
  const int rw = rl % L.RW, rbh = rl / L.RW;
  const int rh = rbh % L.RH, rb = rbh / L.RH;
  const long pg = (long)(e / (unsigned)taps);
  const int tap = (int)(e - (unsigned)pg * (unsigned)taps);
  const long p = pg - (long)L.chunk0 * 32;                           // the sample belongs to this region's level
  int4 ix; float2 fr;
  sample_point(P, L, p, tap, taps, L.Ho * L.Wo, ix, fr);
  const float mm = sample_mask(L, p, tap, taps, L.Ho * L.Wo);        // DCNv2: the sample's modulation scales its scatter
  const float lh = fr.x, lw = fr.y, uh = 1.f - lh, uw = 1.f - lw;
  auto local = [&](int q) {                                          // pixel -> row of this region, 64 = another region's / outside
    if (q < 0) return 64;
    const int w = q % L.W, bh = q / L.W;
    const int h = bh % L.H, b = bh / L.H;
    return (b == rb && (h >> 3) == rh && (w >> 3) == rw) ? ((h & 7) * 8 + (w & 7)) : 64;
  };
  const int i0 = local(ix.x), i1 = local(ix.y), i2 = local(ix.z), i3 = local(ix.w);
  SampleDesc d;
  d.row = e; d.pad0 = d.pad1 = d.pad2 = 0;
  d.q[0] = i0 * CH; d.q[1] = i1 * CH; d.q[2] = i2 * CH; d.q[3] = i3 * CH;
  d.w[0] = i0 < 64 ? uh * uw * mm : 0.f; d.w[1] = i1 < 64 ? uh * lw * mm : 0.f;
  d.w[2] = i2 < 64 ? lh * uw * mm : 0.f; d.w[3] = i3 < 64 ? lh * lw * mm : 0.f;
  desc[i] = d;
}

__global__ void __launch_bounds__(kScatterThreads)
dcn_bwd_scatter_kernel(const BwdParams P, const SampleDesc* __restrict__ desc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* acc = reinterpret_cast<float*>(smem);                       // [65][256]
  const int tid = threadIdx.x, lane = tid & 63;
  // regions are visited in XCD-contiguous blocks (workgroup b runs on XCD b % 8): neighbouring regions share G rows
  int reg;
  {
    const int per = (P.nregions + 7) >> 3;
    reg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || reg >= P.nregions) return;
  }
  int lvl = 0;
#pragma unroll 1
  for (int i = 1; i < P.nlev; i++) if (reg >= P.lv[i].reg0) lvl = i;
  const BLevel L = P.lv[lvl];
  const int rl = reg - L.reg0;
  const int rw = rl % L.RW, rbh = rl / L.RW;
  const int rh = rbh % L.RH, rb = rbh / L.RH;
  // Thread c only ever touches acc[.][c], and every wave keeps its own copy of 64 list entries in registers (lane t <->
  // entry s0 + t, handed to the other lanes by v_readlane): no barrier, no atomics, no shared metadata.
  const int c = tid;
  float* mine = acc + c;
  const int l0 = P.rcount[reg], l1 = P.rcount[reg + 1];
  SampleDesc nxt;                                                    // the next 64 entries are in flight during a block
  auto fetch = [&](int s0) {
    SampleDesc d;
    d.q[0] = d.q[1] = d.q[2] = d.q[3] = 64 * CH; d.w[0] = d.w[1] = d.w[2] = d.w[3] = 0.f; d.row = 0;   // past the list: G row 0 x 0 into the dummy row
    if (s0 + lane < l1) d = desc[s0 + lane];
    return d;
  };
  nxt = fetch(l0);
#pragma unroll 5
  for (int l = 0; l < kAccRows; l++) mine[l * CH] = 0.f;
  constexpr int U = 16;                                              // list entries per step (2 x U G rows in flight per thread)
  const float* Gc = P.G + c;
#if ORP_BWD_DBG & 32
  float dbg_sum = 0.f;
#endif
  for (int s0 = l0; s0 < l1; s0 += 64) {
    const SampleDesc cur = nxt;
    if (s0 + 64 < l1) nxt = fetch(s0 + 64);
    const int o0 = cur.q[0], o1 = cur.q[1], o2 = cur.q[2], o3 = cur.q[3];
    const float w0 = cur.w[0], w1 = cur.w[1], w2 = cur.w[2], w3 = cur.w[3];
    const unsigned e = cur.row;
    const int nb = (l1 - s0) < 64 ? (l1 - s0) : 64;
    // G rows of step j0 + U are in flight while the read-modify-writes of step j0 run (double buffer)
    float g[U], gn[U];
#pragma unroll
    for (int u = 0; u < U; u++) g[u] = (ORP_BWD_DBG & 16) ? 1.f : Gc[(size_t)(unsigned)__builtin_amdgcn_readlane((int)e, u) * CH];
#pragma unroll 1
    for (int j0 = 0; j0 < nb; j0 += U) {
      const bool more = j0 + U < nb;                                 // uniform; j0 + U + u < 64 then
      if (more) {
#pragma unroll
        for (int u = 0; u < U; u++) gn[u] = (ORP_BWD_DBG & 16) ? 1.f : Gc[(size_t)(unsigned)__builtin_amdgcn_readlane((int)e, j0 + U + u) * CH];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int q0 = __builtin_amdgcn_readlane(o0, j0 + u), q1 = __builtin_amdgcn_readlane(o1, j0 + u);
        const int q2 = __builtin_amdgcn_readlane(o2, j0 + u), q3 = __builtin_amdgcn_readlane(o3, j0 + u);
        const float x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w0), j0 + u));
        const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w1), j0 + u));
        const float x2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w2), j0 + u));
        const float x3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w3), j0 + u));
        // the four corners of a sample are four different pixels (or the dummy row): read all, then write all; the next
        // sample's reads follow these writes in program order and the LDS executes a wave's accesses in order
#if ORP_BWD_DBG & 32
        dbg_sum += x0 * g[u] + x1 * g[u] + x2 * g[u] + x3 * g[u] + (float)(q0 + q1 + q2 + q3);
#else
        const float a0 = mine[q0], a1 = mine[q1], a2 = mine[q2], a3 = mine[q3];
        mine[q0] = a0 + x0 * g[u];
        mine[q1] = a1 + x1 * g[u];
        mine[q2] = a2 + x2 * g[u];
        mine[q3] = a3 + x3 * g[u];
#endif
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < U; u++) g[u] = gn[u];
      }
    }
  }
#if ORP_BWD_DBG & 32
  mine[0] = dbg_sum;
#endif
  // every pixel of the region is written exactly once (zeros included): no memset, no atomics
  for (int l = 0; l < 64; l++) {
    const int h = rh * 8 + (l >> 3), w = rw * 8 + (l & 7);
    if (h < L.H && w < L.W) L.gx[((size_t)(rb * L.H + h) * L.W + w) * CH + c] = mine[l * CH];
  }
}

// ---- kernel B: grad_weight partial sums ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
dcn_bwd_weight_kernel(const BwdParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sG = reinterpret_cast<float*>(smem);                       // [32][RS] grad_out rows of the chunk
  float* sC = sG + 32 * RS;                                         // [32][RS] sampled columns of the chunk (this tap)
  float4* sCw = reinterpret_cast<float4*>(sC + 32 * RS);            // [2][32] bilinear weights
  int4* sCi = reinterpret_cast<int4*>(sCw + 64);                    // [2][32] pixel indices (clamped)
  long* sRow = reinterpret_cast<long*>(sCi + 64);                   // [2][32] grad_out row (level-linear position), -1 = none

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int taps = P.kh * P.kw, tap = blockIdx.y;
  const int n_active = P.active[P.total_chunks];                    // this split walks active[c_begin, c_end)
  const int per = (n_active + P.nsplit - 1) / P.nsplit;
  const int c_begin = blockIdx.x * per;
  const int c_end = (c_begin + per < n_active) ? c_begin + per : n_active;

  auto level_of = [&](int chunk) {
    int lvl = 0;
#pragma unroll 1
    for (int i = 1; i < P.nlev; i++) if (chunk >= P.lv[i].chunk0) lvl = i;
    return lvl;
  };
  auto make_coef = [&](int ci, int buf) {                            // threads 0..31; ci indexes the active list
    const int chunk = P.active[ci];
    const BLevel& L = P.lv[level_of(chunk)];
    const int HoWo = L.Ho * L.Wo;
    const long p = (long)(chunk - L.chunk0) * 32 + tid;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 ixc = make_int4(0, 0, 0, 0);
    long row = -1;
    if (p < (long)P.B * HoWo) {
      int4 ix; float2 fr;
      sample_point(P, L, p, tap, taps, HoWo, ix, fr);
      const float mm = sample_mask(L, p, tap, taps, HoWo);           // DCNv2: the column is the modulated sample
      const float lh = fr.x, lw = fr.y, uh = 1.f - lh, uw = 1.f - lw;
      w.x = ix.x >= 0 ? uh * uw * mm : 0.f; w.y = ix.y >= 0 ? uh * lw * mm : 0.f;
      w.z = ix.z >= 0 ? lh * uw * mm : 0.f; w.w = ix.w >= 0 ? lh * lw * mm : 0.f;
      ixc = make_int4(ix.x < 0 ? 0 : ix.x, ix.y < 0 ? 0 : ix.y, ix.z < 0 ? 0 : ix.z, ix.w < 0 ? 0 : ix.w);
      row = p;
    }
    sCw[buf * 32 + tid] = w; sCi[buf * 32 + tid] = ixc; sRow[buf * 32 + tid] = row;
  };
  auto gather_issue = [&](int ci, int buf, int row, float4 (&g)[4], float4& gr) {
    const BLevel& L = P.lv[level_of(P.active[ci])];
    const int4 ix = sCi[buf * 32 + row];
    const long prow = sRow[buf * 32 + row];
    const float* base = L.x + lane * 4;
    g[0] = *reinterpret_cast<const float4*>(base + (size_t)ix.x * CH);
    g[1] = *reinterpret_cast<const float4*>(base + (size_t)ix.y * CH);
    g[2] = *reinterpret_cast<const float4*>(base + (size_t)ix.z * CH);
    g[3] = *reinterpret_cast<const float4*>(base + (size_t)ix.w * CH);
    gr = prow >= 0 ? *reinterpret_cast<const float4*>(L.go + (size_t)prow * CH + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto combine = [&](int buf, int row, const float4 (&g)[4]) {
    const float4 wgt = sCw[buf * 32 + row];
    float4 v;
    v.x = __builtin_fmaf(wgt.w, g[3].x, __builtin_fmaf(wgt.z, g[2].x, __builtin_fmaf(wgt.y, g[1].x, wgt.x * g[0].x)));
    v.y = __builtin_fmaf(wgt.w, g[3].y, __builtin_fmaf(wgt.z, g[2].y, __builtin_fmaf(wgt.y, g[1].y, wgt.x * g[0].y)));
    v.z = __builtin_fmaf(wgt.w, g[3].z, __builtin_fmaf(wgt.z, g[2].z, __builtin_fmaf(wgt.y, g[1].z, wgt.x * g[0].z)));
    v.w = __builtin_fmaf(wgt.w, g[3].w, __builtin_fmaf(wgt.z, g[2].w, __builtin_fmaf(wgt.y, g[1].w, wgt.x * g[0].w)));
    return v;
  };

  floatx16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = floatx16{0};
  const int mrow = lane & 31, kh = lane >> 5;
  const int wo = wave >> 1, wc = wave & 1;                          // output-channel blocks 2wo, 2wo+1; input-channel blocks 4wc..4wc+3

  if (c_begin < c_end) {
    if (tid < 32) { make_coef(c_begin, 0); if (c_begin + 1 < c_end) make_coef(c_begin + 1, 1); }
    __syncthreads();
    {
      float4 g[4][4], gr[4];
#pragma unroll
      for (int u = 0; u < 4; u++) gather_issue(c_begin, 0, u * 8 + wave, g[u], gr[u]);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int row = u * 8 + wave;
        *reinterpret_cast<float4*>(sC + (size_t)row * RS + lane * 4) = combine(0, row, g[u]);
        *reinterpret_cast<float4*>(sG + (size_t)row * RS + lane * 4) = gr[u];
      }
    }
    __syncthreads();

#pragma unroll 1
    for (int chunk = c_begin; chunk < c_end; chunk++) {
      const int it = chunk - c_begin;
      const bool have_next = chunk + 1 < c_end;
      const int nbuf = (it + 1) & 1;
      float4 holdc[4], holdg[4];
      // operands of k-step j + 1 are read from LDS before the MFMAs of step j are issued (the wave's own LDS latency
      // otherwise sits in front of every group of eight MFMAs)
      float a0, a1, b0, b1, b2, b3;
      {
        const float* ga = sG + (size_t)kh * RS + (2 * wo) * 32 + mrow;
        const float* cb = sC + (size_t)kh * RS + (4 * wc) * 32 + mrow;
        a0 = ga[0]; a1 = ga[32]; b0 = cb[0]; b1 = cb[32]; b2 = cb[64]; b3 = cb[96];
      }
#pragma unroll
      for (int j = 0; j < 16; j++) {
        float4 g[4];
        const bool do_row = have_next && j < 4;
        if (do_row) gather_issue(chunk + 1, nbuf, j * 8 + wave, g, holdg[j < 4 ? j : 0]);
        const float ca0 = a0, ca1 = a1, cb0 = b0, cb1 = b1, cb2 = b2, cb3 = b3;
        if (j + 1 < 16) {
          const float* ga = sG + (size_t)(2 * (j + 1) + kh) * RS + (2 * wo) * 32 + mrow;
          const float* cb = sC + (size_t)(2 * (j + 1) + kh) * RS + (4 * wc) * 32 + mrow;
          a0 = ga[0]; a1 = ga[32]; b0 = cb[0]; b1 = cb[32]; b2 = cb[64]; b3 = cb[96];
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0, cb0, acc[0], 0, 0, 0);             // D[o][c]
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0, cb1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0, cb2, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0, cb3, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1, cb0, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1, cb1, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1, cb2, acc[6], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1, cb3, acc[7], 0, 0, 0);
        if (do_row) holdc[j < 4 ? j : 0] = combine(nbuf, j * 8 + wave, g);
      }
      if (have_next) {
        __syncthreads();                                             // every wave is past its last read of this chunk
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int row = u * 8 + wave;
          *reinterpret_cast<float4*>(sC + (size_t)row * RS + lane * 4) = holdc[u];
          *reinterpret_cast<float4*>(sG + (size_t)row * RS + lane * 4) = holdg[u];
        }
        if (tid < 32 && chunk + 2 < c_end) make_coef(chunk + 2, it & 1);
        __syncthreads();
      }
    }
  }

  float* outp = P.partial + ((size_t)blockIdx.x * taps + tap) * CH * CH;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int o = (2 * wo + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        outp[(size_t)o * CH + (4 * wc + q) * 32 + mrow] = acc[a * 4 + q][r];
      }
}

// ---- kernel B on the 16-bit matrix pipe (round 6) ------------------------------------------------------------------------------
// The same contraction, gW_t[o, c] = sum_p go[p, o] * col_t[p, c], in the fp16-pieces arithmetic of the forward (csrc/orp_dcn_split.hip)
// and of the towers' weight gradient (csrc/orp_conv_wgrad.hip): both operands scaled by a power of two that puts the tensor's largest
// magnitude into [2^14, 2^15), split into hi = fp16(v), lo = fp16(v - hi) (2^-22 relative), products lo*hi, hi*lo, hi*hi into one fp32
// accumulator, smallest first.  The contraction runs over POSITIONS, so a lane's eight k-values of an MFMA operand are eight consecutive
// positions of one channel, and LDS holds [piece][channel][32 positions] (80-byte rows: conflict-free 16-byte fragment reads).
// Gathers: thread = a PAIR of channels (8-byte loads: a wave fetches half an NHWC row, 512 B, per request) and four positions of every
// chunk.  A position's sampling corners and bilinear weights are wave-uniform: they come from a table a pre-pass leaves
// (sample_table_kernel; 32 B per (position, tap), read with scalar loads; the row offsets are the SCALAR offsets of buffer loads, the
// lane's channels the vector offset -- no per-lane 64-bit addresses); the four corner values are combined in the lane, split, and
// written as 8-byte row pieces.  grad_out arrives ready-made: its two pieces are written once per call in the staging order
// (pack_go16_kernel) instead of being split by every tap's workgroups.
// Workgroup = (split, tap, half of the input channels): tile 256 (o) x 128 (c), 64 accumulator registers per wave; the corner values of
// the chunk after next are requested one load behind every MFMA of the current chunk (two register sets that swap roles from chunk to
// chunk), the next chunk's are combined and split between the MFMAs.  LDS is double buffered (2 x 60 KB); one barrier per chunk.
// Measured (2 x 21 824 positions, dense gradient): weight kernel + reduction 542 us (exact fp32) -> 253 us.  Anatomy (ORP_BW16_DBG builds):
// without the MFMAs 239 us, without the gathers 182 us -- the kernel sits on the gathers, and what limits them is the vector memory
// unit's REQUEST rate, not latency and not bytes: with 4-byte loads per lane (thread = channel) 2.4 GB moved at ~27 B/clk per CU and the
// same kernel took 300 us, whether 16 or 64 loads per thread were in flight and whether they were issued in a burst or spread between the
// MFMAs (286 - 312 us); 8-byte loads: 253.  Next lever: 16-byte loads (thread = channel quad; a wave then spans two positions and the
// row offsets become per-lane).
// The exact-fp32 kernel above stays for DCNv2 (a modulated column has no known range) and behind ORP_DCN_BWD_SPLIT=0 / ORP_DCN_BWD_W16=0.
struct __attribute__((aligned(32))) SampleTab { int ix[4]; float w[4]; };     // corner pixels (clamped; BYTE offsets of their NHWC rows) and weights x the range scale of x (0: outside)

__global__ void __launch_bounds__(256) sample_table_kernel(const BwdParams P, const unsigned* __restrict__ amax_x, SampleTab* __restrict__ tab) {
  const int taps = P.kh * P.kw;
  const long per_tap = (long)P.total_chunks * 32;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_tap * taps) return;
  const int tap = (int)(i / per_tap);
  const long e = i - (long)tap * per_tap;
  const int chunk = (int)(e >> 5), m = (int)(e & 31);
  int lvl = 0;
  for (int k = 1; k < P.nlev; k++) if (chunk >= P.lv[k].chunk0) lvl = k;
  const BLevel& L = P.lv[lvl];
  const int HoWo = L.Ho * L.Wo;
  const long p = (long)(chunk - L.chunk0) * 32 + m;
  SampleTab t;
  t.ix[0] = t.ix[1] = t.ix[2] = t.ix[3] = 0;
  t.w[0] = t.w[1] = t.w[2] = t.w[3] = 0.f;
  if (p < (long)P.B * HoWo) {
    int4 ix; float2 fr;
    sample_point(P, L, p, tap, taps, HoWo, ix, fr);
    const float sx = range_scale(*amax_x);                            // a power of two: scaling the weights scales the column exactly
    const float lh = fr.x, lw = fr.y, uh = 1.f - lh, uw = 1.f - lw;
    t.w[0] = ix.x >= 0 ? uh * uw * sx : 0.f; t.w[1] = ix.y >= 0 ? uh * lw * sx : 0.f;
    t.w[2] = ix.z >= 0 ? lh * uw * sx : 0.f; t.w[3] = ix.w >= 0 ? lh * lw * sx : 0.f;
    const int rb = CH * (int)sizeof(float);                           // (the host checked that every level's rows fit 2^31 bytes)
    t.ix[0] = ix.x < 0 ? 0 : ix.x * rb; t.ix[1] = ix.y < 0 ? 0 : ix.y * rb; t.ix[2] = ix.z < 0 ? 0 : ix.z * rb; t.ix[3] = ix.w < 0 ? 0 : ix.w * rb;
  }
  tab[i] = t;
}

constexpr int RS16 = 40;                    // LDS row stride in halves (64 B of payload + 16 B)
constexpr int CX16 = 128;                   // input channels (columns of gW) per workgroup
constexpr int GPL16 = CH * RS16;            // halves per grad_out piece plane
constexpr int XPL16 = CX16 * RS16;          // halves per column piece plane
constexpr int BUF16 = 2 * GPL16 + 2 * XPL16;          // G hi | G lo | X hi | X lo
constexpr size_t weight16_smem() { return sizeof(_Float16) * 2 * BUF16; }
constexpr size_t kGo16ChunkHalves = (size_t)2 * 4 * CH * 8;       // grad_out planes of one chunk: [piece][octet][channel][8 positions]

// grad_out (NHWC fp32 in the workspace) -> its two fp16 pieces, scaled, laid out as kernel B stages them: thread = (chunk, octet, channel)
// reads eight positions of its channel (consecutive channels: coalesced) and writes 16 bytes per piece (consecutive channels: contiguous).
// Rows past a level's end are zero.
__global__ void __launch_bounds__(256) pack_go16_kernel(const BwdParams P, _Float16* __restrict__ planes) {
  const int chunk = blockIdx.x >> 2, q = blockIdx.x & 3, c = threadIdx.x;
  int lvl = 0;
  for (int k = 1; k < P.nlev; k++) if (chunk >= P.lv[k].chunk0) lvl = k;
  const BLevel& L = P.lv[lvl];
  const long npos = (long)P.B * L.Ho * L.Wo;
  const long p0 = (long)(chunk - L.chunk0) * 32 + q * 8;
  const float sg = range_scale(*P.go_amax);
  h8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const float v = (p0 + e < npos ? L.go[(size_t)(p0 + e) * CH + c] : 0.f) * sg;
    hi[e] = (_Float16)v;
    lo[e] = (_Float16)(v - (float)hi[e]);
  }
  _Float16* dst = planes + (size_t)chunk * kGo16ChunkHalves + ((size_t)q * CH + c) * 8;
  *reinterpret_cast<h8*>(dst) = hi;
  *reinterpret_cast<h8*>(dst + (size_t)4 * CH * 8) = lo;
}

#ifndef ORP_BW16_DBG
#define ORP_BW16_DBG 0      // dev aid (timing only, wrong results): 1 = no gathers after the prologue, 2 = no MFMA, 4 = no combine / split / LDS writes
#endif

__global__ void __launch_bounds__(kThreads)
dcn_bwd_weight16_kernel(const BwdParams P, const SampleTab* __restrict__ tab, const _Float16* __restrict__ go16,
                        const unsigned* __restrict__ amax_x, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* sT = reinterpret_cast<_Float16*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cz = blockIdx.z;                                          // which 128 input channels
  const int cp = lane;                                                // this thread's two column rows: channels cz * 128 + 2 cp, + 1 ...
  const int px = wave * 4;                                            // ... and its four positions of every chunk (8-byte loads: twice the bytes per request)
  const int cg = tid & (CH - 1), hf = wave >> 2;                      // grad_out staging: channel cg, octets hf and hf + 2
  const int taps = P.kh * P.kw, tap = blockIdx.y;
  const int n_active = P.active[P.total_chunks];
  const int per = (n_active + nsplit - 1) / nsplit;
  const int c_begin = blockIdx.x * per;
  const int c_end = (c_begin + per < n_active) ? c_begin + per : n_active;
  const float sx = range_scale(*amax_x), sg = range_scale(*P.go_amax);
  const SampleTab* tap_tab = tab + (size_t)tap * P.total_chunks * 32 + px;
  const int c4 = (cz * CX16 + 2 * cp) * (int)sizeof(float);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(go16), 0, P.total_chunks * (int)(kGo16ChunkHalves * sizeof(_Float16)), 0x00020000);

  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  struct Raw { f2 x[4][4]; };
  struct GRaw { h8 v[2][2]; };
  auto level_of = [&](int chunk) {
    int lvl = 0;
#pragma unroll 1
    for (int i = 1; i < P.nlev; i++) if (chunk >= P.lv[i].chunk0) lvl = i;
    return lvl;
  };
  auto issue_half = [&](int ci, int half, Raw& r) {                   // 2 positions x 4 corners of this thread's channel pair: positions px + 2 half, + 1
    const int chunk = P.active[ci];
    const BLevel& L = P.lv[level_of(chunk)];
    const SampleTab* t = tap_tab + (size_t)chunk * 32 + 2 * half;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.x), 0, P.B * L.H * L.W * CH * (int)sizeof(float), 0x00020000);
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
      for (int k = 0; k < 4; k++) r.x[2 * half + e][k] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs, c4, t[e].ix[k], 0));
  };
  auto issue = [&](int ci, Raw& r) { issue_half(ci, 0, r); issue_half(ci, 1, r); };
  // positions px + 2 half, + 1: [channel of the pair][piece] -> two halves each
  auto convert = [&](int ci, int half, const Raw& r, h2 (&out)[2][2]) {
    const SampleTab* t = tap_tab + (size_t)P.active[ci] * 32 + 2 * half;
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const f2* x = r.x[2 * half + e];
#pragma unroll
      for (int cc = 0; cc < 2; cc++) {
        const float sv = __builtin_fmaf(t[e].w[3], x[3][cc], __builtin_fmaf(t[e].w[2], x[2][cc], __builtin_fmaf(t[e].w[1], x[1][cc], t[e].w[0] * x[0][cc])));
        const _Float16 hi = (_Float16)sv;
        out[cc][0][e] = hi;
        out[cc][1][e] = (_Float16)(sv - (float)hi);                   // the residual is exact in fp32
      }
    }
  };
  auto put_x = [&](int buf, const h2 (&a)[2][2], const h2 (&b)[2][2]) {      // a: positions px, px + 1; b: px + 2, px + 3
#pragma unroll
    for (int cc = 0; cc < 2; cc++)
#pragma unroll
      for (int pl = 0; pl < 2; pl++) {
        h4 v; v[0] = a[cc][pl][0]; v[1] = a[cc][pl][1]; v[2] = b[cc][pl][0]; v[3] = b[cc][pl][1];
        *reinterpret_cast<h4*>(sT + (size_t)buf * BUF16 + (size_t)2 * GPL16 + (size_t)pl * XPL16 + (size_t)(2 * cp + cc) * RS16 + px) = v;
      }
  };
  auto get_g = [&](int ci, GRaw& g) {                                 // octets hf, hf + 2 of grad_out, both pieces
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int so = P.active[ci] * (int)(kGo16ChunkHalves * sizeof(_Float16)) + (hf + 2 * u) * CH * 16;
      g.v[u][0] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs_g, cg * 16, so, 0));
      g.v[u][1] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs_g, cg * 16, so + 4 * CH * 16, 0));
    }
  };
  auto put_g = [&](int buf, const GRaw& g) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      _Float16* dst = sT + (size_t)buf * BUF16 + (size_t)cg * RS16 + (hf + 2 * u) * 8;
      *reinterpret_cast<h8*>(dst) = g.v[u][0];
      *reinterpret_cast<h8*>(dst + GPL16) = g.v[u][1];
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int q = 0; q < 2; q++) acc[a][q] = floatx16{0};
  const int m = lane & 31, kg = lane >> 5;
  const int wo = wave >> 1, wc = wave & 1;                            // o rows [64 wo, +64), c columns [64 wc, +64) of the workgroup's 128

  if (c_begin < c_end) {
    Raw r1, r2;
    {                                                                 // the first chunk, unpipelined
      h2 cv[2][2][2];
      GRaw g;
      issue(c_begin, r1); get_g(c_begin, g);
      convert(c_begin, 0, r1, cv[0]); convert(c_begin, 1, r1, cv[1]);
      put_x(0, cv[0], cv[1]);
      put_g(0, g);
    }
    issue(min(c_begin + 1, c_end - 1), r1);                           // the second chunk's corners are in flight when the loop starts
    __syncthreads();
    // one chunk: MFMAs out of buffer `cur`; `ra` holds the next chunk's corner values (requested a chunk ago), `rb` receives those of
    // the chunk after next.  The two register sets swap roles from chunk to chunk (the loop below is unrolled by two): a copy would
    // wait for the loads it copies.
    auto step = [&](int ci, int cur, Raw& ra, Raw& rb) {
      const bool more = ci + 1 < c_end;
      const int nx = min(ci + 1, c_end - 1), nx2 = min(ci + 2, c_end - 1);
      const _Float16* sB = sT + (size_t)cur * BUF16;
      GRaw g;
      h2 cv[2][2][2];
      __builtin_amdgcn_sched_barrier(0);
      get_g(nx, g);                                                   // (first: the memory counter retires in order, and these are written to LDS first)
#pragma unroll
      for (int j = 0; j < 2; j++) {                                   // the chunk's two 16-position k-steps
        h8 ga[2][2], xb[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
          for (int pl = 0; pl < 2; pl++)
            ga[a][pl] = *reinterpret_cast<const h8*>(sB + (size_t)pl * GPL16 + (size_t)(wo * 64 + a * 32 + m) * RS16 + j * 16 + kg * 8);
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
          for (int pl = 0; pl < 2; pl++)
            xb[q][pl] = *reinterpret_cast<const h8*>(sB + (size_t)2 * GPL16 + (size_t)pl * XPL16 + (size_t)(wc * 64 + q * 32 + m) * RS16 + j * 16 + kg * 8);
        __builtin_amdgcn_sched_barrier(0);
        // two chunks ahead: 16 of the 32 corner loads per k-step, a load or two behind every MFMA (issued in one burst they hold every
        // wave of the workgroup at the vector memory unit's door at the same time, with the matrix pipe idle behind them)
        if (!(ORP_BW16_DBG & 1)) issue_half(nx2, j, rb);
        // the next chunk's corner values: half an octet per k-step, between the MFMAs
        if (!(ORP_BW16_DBG & 4)) convert(nx, j, ra, cv[j]);
        // smallest products first; four independent accumulators between two MFMAs into the same one
#pragma unroll
        for (int pr = 0; pr < 3; pr++)
#pragma unroll
          for (int a = 0; a < 2; a++)
#pragma unroll
            for (int q = 0; q < 2; q++)
              if (ORP_BW16_DBG & 2) acc[a][q][0] += (float)ga[a][pr == 0 ? 1 : 0][0] * (float)xb[q][pr == 1 ? 1 : 0][0];
              else acc[a][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[a][pr == 0 ? 1 : 0], xb[q][pr == 1 ? 1 : 0], acc[a][q], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 12; i++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more && !(ORP_BW16_DBG & 4)) { put_x(cur ^ 1, cv[0], cv[1]); put_g(cur ^ 1, g); }
      __syncthreads();
    };
    int ci = c_begin;
#pragma unroll 1
    for (; ci + 1 < c_end; ci += 2) { step(ci, 0, r1, r2); step(ci + 1, 1, r2, r1); }
    if (ci < c_end) step(ci, 0, r1, r2);
  }

  const float osc = 1.f / (sx * sg);
  float* outp = P.partial + ((size_t)blockIdx.x * taps + tap) * CH * CH;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int o = wo * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        outp[(size_t)o * CH + cz * CX16 + wc * 64 + q * 32 + m] = acc[a][q][r] * osc;
      }
}

constexpr size_t weight_smem() { return sizeof(float) * 2 * 32 * RS + (sizeof(float4) + sizeof(int4) + sizeof(long)) * 64; }

int device_cus() {
  static int cus[32] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

struct Plan {
  size_t x_off[MAXL], go_off[MAXL], gx_off[MAXL];
  size_t gx_begin, gx_bytes, wT_off, partial_off, flags_off, scale_off, list_off, total;
  size_t G_off, keys_in_off, keys_out_off, vals_in_off, vals_out_off, rcount_off, desc_off, cub_off, cub_bytes;
  int Ho[MAXL], Wo[MAXL], reg0[MAXL], RH[MAXL], RW[MAXL];
  int total_chunks, nsplit, nregions, key_bits;
  long nslots;
};
int make_plan(const orp_dcn_bwd_level* lv, int nlevels, int batch, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
              int dw, int cus, Plan& pl) {
  size_t cur = 0;
  pl.total_chunks = 0;
  for (int i = 0; i < nlevels; i++) {
    if (lv[i].height <= 0 || lv[i].width <= 0) return ORP_EINVAL;
    pl.Ho[i] = out_dim(lv[i].height, ph, dh, kh, sh); pl.Wo[i] = out_dim(lv[i].width, pw, dw, kw, sw);
    if (pl.Ho[i] <= 0 || pl.Wo[i] <= 0) return ORP_EINVAL;
    if ((long)batch * lv[i].height * lv[i].width >= (1L << 31) / CH) return ORP_ETOOBIG;
    pl.x_off[i] = cur; cur += align256(sizeof(float) * (size_t)batch * lv[i].height * lv[i].width * CH);
    pl.go_off[i] = cur; cur += align256(sizeof(float) * (size_t)batch * pl.Ho[i] * pl.Wo[i] * CH);
    pl.total_chunks += (int)(((long)batch * pl.Ho[i] * pl.Wo[i] + 31) / 32);
  }
  pl.gx_begin = cur;
  for (int i = 0; i < nlevels; i++) { pl.gx_off[i] = cur; cur += align256(sizeof(float) * (size_t)batch * lv[i].height * lv[i].width * CH); }
  pl.gx_bytes = cur - pl.gx_begin;
  pl.wT_off = cur; cur += align256(sizeof(float) * (size_t)CH * CH * kh * kw);
  int ns = cus / (kh * kw); if (ns < 1) ns = 1;
  if (ns > pl.total_chunks) ns = pl.total_chunks;
  pl.nsplit = ns;
  pl.partial_off = cur; cur += align256(sizeof(float) * (size_t)ns * kh * kw * CH * CH);
  pl.flags_off = cur; cur += align256(sizeof(int) * (size_t)pl.total_chunks);
  pl.scale_off = cur; cur += 256;       // [0] max |grad_out| bits, [1] max |W| bits, [2] the weights' scale (zeroed together with the flags)
  pl.list_off = cur; cur += align256(sizeof(int) * ((size_t)pl.total_chunks + 1));
  // kernel A2: regions, the G rows, the (region, sample) slots and the radix sort's scratch
  pl.nregions = 0;
  for (int i = 0; i < nlevels; i++) {
    pl.RH[i] = (lv[i].height + 7) / 8; pl.RW[i] = (lv[i].width + 7) / 8;
    pl.reg0[i] = pl.nregions; pl.nregions += batch * pl.RH[i] * pl.RW[i];
  }
  pl.key_bits = 1;
  while ((1L << pl.key_bits) <= (long)pl.nregions) pl.key_bits++;   // keys 0 .. nregions (nregions = unused slot)
  const long E = (long)pl.total_chunks * 32 * kh * kw;
  if (4 * E >= (1L << 31)) return ORP_ETOOBIG;
  pl.nslots = 4 * E;
  pl.G_off = cur; cur += align256(sizeof(float) * (size_t)E * CH);
  pl.keys_in_off = cur; cur += align256(sizeof(unsigned) * (size_t)pl.nslots);
  pl.keys_out_off = cur; cur += align256(sizeof(unsigned) * (size_t)pl.nslots);
  pl.vals_in_off = cur; cur += align256(sizeof(unsigned) * (size_t)pl.nslots);
  pl.vals_out_off = cur; cur += align256(sizeof(unsigned) * (size_t)pl.nslots);
  pl.rcount_off = cur; cur += align256(sizeof(int) * ((size_t)pl.nregions + 1));
  pl.desc_off = cur; cur += align256(sizeof(SampleDesc) * (size_t)pl.nslots);
  size_t cub = 0;
  if (hipcub::DeviceRadixSort::SortPairs((void*)nullptr, cub, (const unsigned*)nullptr, (unsigned*)nullptr,
                                         (const unsigned*)nullptr, (unsigned*)nullptr, (int)pl.nslots, 0, pl.key_bits,
                                         (hipStream_t)0) != hipSuccess) return ORP_EINVAL;
  pl.cub_bytes = cub;
  pl.cub_off = cur; cur += align256(cub);
  pl.total = cur + 256;
  return ORP_OK;
}

}  // namespace

extern "C" {

int orp_dcn_backward_mfma_ok(int c_in, int c_out, int kh, int kw, int groups, int deformable_groups) {
  return (groups == 1 && deformable_groups == 1 && c_in == CH && c_out == CH && kh > 0 && kw > 0 && kh * kw <= MAXT) ? 1 : 0;
}

size_t orp_dcn_backward_workspace_bytes(const orp_dcn_bwd_level* levels_host, int nlevels, int batch, int kh, int kw,
                                        int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w) {
  if (!levels_host || nlevels <= 0 || nlevels > MAXL || batch <= 0) return 0;
  Plan pl;
  if (make_plan(levels_host, nlevels, batch, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, device_cus(), pl) != ORP_OK)
    return 0;
  return pl.total;
}

int orp_dcn_backward_multi(const orp_dcn_bwd_level* levels_host, int nlevels, int batch, int c_in, int c_out,
                           const float* weight, float* grad_weight, int need_input_grads, int kh, int kw, int stride_h,
                           int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, void* workspace,
                           size_t workspace_bytes, void* stream) {
  return orp_dcn_backward_multi_ex(levels_host, nullptr, nullptr, 0, nlevels, batch, c_in, c_out, weight, grad_weight,
                                   need_input_grads, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, workspace,
                                   workspace_bytes, stream);
}

int orp_dcn_backward_multi_ex(const orp_dcn_bwd_level* levels_host, const float* const* masks_host,
                              float* const* grad_masks_host, int io_dtype, int nlevels, int batch, int c_in, int c_out,
                              const float* weight, float* grad_weight, int need_input_grads, int kh, int kw, int stride_h,
                              int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (!levels_host || nlevels <= 0 || nlevels > MAXL || batch <= 0 || !weight || !workspace) return ORP_EINVAL;
  if (!orp_dcn_backward_mfma_ok(c_in, c_out, kh, kw, 1, 1)) return ORP_EINVAL;
  if (io_dtype < 0 || io_dtype > 2 || (need_input_grads && masks_host && !grad_masks_host)) return ORP_EINVAL;
  if (!grad_weight && !need_input_grads) return ORP_OK;
  hipStream_t st = (hipStream_t)stream;
  Plan pl;
  int rc = make_plan(levels_host, nlevels, batch, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, device_cus(), pl);
  if (rc != ORP_OK) return rc;
  if (workspace_bytes < pl.total) return ORP_EWORKSPACE;
  char* ws = reinterpret_cast<char*>(workspace);
  const int taps = kh * kw;

  constexpr int MT = 1;
  BwdParams P;
  P.nlev = nlevels; P.B = batch;
  P.kh = kh; P.kw = kw; P.sh = stride_h; P.sw = stride_w; P.ph = pad_h; P.pw = pad_w; P.dh = dil_h; P.dw = dil_w;
  P.wT = reinterpret_cast<float*>(ws + pl.wT_off);
  P.partial = reinterpret_cast<float*>(ws + pl.partial_off);
  P.nsplit = pl.nsplit; P.total_chunks = pl.total_chunks;
  P.G = nullptr; P.flags = nullptr; P.nregions = 0; P.keys = nullptr; P.vals = nullptr; P.rcount = nullptr; P.sorted_vals = nullptr;
  int* flags = reinterpret_cast<int*>(ws + pl.flags_off);
  P.active = reinterpret_cast<int*>(ws + pl.list_off);
  TransposeSet TI, TO;
  int ti = 0, tiles = 0, chunks = 0, tin = 0, tout = 0;
  for (int i = 0; i < nlevels; i++) {
    const orp_dcn_bwd_level& lv = levels_host[i];
    if (!lv.input || !lv.offset || !lv.grad_output) return ORP_EINVAL;
    if (need_input_grads && (!lv.grad_input || !lv.grad_offset)) return ORP_EINVAL;
    BLevel& D = P.lv[i];
    D.H = lv.height; D.W = lv.width; D.Ho = pl.Ho[i]; D.Wo = pl.Wo[i];
    D.x = reinterpret_cast<float*>(ws + pl.x_off[i]);
    D.go = reinterpret_cast<float*>(ws + pl.go_off[i]);
    D.gx = reinterpret_cast<float*>(ws + pl.gx_off[i]);
    D.off = lv.offset; D.goff = lv.grad_offset;
    D.mask = masks_host ? masks_host[i] : nullptr;
    D.gmask = (masks_host && grad_masks_host && need_input_grads) ? grad_masks_host[i] : nullptr;
    if (masks_host && !D.mask) return ORP_EINVAL;
    if (masks_host && need_input_grads && !D.gmask) return ORP_EINVAL;
    D.tile0 = tiles; D.chunk0 = chunks;
    D.reg0 = pl.reg0[i]; D.RH = pl.RH[i]; D.RW = pl.RW[i];
    const long npos = (long)batch * D.Ho * D.Wo;
    tiles += (int)((npos + 32 * MT - 1) / (32 * MT));
    chunks += (int)((npos + 31) / 32);
    const int HW = lv.height * lv.width, HoWo = D.Ho * D.Wo;
    TI.in[ti] = lv.input; TI.out[ti] = const_cast<float*>(D.x); TI.R[ti] = CH; TI.S[ti] = HW; TI.t0[ti] = tin;
    TI.chunk0[ti] = -1;
    tin += ((HW + 31) / 32) * (CH / 32); ti++;
    TI.in[ti] = lv.grad_output; TI.out[ti] = const_cast<float*>(D.go); TI.R[ti] = CH; TI.S[ti] = HoWo; TI.t0[ti] = tin;
    TI.chunk0[ti] = D.chunk0;
    tin += ((HoWo + 31) / 32) * (CH / 32); ti++;
    TO.in[i] = D.gx; TO.out[i] = lv.grad_input; TO.R[i] = HW; TO.S[i] = CH; TO.t0[i] = tout; TO.chunk0[i] = -1;
    tout += ((HW + 31) / 32) * (CH / 32);
  }
  for (int i = nlevels; i < MAXL; i++) { P.lv[i] = P.lv[0]; P.lv[i].tile0 = 0x7fffffff; P.lv[i].chunk0 = 0x7fffffff; P.lv[i].reg0 = 0x7fffffff; }
  TI.n = ti; TO.n = nlevels;
  for (int i = ti; i <= 2 * MAXL; i++) TI.t0[i] = tin;
  for (int i = nlevels; i <= 2 * MAXL; i++) TO.t0[i] = tout;
  for (int i = ti; i < 2 * MAXL; i++) { TI.in[i] = TI.in[0]; TI.out[i] = TI.out[0]; TI.R[i] = TI.S[i] = 0; TI.chunk0[i] = -1; }
  for (int i = nlevels; i < 2 * MAXL; i++) { TO.in[i] = TO.in[0]; TO.out[i] = TO.out[0]; TO.R[i] = TO.S[i] = 0; TO.chunk0[i] = -1; }
  TI.flags = flags; TO.flags = flags;
  // the contraction of kernel A: 1 (default) = fp16 pieces on the 16-bit matrix pipe, 0 = exact fp32 (v_mfma_f32_32x32x2_f32)
  static const int split_mode = getenv("ORP_DCN_BWD_SPLIT") ? atoi(getenv("ORP_DCN_BWD_SPLIT")) : 1;
  const bool f16 = split_mode != 0 && need_input_grads;
  // ... of kernel B: 1 (default) = fp16 pieces as well (DCNv1: the sampled columns are bounded by max |x|), 0 / DCNv2 = exact fp32
  static const int w16_env = getenv("ORP_DCN_BWD_W16") ? atoi(getenv("ORP_DCN_BWD_W16")) : 1;        // dev aid (A/B timing)
  bool w16 = split_mode != 0 && w16_env != 0 && grad_weight && !masks_host && taps >= 2 && (long)pl.total_chunks * 32768 < (1L << 31);
  for (int i = 0; i < nlevels; i++) w16 = w16 && (long)batch * levels_host[i].height * levels_host[i].width * CH * 4 < (1L << 31);   // 32-bit buffer offsets
  unsigned* scale_words = reinterpret_cast<unsigned*>(ws + pl.scale_off);    // [0] max |grad_out|, [1] max |W|, [2] the weights' scale, [3] max |x|
  TI.amax = (f16 || w16) ? scale_words : nullptr; TO.amax = nullptr;
  TI.amax_x = w16 ? scale_words + 3 : nullptr; TO.amax_x = nullptr;
  P.wT16 = reinterpret_cast<const uint16_t*>(ws + pl.wT_off);      // (the fp32 layout and the two fp16 planes have the same size)
  P.w16_plane = (size_t)CH * CH * taps;
  P.wscale = reinterpret_cast<const float*>(scale_words + 2);
  P.go_amax = scale_words;
  TI.in_code = io_dtype; TI.out_code = 0;                   // x / grad_out arrive in the I/O type, the workspace is fp32
  TO.in_code = 0; TO.out_code = io_dtype;                   // grad_input leaves in the I/O type

  OrpProfScope prof(ORP_PROF_DCN_BWD, st);
  hipError_t e = orp::fill_async(flags, 0, pl.scale_off + 256 - pl.flags_off, st);      // flags + range words
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(transpose_set_kernel, dim3(tin, batch), dim3(256), 0, st, TI);
  hipLaunchKernelGGL(compact_flags_kernel, dim3(1), dim3(1024), 0, st, flags, pl.total_chunks, const_cast<int*>(P.active));
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;

  if (need_input_grads) {
    static const int force = getenv("ORP_DCN_BWD_ATOMIC") ? atoi(getenv("ORP_DCN_BWD_ATOMIC")) : -1;   // dev aid: 1 / 0 force a path
    const bool use_atomics = force >= 0 ? force != 0 : (need_input_grads & ORP_DCN_BWD_SPARSE) != 0;
    if (f16) {
      hipLaunchKernelGGL(absmax_w_kernel, dim3(256), dim3(256), 0, st, weight, CH * CH * taps, scale_words + 1);
      hipLaunchKernelGGL(pack_wT16_kernel, dim3(1024), dim3(256), 0, st, weight, taps, scale_words + 1, const_cast<uint16_t*>(P.wT16),
                         reinterpret_cast<float*>(scale_words + 2));
    } else {
      hipLaunchKernelGGL(pack_wT_kernel, dim3(1024), dim3(256), 0, st, weight, taps, const_cast<float*>(P.wT));
    }
    const int per = (tiles + 7) >> 3;
    P.flags = flags;
    P.nregions = pl.nregions;
    P.keys = reinterpret_cast<unsigned*>(ws + pl.keys_in_off);
    P.vals = reinterpret_cast<unsigned*>(ws + pl.vals_in_off);
    P.rcount = reinterpret_cast<int*>(ws + pl.rcount_off);
    P.sorted_vals = reinterpret_cast<unsigned*>(ws + pl.vals_out_off);
    if (use_atomics) {
      P.G = nullptr;
      e = orp::fill_async(ws + pl.gx_begin, 0, pl.gx_bytes, st);
      if (e != hipSuccess) return (int)e;
      for (int i = 0; i < nlevels; i++) {                            // grad_offset of the chunks that are skipped is zero
        e = orp::fill_async(levels_host[i].grad_offset, 0, sizeof(float) * (size_t)batch * 2 * taps * pl.Ho[i] * pl.Wo[i], st);
        if (e != hipSuccess) return (int)e;
        if (P.lv[i].gmask) {
          e = orp::fill_async(P.lv[i].gmask, 0, sizeof(float) * (size_t)batch * taps * pl.Ho[i] * pl.Wo[i], st);
          if (e != hipSuccess) return (int)e;
        }
      }
      struct T1 { int unused; };
      struct T1h { int unused; };
      e = f16 ? orp::set_max_dynamic_lds_once<T1h>(reinterpret_cast<const void*>(&dcn_bwd_input_kernel<MT, false, true>), input_smem<MT>())
              : orp::set_max_dynamic_lds_once<T1>(reinterpret_cast<const void*>(&dcn_bwd_input_kernel<MT, false, false>), input_smem<MT>());
      if (e != hipSuccess) return (int)e;
      OrpProfScope prof_in(ORP_PROF_DCN_BWD_INPUT, st);
      if (f16) hipLaunchKernelGGL((dcn_bwd_input_kernel<MT, false, true>), dim3(per * 8), dim3(kThreads), input_smem<MT>(), st, P);
      else hipLaunchKernelGGL((dcn_bwd_input_kernel<MT, false, false>), dim3(per * 8), dim3(kThreads), input_smem<MT>(), st, P);
    } else {
      P.G = reinterpret_cast<float*>(ws + pl.G_off);
      // (region, sample) slots -> stable sort by region: every region's list in ascending sample order
      const long E = pl.nslots / 4;
      orp_prof_begin(ORP_PROF_DCN_BWD_SCATTER, st);          // pre-passes + scatter kernel (ends behind the scatter launch)
      hipLaunchKernelGGL(bin_samples_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, P);
      size_t cub_bytes = pl.cub_bytes;
      e = hipcub::DeviceRadixSort::SortPairs(ws + pl.cub_off, cub_bytes, P.keys, reinterpret_cast<unsigned*>(ws + pl.keys_out_off),
                                             P.vals, reinterpret_cast<unsigned*>(ws + pl.vals_out_off), (int)pl.nslots, 0,
                                             pl.key_bits, st);
      if (e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(region_bounds_kernel, dim3((unsigned)((pl.nslots + 255) / 256)), dim3(256), 0, st,
                         reinterpret_cast<const unsigned*>(ws + pl.keys_out_off), pl.nslots, pl.nregions, P.rcount);
      struct T2 { int unused; };
      struct T2h { int unused; };
      e = f16 ? orp::set_max_dynamic_lds_once<T2h>(reinterpret_cast<const void*>(&dcn_bwd_input_kernel<MT, true, true>), input_smem<MT>())
              : orp::set_max_dynamic_lds_once<T2>(reinterpret_cast<const void*>(&dcn_bwd_input_kernel<MT, true, false>), input_smem<MT>());
      if (e != hipSuccess) return (int)e;
      orp_prof_end(ORP_PROF_DCN_BWD_SCATTER, st);            // (paused around the GEMM kernel, which has its own slot)
      {
        OrpProfScope prof_in(ORP_PROF_DCN_BWD_INPUT, st);
        if (f16) hipLaunchKernelGGL((dcn_bwd_input_kernel<MT, true, true>), dim3(per * 8), dim3(kThreads), input_smem<MT>(), st, P);
        else hipLaunchKernelGGL((dcn_bwd_input_kernel<MT, true, false>), dim3(per * 8), dim3(kThreads), input_smem<MT>(), st, P);
      }
      orp_prof_begin(ORP_PROF_DCN_BWD_SCATTER, st);
      struct T3 { int unused; };
      e = orp::set_max_dynamic_lds_once<T3>(reinterpret_cast<const void*>(&dcn_bwd_scatter_kernel), scatter_smem());
      if (e != hipSuccess) return (int)e;
      const int rper = (pl.nregions + 7) >> 3;
      SampleDesc* desc = reinterpret_cast<SampleDesc*>(ws + pl.desc_off);
      hipLaunchKernelGGL(build_desc_kernel, dim3((unsigned)((pl.nslots + 255) / 256)), dim3(256), 0, st, P,
                         reinterpret_cast<const unsigned*>(ws + pl.keys_out_off), desc);
      hipLaunchKernelGGL(dcn_bwd_scatter_kernel, dim3(rper * 8), dim3(kScatterThreads), scatter_smem(), st, P, desc);
      orp_prof_end(ORP_PROF_DCN_BWD_SCATTER, st);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(transpose_set_kernel, dim3(tout, batch), dim3(256), 0, st, TO);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  if (grad_weight) {
    struct TW { int unused; };
    e = orp::set_max_dynamic_lds_once<TW>(reinterpret_cast<const void*>(&dcn_bwd_weight_kernel), weight_smem());
    if (e != hipSuccess) return (int)e;
    OrpProfScope prof_w(ORP_PROF_DCN_BWD_WEIGHT, st);
    int ns_w = pl.nsplit;
    if (w16) {
      struct TW16 { int unused; };
      e = orp::set_max_dynamic_lds_once<TW16>(reinterpret_cast<const void*>(&dcn_bwd_weight16_kernel), weight16_smem());
      if (e != hipSuccess) return (int)e;
      // the sampling table lives where kernel A left the G rows for A2 (both are done with them by now, in stream order)
      // and the two fp16 pieces of grad_out behind it (taps >= 2: both fit into the G rows' space)
      SampleTab* tab = reinterpret_cast<SampleTab*>(ws + pl.G_off);
      const long nsamp = (long)pl.total_chunks * 32 * taps;
      _Float16* go16 = reinterpret_cast<_Float16*>(ws + pl.G_off + align256(sizeof(SampleTab) * (size_t)nsamp));
      ns_w = pl.nsplit >= 2 ? pl.nsplit / 2 : 1;                       // two workgroups (column halves) per (split, tap)
      hipLaunchKernelGGL(sample_table_kernel, dim3((unsigned)((nsamp + 255) / 256)), dim3(256), 0, st, P, scale_words + 3, tab);
      hipLaunchKernelGGL(pack_go16_kernel, dim3((unsigned)pl.total_chunks * 4), dim3(256), 0, st, P, go16);
      hipLaunchKernelGGL(dcn_bwd_weight16_kernel, dim3(ns_w, taps, CH / CX16), dim3(kThreads), weight16_smem(), st, P, tab, go16, scale_words + 3, ns_w);
    } else {
      hipLaunchKernelGGL(dcn_bwd_weight_kernel, dim3(pl.nsplit, taps), dim3(kThreads), weight_smem(), st, P);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(reduce_partial_kernel, dim3(1024), dim3(256), 0, st, P.partial, ns_w, taps, grad_weight);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return ORP_OK;
}

}  // extern "C"
