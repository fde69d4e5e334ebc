// Development aid (GPU box), self-contained: a VMEM load that returns into a VGPR an MFMA has just read -- the split convolution
// kernel's in-place refill of its weight registers -- does not disturb that MFMA (mfma_war.hip part B: 6.5e9 refills, every
// accumulator exact), but it DISTURBS OTHER WAVES: a second kernel on another stream (the compare / mask / select sequence of
// sgpr_mask_probe.hip, 0 errors in 2e11 evaluations on its own and next to a library GEMM) gets the mask of its third select wrong in
// lanes 48..63 while the refilling kernel shares its CUs.  This program shows it without the library: aggressor = mfma_war.hip's
// part-B kernel (A from LDS in place / B from global memory in place / compiler-placed loads), optionally with GAP independent MFMA
// pairs between the MFMAs that read the registers and the refill; victim = the mask probe on a second stream.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_refill_victim mfma_refill_victim.hip && ./mfma_refill_victim
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bf8 splat(float v) {
  const __bf16 h = (__bf16)v;
  return bf8{h, h, h, h, h, h, h, h};
}

// ---- part B ---------------------------------------------------------------------------------------------------------------------
// MODE bit 0: A refilled in place from LDS, bit 1: B refilled in place from global memory.  The refill is ISSUED right behind the six
// MFMAs that read the registers (no wait), twelve more MFMAs on other registers follow (the matrix pipe stays backlogged while the
// loads land, as in the split kernel's next chunks), then the wave waits for the loads.  SAFE: the same data flow through
// compiler-placed loads into registers of its choice, behind a compiler-visible read of the accumulators.
constexpr int kThreads = 512;
template <int MODE, bool SAFE>
__global__ void __launch_bounds__(kThreads) war_probe(const uint4* __restrict__ gB, int iters, unsigned* __restrict__ bad, float* __restrict__ first_bad,
                                                      int pad_lds_dwords) {
  extern __shared__ uint4 lds[];                         // [2 tiles][64 lanes] A operands: tile t = splat(t + 1)
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 128) {
    const bf8 v = splat((float)((threadIdx.x >> 6) + 1));
    lds[threadIdx.x] = __builtin_bit_cast(uint4, v);
  }
  if (pad_lds_dwords < 0) lds[1000] = lds[0];            // (never: keeps the dynamic allocation)
  __syncthreads();
  // B tiles in global memory: tile t, lane l = splat(((l % 32) % 5 + 1) * (t + 1))
  const uint4* gb = gB + lane;
  const unsigned lds_addr = (unsigned)(lane * 16);
  floatx16 acc0 = floatx16{0}, acc1 = floatx16{0}, acc2 = floatx16{0}, acc3 = floatx16{0};
  const bf8 a_t[2] = {__builtin_bit_cast(bf8, lds[lane]), __builtin_bit_cast(bf8, lds[64 + lane])};
  const bf8 b_t[2] = {__builtin_bit_cast(bf8, gb[0]), __builtin_bit_cast(bf8, gb[64])};
  const bf8 c = splat(1.f), d = splat(1.f);
  bf8 a = a_t[0], b = b_t[0];
  for (int it = 0; it < iters; it++) {
    const int nt = (it + 1) & 1;                                               // the next iteration's tiles
    if (SAFE) {
      // the round-4 chunk: five products into one chain, one into the other
#pragma unroll
      for (int u = 0; u < 5; u++) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      float s = acc0[15] + acc1[15];
      asm volatile("" :: "v"(s));
      a = (MODE & 1) ? __builtin_bit_cast(bf8, lds[nt * 64 + lane]) : (nt ? a_t[1] : a_t[0]);
      b = (MODE & 2) ? __builtin_bit_cast(bf8, gb[nt * 64]) : (nt ? b_t[1] : b_t[0]);
#pragma unroll
      for (int u = 0; u < 6; u++) {
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, d, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, d, acc3, 0, 0, 0);
      }
    } else {
      // ONE asm statement, so that the refills land in exactly the registers the six MFMAs read (a tied operand of a separate
      // statement gets copied by the register allocator)
#define WMF(acc) "v_mfma_f32_32x32x16_bf16 %[" #acc "], %[a], %[b], %[" #acc "]\n\t"
#define WMF2 "v_mfma_f32_32x32x16_bf16 %[acc2], %[c], %[d], %[acc2]\n\tv_mfma_f32_32x32x16_bf16 %[acc3], %[c], %[d], %[acc3]\n\t"
      const uint4* pb = gb + nt * 64;
      const unsigned pa = lds_addr + nt * 1024;
      if (MODE == 3)
        asm volatile(WMF(acc1) WMF(acc1) WMF(acc1) WMF(acc1) WMF(acc1) WMF(acc0)
                     "global_load_dwordx4 %[b], %[pb], off\n\tds_read_b128 %[a], %[pa]\n\t"
                     WMF2 WMF2 WMF2 WMF2 WMF2 WMF2 "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\t"
                     : [acc0] "+v"(acc0), [acc1] "+v"(acc1), [acc2] "+v"(acc2), [acc3] "+v"(acc3), [a] "+v"(a), [b] "+v"(b)
                     : [pb] "v"(pb), [pa] "v"(pa), [c] "v"(c), [d] "v"(d) : "memory");
      else if (MODE == 1)
        asm volatile(WMF(acc1) WMF(acc1) WMF(acc1) WMF(acc1) WMF(acc1) WMF(acc0)
                     "ds_read_b128 %[a], %[pa]\n\t"
                     WMF2 WMF2 WMF2 WMF2 WMF2 WMF2 "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\t"
                     : [acc0] "+v"(acc0), [acc1] "+v"(acc1), [acc2] "+v"(acc2), [acc3] "+v"(acc3), [a] "+v"(a), [b] "+v"(b)
                     : [pb] "v"(pb), [pa] "v"(pa), [c] "v"(c), [d] "v"(d) : "memory");
      else
        asm volatile(WMF(acc1) WMF(acc1) WMF(acc1) WMF(acc1) WMF(acc1) WMF(acc0)
                     "global_load_dwordx4 %[b], %[pb], off\n\t"
                     WMF2 WMF2 WMF2 WMF2 WMF2 WMF2 "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\t"
                     : [acc0] "+v"(acc0), [acc1] "+v"(acc1), [acc2] "+v"(acc2), [acc3] "+v"(acc3), [a] "+v"(a), [b] "+v"(b)
                     : [pb] "v"(pb), [pa] "v"(pa), [c] "v"(c), [d] "v"(d) : "memory");
      if (!(MODE & 1)) a = nt ? a_t[1] : a_t[0];
      if (!(MODE & 2)) b = nt ? b_t[1] : b_t[0];
    }
  }
  // expected: iteration `it` uses tiles t = it & 1 on both sides: every element of D gains 16 * (t+1) * beta * (t+1) per MFMA
  const float beta = (float)((lane & 31) % 5 + 1);
  const int n0 = (iters + 1) / 2, n1 = iters / 2;                              // iterations on tile 0 / tile 1
  const float per = 16.f * beta * ((float)n0 * 1.f + (float)n1 * 4.f);
  const float want1 = 5.f * per, want0 = per, want2 = 16.f * 6.f * (float)iters;
  bool wrong = false;
#pragma unroll
  for (int r = 0; r < 16; r++) wrong |= (acc0[r] != want0) | (acc1[r] != want1) | (acc2[r] != want2) | (acc3[r] != want2);
  if (wrong) {
    const unsigned k = atomicAdd(bad, 1u);
    if (k < 8) { first_bad[k * 4] = acc0[0]; first_bad[k * 4 + 1] = want0; first_bad[k * 4 + 2] = acc1[0]; first_bad[k * 4 + 3] = want1; }
  }
}


// ---- the victim (sgpr_mask_probe.hip) ---------------------------------------------------------------------------------------------


constexpr int kVictimThreads = 512;
#undef PRE
#undef POST
#define PRE ""
#define POST ""
__global__ void __launch_bounds__(kVictimThreads) mask_probe_v0(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 12345u;
  const __bf16 one = (__bf16)1.f;
  const bf8 a = bf8{one, one, one, one, one, one, one, one};
  floatx16 acc = floatx16{0};
  unsigned nbad[4] = {0, 0, 0, 0};
  const int Hm1 = H - 1, Wm1 = W - 1;
  for (int it = 0; it < iters; it++) {
    // sample coordinates inside (-1, H) x (-1, W): mostly interior, some on every border
    seed = seed * 1664525u + 1013904223u;
    const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)H + 0.998f);
    seed = seed * 1664525u + 1013904223u;
    const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)W + 0.998f);
    float wx, wy, wz, ww;
    // v16 = h_im, v17 = w_im (a register pair for the packed instructions); the block is the compiler's own code for
    // orp_dcn_split.hip's coefficient table (%bb.20 of dcn_fwd_split_kernel<1, 6, true, false>), registers as there
    asm volatile(
        "v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\t"
        "s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\ts_mov_b64 s[14:15], 0\n\t"
        "v_floor_f32_e32 v2, v16\n\t"
        "v_floor_f32_e32 v3, v17\n\t"
        "v_cvt_i32_f32_e32 v6, v3\n\t"
        "v_cvt_i32_f32_e32 v7, v2\n\t"
        "v_cvt_f32_i32_e32 v3, v6\n\t"
        "v_cvt_f32_i32_e32 v2, v7\n\t"
        "v_or_b32_e32 v19, v7, v6\n\t"
        "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\t"
        "v_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"
        "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_lt_i32_e32 vcc, -1, v19\n\t"
        "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
        "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\t"
        "v_mul_f32_e32 v2, v4, v5\n\t"
        "v_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"
        "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t"
        "v_pk_mul_f32 v[4:5], v[8:9], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "s_and_b64 vcc, s[6:7], s[8:9]\n\t"
        "v_cndmask_b32_e32 v3, 0, v5, vcc\n\t"
        PRE "s_and_b64 vcc, s[10:11], s[4:5]\n\t" POST
        "v_mul_f32_e32 v5, v8, v9\n\t"
        "s_and_b64 s[12:13], s[10:11], s[8:9]\n\t"
        "v_cndmask_b32_e32 v4, 0, v4, vcc\n\t"
        "s_andn2_b64 vcc, exec, s[14:15]\n\t"
        "v_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"
        : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)
        : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13",
          "s14", "s15", "vcc");
    // select-free evaluation of the same values: the borders as 0 / 1 factors (exact)
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
    const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(Hm1 - h_low, 1);
    const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(Wm1 - w_low, 1);
    const float ex = (hh * hw) * (t_ok * l_ok), ey = (hh * lw) * (t_ok * r_ok), ez = (lh * hw) * (b_ok * l_ok), ew = (lh * lw) * (b_ok * r_ok);
    nbad[0] += (wx != ex); nbad[1] += (wy != ey); nbad[2] += (wz != ez); nbad[3] += (ww != ew);
    // a burst of MFMAs now and then, out of step between the waves of a SIMD (the other workgroup of the CU is in its K loop
    // while this one builds its table)
    if (mfma_burst && ((it + wave * 7) & 15) == 0) {
#pragma unroll
      for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
  }
  for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad_lane[lane * 4 + k], nbad[k]);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}


#undef PRE
#undef POST
#define PRE ""
#define POST ""
__global__ void __launch_bounds__(kVictimThreads) mask_probe_v7(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 12345u;
  const __bf16 one = (__bf16)1.f;
  const bf8 a = bf8{one, one, one, one, one, one, one, one};
  floatx16 acc = floatx16{0};
  unsigned nbad[4] = {0, 0, 0, 0};
  const int Hm1 = H - 1, Wm1 = W - 1;
  for (int it = 0; it < iters; it++) {
    // sample coordinates inside (-1, H) x (-1, W): mostly interior, some on every border
    seed = seed * 1664525u + 1013904223u;
    const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)H + 0.998f);
    seed = seed * 1664525u + 1013904223u;
    const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)W + 0.998f);
    float wx, wy, wz, ww;
    // v16 = h_im, v17 = w_im (a register pair for the packed instructions); the block is the compiler's own code for
    // orp_dcn_split.hip's coefficient table (%bb.20 of dcn_fwd_split_kernel<1, 6, true, false>), registers as there
    asm volatile(
        "v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\t"
        "s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\ts_mov_b64 s[14:15], 0\n\t"
        "v_floor_f32_e32 v2, v16\n\t"
        "v_floor_f32_e32 v3, v17\n\t"
        "v_cvt_i32_f32_e32 v6, v3\n\t"
        "v_cvt_i32_f32_e32 v7, v2\n\t"
        "v_cvt_f32_i32_e32 v3, v6\n\t"
        "v_cvt_f32_i32_e32 v2, v7\n\t"
        "v_or_b32_e32 v19, v7, v6\n\t"
        "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\t"
        "v_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"
        "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_lt_i32_e32 vcc, -1, v19\n\t"
        "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
        "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\t"
        "v_mul_f32_e32 v2, v4, v5\n\t"
        "v_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"
        "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t"
        "v_pk_mul_f32 v[20:21], v[8:9], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32_e32 v4, v20\n\tv_mov_b32_e32 v5, v21\n\t"
        "s_and_b64 vcc, s[6:7], s[8:9]\n\t"
        "v_cndmask_b32_e32 v3, 0, v5, vcc\n\t"
        PRE "s_and_b64 vcc, s[10:11], s[4:5]\n\t" POST
        "v_mul_f32_e32 v5, v8, v9\n\t"
        "s_and_b64 s[12:13], s[10:11], s[8:9]\n\t"
        "v_cndmask_b32_e32 v4, 0, v4, vcc\n\t"
        "s_andn2_b64 vcc, exec, s[14:15]\n\t"
        "v_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"
        : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)
        : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)
        : "v20", "v21", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13",
          "s14", "s15", "vcc");
    // select-free evaluation of the same values: the borders as 0 / 1 factors (exact)
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
    const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(Hm1 - h_low, 1);
    const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(Wm1 - w_low, 1);
    const float ex = (hh * hw) * (t_ok * l_ok), ey = (hh * lw) * (t_ok * r_ok), ez = (lh * hw) * (b_ok * l_ok), ew = (lh * lw) * (b_ok * r_ok);
    nbad[0] += (wx != ex); nbad[1] += (wy != ey); nbad[2] += (wz != ez); nbad[3] += (ww != ew);
    // a burst of MFMAs now and then, out of step between the waves of a SIMD (the other workgroup of the CU is in its K loop
    // while this one builds its table)
    if (mfma_burst && ((it + wave * 7) & 15) == 0) {
#pragma unroll
      for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
  }
  for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad_lane[lane * 4 + k], nbad[k]);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}


#undef PRE
#undef POST
#define PRE ""
#define POST ""
__global__ void __launch_bounds__(kVictimThreads) mask_probe_v8(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 12345u;
  const __bf16 one = (__bf16)1.f;
  const bf8 a = bf8{one, one, one, one, one, one, one, one};
  floatx16 acc = floatx16{0};
  unsigned nbad[4] = {0, 0, 0, 0};
  const int Hm1 = H - 1, Wm1 = W - 1;
  for (int it = 0; it < iters; it++) {
    // sample coordinates inside (-1, H) x (-1, W): mostly interior, some on every border
    seed = seed * 1664525u + 1013904223u;
    const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)H + 0.998f);
    seed = seed * 1664525u + 1013904223u;
    const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)W + 0.998f);
    float wx, wy, wz, ww;
    // v16 = h_im, v17 = w_im (a register pair for the packed instructions); the block is the compiler's own code for
    // orp_dcn_split.hip's coefficient table (%bb.20 of dcn_fwd_split_kernel<1, 6, true, false>), registers as there
    asm volatile(
        "v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\t"
        "s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\ts_mov_b64 s[14:15], 0\n\t"
        "v_floor_f32_e32 v2, v16\n\t"
        "v_floor_f32_e32 v3, v17\n\t"
        "v_cvt_i32_f32_e32 v6, v3\n\t"
        "v_cvt_i32_f32_e32 v7, v2\n\t"
        "v_cvt_f32_i32_e32 v3, v6\n\t"
        "v_cvt_f32_i32_e32 v2, v7\n\t"
        "v_or_b32_e32 v19, v7, v6\n\t"
        "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\t"
        "v_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"
        "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_lt_i32_e32 vcc, -1, v19\n\t"
        "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
        "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\t"
        "v_mul_f32_e32 v2, v4, v5\n\t"
        "v_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"
        "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t"
        "s_nop 3\n\t" "v_pk_mul_f32 v[4:5], v[8:9], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "s_and_b64 vcc, s[6:7], s[8:9]\n\t"
        "v_cndmask_b32_e32 v3, 0, v5, vcc\n\t"
        PRE "s_and_b64 vcc, s[10:11], s[4:5]\n\t" POST
        "v_mul_f32_e32 v5, v8, v9\n\t"
        "s_and_b64 s[12:13], s[10:11], s[8:9]\n\t"
        "v_cndmask_b32_e32 v4, 0, v4, vcc\n\t"
        "s_andn2_b64 vcc, exec, s[14:15]\n\t"
        "v_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"
        : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)
        : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13",
          "s14", "s15", "vcc");
    // select-free evaluation of the same values: the borders as 0 / 1 factors (exact)
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
    const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(Hm1 - h_low, 1);
    const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(Wm1 - w_low, 1);
    const float ex = (hh * hw) * (t_ok * l_ok), ey = (hh * lw) * (t_ok * r_ok), ez = (lh * hw) * (b_ok * l_ok), ew = (lh * lw) * (b_ok * r_ok);
    nbad[0] += (wx != ex); nbad[1] += (wy != ey); nbad[2] += (wz != ez); nbad[3] += (ww != ew);
    // a burst of MFMAs now and then, out of step between the waves of a SIMD (the other workgroup of the CU is in its K loop
    // while this one builds its table)
    if (mfma_burst && ((it + wave * 7) & 15) == 0) {
#pragma unroll
      for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
  }
  for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad_lane[lane * 4 + k], nbad[k]);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}


#undef PRE
#undef POST
#define PRE ""
#define POST ""
__global__ void __launch_bounds__(kVictimThreads) mask_probe_v9(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 12345u;
  const __bf16 one = (__bf16)1.f;
  const bf8 a = bf8{one, one, one, one, one, one, one, one};
  floatx16 acc = floatx16{0};
  unsigned nbad[4] = {0, 0, 0, 0};
  const int Hm1 = H - 1, Wm1 = W - 1;
  for (int it = 0; it < iters; it++) {
    // sample coordinates inside (-1, H) x (-1, W): mostly interior, some on every border
    seed = seed * 1664525u + 1013904223u;
    const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)H + 0.998f);
    seed = seed * 1664525u + 1013904223u;
    const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)W + 0.998f);
    float wx, wy, wz, ww;
    // v16 = h_im, v17 = w_im (a register pair for the packed instructions); the block is the compiler's own code for
    // orp_dcn_split.hip's coefficient table (%bb.20 of dcn_fwd_split_kernel<1, 6, true, false>), registers as there
    asm volatile(
        "v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\t"
        "s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\ts_mov_b64 s[14:15], 0\n\t"
        "v_floor_f32_e32 v2, v16\n\t"
        "v_floor_f32_e32 v3, v17\n\t"
        "v_cvt_i32_f32_e32 v6, v3\n\t"
        "v_cvt_i32_f32_e32 v7, v2\n\t"
        "v_cvt_f32_i32_e32 v3, v6\n\t"
        "v_cvt_f32_i32_e32 v2, v7\n\t"
        "v_or_b32_e32 v19, v7, v6\n\t"
        "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\t"
        "v_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"
        "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_lt_i32_e32 vcc, -1, v19\n\t"
        "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
        "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\t"
        "v_mul_f32_e32 v2, v4, v5\n\t"
        "v_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"
        "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t"
        "v_pk_mul_f32 v[4:5], v[8:9], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]\n\t" "s_nop 3\n\t"
        "s_and_b64 vcc, s[6:7], s[8:9]\n\t"
        "v_cndmask_b32_e32 v3, 0, v5, vcc\n\t"
        PRE "s_and_b64 vcc, s[10:11], s[4:5]\n\t" POST
        "v_mul_f32_e32 v5, v8, v9\n\t"
        "s_and_b64 s[12:13], s[10:11], s[8:9]\n\t"
        "v_cndmask_b32_e32 v4, 0, v4, vcc\n\t"
        "s_andn2_b64 vcc, exec, s[14:15]\n\t"
        "v_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"
        : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)
        : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13",
          "s14", "s15", "vcc");
    // select-free evaluation of the same values: the borders as 0 / 1 factors (exact)
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
    const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(Hm1 - h_low, 1);
    const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(Wm1 - w_low, 1);
    const float ex = (hh * hw) * (t_ok * l_ok), ey = (hh * lw) * (t_ok * r_ok), ez = (lh * hw) * (b_ok * l_ok), ew = (lh * lw) * (b_ok * r_ok);
    nbad[0] += (wx != ex); nbad[1] += (wy != ey); nbad[2] += (wz != ez); nbad[3] += (ww != ew);
    // a burst of MFMAs now and then, out of step between the waves of a SIMD (the other workgroup of the CU is in its K loop
    // while this one builds its table)
    if (mfma_burst && ((it + wave * 7) & 15) == 0) {
#pragma unroll
      for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
  }
  for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad_lane[lane * 4 + k], nbad[k]);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}


#undef PRE
#undef POST
#define PRE ""
#define POST ""
__global__ void __launch_bounds__(kVictimThreads) mask_probe_v10(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 12345u;
  const __bf16 one = (__bf16)1.f;
  const bf8 a = bf8{one, one, one, one, one, one, one, one};
  floatx16 acc = floatx16{0};
  unsigned nbad[4] = {0, 0, 0, 0};
  const int Hm1 = H - 1, Wm1 = W - 1;
  for (int it = 0; it < iters; it++) {
    // sample coordinates inside (-1, H) x (-1, W): mostly interior, some on every border
    seed = seed * 1664525u + 1013904223u;
    const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)H + 0.998f);
    seed = seed * 1664525u + 1013904223u;
    const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)W + 0.998f);
    float wx, wy, wz, ww;
    // v16 = h_im, v17 = w_im (a register pair for the packed instructions); the block is the compiler's own code for
    // orp_dcn_split.hip's coefficient table (%bb.20 of dcn_fwd_split_kernel<1, 6, true, false>), registers as there
    asm volatile(
        "v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\t"
        "s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\ts_mov_b64 s[14:15], 0\n\t"
        "v_floor_f32_e32 v2, v16\n\t"
        "v_floor_f32_e32 v3, v17\n\t"
        "v_cvt_i32_f32_e32 v6, v3\n\t"
        "v_cvt_i32_f32_e32 v7, v2\n\t"
        "v_cvt_f32_i32_e32 v3, v6\n\t"
        "v_cvt_f32_i32_e32 v2, v7\n\t"
        "v_or_b32_e32 v19, v7, v6\n\t"
        "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\t"
        "v_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"
        "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_lt_i32_e32 vcc, -1, v19\n\t"
        "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
        "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\t"
        "v_mul_f32_e32 v2, v4, v5\n\t"
        "v_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"
        "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t"
        "v_mov_b32_e32 v20, v5\n\tv_mov_b32_e32 v21, v4\n\tv_pk_mul_f32 v[4:5], v[8:9], v[20:21]\n\t"
        "s_and_b64 vcc, s[6:7], s[8:9]\n\t"
        "v_cndmask_b32_e32 v3, 0, v5, vcc\n\t"
        PRE "s_and_b64 vcc, s[10:11], s[4:5]\n\t" POST
        "v_mul_f32_e32 v5, v8, v9\n\t"
        "s_and_b64 s[12:13], s[10:11], s[8:9]\n\t"
        "v_cndmask_b32_e32 v4, 0, v4, vcc\n\t"
        "s_andn2_b64 vcc, exec, s[14:15]\n\t"
        "v_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"
        : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)
        : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)
        : "v20", "v21", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13",
          "s14", "s15", "vcc");
    // select-free evaluation of the same values: the borders as 0 / 1 factors (exact)
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
    const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(Hm1 - h_low, 1);
    const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(Wm1 - w_low, 1);
    const float ex = (hh * hw) * (t_ok * l_ok), ey = (hh * lw) * (t_ok * r_ok), ez = (lh * hw) * (b_ok * l_ok), ew = (lh * lw) * (b_ok * r_ok);
    nbad[0] += (wx != ex); nbad[1] += (wy != ey); nbad[2] += (wz != ez); nbad[3] += (ww != ew);
    // a burst of MFMAs now and then, out of step between the waves of a SIMD (the other workgroup of the CU is in its K loop
    // while this one builds its table)
    if (mfma_burst && ((it + wave * 7) & 15) == 0) {
#pragma unroll
      for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
  }
  for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad_lane[lane * 4 + k], nbad[k]);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}


#undef PRE
#undef POST
#define PRE ""
#define POST ""
__global__ void __launch_bounds__(kVictimThreads) mask_probe_v6(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 12345u;
  const __bf16 one = (__bf16)1.f;
  const bf8 a = bf8{one, one, one, one, one, one, one, one};
  floatx16 acc = floatx16{0};
  unsigned nbad[4] = {0, 0, 0, 0};
  const int Hm1 = H - 1, Wm1 = W - 1;
  for (int it = 0; it < iters; it++) {
    // sample coordinates inside (-1, H) x (-1, W): mostly interior, some on every border
    seed = seed * 1664525u + 1013904223u;
    const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)H + 0.998f);
    seed = seed * 1664525u + 1013904223u;
    const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)W + 0.998f);
    float wx, wy, wz, ww;
    // v16 = h_im, v17 = w_im (a register pair for the packed instructions); the block is the compiler's own code for
    // orp_dcn_split.hip's coefficient table (%bb.20 of dcn_fwd_split_kernel<1, 6, true, false>), registers as there
    asm volatile(
        "v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\t"
        "s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\ts_mov_b64 s[14:15], 0\n\t"
        "v_floor_f32_e32 v2, v16\n\t"
        "v_floor_f32_e32 v3, v17\n\t"
        "v_cvt_i32_f32_e32 v6, v3\n\t"
        "v_cvt_i32_f32_e32 v7, v2\n\t"
        "v_cvt_f32_i32_e32 v3, v6\n\t"
        "v_cvt_f32_i32_e32 v2, v7\n\t"
        "v_or_b32_e32 v19, v7, v6\n\t"
        "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\t"
        "v_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"
        "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_lt_i32_e32 vcc, -1, v19\n\t"
        "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
        "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\t"
        "v_mul_f32_e32 v2, v4, v5\n\t"
        "v_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"
        "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t"
        "v_mul_f32_e32 v19, v8, v5\n\tv_mul_f32_e32 v5, v9, v4\n\tv_mov_b32_e32 v4, v19\n\t"
        "s_and_b64 vcc, s[6:7], s[8:9]\n\t"
        "v_cndmask_b32_e32 v3, 0, v5, vcc\n\t"
        PRE "s_and_b64 vcc, s[10:11], s[4:5]\n\t" POST
        "v_mul_f32_e32 v5, v8, v9\n\t"
        "s_and_b64 s[12:13], s[10:11], s[8:9]\n\t"
        "v_cndmask_b32_e32 v4, 0, v4, vcc\n\t"
        "s_andn2_b64 vcc, exec, s[14:15]\n\t"
        "v_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"
        : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)
        : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13",
          "s14", "s15", "vcc");
    // select-free evaluation of the same values: the borders as 0 / 1 factors (exact)
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
    const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(Hm1 - h_low, 1);
    const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(Wm1 - w_low, 1);
    const float ex = (hh * hw) * (t_ok * l_ok), ey = (hh * lw) * (t_ok * r_ok), ez = (lh * hw) * (b_ok * l_ok), ew = (lh * lw) * (b_ok * r_ok);
    nbad[0] += (wx != ex); nbad[1] += (wy != ey); nbad[2] += (wz != ez); nbad[3] += (ww != ew);
    // a burst of MFMAs now and then, out of step between the waves of a SIMD (the other workgroup of the CU is in its K loop
    // while this one builds its table)
    if (mfma_burst && ((it + wave * 7) & 15) == 0) {
#pragma unroll
      for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
  }
  for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad_lane[lane * 4 + k], nbad[k]);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}


#undef PRE
#undef POST
#define PRE "s_nop 0\n\t"
#define POST ""
__global__ void __launch_bounds__(kVictimThreads) mask_probe_v1(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 12345u;
  const __bf16 one = (__bf16)1.f;
  const bf8 a = bf8{one, one, one, one, one, one, one, one};
  floatx16 acc = floatx16{0};
  unsigned nbad[4] = {0, 0, 0, 0};
  const int Hm1 = H - 1, Wm1 = W - 1;
  for (int it = 0; it < iters; it++) {
    // sample coordinates inside (-1, H) x (-1, W): mostly interior, some on every border
    seed = seed * 1664525u + 1013904223u;
    const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)H + 0.998f);
    seed = seed * 1664525u + 1013904223u;
    const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)W + 0.998f);
    float wx, wy, wz, ww;
    // v16 = h_im, v17 = w_im (a register pair for the packed instructions); the block is the compiler's own code for
    // orp_dcn_split.hip's coefficient table (%bb.20 of dcn_fwd_split_kernel<1, 6, true, false>), registers as there
    asm volatile(
        "v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\t"
        "s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\ts_mov_b64 s[14:15], 0\n\t"
        "v_floor_f32_e32 v2, v16\n\t"
        "v_floor_f32_e32 v3, v17\n\t"
        "v_cvt_i32_f32_e32 v6, v3\n\t"
        "v_cvt_i32_f32_e32 v7, v2\n\t"
        "v_cvt_f32_i32_e32 v3, v6\n\t"
        "v_cvt_f32_i32_e32 v2, v7\n\t"
        "v_or_b32_e32 v19, v7, v6\n\t"
        "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\t"
        "v_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"
        "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_lt_i32_e32 vcc, -1, v19\n\t"
        "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
        "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\t"
        "v_mul_f32_e32 v2, v4, v5\n\t"
        "v_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"
        "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t"
        "v_pk_mul_f32 v[4:5], v[8:9], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "s_and_b64 vcc, s[6:7], s[8:9]\n\t"
        "v_cndmask_b32_e32 v3, 0, v5, vcc\n\t"
        PRE "s_and_b64 vcc, s[10:11], s[4:5]\n\t" POST
        "v_mul_f32_e32 v5, v8, v9\n\t"
        "s_and_b64 s[12:13], s[10:11], s[8:9]\n\t"
        "v_cndmask_b32_e32 v4, 0, v4, vcc\n\t"
        "s_andn2_b64 vcc, exec, s[14:15]\n\t"
        "v_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"
        : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)
        : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13",
          "s14", "s15", "vcc");
    // select-free evaluation of the same values: the borders as 0 / 1 factors (exact)
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
    const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(Hm1 - h_low, 1);
    const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(Wm1 - w_low, 1);
    const float ex = (hh * hw) * (t_ok * l_ok), ey = (hh * lw) * (t_ok * r_ok), ez = (lh * hw) * (b_ok * l_ok), ew = (lh * lw) * (b_ok * r_ok);
    nbad[0] += (wx != ex); nbad[1] += (wy != ey); nbad[2] += (wz != ez); nbad[3] += (ww != ew);
    // a burst of MFMAs now and then, out of step between the waves of a SIMD (the other workgroup of the CU is in its K loop
    // while this one builds its table)
    if (mfma_burst && ((it + wave * 7) & 15) == 0) {
#pragma unroll
      for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
  }
  for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad_lane[lane * 4 + k], nbad[k]);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}


#undef PRE
#undef POST
#define PRE "s_nop 1\n\t"
#define POST ""
__global__ void __launch_bounds__(kVictimThreads) mask_probe_v2(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 12345u;
  const __bf16 one = (__bf16)1.f;
  const bf8 a = bf8{one, one, one, one, one, one, one, one};
  floatx16 acc = floatx16{0};
  unsigned nbad[4] = {0, 0, 0, 0};
  const int Hm1 = H - 1, Wm1 = W - 1;
  for (int it = 0; it < iters; it++) {
    // sample coordinates inside (-1, H) x (-1, W): mostly interior, some on every border
    seed = seed * 1664525u + 1013904223u;
    const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)H + 0.998f);
    seed = seed * 1664525u + 1013904223u;
    const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)W + 0.998f);
    float wx, wy, wz, ww;
    // v16 = h_im, v17 = w_im (a register pair for the packed instructions); the block is the compiler's own code for
    // orp_dcn_split.hip's coefficient table (%bb.20 of dcn_fwd_split_kernel<1, 6, true, false>), registers as there
    asm volatile(
        "v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\t"
        "s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\ts_mov_b64 s[14:15], 0\n\t"
        "v_floor_f32_e32 v2, v16\n\t"
        "v_floor_f32_e32 v3, v17\n\t"
        "v_cvt_i32_f32_e32 v6, v3\n\t"
        "v_cvt_i32_f32_e32 v7, v2\n\t"
        "v_cvt_f32_i32_e32 v3, v6\n\t"
        "v_cvt_f32_i32_e32 v2, v7\n\t"
        "v_or_b32_e32 v19, v7, v6\n\t"
        "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\t"
        "v_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"
        "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_lt_i32_e32 vcc, -1, v19\n\t"
        "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
        "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\t"
        "v_mul_f32_e32 v2, v4, v5\n\t"
        "v_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"
        "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t"
        "v_pk_mul_f32 v[4:5], v[8:9], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "s_and_b64 vcc, s[6:7], s[8:9]\n\t"
        "v_cndmask_b32_e32 v3, 0, v5, vcc\n\t"
        PRE "s_and_b64 vcc, s[10:11], s[4:5]\n\t" POST
        "v_mul_f32_e32 v5, v8, v9\n\t"
        "s_and_b64 s[12:13], s[10:11], s[8:9]\n\t"
        "v_cndmask_b32_e32 v4, 0, v4, vcc\n\t"
        "s_andn2_b64 vcc, exec, s[14:15]\n\t"
        "v_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"
        : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)
        : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13",
          "s14", "s15", "vcc");
    // select-free evaluation of the same values: the borders as 0 / 1 factors (exact)
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
    const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(Hm1 - h_low, 1);
    const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(Wm1 - w_low, 1);
    const float ex = (hh * hw) * (t_ok * l_ok), ey = (hh * lw) * (t_ok * r_ok), ez = (lh * hw) * (b_ok * l_ok), ew = (lh * lw) * (b_ok * r_ok);
    nbad[0] += (wx != ex); nbad[1] += (wy != ey); nbad[2] += (wz != ez); nbad[3] += (ww != ew);
    // a burst of MFMAs now and then, out of step between the waves of a SIMD (the other workgroup of the CU is in its K loop
    // while this one builds its table)
    if (mfma_burst && ((it + wave * 7) & 15) == 0) {
#pragma unroll
      for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
  }
  for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad_lane[lane * 4 + k], nbad[k]);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}


#undef PRE
#undef POST
#define PRE "s_nop 3\n\t"
#define POST ""
__global__ void __launch_bounds__(kVictimThreads) mask_probe_v3(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 12345u;
  const __bf16 one = (__bf16)1.f;
  const bf8 a = bf8{one, one, one, one, one, one, one, one};
  floatx16 acc = floatx16{0};
  unsigned nbad[4] = {0, 0, 0, 0};
  const int Hm1 = H - 1, Wm1 = W - 1;
  for (int it = 0; it < iters; it++) {
    // sample coordinates inside (-1, H) x (-1, W): mostly interior, some on every border
    seed = seed * 1664525u + 1013904223u;
    const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)H + 0.998f);
    seed = seed * 1664525u + 1013904223u;
    const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)W + 0.998f);
    float wx, wy, wz, ww;
    // v16 = h_im, v17 = w_im (a register pair for the packed instructions); the block is the compiler's own code for
    // orp_dcn_split.hip's coefficient table (%bb.20 of dcn_fwd_split_kernel<1, 6, true, false>), registers as there
    asm volatile(
        "v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\t"
        "s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\ts_mov_b64 s[14:15], 0\n\t"
        "v_floor_f32_e32 v2, v16\n\t"
        "v_floor_f32_e32 v3, v17\n\t"
        "v_cvt_i32_f32_e32 v6, v3\n\t"
        "v_cvt_i32_f32_e32 v7, v2\n\t"
        "v_cvt_f32_i32_e32 v3, v6\n\t"
        "v_cvt_f32_i32_e32 v2, v7\n\t"
        "v_or_b32_e32 v19, v7, v6\n\t"
        "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\t"
        "v_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"
        "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_lt_i32_e32 vcc, -1, v19\n\t"
        "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
        "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\t"
        "v_mul_f32_e32 v2, v4, v5\n\t"
        "v_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"
        "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t"
        "v_pk_mul_f32 v[4:5], v[8:9], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "s_and_b64 vcc, s[6:7], s[8:9]\n\t"
        "v_cndmask_b32_e32 v3, 0, v5, vcc\n\t"
        PRE "s_and_b64 vcc, s[10:11], s[4:5]\n\t" POST
        "v_mul_f32_e32 v5, v8, v9\n\t"
        "s_and_b64 s[12:13], s[10:11], s[8:9]\n\t"
        "v_cndmask_b32_e32 v4, 0, v4, vcc\n\t"
        "s_andn2_b64 vcc, exec, s[14:15]\n\t"
        "v_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"
        : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)
        : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13",
          "s14", "s15", "vcc");
    // select-free evaluation of the same values: the borders as 0 / 1 factors (exact)
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
    const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(Hm1 - h_low, 1);
    const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(Wm1 - w_low, 1);
    const float ex = (hh * hw) * (t_ok * l_ok), ey = (hh * lw) * (t_ok * r_ok), ez = (lh * hw) * (b_ok * l_ok), ew = (lh * lw) * (b_ok * r_ok);
    nbad[0] += (wx != ex); nbad[1] += (wy != ey); nbad[2] += (wz != ez); nbad[3] += (ww != ew);
    // a burst of MFMAs now and then, out of step between the waves of a SIMD (the other workgroup of the CU is in its K loop
    // while this one builds its table)
    if (mfma_burst && ((it + wave * 7) & 15) == 0) {
#pragma unroll
      for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
  }
  for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad_lane[lane * 4 + k], nbad[k]);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}


#undef PRE
#undef POST
#define PRE "s_nop 7\n\t"
#define POST ""
__global__ void __launch_bounds__(kVictimThreads) mask_probe_v4(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 12345u;
  const __bf16 one = (__bf16)1.f;
  const bf8 a = bf8{one, one, one, one, one, one, one, one};
  floatx16 acc = floatx16{0};
  unsigned nbad[4] = {0, 0, 0, 0};
  const int Hm1 = H - 1, Wm1 = W - 1;
  for (int it = 0; it < iters; it++) {
    // sample coordinates inside (-1, H) x (-1, W): mostly interior, some on every border
    seed = seed * 1664525u + 1013904223u;
    const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)H + 0.998f);
    seed = seed * 1664525u + 1013904223u;
    const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)W + 0.998f);
    float wx, wy, wz, ww;
    // v16 = h_im, v17 = w_im (a register pair for the packed instructions); the block is the compiler's own code for
    // orp_dcn_split.hip's coefficient table (%bb.20 of dcn_fwd_split_kernel<1, 6, true, false>), registers as there
    asm volatile(
        "v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\t"
        "s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\ts_mov_b64 s[14:15], 0\n\t"
        "v_floor_f32_e32 v2, v16\n\t"
        "v_floor_f32_e32 v3, v17\n\t"
        "v_cvt_i32_f32_e32 v6, v3\n\t"
        "v_cvt_i32_f32_e32 v7, v2\n\t"
        "v_cvt_f32_i32_e32 v3, v6\n\t"
        "v_cvt_f32_i32_e32 v2, v7\n\t"
        "v_or_b32_e32 v19, v7, v6\n\t"
        "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\t"
        "v_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"
        "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_lt_i32_e32 vcc, -1, v19\n\t"
        "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
        "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\t"
        "v_mul_f32_e32 v2, v4, v5\n\t"
        "v_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"
        "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t"
        "v_pk_mul_f32 v[4:5], v[8:9], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "s_and_b64 vcc, s[6:7], s[8:9]\n\t"
        "v_cndmask_b32_e32 v3, 0, v5, vcc\n\t"
        PRE "s_and_b64 vcc, s[10:11], s[4:5]\n\t" POST
        "v_mul_f32_e32 v5, v8, v9\n\t"
        "s_and_b64 s[12:13], s[10:11], s[8:9]\n\t"
        "v_cndmask_b32_e32 v4, 0, v4, vcc\n\t"
        "s_andn2_b64 vcc, exec, s[14:15]\n\t"
        "v_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"
        : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)
        : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13",
          "s14", "s15", "vcc");
    // select-free evaluation of the same values: the borders as 0 / 1 factors (exact)
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
    const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(Hm1 - h_low, 1);
    const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(Wm1 - w_low, 1);
    const float ex = (hh * hw) * (t_ok * l_ok), ey = (hh * lw) * (t_ok * r_ok), ez = (lh * hw) * (b_ok * l_ok), ew = (lh * lw) * (b_ok * r_ok);
    nbad[0] += (wx != ex); nbad[1] += (wy != ey); nbad[2] += (wz != ez); nbad[3] += (ww != ew);
    // a burst of MFMAs now and then, out of step between the waves of a SIMD (the other workgroup of the CU is in its K loop
    // while this one builds its table)
    if (mfma_burst && ((it + wave * 7) & 15) == 0) {
#pragma unroll
      for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
  }
  for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad_lane[lane * 4 + k], nbad[k]);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}


#undef PRE
#undef POST
#define PRE ""
#define POST "s_nop 7\n\t"
__global__ void __launch_bounds__(kVictimThreads) mask_probe_v5(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 12345u;
  const __bf16 one = (__bf16)1.f;
  const bf8 a = bf8{one, one, one, one, one, one, one, one};
  floatx16 acc = floatx16{0};
  unsigned nbad[4] = {0, 0, 0, 0};
  const int Hm1 = H - 1, Wm1 = W - 1;
  for (int it = 0; it < iters; it++) {
    // sample coordinates inside (-1, H) x (-1, W): mostly interior, some on every border
    seed = seed * 1664525u + 1013904223u;
    const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)H + 0.998f);
    seed = seed * 1664525u + 1013904223u;
    const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)W + 0.998f);
    float wx, wy, wz, ww;
    // v16 = h_im, v17 = w_im (a register pair for the packed instructions); the block is the compiler's own code for
    // orp_dcn_split.hip's coefficient table (%bb.20 of dcn_fwd_split_kernel<1, 6, true, false>), registers as there
    asm volatile(
        "v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\t"
        "s_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\ts_mov_b64 s[14:15], 0\n\t"
        "v_floor_f32_e32 v2, v16\n\t"
        "v_floor_f32_e32 v3, v17\n\t"
        "v_cvt_i32_f32_e32 v6, v3\n\t"
        "v_cvt_i32_f32_e32 v7, v2\n\t"
        "v_cvt_f32_i32_e32 v3, v6\n\t"
        "v_cvt_f32_i32_e32 v2, v7\n\t"
        "v_or_b32_e32 v19, v7, v6\n\t"
        "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\t"
        "v_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"
        "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_cmp_lt_i32_e32 vcc, -1, v19\n\t"
        "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
        "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\t"
        "v_mul_f32_e32 v2, v4, v5\n\t"
        "v_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"
        "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t"
        "v_pk_mul_f32 v[4:5], v[8:9], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
        "s_and_b64 vcc, s[6:7], s[8:9]\n\t"
        "v_cndmask_b32_e32 v3, 0, v5, vcc\n\t"
        PRE "s_and_b64 vcc, s[10:11], s[4:5]\n\t" POST
        "v_mul_f32_e32 v5, v8, v9\n\t"
        "s_and_b64 s[12:13], s[10:11], s[8:9]\n\t"
        "v_cndmask_b32_e32 v4, 0, v4, vcc\n\t"
        "s_andn2_b64 vcc, exec, s[14:15]\n\t"
        "v_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"
        : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)
        : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13",
          "s14", "s15", "vcc");
    // select-free evaluation of the same values: the borders as 0 / 1 factors (exact)
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int h_low = (int)fh, w_low = (int)fw;
    const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
    const float t_ok = (float)min(h_low + 1, 1), b_ok = (float)min(Hm1 - h_low, 1);
    const float l_ok = (float)min(w_low + 1, 1), r_ok = (float)min(Wm1 - w_low, 1);
    const float ex = (hh * hw) * (t_ok * l_ok), ey = (hh * lw) * (t_ok * r_ok), ez = (lh * hw) * (b_ok * l_ok), ew = (lh * lw) * (b_ok * r_ok);
    nbad[0] += (wx != ex); nbad[1] += (wy != ey); nbad[2] += (wz != ez); nbad[3] += (ww != ew);
    // a burst of MFMAs now and then, out of step between the waves of a SIMD (the other workgroup of the CU is in its K loop
    // while this one builds its table)
    if (mfma_burst && ((it + wave * 7) & 15) == 0) {
#pragma unroll
      for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
  }
  for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad_lane[lane * 4 + k], nbad[k]);
  if (acc[0] == 12345.f) sink[0] = acc[1];
}



typedef void (*victim_fn)(int, int, int, unsigned*, float*, int);

// ---- aggressor family 2: a plain C++ loop (compiler-scheduled), to see WHICH ingredients disturb the victim ---------------------------
// KIND 0: bf16 32x32x16 MFMAs, 1: fp32 32x32x2 MFMAs, 2: f16 32x32x16 MFMAs, 3: no MFMA (v_fma chains instead)
// LOADS: a global load (L1 / L2 hit) per iteration feeding the next iteration's operand; READACC: a VALU read of the accumulator per
// iteration; FULLVGPR: the kernel claims 256 VGPRs (two of its waves fill a SIMD's register file: no foreign wave fits beside them)
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
template <int KIND, bool LOADS, bool READACC, bool FULLVGPR>
__global__ void __launch_bounds__(kThreads) aggressor2(const uint4* __restrict__ gB, int iters, float* __restrict__ sink) {
  if (FULLVGPR) asm volatile("" ::: "v255");
  const int lane = threadIdx.x & 63;
  const uint4* gb = gB + lane;
  floatx16 acc0 = floatx16{0}, acc1 = floatx16{0};
  uint4 bu = gb[0];
  float f0 = 1.f, f1 = 2.f;
  float keep = 0.f;
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) {
      const bf8 a = __builtin_bit_cast(bf8, bu);
#pragma unroll
      for (int u = 0; u < 6; u++) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc1, 0, 0, 0); }
    } else if (KIND == 1) {
      const float a = __uint_as_float(bu.x);
#pragma unroll
      for (int u = 0; u < 6; u++) { acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc1, 0, 0, 0); }
    } else if (KIND == 2) {
      const h8v a = __builtin_bit_cast(h8v, bu);
#pragma unroll
      for (int u = 0; u < 6; u++) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc1, 0, 0, 0); }
    } else {
#pragma unroll
      for (int u = 0; u < 48; u++) { f0 = fmaf(f0, 0.999f, __uint_as_float(bu.x)); f1 = fmaf(f1, 1.001f, f0); }
    }
    if (READACC) { keep += acc0[15] + acc1[15] + f1; asm volatile("" : "+v"(keep)); }
    if (LOADS) bu = gb[((it + 1) & 1) * 64];
  }
  if (keep + acc0[0] + acc1[3] + f1 == 12345.f) sink[0] = keep;
}

template <int KIND, bool LOADS, bool READACC, bool FULLVGPR>
void run2(const char* name, const uint4* gB, int rounds) {
  unsigned* vbad; float* sink;
  CHK(hipMalloc(&vbad, 1024)); CHK(hipMalloc(&sink, 64)); CHK(hipMemset(vbad, 0, 1024));
  hipStream_t s[2]; CHK(hipStreamCreate(&s[0])); CHK(hipStreamCreate(&s[1]));
  int occ = 0; CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, aggressor2<KIND, LOADS, READACC, FULLVGPR>, kThreads, 0));
  for (int r = 0; r < rounds; r++) {
    hipLaunchKernelGGL((aggressor2<KIND, LOADS, READACC, FULLVGPR>), dim3(256 * 2), dim3(kThreads), 0, s[0], gB, 400, sink);
    hipLaunchKernelGGL(mask_probe_v0, dim3(512), dim3(kVictimThreads), 0, s[1], 2000, 32, 32, vbad, sink, 0);
  }
  CHK(hipGetLastError()); CHK(hipDeviceSynchronize());
  unsigned h[256]; CHK(hipMemcpy(h, vbad, 1024, hipMemcpyDeviceToHost));
  unsigned long long q3z = 0, tot = 0;
  for (int l = 0; l < 64; l++) for (int k = 0; k < 4; k++) { tot += h[l * 4 + k]; if (l >= 48 && k == 2) q3z += h[l * 4 + k]; }
  printf("aggressor %-78s (%d of its workgroups fit a CU) | victim, %d launches: wrong weights %llu (of them w.z in lanes 48..63: %llu)\n", name, occ, rounds, tot, q3z);
  CHK(hipFree(vbad)); CHK(hipFree(sink)); CHK(hipStreamDestroy(s[0])); CHK(hipStreamDestroy(s[1]));
}
template <int KIND, bool LOADS, bool READACC, bool FULLVGPR>
void run2v(const char* name, const uint4* gB, int rounds, void (*victim)(int, int, int, unsigned*, float*, int), const char* vname) {
  unsigned* vbad; float* sink;
  CHK(hipMalloc(&vbad, 1024)); CHK(hipMalloc(&sink, 64)); CHK(hipMemset(vbad, 0, 1024));
  hipStream_t s[2]; CHK(hipStreamCreate(&s[0])); CHK(hipStreamCreate(&s[1]));
  int occ = 0; CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, aggressor2<KIND, LOADS, READACC, FULLVGPR>, kThreads, 0));
  for (int r = 0; r < rounds; r++) {
    hipLaunchKernelGGL((aggressor2<KIND, LOADS, READACC, FULLVGPR>), dim3(256 * 2), dim3(kThreads), 0, s[0], gB, 400, sink);
    hipLaunchKernelGGL(victim, dim3(512), dim3(kVictimThreads), 0, s[1], 2000, 32, 32, vbad, sink, 0);
  }
  CHK(hipGetLastError()); CHK(hipDeviceSynchronize());
  unsigned h[256]; CHK(hipMemcpy(h, vbad, 1024, hipMemcpyDeviceToHost));
  unsigned long long q3z = 0, tot = 0;
  for (int l = 0; l < 64; l++) for (int k = 0; k < 4; k++) { tot += h[l * 4 + k]; if (l >= 48 && k == 2) q3z += h[l * 4 + k]; }
  printf("aggressor %-78s (%d of its workgroups fit a CU) | victim %s, %d launches: wrong weights %llu (of them w.z in lanes 48..63: %llu)\n", name, occ, vname, rounds, tot, q3z);
  CHK(hipFree(vbad)); CHK(hipFree(sink)); CHK(hipStreamDestroy(s[0])); CHK(hipStreamDestroy(s[1]));
}

template <int MODE, bool SAFE>
void run(const char* name, const uint4* gB, int rounds, int lds_bytes, victim_fn victim = mask_probe_v0, const char* vname = "") {
  unsigned *bad, *vbad; float *fb, *sink;
  CHK(hipMalloc(&bad, 4)); CHK(hipMalloc(&fb, 128)); CHK(hipMalloc(&vbad, 1024)); CHK(hipMalloc(&sink, 64));
  CHK(hipMemset(bad, 0, 4)); CHK(hipMemset(fb, 0, 128)); CHK(hipMemset(vbad, 0, 1024));
  hipStream_t s[2]; CHK(hipStreamCreate(&s[0])); CHK(hipStreamCreate(&s[1]));
  if (MODE >= 0) CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&war_probe<(MODE < 0 ? 3 : MODE), SAFE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  for (int r = 0; r < rounds; r++) {
    if (MODE >= 0) hipLaunchKernelGGL((war_probe<(MODE < 0 ? 3 : MODE), SAFE>), dim3(256 * 2), dim3(kThreads), lds_bytes, s[0], gB, 400, bad, fb, 0);
    hipLaunchKernelGGL(victim, dim3(512), dim3(kVictimThreads), 0, s[1], 2000, 32, 32, vbad, sink, 0);
  }
  CHK(hipGetLastError()); CHK(hipDeviceSynchronize());
  unsigned hb, h[256];
  CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(h, vbad, 1024, hipMemcpyDeviceToHost));
  unsigned long long q[4][4] = {};
  for (int l = 0; l < 64; l++) for (int k = 0; k < 4; k++) q[l >> 4][k] += h[l * 4 + k];
  printf("aggressor %-44s (its own wrong accumulators: %u) | victim %s, %d launches: wrong w.x / w.y / w.z / w.w by lane quarter:", name, hb, vname, rounds);
  for (int qq = 0; qq < 4; qq++) printf("  [%llu %llu %llu %llu]", q[qq][0], q[qq][1], q[qq][2], q[qq][3]);
  printf("\n");
  CHK(hipFree(bad)); CHK(hipFree(fb)); CHK(hipFree(vbad)); CHK(hipFree(sink));
  CHK(hipStreamDestroy(s[0])); CHK(hipStreamDestroy(s[1]));
}


// ---- WHICH packed instructions are affected: one packed op on small integers (every result exact) per iteration, checked per half ----
typedef float f2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void __launch_bounds__(kVictimThreads) packed_victim(int iters, unsigned* __restrict__ bad /* [4 quarters][2 halves] */) {
  const int lane = threadIdx.x & 63;
  unsigned seed = (blockIdx.x * kVictimThreads + threadIdx.x) * 2654435761u + 777u;
  unsigned nlo = 0, nhi = 0;
  for (int it = 0; it < iters; it++) {
    seed = seed * 1664525u + 1013904223u;
    const int a0 = (seed >> 4) & 15, a1 = (seed >> 8) & 15, b0 = (seed >> 12) & 15, b1 = (seed >> 16) & 15, c0 = (seed >> 20) & 15, c1 = (seed >> 24) & 15;
    if (OP < 3 || OP >= 6) {
      f2v a = {(float)a0, (float)a1}, b = {(float)b0, (float)b1}, c = {(float)c0, (float)c1}, r;
      if (OP == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      else if (OP == 6) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));   // lo = a.lo * b.hi, hi = a.hi * b.lo
      else if (OP == 7) { r = b; asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(r) : "v"(a)); }                             // destination = second source
      else if (OP == 8) { r = b; asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(r) : "v"(a)); }
      else if (OP == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      else asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
      asm volatile("s_nop 2");
      const bool sw = OP == 6 || OP == 8;
      const float e0 = sw ? (float)(a0 * b1) : (OP == 0 || OP == 7) ? (float)(a0 * b0) : OP == 1 ? (float)(a0 + b0) : (float)(a0 * b0 + c0);
      const float e1 = sw ? (float)(a1 * b0) : (OP == 0 || OP == 7) ? (float)(a1 * b1) : OP == 1 ? (float)(a1 + b1) : (float)(a1 * b1 + c1);
      nlo += r[0] != e0; nhi += r[1] != e1;
    } else {
      typedef _Float16 h2v __attribute__((ext_vector_type(2)));
      h2v a = {(_Float16)a0, (_Float16)a1}, b = {(_Float16)b0, (_Float16)b1}, c = {(_Float16)c0, (_Float16)c1}, r;
      if (OP == 3) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      else if (OP == 4) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      else asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
      asm volatile("s_nop 2");
      const float e0 = OP == 3 ? (float)(a0 * b0) : OP == 4 ? (float)(a0 + b0) : (float)(a0 * b0 + c0);
      const float e1 = OP == 3 ? (float)(a1 * b1) : OP == 4 ? (float)(a1 + b1) : (float)(a1 * b1 + c1);
      nlo += (float)r[0] != e0; nhi += (float)r[1] != e1;
    }
  }
  if (nlo) atomicAdd(&bad[(lane >> 4) * 2], nlo);
  if (nhi) atomicAdd(&bad[(lane >> 4) * 2 + 1], nhi);
}

template <int OP>
void run_packed(const char* name, const uint4* gB, int rounds, bool with_aggressor) {
  unsigned* vbad; float* sink;
  CHK(hipMalloc(&vbad, 64)); CHK(hipMalloc(&sink, 64)); CHK(hipMemset(vbad, 0, 64));
  hipStream_t s[2]; CHK(hipStreamCreate(&s[0])); CHK(hipStreamCreate(&s[1]));
  for (int r = 0; r < rounds; r++) {
    if (with_aggressor) hipLaunchKernelGGL((aggressor2<0, false, true, false>), dim3(256 * 2), dim3(kThreads), 0, s[0], gB, 400, sink);
    hipLaunchKernelGGL(packed_victim<OP>, dim3(512), dim3(kVictimThreads), 0, s[1], 4000, vbad);
  }
  CHK(hipGetLastError()); CHK(hipDeviceSynchronize());
  unsigned h[8]; CHK(hipMemcpy(h, vbad, 32, hipMemcpyDeviceToHost));
  printf("%-14s %s: %.1e results per half; wrong (low half | high half) by lane quarter: [%u | %u] [%u | %u] [%u | %u] [%u | %u]\n", name,
         with_aggressor ? "next to bf16 MFMAs + accumulator reads" : "alone                                 ", (double)rounds * 512 * 512 * 4000,
         h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  CHK(hipFree(vbad)); CHK(hipFree(sink)); CHK(hipStreamDestroy(s[0])); CHK(hipStreamDestroy(s[1]));
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  uint4* gB; CHK(hipMalloc(&gB, sizeof(uint4) * 128));
  {
    uint16_t h[128 * 8];
    for (int t = 0; t < 2; t++)
      for (int l = 0; l < 64; l++) {
        const float v = (float)(((l & 31) % 5 + 1) * (t + 1));
        uint32_t u; memcpy(&u, &v, 4);
        for (int e = 0; e < 8; e++) h[(t * 64 + l) * 8 + e] = (uint16_t)(u >> 16);
      }
    CHK(hipMemcpy(gB, h, sizeof(h), hipMemcpyHostToDevice));
  }
  const int lds2 = 36 * 1024;
  run<-1, true>("none", gB, rounds, lds2);
  run<3, true>("MFMAs + compiler-placed loads (SAFE)", gB, rounds, lds2);
  run<1, false>("MFMAs + A refilled in place from LDS", gB, rounds, lds2);
  run<2, false>("MFMAs + B refilled in place from global", gB, rounds, lds2);
  run<3, false>("MFMAs + A and B refilled in place", gB, rounds, lds2);
  printf("== the victim with wait states around the SALU instruction that combines the two compare results (aggressor: MFMAs + compiler-placed loads)\n");
  run<3, true>("MFMAs + compiler-placed loads (SAFE)", gB, rounds, lds2, mask_probe_v0, "(as the compiler scheduled it)");
  run<3, true>("MFMAs + compiler-placed loads (SAFE)", gB, rounds, lds2, mask_probe_v1, "(1 wait state in front of the s_and_b64 that reads the v_cmp results)");
  run<3, true>("MFMAs + compiler-placed loads (SAFE)", gB, rounds, lds2, mask_probe_v2, "(2 wait states in front)");
  run<3, true>("MFMAs + compiler-placed loads (SAFE)", gB, rounds, lds2, mask_probe_v3, "(4 wait states in front)");
  run<3, true>("MFMAs + compiler-placed loads (SAFE)", gB, rounds, lds2, mask_probe_v4, "(8 wait states in front)");
  run<3, true>("MFMAs + compiler-placed loads (SAFE)", gB, rounds, lds2, mask_probe_v5, "(8 wait states BEHIND the s_and_b64 (in front of the v_cndmask that reads VCC))");
  printf("== which ingredients of the aggressor matter (plain C++ loops)\n");
  run2<0, true, true, false>("bf16 32x32x16 MFMAs + a global load + a read of the accumulator per iteration", gB, rounds);
  run2<0, true, false, false>("bf16 32x32x16 MFMAs + a global load per iteration", gB, rounds);
  run2<0, false, true, false>("bf16 32x32x16 MFMAs + a read of the accumulator per iteration, no loads", gB, rounds);
  run2<0, false, false, false>("bf16 32x32x16 MFMAs only", gB, rounds);
  run2<3, true, true, false>("no MFMA (v_fma chains) + a global load per iteration", gB, rounds);
  run2<1, true, true, false>("fp32 32x32x2 MFMAs + a global load + a read of the accumulator", gB, rounds);
  run2<2, true, true, false>("f16 32x32x16 MFMAs + a global load + a read of the accumulator", gB, rounds);
  run2<1, false, true, false>("fp32 32x32x2 MFMAs + a read of the accumulator per iteration, no loads", gB, rounds);
  run2<2, false, true, false>("f16 32x32x16 MFMAs + a read of the accumulator per iteration, no loads", gB, rounds);
  run2<3, false, true, false>("no MFMA (v_fma chains), no loads", gB, rounds);
  run2<0, false, true, true>("bf16 32x32x16 MFMAs + read, no loads, kernel claims 256 VGPRs (owns its SIMDs)", gB, rounds);
  run2<0, true, true, true>("bf16 32x32x16 MFMAs + load + read, kernel claims 256 VGPRs (owns its SIMDs)", gB, rounds);
  printf("== the victim without its packed-fp32 multiply (two v_mul_f32 instead of v_pk_mul_f32)\n");
  run<3, true>("MFMAs + compiler-placed loads (SAFE)", gB, rounds, lds2, mask_probe_v6, "(v_pk_mul_f32 replaced by two v_mul_f32)");
  run2v<0, false, true, false>("bf16 32x32x16 MFMAs + a read of the accumulator per iteration, no loads", gB, rounds, mask_probe_v0, "as compiled");
  run2v<0, false, true, false>("bf16 32x32x16 MFMAs + a read of the accumulator per iteration, no loads", gB, rounds, mask_probe_v6, "without v_pk_mul_f32");
  run2v<0, false, true, false>("same", gB, rounds, mask_probe_v7, "packed multiply into OTHER registers, then two v_mov");
  run2v<0, false, true, false>("same", gB, rounds, mask_probe_v8, "s_nop 3 in front of the packed multiply");
  run2v<0, false, true, false>("same", gB, rounds, mask_probe_v9, "s_nop 3 behind the packed multiply");
  run2v<0, false, true, false>("same", gB, rounds, mask_probe_v10, "operands swapped by v_mov, packed multiply WITHOUT op_sel");
  printf("== which packed instructions are affected\n");
  run_packed<0>("v_pk_mul_f32", gB, rounds / 4, false);
  run_packed<0>("v_pk_mul_f32", gB, rounds / 4, true);
  run_packed<1>("v_pk_add_f32", gB, rounds / 4, true);
  run_packed<2>("v_pk_fma_f32", gB, rounds / 4, true);
  run_packed<6>("pk_mul op_sel", gB, rounds / 4, true);
  run_packed<7>("pk_mul inplace", gB, rounds / 4, true);
  run_packed<8>("pk_mul both", gB, rounds / 4, true);
  run_packed<3>("v_pk_mul_f16", gB, rounds / 4, true);
  run_packed<4>("v_pk_add_f16", gB, rounds / 4, true);
  run_packed<5>("v_pk_fma_f16", gB, rounds / 4, true);
  CHK(hipFree(gB));
  return 0;
}
