// MI355X (gfx950): v_pk_mul_f32 with op_sel returns a wrong LOW half in lanes 48..63 while other waves of the CU issue dense
// 16-bit MFMAs and read their accumulators.  Stand-alone: hipcc --offload-arch=gfx950 -O3 -o repro THIS && ./repro [launches]
// Aggressor (stream 0): v_mfma_f32_32x32x16_bf16 chains + one VALU read of the accumulators per 12 MFMAs, no memory traffic.
// Victim (stream 1): hipcc's own code for four bilinear weights with border masks (floor, 1 - frac as v_pk_add_f32, the cross
// products as ONE v_pk_mul_f32 v[4:5], v[8:9], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]), checked against scalar arithmetic.
// Variants: as compiled; the packed multiply as two v_mul_f32; s_nop 0 / 1 / 3 / 7 in front of the packed multiply (the failure is
// timing dependent: which variant fails moves with code placement -- in the long program tests/checks/mfma_refill_victim.hip it is the
// as-compiled one, 5e6 wrong per 5e10; here (ROCm 7.2.0, two leases) s_nop 3: 3e3 .. 4e5 wrong per 5e10, ALL w.z in lanes 48..63).
// The scalar form never fails; no variant fails without the aggressor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void __launch_bounds__(512) aggressor(const uint4* __restrict__ gB, int iters, float* sink) {
  const bf8 a = __builtin_bit_cast(bf8, gB[threadIdx.x & 63]);        // bf16 splat of (lane % 32) % 5 + 1
  floatx16 acc0 = floatx16{0}, acc1 = floatx16{0};
  float keep = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 6; u++) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc1, 0, 0, 0); }
    keep += acc0[15] + acc1[15];
    asm volatile("" : "+v"(keep));
  }
  if (keep + acc0[0] + acc1[3] == 12345.f) sink[0] = keep;
}

#define PK0 "v_pk_mul_f32 v[4:5], v[8:9], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
#define PK1 "v_mul_f32_e32 v19, v8, v5\n\tv_mul_f32_e32 v5, v9, v4\n\tv_mov_b32_e32 v4, v19\n\t"
#define PK2 "s_nop 3\n\t" PK0
#define PK3 "s_nop 0\n\t" PK0
#define PK4 "s_nop 1\n\t" PK0
#define PK5 "s_nop 7\n\t" PK0
#define VICTIM(NAME, PK)                                                                                                          \
  __global__ void __launch_bounds__(512) NAME(int iters, int Hm1, int Wm1, unsigned* __restrict__ bad /* [64 lanes][4 weights] */, \
                                              float* __restrict__ sink, int mfma_burst /* 0: never taken */) {                      \
    unsigned seed = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u, nbad[4] = {0, 0, 0, 0};                               \
    const __bf16 one = (__bf16)1.f; const bf8 a1 = {one, one, one, one, one, one, one, one}; floatx16 acc = floatx16{0};            \
    for (int it = 0; it < iters; it++) {                                                                                          \
      seed = seed * 1664525u + 1013904223u;                                                                                       \
      const float h_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)(Hm1 + 1) + 0.998f);                         \
      seed = seed * 1664525u + 1013904223u;                                                                                       \
      const float w_im = -0.999f + (float)(seed >> 8) * (1.f / 16777216.f) * ((float)(Wm1 + 1) + 0.998f);                         \
      float wx, wy, wz, ww;                                                                                                       \
      asm volatile("v_mov_b32 v16, %[him]\n\tv_mov_b32 v17, %[wim]\n\ts_mov_b64 s[10:11], 0\n\ts_mov_b64 s[4:5], 0\n\t"           \
                   "s_mov_b64 s[14:15], 0\n\tv_floor_f32_e32 v2, v16\n\tv_floor_f32_e32 v3, v17\n\tv_cvt_i32_f32_e32 v6, v3\n\t"  \
                   "v_cvt_i32_f32_e32 v7, v2\n\tv_cvt_f32_i32_e32 v3, v6\n\tv_cvt_f32_i32_e32 v2, v7\n\tv_or_b32_e32 v19, v7, v6\n\t" \
                   "v_cmp_lt_i32_e64 s[6:7], -1, v7\n\tv_cmp_gt_i32_e64 s[8:9], %[Wm1], v6\n\t"                                  \
                   "v_pk_add_f32 v[8:9], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\tv_cmp_lt_i32_e32 vcc, -1, v19\n\t"         \
                   "v_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"                               \
                   "v_cmp_gt_i32_e64 s[10:11], %[Hm1], v7\n\tv_mul_f32_e32 v2, v4, v5\n\tv_cmp_lt_i32_e64 s[4:5], -1, v6\n\t"     \
                   "v_cndmask_b32_e32 v2, 0, v2, vcc\n\t" PK "s_and_b64 vcc, s[6:7], s[8:9]\n\tv_cndmask_b32_e32 v3, 0, v5, vcc\n\t" \
                   "s_and_b64 vcc, s[10:11], s[4:5]\n\tv_mul_f32_e32 v5, v8, v9\n\ts_and_b64 s[12:13], s[10:11], s[8:9]\n\t"      \
                   "v_cndmask_b32_e32 v4, 0, v4, vcc\n\ts_andn2_b64 vcc, exec, s[14:15]\n\tv_cndmask_b32_e64 v5, 0, v5, s[12:13]\n\t" \
                   "s_nop 4\n\tv_mov_b32 %[wx], v2\n\tv_mov_b32 %[wy], v3\n\tv_mov_b32 %[wz], v4\n\tv_mov_b32 %[ww], v5\n\t"      \
                   : [wx] "=&v"(wx), [wy] "=&v"(wy), [wz] "=&v"(wz), [ww] "=&v"(ww)                                               \
                   : [him] "v"(h_im), [wim] "v"(w_im), [Hm1] "s"(Hm1), [Wm1] "s"(Wm1)                                             \
                   : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v19", "s4", "s5", "s6", "s7", "s8", "s9",     \
                     "s10", "s11", "s12", "s13", "s14", "s15", "vcc");                                                            \
      const float fh = floorf(h_im), fw = floorf(w_im), lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;             \
      const int hl = (int)fh, wl = (int)fw;                                                                                       \
      const float t = (float)min(hl + 1, 1), b = (float)min(Hm1 - hl, 1), l = (float)min(wl + 1, 1), r = (float)min(Wm1 - wl, 1); \
      nbad[0] += wx != (hh * hw) * (t * l); nbad[1] += wy != (hh * lw) * (t * r);                                                 \
      nbad[2] += wz != (lh * hw) * (b * l); nbad[3] += ww != (lh * lw) * (b * r);                                                 \
      if (mfma_burst && ((it + (threadIdx.x >> 6) * 7) & 15) == 0)                                                                \
        for (int u = 0; u < 12; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, a1, acc, 0, 0, 0);                          \
    }                                                                                                                             \
    for (int k = 0; k < 4; k++) if (nbad[k]) atomicAdd(&bad[(threadIdx.x & 63) * 4 + k], nbad[k]);                                \
    if (acc[0] == 12345.f) sink[0] = acc[1];                                                                                      \
  }
VICTIM(victim_as_compiled, PK0)
VICTIM(victim_two_v_mul, PK1)
VICTIM(victim_s_nop_3, PK2)
VICTIM(victim_s_nop_0, PK3)
VICTIM(victim_s_nop_1, PK4)
VICTIM(victim_s_nop_7, PK5)

static void run(const char* name, void (*victim)(int, int, int, unsigned*, float*, int), int launches, bool with_aggressor) {
  unsigned* bad; float* sink; uint4* gB; hipStream_t s[2];
  CHK(hipMalloc(&bad, 1024)); CHK(hipMalloc(&sink, 64)); CHK(hipMemset(bad, 0, 1024)); CHK(hipMalloc(&gB, 1024));
  unsigned short hb[512];
  for (int l = 0; l < 64; l++) { const float v = (float)((l & 31) % 5 + 1); unsigned u; memcpy(&u, &v, 4); for (int e = 0; e < 8; e++) hb[l * 8 + e] = (unsigned short)(u >> 16); }
  CHK(hipMemcpy(gB, hb, 1024, hipMemcpyHostToDevice));
  CHK(hipStreamCreate(&s[0])); CHK(hipStreamCreate(&s[1]));
  for (int r = 0; r < launches; r++) {
    if (with_aggressor) hipLaunchKernelGGL(aggressor, dim3(512), dim3(512), 0, s[0], gB, 400, sink);
    hipLaunchKernelGGL(victim, dim3(512), dim3(512), 0, s[1], 2000, 31, 31, bad, sink, 0);
  }
  CHK(hipGetLastError()); CHK(hipDeviceSynchronize());
  unsigned h[256]; CHK(hipMemcpy(h, bad, 1024, hipMemcpyDeviceToHost));
  unsigned long long tot = 0, q3z = 0;
  for (int l = 0; l < 64; l++) for (int k = 0; k < 4; k++) { tot += h[l * 4 + k]; if (l >= 48 && k == 2) q3z += h[l * 4 + k]; }
  printf("RESULT %-16s %-14s evaluations %.2e wrong %llu wrong_wz_lanes_48_63 %llu\n", name, with_aggressor ? "next_to_mfma" : "alone",
         (double)launches * 512 * 512 * 2000, tot, q3z);
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 100;
  run("as_compiled", victim_as_compiled, n, false);
  run("as_compiled", victim_as_compiled, n, true);
  run("two_v_mul_f32", victim_two_v_mul, n, true);
  run("s_nop_3_in_front", victim_s_nop_3, n, true);
  run("s_nop_0_in_front", victim_s_nop_0, n, true);
  run("s_nop_1_in_front", victim_s_nop_1, n, true);
  run("s_nop_7_in_front", victim_s_nop_7, n, true);
  run("s_nop_3_in_front", victim_s_nop_3, n, false);
  return 0;
}
