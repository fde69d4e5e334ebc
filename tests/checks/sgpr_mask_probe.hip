// Development aid (GPU box): the instruction sequence behind the wrong rows of the MT = 1 DeformConv split launch, in isolation.
//
// tests/checks/split_trace.py showed WHAT goes wrong: in a failing launch the bilinear coefficient table of a workgroup has the third
// weight (w.z = lh * hw, kept when b_ok && l_ok) equal to 0 in exactly the entries built by lanes 48..63 of one wave -- every other
// word of the entry correct.  In the compiler's code for that block (llvm, gfx950, -O3) the select reads VCC = s[10:11] & s[4:5], both
// written by v_cmp_*_e64 a few instructions earlier, between packed-fp32 instructions:
//
//     v_cmp_gt_i32_e64 s[10:11], H-1, h_low ; v_mul_f32 ; v_cmp_lt_i32_e64 s[4:5], -1, w_low ; v_cndmask ; v_pk_mul_f32 ;
//     s_and_b64 vcc, s[6:7], s[8:9] ; v_cndmask (w.y) ; s_and_b64 vcc, s[10:11], s[4:5] ; v_mul_f32 ; s_and_b64 s[12:13], ... ;
//     v_cndmask (w.z) ; s_andn2_b64 vcc, exec, s[..] ; v_cndmask_e64 (w.w)
//
// This program runs that sequence (the compiler's own instruction order, fixed registers, s[10:11] zeroed before every pass so that a
// stale quarter is visible) millions of times per lane, two 512-thread workgroups per CU, with bursts of MFMAs of the other waves of
// the SIMD in between, and checks the four weights of every lane against a select-free evaluation.
//   hipcc --offload-arch=gfx950 -O3 -o sgpr_mask_probe sgpr_mask_probe.hip && ./sgpr_mask_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kThreads = 512;

__global__ void __launch_bounds__(kThreads) mask_probe(int iters, int H, int W, unsigned* __restrict__ bad_lane /* [64][4] */, float* __restrict__ sink,
                                                       int mfma_burst) {
  extern __shared__ float lds[];
  if (iters < 0) lds[threadIdx.x] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * kThreads + threadIdx.x) * 2654435761u + 12345u;
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
        "s_and_b64 vcc, s[10:11], s[4:5]\n\t"
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

#ifdef VICTIM_LIB
// as a shared library (tests/checks/victim_probe.py): the probe kernel as a VICTIM next to another stream's kernels
extern "C" int victim_launch(int blocks, int iters, unsigned* bad_lane, float* sink, int mfma_burst, void* stream) {
  hipLaunchKernelGGL(mask_probe, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, iters, 32, 32, bad_lane, sink, mfma_burst);
  return (int)hipGetLastError();
}
#else
int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  unsigned* bad; float* sink;
  CHK(hipMalloc(&bad, sizeof(unsigned) * 256)); CHK(hipMalloc(&sink, 64));
  hipStream_t s[2]; CHK(hipStreamCreate(&s[0])); CHK(hipStreamCreate(&s[1]));
  for (int variant = 0; variant < 4; variant++) {
    const int lds_bytes = (variant & 1) ? 84 * 1024 : 36 * 1024;           // two / one workgroup(s) per CU
    const int burst = (variant & 2) ? 0 : 1;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CHK(hipMemset(bad, 0, sizeof(unsigned) * 256));
    int occ = 0; CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mask_probe, kThreads, lds_bytes));
    for (int l = 0; l < launches; l++)
      hipLaunchKernelGGL(mask_probe, dim3(256 * 4), dim3(kThreads), lds_bytes, s[l & 1], iters, 32, 32, bad, sink, burst);
    CHK(hipGetLastError()); CHK(hipDeviceSynchronize());
    unsigned h[256]; CHK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
    unsigned long long tot[4] = {0, 0, 0, 0}, q[4][4] = {};
    for (int l = 0; l < 64; l++) for (int k = 0; k < 4; k++) { tot[k] += h[l * 4 + k]; q[k][l >> 4] += h[l * 4 + k]; }
    printf("%d workgroup(s)/CU, MFMA bursts %s: %.2e evaluations per weight; wrong w.x %llu  w.y %llu  w.z %llu  w.w %llu", occ, burst ? "on " : "off",
           (double)launches * 1024 * kThreads * iters, tot[0], tot[1], tot[2], tot[3]);
    if (tot[0] + tot[1] + tot[2] + tot[3])
      printf("   by lane quarter (0-15 | 16-31 | 32-47 | 48-63): w.x %llu|%llu|%llu|%llu  w.y %llu|%llu|%llu|%llu  w.z %llu|%llu|%llu|%llu  w.w %llu|%llu|%llu|%llu",
             q[0][0], q[0][1], q[0][2], q[0][3], q[1][0], q[1][1], q[1][2], q[1][3], q[2][0], q[2][1], q[2][2], q[2][3], q[3][0], q[3][1], q[3][2], q[3][3]);
    printf("\n");
  }
  return 0;
}
#endif
