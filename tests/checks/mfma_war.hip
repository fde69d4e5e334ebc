// Development aid (GPU box): does gfx950 need software protection between an ISSUED v_mfma and a later asynchronous write of its
// SrcA / SrcB registers (an LDS or global load returning into them)?  Round 4's split convolution kernel carried an "accumulator
// drain" (an inline-asm v_mov of one accumulator element before the in-place refill of the weight registers) on the hypothesis
// that such writes can overtake queued MFMAs when four waves per SIMD keep the matrix pipe backlogged.  This program tests the
// hypothesis directly, and what that v_mov actually does:
//
//   part A  one wave: K dependent MFMAs, then  (0) nothing  (1) the inline-asm v_mov of acc[15]  (2) a compiler-visible read of
//           acc[15]; s_memtime ticks from the first MFMA to after the read, and WHICH value the read returned (every MFMA adds 16:
//           16 * j = the j-th MFMA had written back when the read executed).
//   part B  the hazard itself, forced: every iteration issues 6 dependent MFMAs on operand registers A / B and IMMEDIATELY
//           overwrites those same physical registers with the NEXT iteration's operands by an LDS read (A) and / or a global
//           load (B) -- the tied asm operand guarantees the destination is the register set the MFMAs just read.  512-thread
//           workgroups, two per CU (four waves per SIMD, the residency in which the wrong rows were seen), every CU busy, optionally
//           twice concurrently on two streams.  Operands are small integers in bf16: every sum is exact, the expected accumulator is
//           a closed form, every lane of every wave is checked on the device.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_war mfma_war.hip && ./mfma_war
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

// ---- part A ---------------------------------------------------------------------------------------------------------------------
// The timed region is ONE asm statement on fixed registers (v[32:47] = the accumulator, in VGPRs as in the split kernel), so that
// the compiler's hazard recogniser cannot add wait states: what is measured is the hardware's own behaviour.
#define MF1 "v_mfma_f32_32x32x16_bf16 v[32:47], %[a], %[b], v[32:47]\n\t"
#define MF2 MF1 MF1
#define MF4 MF2 MF2
#define MF6 MF4 MF2
#define MF16 MF4 MF4 MF4 MF4
#define NOP8 "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#define DRAIN_BODY(MFS, READ)                                                                                                   \
  asm volatile("v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\tv_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\t"   \
               "v_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\tv_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"   \
               "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\tv_mov_b32 %[got], -1.0\n\t"                   \
               MF4 NOP8 NOP8                                                       /* warm-up, then an idle pipe */                  \
               "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\tv_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\t"   \
               "v_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\tv_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"   \
               "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"                                           \
               "s_nop 4\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"                                                              \
               MFS READ                                                                                                               \
               "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"                                                                          \
               NOP8 NOP8 NOP8 NOP8 NOP8                                            /* 640 cycles: every MFMA has written back */       \
               "v_mov_b32 %[fin], v47\n\t"                                                                                            \
               : [t0] "=&s"(t0), [t1] "=&s"(t1), [got] "=&v"(got), [fin] "=&v"(fin)                                                   \
               : [a] "v"(a), [b] "v"(b)                                                                                               \
               : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory")

template <int K, int VARIANT>
__global__ void __launch_bounds__(64) drain_probe(long long* ticks, float* seen, float* final_) {
  const bf8 a = splat(1.f), b = splat(1.f);
  long long t0, t1;
  float got, fin;
  // VARIANT 0: nothing behind the MFMAs; 1: the v_mov of the last accumulator register, unprotected (round 4's "drain");
  // 2: the same read behind the software wait states the ISA asks for (s_nop 11 = 12 wait states for an 8-pass MFMA on gfx950)
#define READ0 ""
#define READ1 "v_mov_b32 %[got], v47\n\t"
#define READ2 "s_nop 11\n\tv_mov_b32 %[got], v47\n\t"
#define DRAIN_K(MFS)                                                                   \
  if (VARIANT == 0) { DRAIN_BODY(MFS, READ0); }                                        \
  else if (VARIANT == 1) { DRAIN_BODY(MFS, READ1); }                                   \
  else { DRAIN_BODY(MFS, READ2); }
  if (K == 1) { DRAIN_K(MF1) } else if (K == 2) { DRAIN_K(MF2) } else if (K == 4) { DRAIN_K(MF4) } else if (K == 6) { DRAIN_K(MF6) } else { DRAIN_K(MF16) }
  if (threadIdx.x == 0) { ticks[0] = t1 - t0; seen[0] = got; }
  final_[threadIdx.x] = fin;
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

template <int MODE, bool SAFE>
void run_war(const char* name, const uint4* gB, int iters, int launches, int lds_bytes, bool two_streams) {
  unsigned* bad; float* fb;
  CHK(hipMalloc(&bad, sizeof(unsigned))); CHK(hipMalloc(&fb, sizeof(float) * 32));
  CHK(hipMemset(bad, 0, sizeof(unsigned))); CHK(hipMemset(fb, 0, sizeof(float) * 32));
  hipStream_t s[2];
  CHK(hipStreamCreate(&s[0])); CHK(hipStreamCreate(&s[1]));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&war_probe<MODE, SAFE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  int occ = 0;
  CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, war_probe<MODE, SAFE>, kThreads, lds_bytes));
  const int blocks = 256 * 2 * 4;                                               // four rounds of two workgroups per CU
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  CHK(hipEventRecord(e0, s[0]));
  for (int l = 0; l < launches; l++)
    hipLaunchKernelGGL((war_probe<MODE, SAFE>), dim3(blocks), dim3(kThreads), lds_bytes, s[two_streams ? (l & 1) : 0], gB, iters, bad, fb, 0);
  CHK(hipGetLastError());
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e1, s[0])); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  unsigned hb; float hf[32];
  CHK(hipMemcpy(&hb, bad, sizeof(unsigned), hipMemcpyDeviceToHost)); CHK(hipMemcpy(hf, fb, sizeof(hf), hipMemcpyDeviceToHost));
  const double waves = (double)launches * blocks * (kThreads / 64);
  printf("%-34s LDS %3d KB -> %d workgroups/CU, %s: %d launches x %d blocks x %d iterations (%.2e in-place refills right behind 6 MFMAs) in %.0f ms: "
         "lanes with a wrong accumulator: %u", name, lds_bytes / 1024, occ, two_streams ? "two streams" : "one stream", launches, blocks, iters,
         waves * iters, ms, hb);
  if (hb) printf("  first: acc0 %.0f want %.0f, acc1 %.0f want %.0f", hf[0], hf[1], hf[2], hf[3]);
  printf("\n");
  CHK(hipFree(bad)); CHK(hipFree(fb));
  CHK(hipStreamDestroy(s[0])); CHK(hipStreamDestroy(s[1]));
}


// ---- part C ---------------------------------------------------------------------------------------------------------------------
// VALU writes to the source registers of a JUST-ISSUED MFMA (the compiler reuses dead weight registers as temporaries of the bilinear
// combine, one instruction behind the MFMA that read them: the -DORP_DCNS_DRAIN=0 fp16-pieces build of the split kernel, whose wrong
// rows are the rows combined by lanes 48..63).  One asm statement per iteration: a dependent MFMA chain, the MFMA on A = v[44:47],
// then v_add / v_mul into v44..v47 with GAP s_nop wait states in front, then both the MFMA result (exact small integers) and the
// VALU results are checked, per lane quarter.
template <int GAP>
__global__ void __launch_bounds__(kThreads) valu_after_mfma_probe(int iters, unsigned* __restrict__ bad /* [2][4]: VALU result / MFMA result by lane quarter */) {
  extern __shared__ uint4 lds[];
  if (iters < 0) lds[threadIdx.x] = uint4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  const bf8 b = splat(1.f);
  floatx16 acc = floatx16{0};
  unsigned nv = 0, nm = 0;
  float expect_acc = 0.f;
  unsigned seed = (blockIdx.x * kThreads + threadIdx.x) * 2654435761u + 99u;
  for (int it = 0; it < iters; it++) {
    seed = seed * 1664525u + 1013904223u;
    const float x0 = (float)((seed >> 8) & 1023), x1 = (float)((seed >> 18) & 1023);
    const float alpha = (float)((it & 3) + 1);
    const bf8 a = splat(alpha);                           // A operand of this iteration, placed in v[44:47]
    float r0, r1, r2, r3;
#define VAM_ASM(GAPSTR)                                                                                                          \
    asm volatile(                                                                                                                \
        "v_mov_b32 v44, %[a0]\n\tv_mov_b32 v45, %[a1]\n\tv_mov_b32 v46, %[a2]\n\tv_mov_b32 v47, %[a3]\n\t"                        \
        "s_nop 4\n\t"                                                                                                            \
        "v_mfma_f32_32x32x16_bf16 %[acc], %[b], %[b], %[acc]\n\t"          /* the chain: the next MFMA waits for this one */      \
        "v_mfma_f32_32x32x16_bf16 %[acc], v[44:47], %[b], %[acc]\n\t"      /* reads v[44:47] */                                   \
        GAPSTR                                                                                                                   \
        "v_add_f32_e64 v44, %[x0], %[x1]\n\t"                              /* ... and the VALU writes them right behind it */     \
        "v_add_f32_e64 v45, %[x1], %[x1]\n\t"                                                                                    \
        "v_mul_f32_e64 v46, %[x0], %[x1]\n\t"                                                                                    \
        "v_mul_f32_e64 v47, %[x0], %[x0]\n\t"                                                                                    \
        "v_mfma_f32_32x32x16_bf16 %[acc], %[b], %[b], %[acc]\n\t"                                                                \
        "s_nop 7\n\t"                                                                                                            \
        "v_mov_b32 %[r0], v44\n\tv_mov_b32 %[r1], v45\n\tv_mov_b32 %[r2], v46\n\tv_mov_b32 %[r3], v47\n\t"                        \
        "s_nop 15\n\t"                                                                                                           \
        : [acc] "+v"(acc), [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3)                                         \
        : [b] "v"(b), [x0] "v"(x0), [x1] "v"(x1), [a0] "v"(au[0]), [a1] "v"(au[1]), [a2] "v"(au[2]), [a3] "v"(au[3])             \
        : "v44", "v45", "v46", "v47", "memory")
    const u4 au = __builtin_bit_cast(u4, a);
    if (GAP == 0) { VAM_ASM(""); }
    else if (GAP == 1) { VAM_ASM("s_nop 0\n\t"); }
    else if (GAP == 2) { VAM_ASM("s_nop 1\n\t"); }
    else if (GAP == 4) { VAM_ASM("s_nop 3\n\t"); }
    else { VAM_ASM("s_nop 7\n\t"); }
    nv += (r0 != x0 + x1) | (r1 != x1 + x1) | (r2 != x0 * x1) | (r3 != x0 * x0);
    expect_acc += 16.f + 16.f * alpha + 16.f;
    if ((it & 255) == 255) {                              // every 256 iterations: is the accumulator still exact?
      bool wrong = false;
#pragma unroll
      for (int r = 0; r < 16; r++) wrong |= acc[r] != expect_acc;
      nm += wrong;
      acc = floatx16{0}; expect_acc = 0.f;
    }
  }
  if (nv) atomicAdd(&bad[lane >> 4], nv);
  if (nm) atomicAdd(&bad[4 + (lane >> 4)], nm);
}

template <int GAP> void run_vam(int launches, int iters, int lds_bytes) {
  unsigned* bad; CHK(hipMalloc(&bad, 32)); CHK(hipMemset(bad, 0, 32));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&valu_after_mfma_probe<GAP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  int occ = 0; CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, valu_after_mfma_probe<GAP>, kThreads, lds_bytes));
  for (int l = 0; l < launches; l++) hipLaunchKernelGGL((valu_after_mfma_probe<GAP>), dim3(2048), dim3(kThreads), lds_bytes, 0, iters, bad);
  CHK(hipGetLastError()); CHK(hipDeviceSynchronize());
  unsigned h[8]; CHK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
  printf("VALU writes v44..v47 %d wait state(s) behind the MFMA that reads v[44:47], %d workgroups/CU: %.2e sequences per lane; wrong VALU results by lane "
         "quarter %u | %u | %u | %u; inexact accumulators (checked every 256 MFMA triples) by lane quarter %u | %u | %u | %u\n", GAP, occ,
         (double)launches * 2048 * 8 * iters, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  CHK(hipFree(bad));
}

template <int K> void run_drain() {
  long long* t; float* seen; float* fin;
  CHK(hipMalloc(&t, 8)); CHK(hipMalloc(&seen, 4)); CHK(hipMalloc(&fin, 4 * 64));
  long long ht[3]; float hs[3]; float hf;
  for (int rep = 0; rep < 2; rep++) {                     // (second pass: warm instruction cache)
    hipLaunchKernelGGL((drain_probe<K, 0>), dim3(1), dim3(64), 0, 0, t, seen, fin); CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(&ht[0], t, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hs[0], seen, 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL((drain_probe<K, 1>), dim3(1), dim3(64), 0, 0, t, seen, fin); CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(&ht[1], t, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hs[1], seen, 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL((drain_probe<K, 2>), dim3(1), dim3(64), 0, 0, t, seen, fin); CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(&ht[2], t, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hs[2], seen, 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(&hf, fin, 4, hipMemcpyDeviceToHost));
  }
  printf("K = %2d dependent MFMAs (final acc[15] = %.0f = 16 K): ticks to after  nothing %lld | asm v_mov of acc[15] %lld (read %.0f) | "
         "compiler-visible read %lld (read %.0f)\n", K, hf, ht[0], ht[1], hs[1], ht[2], hs[2]);
  CHK(hipFree(t)); CHK(hipFree(seen)); CHK(hipFree(fin));
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 40;
  const int iters = argc > 2 ? atoi(argv[2]) : 2000;
  printf("== part A: what a read of the accumulator right behind K dependent v_mfma_f32_32x32x16_bf16 waits for\n");
  run_drain<1>(); run_drain<2>(); run_drain<4>(); run_drain<6>(); run_drain<16>();
  printf("== part B: in-place refill of MFMA source registers right behind their MFMAs, four waves per SIMD, all CUs\n");
  // B tiles
  uint4* gB; CHK(hipMalloc(&gB, sizeof(uint4) * 128));
  {
    uint16_t h[128 * 8];
    for (int t = 0; t < 2; t++)
      for (int l = 0; l < 64; l++) {
        const float v = (float)(((l & 31) % 5 + 1) * (t + 1));
        uint32_t u; memcpy(&u, &v, 4);
        for (int e = 0; e < 8; e++) h[(t * 64 + l) * 8 + e] = (uint16_t)(u >> 16);      // exact: small integers
      }
    CHK(hipMemcpy(gB, h, sizeof(h), hipMemcpyHostToDevice));
  }
  const int lds2 = 36 * 1024, lds1 = 84 * 1024;                                         // two / one workgroup(s) per CU
  run_war<3, true>("SAFE (compiler-placed refills)", gB, iters, launches, lds2, false);
  run_war<1, false>("A in place from LDS", gB, iters, launches, lds2, false);
  run_war<2, false>("B in place from global memory", gB, iters, launches, lds2, false);
  run_war<3, false>("A and B in place", gB, iters, launches, lds2, false);
  run_war<3, false>("A and B in place", gB, iters, launches, lds2, true);
  run_war<3, false>("A and B in place", gB, iters, launches, lds1, false);
  run_war<3, false>("A and B in place", gB, iters, launches, lds1, true);
  CHK(hipFree(gB));
  printf("== part C: VALU writes to the source registers of a just-issued MFMA\n");
  run_vam<0>(launches / 4, 4096, lds2); run_vam<1>(launches / 4, 4096, lds2); run_vam<2>(launches / 4, 4096, lds2); run_vam<4>(launches / 4, 4096, lds2);
  run_vam<8>(launches / 4, 4096, lds2); run_vam<0>(launches / 4, 4096, lds1);
  return 0;
}
