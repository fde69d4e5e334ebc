// Development aid (GPU box): sustained issue rate of v_mfma_f32_32x32x16_bf16 (and 16x16x32) with NACC independent
// accumulators per wave and 1 / 2 waves per SIMD, operands in registers, on RANDOM bit patterns (toggle-dependent power),
// plus the shader clock it ran at (s_memtime ticks vs wall).  hipcc --offload-arch=gfx950 -O3 mfma_rate_bf16.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__device__ inline bf8 mk(uint32_t seed) {
  union { bf8 v; uint32_t u[4]; } x;
  for (int i = 0; i < 4; i++) { seed = seed * 1664525u + 1013904223u; x.u[i] = (seed & 0x807f807fu) | 0x3f003f00u; }   // magnitudes ~ [0.5, 1)
  return x.v;
}
template <int NACC, int NOP>
__global__ void k32(float* out, long long* clk, int iters, uint32_t seed) {
  floatx16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = floatx16{0};
  bf8 a[NOP], b[NOP];
#pragma unroll
  for (int i = 0; i < NOP; i++) { a[i] = mk(seed + threadIdx.x * 17 + i); b[i] = mk(seed * 3 + threadIdx.x * 5 + i); }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 6; u++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + i) % NOP], b[u % NOP], acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <typename K> void run(const char* name, K kern, int threads, int nacc, int blocks) {
  float* out; hipMalloc(&out, sizeof(float) * 1024 * 1024);
  long long* clk; hipMalloc(&clk, sizeof(long long) * 1024);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, clk, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, clk, iters, 7u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[1024]; hipMemcpy(h, clk, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < blocks; i++) mean += (double)h[i]; mean /= blocks;
  const double mf = (double)blocks * (threads / 64) * iters * 6.0 * nacc;
  const double per_simd = mf / (256.0 * 4);
  printf("%-10s blocks %3d threads %4d nacc %d: %.3f ms  %.0f TFLOP/s;  s_memtime ticks %.3e -> %.1f ticks per MFMA per SIMD, tick rate %.0f MHz\n",
         name, blocks, threads, nacc, ms, mf * 32768.0 / ms / 1e9, mean, mean / (per_simd * 256.0 / blocks), mean / (ms * 1e3));
  hipFree(out); hipFree(clk);
}
int main() {
  run("32x32x16", k32<3, 3>, 512, 3, 256);
  run("32x32x16", k32<3, 3>, 256, 3, 256);
  run("32x32x16", k32<3, 3>, 512, 3, 228);
  run("32x32x16", k32<3, 3>, 512, 3, 128);
  run("32x32x16", k32<3, 3>, 512, 3, 64);
  run("32x32x16", k32<1, 3>, 512, 1, 256);
  run("32x32x16", k32<2, 3>, 512, 2, 256);
  run("32x32x16", k32<4, 3>, 512, 4, 256);
  run("32x32x16", k32<3, 1>, 512, 3, 256);
  return 0;
}
