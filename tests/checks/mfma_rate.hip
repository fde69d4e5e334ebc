// Development aid (GPU box): achievable issue rate of v_mfma_f32_32x32x2_f32 / 16x16x4 with NACC independent accumulators
// per wave and WAVES waves per SIMD, operands in registers (no memory traffic).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
  floatx16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = floatx16{0};
  float av = a + threadIdx.x, bv = b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
  floatx4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = floatx4{0};
  float av = a + threadIdx.x, bv = b;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K> void run(const char* name, K kern, int threads, int nacc, double flop_per_mfma) {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 1024);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = 256.0 * (threads / 64) * iters * 8.0 * nacc;
  printf("%-28s threads %4d nacc %d: %.3f ms  %.1f TFLOP/s  (%.1f clk per MFMA per SIMD at 2.4 GHz)\n", name, threads, nacc, ms,
         mf * flop_per_mfma / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / (256.0 * 4)));
  hipFree(out);
}
int main() {
  run("32x32x2 f32", k32<1>, 256, 1, 4096); run("32x32x2 f32", k32<2>, 256, 2, 4096); run("32x32x2 f32", k32<3>, 256, 3, 4096);
  run("32x32x2 f32", k32<4>, 256, 4, 4096); run("32x32x2 f32", k32<3>, 512, 3, 4096); run("32x32x2 f32", k32<2>, 512, 2, 4096);
  run("32x32x2 f32", k32<1>, 512, 1, 4096); run("32x32x2 f32", k32<1>, 1024, 1, 4096);
  run("16x16x4 f32", k16<4>, 256, 4, 2048); run("16x16x4 f32", k16<8>, 256, 8, 2048); run("16x16x4 f32", k16<4>, 512, 4, 2048);
  return 0;
}
