// Microbenchmark (dev aid): rate of coalesced fp32 atomic adds into NHWC rows, agent scope vs workgroup scope into a
// per-XCD private copy (selected by the hardware XCC_ID), and whether the per-XCD copies add up exactly.
//   hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate && ./atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ inline int xcc_id() {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 15;
}

template <int MODE>
__global__ void __launch_bounds__(512) scatter(float* buf, size_t copy_stride, int rows, int iters, int* xcc_seen) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = xcc_id();
  if (threadIdx.x == 0) xcc_seen[blockIdx.x] = x;
  float* base = buf + (MODE == 1 ? (size_t)x * copy_stride : 0);
  // each workgroup works on a band of rows near its own "tile" (like the conv scatter), with some spread
  unsigned s = blockIdx.x * 9781u + wave * 7919u + 12345u;
  const int band0 = (int)(((long)blockIdx.x * rows) / gridDim.x);
  for (int i = 0; i < iters; i++) {
    s = s * 1664525u + 1013904223u;
    int r = band0 + (int)((s >> 8) % 512u) - 256;
    r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
    float* p = base + (size_t)r * 256 + (lane & 31) + ((lane >> 5) ? 128 : 0) + (wave & 3) * 32;
    if (MODE == 0) atomicAdd(p, 1.0f);
    else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

__global__ void reduce8(const float* copies, size_t copy_stride, size_t n, float* out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int c = 0; c < 8; c++) a += copies[c * copy_stride + i];
    out[i] = a;
  }
}

int main() {
  const int rows = 2 * 21824, iters = 2000, grid = 1364;
  const size_t n = (size_t)rows * 256;
  float *buf, *out; int* seen;
  hipMalloc(&buf, n * 8 * sizeof(float)); hipMalloc(&out, n * sizeof(float)); hipMalloc(&seen, grid * sizeof(int));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 2; mode++) {
    hipMemset(buf, 0, n * 8 * sizeof(float));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipMemset(buf, 0, n * 8 * sizeof(float));
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(scatter<0>, dim3(grid), dim3(512), 0, 0, buf, n, rows, iters, seen);
      else hipLaunchKernelGGL(scatter<1>, dim3(grid), dim3(512), 0, 0, buf, n, rows, iters, seen);
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    }
    const double lanes = (double)grid * 8 * 64 * iters;
    printf("mode %d (%s): %.3f ms, %.1f G lane-atomics/s\n", mode, mode ? "workgroup scope, per-XCC copy" : "agent scope", ms, lanes / ms * 1e-6);
    if (mode == 1) {
      hipLaunchKernelGGL(reduce8, dim3(2048), dim3(256), 0, 0, buf, n, n, out);
      std::vector<float> h(n);
      hipMemcpy(h.data(), out, n * sizeof(float), hipMemcpyDeviceToHost);
      double tot = 0; for (size_t i = 0; i < n; i++) tot += h[i];
      printf("sum over copies %.0f, expected %.0f\n", tot, lanes);
      std::vector<int> hs(grid);
      hipMemcpy(hs.data(), seen, grid * sizeof(int), hipMemcpyDeviceToHost);
      int mism = 0, hist[16] = {0};
      for (int i = 0; i < grid; i++) { hist[hs[i] & 15]++; if ((hs[i] & 15) != (i & 7)) mism++; }
      printf("xcc histogram:"); for (int i = 0; i < 16; i++) if (hist[i]) printf(" %d:%d", i, hist[i]);
      printf("   blocks with xcc != blockIdx%%8: %d\n", mism);
    } else {
      std::vector<float> h(n);
      hipMemcpy(h.data(), buf, n * sizeof(float), hipMemcpyDeviceToHost);
      double tot = 0; for (size_t i = 0; i < n; i++) tot += h[i];
      printf("sum %.0f, expected %.0f\n", tot, lanes);
    }
  }
  return 0;
}
