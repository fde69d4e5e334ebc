// orp_launch.hpp -- launch helpers shared by the kernels that need more than 64 KB of dynamic LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace orp {

// Stream-ordered fill of device memory as a KERNEL launch, never hipMemsetAsync: every entry point of this library may be captured
// into a hipGraph (mmdet_models/graph_inference.py), and a captured memset node was seen to write a wrong pattern in replays that
// followed eager work on the same stream -- range words "zeroed" by a memset node were read back as 0x80808080 by the kernel
// behind it (ROCm 7.2, PyTorch 2.10 graphs; tests/checks/graph_bitwise.py with ORP_FILL=memset).  A kernel node has
// its arguments baked into the graph like every other launch.  ORP_FILL=memset switches back (A/B aid).
namespace {
__global__ void fill_words_kernel(unsigned* __restrict__ p, unsigned v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_bytes_kernel(unsigned char* __restrict__ p, unsigned char v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
}  // namespace
inline hipError_t fill_async(void* p, int byte_value, size_t nbytes, hipStream_t st) {
  if (nbytes == 0) return hipSuccess;
  static const bool use_memset = getenv("ORP_FILL") && getenv("ORP_FILL")[0] == 'm';
  if (use_memset) return hipMemsetAsync(p, byte_value, nbytes, st);
  const unsigned b = (unsigned)byte_value & 0xffu;
  if (((uintptr_t)p & 3) == 0 && (nbytes & 3) == 0) {
    const size_t n = nbytes >> 2;
    const size_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, st, reinterpret_cast<unsigned*>(p), b * 0x01010101u, n);
  } else {
    const size_t nb = (nbytes + 255) / 256;
    hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, st, reinterpret_cast<unsigned char*>(p), (unsigned char)b, nbytes);
  }
  return hipGetLastError();
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property and must not be called while a stream is
// being captured: do it once per (kernel instantiation, device).  `Tag` makes one flag array per call site.
template <typename Tag>
inline hipError_t set_max_dynamic_lds_once(const void* fn, size_t bytes) {
  static bool done[32] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 32) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (done[dev]) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done[dev] = true;
  return e;
}

}  // namespace orp
