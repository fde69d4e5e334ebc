// orp_launch.hpp -- launch helpers shared by the kernels that need more than 64 KB of dynamic LDS.
#pragma once
#include <hip/hip_runtime.h>

namespace orp {

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
