// tests/host_harness/libm_host.cpp -- TEST INFRASTRUCTURE: g++ compiles the device header orp_libm.hpp and counts the floats
// on which its cosf_host / sinf_host differ from the C library this process runs on (tests/test_libm_host.py).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../orientedreppoints_amd/csrc/orp_libm.hpp"

extern "C" {
// every `stride`-th float bit pattern in [lo_bits, hi_bits), both signs; returns the number of mismatches, first one in *first
long host_libm_mismatches(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride, int which, float* first) {
  long bad = 0;
  for (int neg = 0; neg < 2; neg++)
    for (uint64_t u = lo_bits; u < hi_bits; u += stride) {
      uint32_t b = (uint32_t)u | (neg ? 0x80000000u : 0u);
      float x;
      memcpy(&x, &b, 4);
      const float want = which ? sinf(x) : cosf(x);
      const float got = which ? orp::libm::sinf_host(x) : orp::libm::cosf_host(x);
      if (memcmp(&want, &got, 4) != 0) {
        if (bad == 0 && first) *first = x;
        bad++;
      }
    }
  return bad;
}
void host_libm_eval(const float* x, int n, int which, float* out) {
  for (int i = 0; i < n; i++) out[i] = which ? orp::libm::sinf_host(x[i]) : orp::libm::cosf_host(x[i]);
}
}
