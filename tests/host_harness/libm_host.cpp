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
      const float want = which == 0 ? cosf(x) : which == 1 ? sinf(x) : which == 2 ? expf(x) : logf(x);
      const float got = which == 0 ? orp::libm::cosf_host(x) : which == 1 ? orp::libm::sinf_host(x)
                      : which == 2 ? orp::libm::expf_host(x) : orp::libm::logf_host(x);
      if (memcmp(&want, &got, 4) != 0 && !(want != want && got != got)) {
        if (bad == 0 && first) *first = x;
        bad++;
      }
    }
  return bad;
}
void host_libm_eval(const float* x, int n, int which, float* out) {
  for (int i = 0; i < n; i++)
    out[i] = which == 0 ? orp::libm::cosf_host(x[i]) : which == 1 ? orp::libm::sinf_host(x[i])
           : which == 2 ? orp::libm::expf_host(x[i]) : orp::libm::logf_host(x[i]);
}
// powf: every `stride`-th float x of [lo_bits, hi_bits) against each of the n exponents, then `nrand` xorshift (x, y) pairs
long host_powf_mismatches(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride, const float* ys, int n, long nrand, float* first) {
  long bad = 0;
  for (int g = 0; g < n; g++)
    for (uint64_t u = lo_bits; u < hi_bits; u += stride) {
      float x; const uint32_t b = (uint32_t)u; memcpy(&x, &b, 4);
      const float want = powf(x, ys[g]), got = orp::libm::powf_host(x, ys[g]);
      if (memcmp(&want, &got, 4) != 0 && !(want != want && got != got)) { if (bad == 0 && first) { first[0] = x; first[1] = ys[g]; } bad++; }
    }
  uint64_t s = 88172645463325252ull;
  for (long i = 0; i < nrand; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    float x, y; const uint32_t a = (uint32_t)s, c = (uint32_t)(s >> 32); memcpy(&x, &a, 4); memcpy(&y, &c, 4);
    const float want = powf(x, y), got = orp::libm::powf_host(x, y);
    if (memcmp(&want, &got, 4) != 0 && !(want != want && got != got)) { if (bad == 0 && first) { first[0] = x; first[1] = y; } bad++; }
  }
  return bad;
}
}
