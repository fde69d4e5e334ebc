// orp_libm.hpp -- single-precision cos / sin whose results are the HOST C library's, bit for bit.
//
// Why: minareabbox (mmdet/ops/minarearect/src/minarearect_kernel.cu:113-120) builds its rotation from cos(float) and
// keeps "the first strictly smaller area"; on rectangle-like hulls two edge directions give areas that differ in the last
// bit, so ONE ulp of one cosine decides which rectangle (which corner order) comes out.  The parity oracle is the reference
// compiled for the host, i.e. glibc's cosf -- and glibc's cosf is NOT correctly rounded (its header says 0.56 ulp worst
// case; measured here: 0.13 % of all floats in [2^-31, 4) differ from (float)cos((double)x)).  A correctly rounded device
// cosine therefore disagrees with the oracle on exactly the tie cases.  The way to agree is to evaluate the same published
// algorithm: the single-step reduction + degree-8 / degree-7 double-precision polynomials of "optimized-routines" sincosf
// (W. Dijkstra, Arm; glibc >= 2.28 sysdeps/ieee754/flt-32/s_cosf.c, s_sinf.c, sincosf.h, s_sincosf_data.c).  Constants
// below are that table's; tests/test_libm_host.py compiles this header with g++ and checks every float in (-96, 96)
// against the C library it runs on, and a -m gpu test does the same for the gfx950 build.
//
// Arithmetic notes: everything is IEEE double + - * (translation units including this header are built with
// -ffp-contract=off); the one fused operation is the reduction x - n*(pi/2), which the x86-64 C library performs as an FMA
// (its ifunc picks the -mfma build on every CPU with FMA3; for |n| <= 2, i.e. |x| < pi + pi/4, the product is exact and
// fused / unfused agree anyway).  |x| >= 120 (never produced on this path) falls back to the double-precision routine.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define ORP_LIBM_FN __host__ __device__ __forceinline__
#else
#define ORP_LIBM_FN static inline
#endif

namespace orp {
namespace libm {

ORP_LIBM_FN uint32_t top12(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return (u >> 20) & 0x7ffu;
}

// sinf_poly of sincosf.h: quadrant even -> sine polynomial in x, odd -> cosine polynomial in x2; `cs` = +-1 is the sign the
// second table entry carries on its cosine coefficients
ORP_LIBM_FN float sincos_poly(double x, double x2, double cs, int n) {
  if ((n & 1) == 0) {
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double x3 = x * x2;
    const double t1 = s2 + x2 * s3;
    const double x7 = x3 * x2;
    const double s = x + x3 * s1;
    return (float)(s + x7 * t1);
  }
  const double c0 = cs * 0x1p0, c1 = cs * -0x1.ffffffd0c621cp-2, c2 = cs * 0x1.55553e1068f19p-5,
               c3 = cs * -0x1.6c087e89a359dp-10, c4 = cs * 0x1.99343027bf8c3p-16;
  const double x4 = x2 * x2;
  const double t2 = c3 + x2 * c4;
  const double t1 = c0 + x2 * c1;
  const double x6 = x4 * x2;
  const double c = t1 + x4 * c2;
  return (float)(c + x6 * t2);
}

// reduce_fast (the !TOINT_INTRINSICS form x86-64 builds): quadrant from a 2^24-scaled product, truncation made rounding by the
// added half
ORP_LIBM_FN double reduce_fast(double x, int* np) {
  const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
  const double r = x * hpi_inv;
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return __builtin_fma(-(double)n, hpi, x);
}

ORP_LIBM_FN float cosf_host(float y) {
  double x = (double)y;
  const uint32_t t = top12(y);
  if (t < top12(0x1.921FB6p-1f)) {                 // |y| < pi/4
    if (t < top12(0x1p-12f)) return 1.0f;
    return sincos_poly(x, x * x, 1.0, 1);
  }
  if (t < top12(120.0f)) {
    int n;
    x = reduce_fast(x, &n);
    const double s = ((n + 1) & 2) ? -1.0 : 1.0;   // sign[n & 3] = {1, -1, -1, 1}
    const double cs = (n & 2) ? -1.0 : 1.0;
    return sincos_poly(x * s, x * x, cs, n ^ 1);
  }
  return (float)cos(x);
}

ORP_LIBM_FN float sinf_host(float y) {
  double x = (double)y;
  const uint32_t t = top12(y);
  if (t < top12(0x1.921FB6p-1f)) {
    if (t < top12(0x1p-12f)) return y;
    return sincos_poly(x, x * x, 1.0, 0);
  }
  if (t < top12(120.0f)) {
    int n;
    x = reduce_fast(x, &n);
    const double s = ((n + 1) & 2) ? -1.0 : 1.0;
    const double cs = (n & 2) ? -1.0 : 1.0;
    return sincos_poly(x * s, x * x, cs, n);
  }
  return (float)sin(x);
}

}  // namespace libm
}  // namespace orp
