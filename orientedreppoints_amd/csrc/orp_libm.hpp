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


// ---- expf / logf / powf of the host C library (glibc >= 2.28: "optimized-routines" e_expf.c, e_logf.c, e_powf.c with exp2f_data.c,
// logf_data.c, powf_log2_data.c) -- what the sigmoid focal loss of the reference compiled for the host evaluates
// (sigmoid_focal_loss_cuda.cu:36-57, 73-96).  Tables below are those files' constants (read back from the library this was developed
// against and compared with it: tests/test_libm_host.py -- every float for expf and logf, 3.3e9 (x, y) pairs for powf: 0 mismatches).
// One fused operation: expf's reduction r = InvLn2N * x - k is an FMA in the x86-64 library's FMA build (2 of 2^32 floats differ
// otherwise); everything else is unfused IEEE double arithmetic.  Error handling (errno, exceptions) is not reproduced, values are.
ORP_LIBM_FN uint64_t exp2f_tab(unsigned i) {
  static constexpr uint64_t T[32] = {
      0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
      0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
      0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
      0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
      0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
      0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
      0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
      0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
  };
  return T[i & 31];
}
ORP_LIBM_FN double bits_to_double(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
ORP_LIBM_FN uint64_t double_to_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
ORP_LIBM_FN uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
ORP_LIBM_FN float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

ORP_LIBM_FN float expf_host(float x) {
  const double xd = (double)x;
  const uint32_t abstop = (float_to_bits(x) >> 20) & 0x7ff;
  if (abstop >= (float_to_bits(88.0f) >> 20)) {                      // |x| >= 88 or NaN
    if (float_to_bits(x) == 0xff800000u) return 0.0f;
    if (abstop >= (0x7f800000u >> 20)) return x + x;
    if (x > 0x1.62e42ep6f) return __builtin_inff();                  // overflow
    if (x < -0x1.9fe368p6f) return 0.0f;                             // underflow
  }
  const double shift = 0x1.8p+52, invln2n = 0x1.71547652b82fep+5;
  const double c0 = 0x1.c6af84b912394p-20, c1 = 0x1.ebfce50fac4f3p-13, c2 = 0x1.62e42ff0c52d6p-6;
  double z = invln2n * xd;
  double kd = z + shift;
  const uint64_t ki = double_to_bits(kd);
  kd -= shift;
  const double r = __builtin_fma(invln2n, xd, -kd);
  uint64_t t = exp2f_tab((unsigned)(ki & 31));
  t += ki << (52 - 5);
  const double s = bits_to_double(t);
  z = c0 * r + c1;
  const double r2 = r * r;
  double y = c2 * r + 1.0;
  y = z * r2 + y;
  y = y * s;
  return (float)y;
}

ORP_LIBM_FN double logf_tab(unsigned i) {
  static constexpr double T[32] = {                                    // {invc, logc} x 16
      0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,
      0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,
      0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3,
      0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,
      0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1.0000000000000p+0, 0x0.0p+0,
      0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5, 0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4,
      0x1.b2036576afce6p-1, 0x1.526e57720db08p-3, 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3,
      0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2, 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2,
  };
  return T[i & 31];
}
ORP_LIBM_FN float logf_host(float x) {
  uint32_t ix = float_to_bits(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {                // x < 2^-126, inf or NaN
    if (ix * 2 == 0) return -__builtin_inff();
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return (x - x) / (x - x);
    ix = float_to_bits(x * 0x1p23f);
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> (23 - 4)) & 15);
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & (0x1ffu << 23));
  const double invc = logf_tab(2 * i), logc = logf_tab(2 * i + 1);
  const double z = (double)bits_to_float(iz);
  const double ln2 = 0x1.62e42fefa39efp-1, a0 = -0x1.00ea348b88334p-2, a1 = 0x1.5575b0be00b6ap-2, a2 = -0x1.ffffef20a4123p-2;
  const double r = z * invc - 1.0;
  const double y0 = logc + (double)k * ln2;
  const double r2 = r * r;
  double y = a1 * r + a2;
  y = a0 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

ORP_LIBM_FN double powf_log2_tab(unsigned i) {
  static constexpr double T[32] = {                                    // {invc, log2(c)} x 16
      0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2, 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2,
      0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2, 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2,
      0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2, 0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3,
      0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3, 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4,
      0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5, 0x1.0000000000000p+0, 0x0.0p+0,
      0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4, 0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3,
      0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3, 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2,
      0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2, 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2,
  };
  return T[i & 31];
}
ORP_LIBM_FN int powf_checkint(uint32_t iy) {                           // 0: not an integer, 1: odd, 2: even
  const int e = (int)(iy >> 23 & 0xff);
  if (e < 0x7f) return 0;
  if (e > 0x7f + 23) return 2;
  if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
  if (iy & (1u << (0x7f + 23 - e))) return 1;
  return 2;
}
ORP_LIBM_FN bool powf_zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }
ORP_LIBM_FN float powf_host(float x, float y) {
  uint32_t sign_bias = 0;
  uint32_t ix = float_to_bits(x);
  const uint32_t iy = float_to_bits(y);
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy)) {
    if (powf_zeroinfnan(iy)) {
      if (2 * iy == 0) return 1.0f;
      if (ix == 0x3f800000u) return 1.0f;
      if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
      if (2 * ix == 2 * 0x3f800000u) return 1.0f;
      if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
      return y * y;
    }
    if (powf_zeroinfnan(ix)) {
      float x2 = x * x;
      if ((ix & 0x80000000u) && powf_checkint(iy) == 1) x2 = -x2;
      return (iy & 0x80000000u) ? 1 / x2 : x2;
    }
    if (ix & 0x80000000u) {                                            // finite x < 0
      const int yint = powf_checkint(iy);
      if (yint == 0) return (x - x) / (x - x);
      if (yint == 1) sign_bias = 1u << (5 + 11);
      ix &= 0x7fffffffu;
    }
    if (ix < 0x00800000u) {                                            // subnormal x
      ix = float_to_bits(x * 0x1p23f);
      ix &= 0x7fffffffu;
      ix -= 23u << 23;
    }
  }
  // log2(x) in double
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> (23 - 4)) & 15);
  const uint32_t top = tmp & 0xff800000u;
  const uint32_t iz = ix - top;
  const int k = (int32_t)top >> 23;
  const double invc = powf_log2_tab(2 * i), logc = powf_log2_tab(2 * i + 1);
  const double z = (double)bits_to_float(iz);
  const double p0 = 0x1.27616c9496e0bp-2, p1 = -0x1.71969a075c67ap-2, p2 = 0x1.ec70a6ca7baddp-2, p3 = -0x1.7154748bef6c8p-1,
               p4 = 0x1.71547652ab82bp+0;
  const double r = z * invc - 1.0;
  const double y0 = logc + (double)k;
  const double r2 = r * r;
  double yy = p0 * r + p1;
  const double p = p2 * r + p3;
  const double r4 = r2 * r2;
  double q = p4 * r + y0;
  q = p * r2 + q;
  yy = yy * r4 + q;
  const double ylogx = (double)y * yy;
  if ((double_to_bits(ylogx) >> 47 & 0xffff) >= (double_to_bits(126.0) >> 47)) {       // |y log2 x| >= 126
    if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();
    if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
  }
  // 2^(y log2 x)
  const double shift_scaled = 0x1.8p+47, c0 = 0x1.c6af84b912394p-5, c1 = 0x1.ebfce50fac4f3p-3, c2 = 0x1.62e42ff0c52d6p-1;
  double kd = ylogx + shift_scaled;
  const uint64_t ki = double_to_bits(kd);
  kd -= shift_scaled;
  const double rr = ylogx - kd;
  uint64_t t = exp2f_tab((unsigned)(ki & 31));
  const uint64_t ski = ki + sign_bias;
  t += ski << (52 - 5);
  const double s = bits_to_double(t);
  const double zz = c0 * rr + c1;
  const double rr2 = rr * rr;
  double yo = c2 * rr + 1.0;
  yo = zz * rr2 + yo;
  yo = yo * s;
  return (float)yo;
}

}  // namespace libm
}  // namespace orp
