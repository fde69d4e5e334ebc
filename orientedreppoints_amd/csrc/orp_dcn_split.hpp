// orp_dcn_split.hpp -- interface between orp_dcn.hip (entry points, layout conversion, tile table) and orp_dcn_split.hip (the
// fp32-by-bf16-splitting DeformConv forward kernel).  Internal to liborp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace orp_split {

constexpr int kMaxLevels = 8;

struct Level {
  const float* x[2];    // NHWC [B, H, W, Cin] of layer 0 / layer 1 (pair launch: both layers read the same offsets / mask)
  const float* off;     // NCHW [B, 2*taps, Ho, Wo]
  const float* mask;    // DCNv2 modulation [B, taps, Ho, Wo] or nullptr
  float* out[2];        // NCHW [B, Cout, Ho, Wo] or NHWC [B, Ho, Wo, Cout]
  int H, W, Ho, Wo;
  const uint16_t* planes;   // nullptr, or this level's own weight planes / bias (single-layer launches: a different layer per
  const float* bias;        // level, e.g. the FPN's output convolutions) instead of Args::planes[0] / bias[0]
};

struct Args {
  Level lv[kMaxLevels];
  int nlev, B, Cin, Cout;
  int kh, kw, sh, sw, ph, pw, dh, dw;
  const uint16_t* planes[2];   // per layer: [3 planes hi|mid|lo][tap][Cin/16][2][Cout][8] bf16 (pack_planes)
  const float* bias[2];        // [Cout] or nullptr
  int relu, nconv, out_nchw;
  int nprod;                   // 6 or 9 partial products per (a, w) pair
};

// cin % 64 == 0, cout % 64 == 0, taps <= 9
bool shape_ok(int c_in, int c_out, int kh, int kw);
// number of uint16_t elements of one layer's planes
size_t plane_elems(int c_out, int c_in, int taps);
hipError_t pack_planes(const float* weight /* [o][c][tap] */, int c_out, int c_in, int taps, uint16_t* planes, hipStream_t st);
hipError_t launch(const Args& a, hipStream_t st);

}  // namespace orp_split
