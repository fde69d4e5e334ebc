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
  const float* wscale;      //   ... and (nprod == 3) the scale of its fp16 planes
};

struct Args {
  Level lv[kMaxLevels];
  int nlev, B, Cin, Cout;
  int kh, kw, sh, sw, ph, pw, dh, dw;
  const uint16_t* planes[2];   // per layer: planes_of(packed, ..., nprod): [3 planes hi|mid|lo] bf16 or [2 planes hi|lo] fp16,
                               // each [tap][Cin/16][2][Cout][8] (pack_planes)
  const float* bias[2];        // [Cout] or nullptr
  int relu, nconv, out_nchw;
  int nprod;                   // 6 or 9 partial products of three bf16 pieces; 3 = two fp16 pieces (hi*hi, hi*lo, lo*hi)
  const float* wscale[2];      // nprod == 3: wscale_of(packed, ...) per layer (device scalar: the planes' power-of-two scale)
  unsigned* scratch;           // nprod == 3: >= 8 bytes of device memory of the caller's (max |x| of the inputs, per layer)
  const unsigned* amax_in;     // nprod == 3: nullptr (a pre-pass over the inputs fills `scratch`), or float bits of upper bounds of
  int amax_stride;             //   max |x| the producer of the inputs left: layer cv takes the maximum of the amax_count words at
  int amax_count = 0;          //   amax_in[cv * amax_stride] (0 / 1: one word)
  // GroupNorm fused around a PLAIN launch (orp_conv_split_multi_gn): tiles never span two images; the inputs are read as
  // relu?(x * a[c] + b[c]) (coef_in [layer][level][image][Cin] (a, b) pairs, or nullptr); every output tile leaves its per-group
  // statistics in gn_part [layer][tile][group] (mean, M2, max |y|, count), or nullptr
  int per_image = 0;
  const float* coef_in = nullptr;
  int relu_in = 0;
  float* gn_part = nullptr;
  int groups = 0;
};

// the tile table of a launch (what the kernel and orp_conv_split_gn_finish agree on)
struct Plan { int MT, tiles; int tile0[kMaxLevels], tpi[kMaxLevels]; };
Plan plan(const Args& a);

// cin % 64 == 0, cout % 64 == 0, taps <= 9
bool shape_ok(int c_in, int c_out, int kh, int kw);
// number of uint16_t elements of one layer's planes (bf16 planes + fp16 planes + the fp16 planes' scale)
size_t plane_elems(int c_out, int c_in, int taps);
// within the buffer orp_dcn_pack_weight fills (2 N floats of fp32 packings first): the plane set of a mode, the fp16 scale
const uint16_t* planes_of(const float* packed, int c_out, int c_in, int taps, int nprod);
const float* wscale_of(const float* packed, int c_out, int c_in, int taps);
hipError_t pack_planes(const float* weight /* [o][c][tap] */, int c_out, int c_in, int taps, uint16_t* planes, hipStream_t st);
hipError_t launch(const Args& a, hipStream_t st);
// dev aid (orp_debug_amax_log): returns the number of launches logged since the previous call
int set_amax_log(unsigned* log, int capacity_launches);

}  // namespace orp_split
