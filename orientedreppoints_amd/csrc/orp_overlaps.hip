// orp_overlaps.hip -- pairwise rotated / polygon IoU matrices for gfx950.
//
// Replaces
//   DOTA_devkit/poly_nms_gpu/poly_overlaps_kernel.cu:280-427  (RotBox2Poly, devPolyIoU, overlaps_kernel, _overlaps)
// and provides the all-pairs fp32 quad IoU (devrIoU / devPolyIoU arithmetic) used by tests and by the merge /
// evaluation tools.  Layout: lane = column (query / b row) so the [N,K] result is written coalesced; a workgroup
// owns 16 rows x 64 columns and runs the tile phases of orp_tile.hpp (exact-zero pair classifier, per-term exact-zero
// screen, then the register decision tree on the surviving fan terms).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/orp_hip.h"
#include "orp_geom.hpp"
#include "orp_libm.hpp"
#include "orp_quadfast.hpp"
#include "orp_tile.hpp"

namespace {
using orp::Pt;
constexpr int kThreads = 256;

// (cx,cy,w,h,theta) -> 4 corners, mixed precision exactly as poly_overlaps_kernel.cu:280-297
// (cos/sin of the float angle; products with (w / 2.0) in double; one rounding to float per coordinate).
__device__ __forceinline__ void rotbox2poly(const float* dbox, float* p8) {
  const float cs = orp::libm::cosf_host(dbox[4]);       // the host C library's cosf / sinf, bit for bit (orp_libm.hpp)
  const float ss = orp::libm::sinf_host(dbox[4]);
  const float w = dbox[2], h = dbox[3], x_ctr = dbox[0], y_ctr = dbox[1];
  p8[0] = (float)(x_ctr + cs * (w / 2.0) - ss * (-h / 2.0));
  p8[2] = (float)(x_ctr + cs * (w / 2.0) - ss * (h / 2.0));
  p8[4] = (float)(x_ctr + cs * (-w / 2.0) - ss * (h / 2.0));
  p8[6] = (float)(x_ctr + cs * (-w / 2.0) - ss * (-h / 2.0));
  p8[1] = (float)(y_ctr + ss * (w / 2.0) + cs * (-h / 2.0));
  p8[3] = (float)(y_ctr + ss * (w / 2.0) + cs * (h / 2.0));
  p8[5] = (float)(y_ctr + ss * (-w / 2.0) + cs * (h / 2.0));
  p8[7] = (float)(y_ctr + ss * (-w / 2.0) + cs * (-h / 2.0));
}

// MODE 0: rows are quads (stride floats apart); MODE 1: rows are 5-param rotated boxes.
// One workgroup = a tile of kRows rows x 64 columns, evaluated like the NMS mask tile (orp_tile.hpp): every box of
// the tile is prepared once into LDS (orientation, oriented fan triangles, signs, |area|), phase A (lane = column,
// row wave-uniform) resolves the pairs whose intersection is exactly 0 with the division-free classifier and writes
// 0/union directly, the rest are queued and drained term by term (orp_tile::tile_drain_terms).
constexpr int kRows = 16;              // 4 rows per wave

struct RowFar { float vx[4], vy[4]; float mabs; int slow; };

template <int MODE, bool GUARD>
__global__ void __launch_bounds__(kThreads, 4)
iou_matrix_kernel(const float* __restrict__ a, int n, const float* __restrict__ b, int k, int stride,
                  float* __restrict__ out) {
  __shared__ orp_tile::TileLds T;
  __shared__ orp_tile::TermLds X;
  __shared__ RowFar rowF[kRows];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int row_base = blockIdx.y * kRows;

  // every wave prepares its lane's column box (registers: classifier constants); wave 0 also files the LDS record
  orp::FarCol fc;
  {
    float q8[8];
    if (col < k) {
      if (MODE == 0) {
        const float* cp = b + (size_t)col * stride;
#pragma unroll
        for (int i = 0; i < 8; i++) q8[i] = cp[i];
      } else {
        rotbox2poly(b + (size_t)col * 5, q8);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) q8[i] = 0.f;
    }
    orp::QuadPrep cp;
    orp::quad_prepare(q8, cp);
    fc = orp::far_col(cp);
    if (wave == 0) {
#pragma unroll
      for (int e = 0; e < 4; e++) T.colE[e][lane] = make_float4(cp.ax[e], cp.ay[e], cp.bx[e], cp.by[e]);
      T.colS[lane] = orp_tile::pack_signs(cp);
      T.colArea[lane] = cp.area_abs;
      X.colM[lane] = cp.mabs;
    }
  }
  if (tid < kRows) {
    const int r = row_base + tid;
    float p8[8];
    if (r < n) {
      if (MODE == 0) {
        const float* rp = a + (size_t)r * stride;
#pragma unroll
        for (int i = 0; i < 8; i++) p8[i] = rp[i];
      } else {
        rotbox2poly(a + (size_t)r * 5, p8);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) p8[i] = 0.f;
    }
    orp::QuadPrep rp;
    orp::quad_prepare(p8, rp);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      T.rowE[e][tid] = make_float4(rp.ax[e], rp.ay[e], rp.bx[e], rp.by[e]);
      rowF[tid].vx[e] = rp.vx[e]; rowF[tid].vy[e] = rp.vy[e];
    }
    rowF[tid].mabs = rp.mabs; rowF[tid].slow = rp.force_slow;
    T.rowS[tid] = orp_tile::pack_signs(rp);
    T.rowArea[tid] = rp.area_abs;
    X.rowM[tid] = rp.mabs;
  }
  if (tid == 0) T.qcount = 0;
  orp_tile::term_lds_reset(X, tid);
  __syncthreads();
  const bool cslow = (T.colS[lane] >> 8) != 0;
  const float carea = T.colArea[lane];

  // ---- phase A ---------------------------------------------------------------------------------------------------
  const int rl_first = __builtin_amdgcn_readfirstlane(wave * (kRows / 4));
  for (int rr = 0; rr < kRows / 4; rr++) {
    const int rl = rl_first + rr;                        // wave-uniform
    const int r = row_base + rl;
    if (r >= n) break;
    const bool valid = col < k;
    bool resolved = false;
    if (valid && !(cslow | (rowF[rl].slow != 0))) {
      float rvx[4], rvy[4];
#pragma unroll
      for (int e = 0; e < 4; e++) { rvx[e] = rowF[rl].vx[e]; rvy[e] = rowF[rl].vy[e]; }   // uniform address: LDS broadcast
      resolved = orp::pair_is_far(rvx, rvy, rowF[rl].mabs, fc);
    }
    if (resolved) out[(size_t)r * k + col] = orp::iou_of_zero_inter<GUARD>(T.rowArea[rl], carea);
    const bool pend = valid && !resolved;
    const unsigned long long pmask = __ballot(pend);
    if (pmask) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&T.qcount, __popcll(pmask));
      base = __builtin_amdgcn_readfirstlane(base);
      if (pend) T.queue[base + __popcll(pmask & ((1ull << lane) - 1ull))] = (unsigned short)((rl << 6) | lane);
    }
  }
  __syncthreads();

  // ---- phase B: per-term screen, one surviving fan term per lane, ordered sum per pair (orp_tile.hpp) ----------
  const size_t col0 = (size_t)blockIdx.x * 64;
  orp_tile::tile_drain_terms<GUARD>(T, X, T.qcount, [&](int rl, int cl, float iou) {
    out[(size_t)(row_base + rl) * k + col0 + cl] = iou;
  });
}

int launch(int mode, int guard, const float* a, int n, const float* b, int k, int stride, float* out, hipStream_t st) {
  if (n < 0 || k < 0 || ((n > 0 && k > 0) && (!a || !b || !out))) return ORP_EINVAL;
  if (n == 0 || k == 0) return ORP_OK;
  dim3 grid((k + 63) / 64, (n + kRows - 1) / kRows), block(kThreads);
  if (mode == 1) hipLaunchKernelGGL((iou_matrix_kernel<1, true>), grid, block, 0, st, a, n, b, k, 5, out);
  else if (guard) hipLaunchKernelGGL((iou_matrix_kernel<0, true>), grid, block, 0, st, a, n, b, k, stride, out);
  else hipLaunchKernelGGL((iou_matrix_kernel<0, false>), grid, block, 0, st, a, n, b, k, stride, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ORP_OK : (int)e;
}
}  // namespace

extern "C" {
int orp_quad_iou_matrix(const float* a, int n, const float* b, int k, int stride, int guard, float* out, void* stream) {
  if (stride < 8) return ORP_EINVAL;
  return launch(0, guard, a, n, b, k, stride, out, (hipStream_t)stream);
}
int orp_poly_overlaps(const float* boxes, int n, const float* query, int k, float* out, void* stream) {
  return launch(1, 1, boxes, n, query, k, 5, out, (hipStream_t)stream);
}

// Host-pointer API, exact signature of DOTA_devkit/poly_nms_gpu/poly_overlaps.hpp:1 (errors are printed, as the
// reference's CUDA_CHECK does, poly_overlaps_kernel.cu:20-27).
void _overlaps(float* overlaps_host, const float* boxes_host, const float* query_boxes_host, int n, int k,
               int device_id) {
  if (n <= 0 || k <= 0) return;
  float *d_o = nullptr, *d_b = nullptr, *d_q = nullptr;
  int rc;
#define ORP_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "_overlaps: %s\n", hipGetErrorString(e_)); goto done; } } while (0)
  ORP_CHK(hipSetDevice(device_id));
  ORP_CHK(hipMalloc(&d_b, sizeof(float) * 5 * (size_t)n));
  ORP_CHK(hipMalloc(&d_q, sizeof(float) * 5 * (size_t)k));
  ORP_CHK(hipMalloc(&d_o, sizeof(float) * (size_t)n * (size_t)k));
  ORP_CHK(hipMemcpy(d_b, boxes_host, sizeof(float) * 5 * (size_t)n, hipMemcpyHostToDevice));
  ORP_CHK(hipMemcpy(d_q, query_boxes_host, sizeof(float) * 5 * (size_t)k, hipMemcpyHostToDevice));
  rc = orp_poly_overlaps(d_b, n, d_q, k, d_o, nullptr);
  if (rc != ORP_OK) { fprintf(stderr, "_overlaps: launch failed (%d)\n", rc); goto done; }
  ORP_CHK(hipMemcpy(overlaps_host, d_o, sizeof(float) * (size_t)n * (size_t)k, hipMemcpyDeviceToHost));
done:
  if (d_o) (void)hipFree(d_o);
  if (d_b) (void)hipFree(d_b);
  if (d_q) (void)hipFree(d_q);
#undef ORP_CHK
}

#ifndef ORP_BUILD_ID
#define ORP_BUILD_ID "dev"
#endif
const char* orp_version(void) { return "orp_hip gfx950 abi1 " ORP_BUILD_ID; }
}
