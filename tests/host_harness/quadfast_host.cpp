// Host build of the DEVICE header orientedreppoints_amd/csrc/orp_quadfast.hpp (test infrastructure only).
// g++ compiles the very same inline functions hipcc compiles for gfx950, with -ffp-contract=off, so the fp32
// operation order of the register fast path can be compared bit for bit with the oracle on the CPU.
#include <stddef.h>
#include <stdint.h>
#include "../../orientedreppoints_amd/csrc/orp_quadfast.hpp"

extern "C" {

// out[n,k] = IoU(a[i], b[j]) through quad_prepare + quad_iou_two_phase (classifier, register fast path, generic
// fallback); stats[0] += pairs resolved by the classifier, stats[1] += pairs on the generic path
void host_quadfast_matrix(const float* a, int n, const float* b, int k, int guard, float* out, int64_t* stats) {
  orp::QuadPrep* ra = new orp::QuadPrep[n > 0 ? n : 1];
  long long st[2] = {0, 0};
  for (int i = 0; i < n; i++) orp::quad_prepare(a + 8 * (size_t)i, ra[i]);
  for (int j = 0; j < k; j++) {
    orp::QuadPrep pc;
    orp::quad_prepare(b + 8 * (size_t)j, pc);
    for (int i = 0; i < n; i++)
      out[(size_t)i * k + j] = guard ? orp::quad_iou_two_phase<true>(&ra[i], &pc, st)
                                     : orp::quad_iou_two_phase<false>(&ra[i], &pc, st);
  }
  if (stats) { stats[0] += st[0]; stats[1] += st[1]; }
  delete[] ra;
}

// fp64 instantiation (DOTA_devkit/polyiou.cpp arithmetic: the merge NMS): out[n,k] = IoU(a[i], b[j]) in double
void host_quadfast_matrix_f64(const double* a, int n, const double* b, int k, double* out, int64_t* stats) {
  orp::QuadPrepT<double>* ra = new orp::QuadPrepT<double>[n > 0 ? n : 1];
  long long st[2] = {0, 0};
  for (int i = 0; i < n; i++) orp::quad_prepare<double>(a + 8 * (size_t)i, ra[i]);
  for (int j = 0; j < k; j++) {
    orp::QuadPrepT<double> pc;
    orp::quad_prepare<double>(b + 8 * (size_t)j, pc);
    for (int i = 0; i < n; i++) out[(size_t)i * k + j] = orp::quad_iou_two_phase_t<double, false>(&ra[i], &pc, st);
  }
  if (stats) { stats[0] += st[0]; stats[1] += st[1]; }
  delete[] ra;
}

// the generic path of the same header (quad_iou), for a same-compiler cross-check
void host_quadgeneric_matrix(const float* a, int n, const float* b, int k, int guard, float* out) {
  orp::PolyPriv<float, orp::ORP_CLIP_CAP> P, Q;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < k; j++)
      out[(size_t)i * k + j] = guard ? orp::quad_iou<float, true>(P, Q, a + 8 * (size_t)i, b + 8 * (size_t)j)
                                     : orp::quad_iou<float, false>(P, Q, a + 8 * (size_t)i, b + 8 * (size_t)j);
}

}  // extern "C"
