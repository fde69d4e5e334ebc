// orp_soft_rnms.hip -- soft rotated NMS on the HOST (no device code in this file).
//
// Replaces rnms_cpu.soft_rnms (mmdet/ops/nms/src/rnms_cpu.cpp:165-333), which is CPU-only in the reference as well
// (mmdet/ops/nms/nms_wrapper.py:120-175 "Dispatch to only CPU Soft NMS implementations").  The rotated IoU is the same
// fp32 triangle-fan arithmetic as the device kernels: this file instantiates orp::quad_iou from orp_geom.hpp for the
// host (the header is plain C++), compiled with -ffp-contract=off like everything that feeds a `>` decision.
// Algorithm (reference :196-310): selection sort by score with in-place swaps; every later box is re-weighted by
//   method 0: 0 if iou > thr else 1;  1 (linear): 1 - iou if iou > thr;  2 (gaussian): exp(-iou^2 / sigma)
// and boxes whose score falls below min_score are swapped with the last live box and dropped.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/orp_hip.h"
#include "orp_geom.hpp"

extern "C" int orp_soft_rnms_host(const float* dets, int m, float iou_thr, int method, float sigma, float min_score,
                                  float* out, int* num_out) {
  if (m < 0 || !num_out || (m > 0 && (!dets || !out)) || method < 0 || method > 2) return ORP_EINVAL;
  *num_out = 0;
  if (m == 0) return ORP_OK;
  // rows: 8 coordinates, score, original index (kept as float, as the reference's arange(ndets, dets.options()))
  std::vector<float> box((size_t)m * 10);
  for (int i = 0; i < m; i++) {
    memcpy(&box[(size_t)i * 10], dets + (size_t)i * 9, sizeof(float) * 9);
    box[(size_t)i * 10 + 9] = (float)i;
  }
  orp::PolyPriv<float, orp::ORP_CLIP_CAP> P, Q;
  int ndets = m;
  float tmp[10];
  for (int i = 0; i < ndets; i++) {
    float max_score = box[(size_t)i * 10 + 8];
    int max_pos = i;
    for (int pos = i + 1; pos < ndets; pos++) {
      if (max_score < box[(size_t)pos * 10 + 8]) { max_score = box[(size_t)pos * 10 + 8]; max_pos = pos; }
    }
    memcpy(tmp, &box[(size_t)i * 10], sizeof(tmp));
    memcpy(&box[(size_t)i * 10], &box[(size_t)max_pos * 10], sizeof(tmp));
    memcpy(&box[(size_t)max_pos * 10], tmp, sizeof(tmp));
    float cur[8];
    memcpy(cur, &box[(size_t)i * 10], sizeof(cur));
    int pos = i + 1;
    while (pos < ndets) {
      float* o = &box[(size_t)pos * 10];
      const float ovr = orp::quad_iou<float, false>(P, Q, cur, o);
      float weight = 1.f;
      if (method == 1) {
        if (ovr > iou_thr) weight = 1 - ovr;
      } else if (method == 2) {
        weight = expf(-(ovr * ovr) / sigma);
      } else {
        weight = (ovr > iou_thr) ? 0.f : 1.f;
      }
      o[8] = weight * o[8];
      if (o[8] < min_score) {
        memcpy(o, &box[(size_t)(ndets - 1) * 10], sizeof(float) * 10);
        ndets--;
        pos--;
      }
      pos++;
    }
  }
  memcpy(out, box.data(), sizeof(float) * 10 * (size_t)ndets);
  *num_out = ndets;
  return ORP_OK;
}
