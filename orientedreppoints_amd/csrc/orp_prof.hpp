// orp_prof.hpp -- optional per-kernel HIP-event timing inside the library (bench.py's `roofline.achieved`).
// Disabled by default: two relaxed loads per launch.  When enabled, every instrumented launch is bracketed by a
// hipEvent pair recorded ON THE STREAM THE KERNEL IS LAUNCHED ON (no synchronisation); orp_profile_read() later
// synchronises on the recorded events and returns total milliseconds + launch count per slot.
#pragma once
#include <hip/hip_runtime.h>

enum OrpProfSlot {
  ORP_PROF_NMS_MASK = 0, ORP_PROF_NMS_SWEEP = 1, ORP_PROF_NMS_SORT = 2, ORP_PROF_DCN_FWD = 3, ORP_PROF_MINAREARECT = 4,
  ORP_PROF_CONVEX_IOU = 5, ORP_PROF_CONVEX_GIOU = 6, ORP_PROF_IOU_MATRIX = 7, ORP_PROF_DCN_BWD = 8,
  ORP_PROF_DCN_BWD_INPUT = 9, ORP_PROF_DCN_BWD_SCATTER = 10, ORP_PROF_DCN_BWD_WEIGHT = 11,
  ORP_PROF_CONV_SPLIT = 12, ORP_PROF_CONV_WGRAD = 13, ORP_PROF_NSLOTS = 16
};

void orp_prof_begin(int slot, hipStream_t st);
void orp_prof_end(int slot, hipStream_t st);

struct OrpProfScope {
  int slot; hipStream_t st;
  OrpProfScope(int s, hipStream_t t) : slot(s), st(t) { orp_prof_begin(slot, st); }
  ~OrpProfScope() { orp_prof_end(slot, st); }
};
