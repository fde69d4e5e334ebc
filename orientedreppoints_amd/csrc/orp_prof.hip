// orp_prof.hip -- see orp_prof.hpp
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>

#include "../../include/orp_hip.h"
#include "orp_prof.hpp"

namespace {
struct Slot { std::vector<hipEvent_t> beg, end; size_t open = 0; };
Slot g_slots[ORP_PROF_NSLOTS];
volatile int g_enabled = 0;
std::mutex g_mu;
constexpr size_t kMaxPairs = 1 << 15;
}  // namespace

void orp_prof_begin(int slot, hipStream_t st) {
  if (!g_enabled || slot < 0 || slot >= ORP_PROF_NSLOTS) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Slot& s = g_slots[slot];
  if (s.beg.size() >= kMaxPairs) return;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  s.beg.push_back(a); s.end.push_back(b);
  s.open = s.beg.size();
  (void)hipEventRecord(a, st);
}
void orp_prof_end(int slot, hipStream_t st) {
  if (!g_enabled || slot < 0 || slot >= ORP_PROF_NSLOTS) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Slot& s = g_slots[slot];
  if (s.open == 0 || s.open != s.beg.size()) return;
  (void)hipEventRecord(s.end[s.open - 1], st);
  s.open = 0;
}

extern "C" {
int orp_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_enabled = on ? 1 : 0;
  return ORP_OK;
}
// total_ms / count of slot since the last reset; synchronises on the slot's last event
int orp_profile_read(int slot, double* total_ms, int* count, int reset) {
  if (slot < 0 || slot >= ORP_PROF_NSLOTS || !total_ms || !count) return ORP_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  Slot& s = g_slots[slot];
  double tot = 0; int n = 0;
  for (size_t i = 0; i < s.beg.size(); i++) {
    if (hipEventSynchronize(s.end[i]) != hipSuccess) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, s.beg[i], s.end[i]) == hipSuccess) { tot += ms; n++; }
  }
  *total_ms = tot; *count = n;
  if (reset) {
    for (size_t i = 0; i < s.beg.size(); i++) { (void)hipEventDestroy(s.beg[i]); (void)hipEventDestroy(s.end[i]); }
    s.beg.clear(); s.end.clear(); s.open = 0;
  }
  return ORP_OK;
}
}
