// Error reporting and optional HIP-event bracketing of kernel families (used by bench.py's roofline leg).
#include "dx_common.h"
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>

namespace {
thread_local char g_error[512] = "";

struct ProfState {
  bool enabled = false;
  std::vector<hipEvent_t> start, stop;
  size_t used = 0;
};
ProfState g_prof[DX_PROF_NKINDS];
std::mutex g_prof_mutex;   // launches come from the caller's thread AND from autograd engine threads (backward)
}  // namespace

extern "C" {

void dx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

const char* dx_last_error(void) { return g_error; }

int dx_version(void) { return 1; }

// Bracket every launch of kernel family `kind` with HIP events recorded on the launch stream.
// capacity = maximum number of launches recorded between dx_prof_enable and dx_prof_collect.
int dx_prof_enable(int kind, int capacity) {
  DX_REQUIRE(kind >= 0 && kind < DX_PROF_NKINDS && capacity > 0, "dx_prof_enable: bad arguments");
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  ProfState& p = g_prof[kind];
  while ((int)p.start.size() < capacity) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      dx_set_error("dx_prof_enable: hipEventCreate failed");
      return DX_ERR_LAUNCH;
    }
    p.start.push_back(a);
    p.stop.push_back(b);
  }
  p.used = 0;
  p.enabled = true;
  return DX_OK;
}

// Synchronises the recorded events and returns the number of launches and their summed duration (ms).
int dx_prof_collect(int kind, int* launches, double* total_ms) {
  DX_REQUIRE(kind >= 0 && kind < DX_PROF_NKINDS && launches && total_ms, "dx_prof_collect: bad arguments");
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  ProfState& p = g_prof[kind];
  double sum = 0.0;
  for (size_t i = 0; i < p.used; ++i) {
    float ms = 0.f;
    hipEventSynchronize(p.stop[i]);
    if (hipEventElapsedTime(&ms, p.start[i], p.stop[i]) == hipSuccess) sum += ms;
  }
  *launches = (int)p.used;
  *total_ms = sum;
  p.used = 0;
  p.enabled = false;
  return DX_OK;
}

void dx_prof_begin(int kind, hipStream_t s) {
  ProfState& p = g_prof[kind];
  if (!p.enabled) return;                      // the common case takes no lock
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  if (p.enabled && p.used < p.start.size()) hipEventRecord(p.start[p.used], s);
}

void dx_prof_end(int kind, hipStream_t s) {
  ProfState& p = g_prof[kind];
  if (!p.enabled) return;
  std::lock_guard<std::mutex> lock(g_prof_mutex);
  if (p.enabled && p.used < p.start.size()) {
    hipEventRecord(p.stop[p.used], s);
    ++p.used;
  }
}

}  // extern "C"
