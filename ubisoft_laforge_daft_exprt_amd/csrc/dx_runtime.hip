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


// ---- host-side integer frame durations ----------------------------------------------------------------------------------------
// DaftExprt.get_int_durations (reference model.py:950-973) -> duration_to_integer (extract_features.py:69-125), one utterance per row.
// Plain host code: the reference's arithmetic is Python float (IEEE double) with int() truncation, reproduced operation by operation
// (no fused multiply-add can form: no product feeds a sum).  The O(frames x phones) scan of the reference is replaced by counting the
// frame centres filter_length/2 + hop*i, 0 <= i < nb_frames, inside (begin_sample, end_sample] in closed form.
// status[b]: 0 ok, 1 the phones ran out before the frames did (the reference raises IndexError), 2 the number of integer durations
// differs from the number of non-zero symbols (the reference's index assignment raises).  Rows with status != 0 are left zero
// (the Python wrapper re-runs such a row through its own restatement, which raises the reference's exception type).
#pragma STDC FP_CONTRACT OFF
static long long dx_centres_upto(long long sample, long long first, long long hop, long long nb_frames) {
  if (nb_frames <= 0 || sample < first) return 0;
  const long long n = (sample - first) / hop + 1;
  return n < nb_frames ? n : nb_frames;
}

int dx_int_durations(const float* dur, int B, int L, int sampling_rate, int filter_length, int hop_length, int centered,
                     long long* out, long long* totals, int* status) {
  DX_REQUIRE(dur && out && totals && status && B >= 0 && L > 0 && sampling_rate > 0 && hop_length > 0, "dx_int_durations: bad arguments");
  const double sr = (double)sampling_rate;
  const long long first = (long long)((double)filter_length / 2.0);
  int bad = 0;
  for (int b = 0; b < B; ++b) {
    const float* row = dur + (size_t)b * L;
    long long* o = out + (size_t)b * L;
    for (int s = 0; s < L; ++s) o[s] = 0;
    totals[b] = 0;
    status[b] = 0;
    // nb_samples = int(sum(end - begin) * sr): the spans are [end_prev, end_prev + d] with end_prev accumulated in double
    double end_prev = 0.0, total = 0.0;
    int n_nonzero = 0;
    for (int s = 0; s < L; ++s) {
      const double d = (double)row[s];
      if (d != 0.0) { const double end = end_prev + d; total += end - end_prev; end_prev += d; ++n_nonzero; }
    }
    const long long nb_samples = (long long)(total * sr);
    const long long nb_frames = 1 + (long long)((double)(nb_samples - filter_length) / (double)hop_length);
    long long consumed = 1, n_out = 0, first_idx = -1, last_idx = -1;
    int s = 0;
    end_prev = 0.0;
    bool ran_out = false;
    while (consumed <= nb_frames) {
      while (s < L && (double)row[s] == 0.0) ++s;
      if (s >= L) { ran_out = true; break; }
      const double d = (double)row[s];
      const double begin = end_prev, end = end_prev + d;
      end_prev += d;
      if (begin == end) { ran_out = true; break; }      // the reference raises ValueError; reported as status 1 as well
      const long long bs = (long long)(begin * sr), es = (long long)(end * sr);
      const long long n = dx_centres_upto(es, first, hop_length, nb_frames) - dx_centres_upto(bs, first, hop_length, nb_frames);
      o[s] = n;
      if (first_idx < 0) first_idx = s;
      last_idx = s;
      consumed += n;
      ++n_out;
      ++s;
    }
    if (ran_out || n_out == 0) { status[b] = 1; ++bad; for (int q = 0; q < L; ++q) o[q] = 0; continue; }
    long long head, tail;
    if (centered) { head = tail = (long long)((double)filter_length / 2.0 / (double)hop_length); }
    else { const long long extra = (long long)((double)(filter_length - hop_length) / (double)hop_length); head = extra / 2; tail = extra - head; }
    o[first_idx] += head;
    while (s < L && (double)row[s] == 0.0) ++s;
    if (s < L) { o[s] = tail; ++n_out; ++s; }        // phones left over: the next one receives the trailing edge frames
    else o[last_idx] += tail;
    if (n_out != n_nonzero) { status[b] = 2; ++bad; for (int q = 0; q < L; ++q) o[q] = 0; continue; }
    long long t = 0;
    for (int q = 0; q < L; ++q) t += o[q];
    totals[b] = t;
  }
  (void)bad;
  return 0;                     // per-row conditions are reported in status[], like the reference reports them per utterance
}

}  // extern "C"
