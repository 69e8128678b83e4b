// Optional in-library kernel timing with HIP events (feeds bench.py's roofline block).
// When enabled, the Gram SYRK launches and every `stride`-th symmetric matrix-vector launch of
// the tridiagonalisation are bracketed by events on the stream they are launched on.
#include <vector>

#include "common.h"

namespace vivit {

struct ProfSample {
  hipEvent_t start, stop;
  int kind;
  double work;
};
static bool g_prof_on = false;
static int g_prof_stride = 64;
static std::vector<ProfSample> g_samples;
static ProfSample g_open[2];
static bool g_open_valid[2] = {false, false};

// Stage marks of the eigensolver: one event per stage boundary; the time between two consecutive marks of one
// symeig call is attributed to the stage named by the LATER mark (PROF_STAGE_BEGIN opens a call).
struct ProfMark {
  hipEvent_t ev;
  int stage;
};
static std::vector<ProfMark> g_marks;

void prof_mark(int stage, hipStream_t stream) {
  if (!g_prof_on) return;
  ProfMark m;
  m.stage = stage;
  if (hipEventCreate(&m.ev) != hipSuccess) return;
  (void)hipEventRecord(m.ev, stream);
  g_marks.push_back(m);
}

bool prof_enabled() { return g_prof_on; }
int prof_stride() { return g_prof_stride; }

void prof_begin(int kind, double work, hipStream_t stream) {
  if (!g_prof_on) return;
  ProfSample s;
  s.kind = kind;
  s.work = work;
  if (hipEventCreate(&s.start) != hipSuccess || hipEventCreate(&s.stop) != hipSuccess) return;
  (void)hipEventRecord(s.start, stream);
  g_open[kind] = s;
  g_open_valid[kind] = true;
}

void prof_end(int kind, hipStream_t stream) {
  if (!g_prof_on || !g_open_valid[kind]) return;
  (void)hipEventRecord(g_open[kind].stop, stream);
  g_samples.push_back(g_open[kind]);
  g_open_valid[kind] = false;
}

} // namespace vivit

using namespace vivit;

extern "C" {

int vivit_profile_begin(int symv_stride) {
  g_samples.clear();
  for (auto &m : g_marks) (void)hipEventDestroy(m.ev);
  g_marks.clear();
  g_prof_stride = symv_stride > 0 ? symv_stride : 1;
  g_prof_on = true;
  return VIVIT_OK;
}

int vivit_profile_end(double *out) {
  g_prof_on = false;
  double acc[2][3] = {{0, 0, 0}, {0, 0, 0}};
  for (auto &s : g_samples) {
    float ms = 0.f;
    if (hipEventSynchronize(s.stop) == hipSuccess && hipEventElapsedTime(&ms, s.start, s.stop) == hipSuccess) {
      acc[s.kind][0] += 1.0;
      acc[s.kind][1] += ms;
      acc[s.kind][2] += s.work;
    }
    (void)hipEventDestroy(s.start);
    (void)hipEventDestroy(s.stop);
  }
  g_samples.clear();
  if (out)
    for (int k = 0; k < 2; ++k)
      for (int c = 0; c < 3; ++c) out[3 * k + c] = acc[k][c];
  return VIVIT_OK;
}

int vivit_profile_stages(double *out_ms, int num) {
  if (!out_ms || num < 0) return VIVIT_E_BADARG;
  for (int i = 0; i < num; ++i) out_ms[i] = 0.0;
  for (size_t i = 1; i < g_marks.size(); ++i) {
    const int stage = g_marks[i].stage;
    if (stage == PROF_STAGE_BEGIN || stage < 0 || stage >= num) continue;
    float ms = 0.f;
    if (hipEventSynchronize(g_marks[i].ev) == hipSuccess &&
        hipEventElapsedTime(&ms, g_marks[i - 1].ev, g_marks[i].ev) == hipSuccess)
      out_ms[stage] += ms;
  }
  for (auto &m : g_marks) (void)hipEventDestroy(m.ev);
  g_marks.clear();
  return VIVIT_OK;
}

} // extern "C"
