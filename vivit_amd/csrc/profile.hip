// Optional in-library kernel timing with HIP events (feeds bench.py's roofline block).
// When enabled, the Gram SYRK launches and every `stride`-th symmetric matrix-vector launch of
// the tridiagonalisation are bracketed by events on the stream they are launched on.
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
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

// One sticky word per (device, stream): solves on one stream are ordered, so the word a persistent kernel writes is taken
// by the info finalisation of ITS solve; a solve on another stream (the hooks' side stream beside the caller's) has a word
// of its own and cannot be blamed for it.  Slot 0 is shared by whatever comes after the 255th stream.
constexpr int PERSIST_WORDS = 256;
__device__ int g_persist_timeout[PERSIST_WORDS] = {0};

int *persist_timeout_word(hipStream_t stream) {
  static std::mutex mu;
  static int *base[64] = {nullptr};
  static std::map<std::pair<int, hipStream_t>, int> slot_of;
  static int next_slot[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  int *&p = base[dev & 63];
  if (!p) {
    void *q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_persist_timeout)) != hipSuccess) return nullptr;
    p = static_cast<int *>(q);
  }
  auto key = std::make_pair(dev, stream);
  auto it = slot_of.find(key);
  if (it == slot_of.end()) {
    int &nx = next_slot[dev & 63];
    const int slot = nx + 1 < PERSIST_WORDS ? ++nx : 0;
    it = slot_of.emplace(key, slot).first;
  }
  return p + it->second;
}

static int g_persist_override = -1;
int persist_override() { return g_persist_override; }
int persist_fault() {
  static int f = -1;
  if (f < 0) {
    const char *e = getenv("VIVIT_PERSIST_FAULT");
    f = e ? atoi(e) & 3 : 0;
  }
  return f;
}

__global__ void take_timeout_kernel(int32_t *info, int *word) {
  const int bits = *word;
  if (bits) {
    *info = VIVIT_INFO_PERSIST_TIMEOUT;
    *word = 0;
  }
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

// 1 / 0: allow / forbid the persistent kernels (sytrd_persist.hip, the panel QR of sy2sb.hip, sb2st.hip's chase) for the
// following calls of this process, whatever VIVIT_*_PERSIST say; -1: back to the environment's choice.  Returns the
// previous setting.  (The host wrapper retries a solve that ended with VIVIT_INFO_PERSIST_TIMEOUT on the launch chains.)
int vivit_persistent_kernels(int on) {
  const int prev = g_persist_override;
  g_persist_override = on < 0 ? -1 : (on != 0);
  return prev;
}

// *info = VIVIT_INFO_PERSIST_TIMEOUT if a persistent kernel ON THIS STREAM gave up since its word was last taken (and clear it); for
// callers of the stage-level entry points (vivit_sytrd_f32, vivit_sy2sb_f32, vivit_sy2sb_panel_qr_f32, vivit_sb2st_f32),
// which have no info word of their own.  The eigensolver entry points do this themselves.
int vivit_take_persist_timeout(int32_t *info, void *stream) {
  if (!info) return VIVIT_E_BADARG;
  int *word = persist_timeout_word(static_cast<hipStream_t>(stream));
  if (!word) return VIVIT_E_LAUNCH;
  take_timeout_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(info, word);
  return launch_status();
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
