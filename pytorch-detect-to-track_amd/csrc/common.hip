// Error reporting + ABI version for libdtt_hip.so.
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include "common.h"

static thread_local char g_err[512] = "";

void dtt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dtt_last_error(void) { return g_err; }
extern "C" int dtt_abi_version(void) { return 1; }

// ---- per-kernel timing hook --------------------------------------------------------------------------
namespace {
struct ProfState {
  char tag[64];
  hipEvent_t* begin;
  hipEvent_t* end;
  int capacity;
  int used;
  bool open;
} g_prof = {"", nullptr, nullptr, 0, 0, false};
std::mutex g_prof_mu;   // the hook is process-wide state: attach / count / the launchers' begin-end pairs are serialised
}  // namespace

// begin_events / end_events: arrays of `n` hipEvent_t handles owned by the caller.  Launch i of the kernel
// named `tag` records begin_events[i] / end_events[i] around itself, until n launches have been recorded.
// Pass tag = NULL (or n = 0) to detach.  Returns the number of launches recorded so far for the previous tag.
extern "C" int dtt_profile_attach(const char* tag, void** begin_events, void** end_events, int n) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  const int prev = g_prof.used;
  if (!tag || n <= 0) {
    g_prof.tag[0] = 0; g_prof.begin = g_prof.end = nullptr; g_prof.capacity = g_prof.used = 0; g_prof.open = false;
    return prev;
  }
  strncpy(g_prof.tag, tag, sizeof(g_prof.tag) - 1);
  g_prof.tag[sizeof(g_prof.tag) - 1] = 0;
  g_prof.begin = reinterpret_cast<hipEvent_t*>(begin_events);
  g_prof.end = reinterpret_cast<hipEvent_t*>(end_events);
  g_prof.capacity = n;
  g_prof.used = 0;
  g_prof.open = false;
  return prev;
}
extern "C" int dtt_profile_count(void) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  return g_prof.used;
}

void dtt_prof_begin(const char* tag, hipStream_t stream) {
  if (g_prof.capacity == 0) return;   // nothing attached: no lock on the hot path
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (g_prof.capacity == 0 || g_prof.used >= g_prof.capacity || strcmp(tag, g_prof.tag) != 0) return;
  (void)hipEventRecord(g_prof.begin[g_prof.used], stream);
  g_prof.open = true;
}
void dtt_prof_end(const char* tag, hipStream_t stream) {
  if (g_prof.capacity == 0) return;
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (!g_prof.open || strcmp(tag, g_prof.tag) != 0) return;
  (void)hipEventRecord(g_prof.end[g_prof.used], stream);
  g_prof.used++;
  g_prof.open = false;
}

int dtt_device_cus() {
  static int cached[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& c = cached[dev & 63];
  if (c == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    c = v;
  }
  return c;
}
