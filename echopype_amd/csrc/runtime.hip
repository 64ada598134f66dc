// Runtime plumbing of the C ABI: error reporting, device memory helpers, HIP-event timers.
#include "epa_internal.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <set>
#include <string>

namespace epa {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
std::atomic<bool> g_trace_on{false};
static thread_local char g_trace[2048] = "";
static thread_local size_t g_trace_len = 0;
void note_launch(const char* what) {
  const size_t n = strlen(what);
  if (g_trace_len + n + 2 > sizeof(g_trace)) return;  // full: the first launches are the ones kept
  memcpy(g_trace + g_trace_len, what, n);
  g_trace_len += n;
  g_trace[g_trace_len++] = ';';
  g_trace[g_trace_len] = 0;
}
// distinct names seen, process-wide.  The caller passes string literals: the common case (seen before, same pointer) is a
// lock-free compare against a small per-thread cache of pointers.
static std::mutex g_seen_mu;
static std::set<std::string> g_seen;
static std::string g_seen_joined;
void note_seen(const char* what) {
  // (larger than the number of distinct launch names in the library -- about 70 -- so that every name ends up cached and
  //  the mutex below is taken once per name and thread, never on the steady-state launch path)
  constexpr int kCache = 256;
  static thread_local const char* cache[kCache];
  static thread_local int ncache = 0;
  for (int i = 0; i < ncache; ++i)
    if (cache[i] == what) return;
  {
    std::lock_guard<std::mutex> lk(g_seen_mu);
    g_seen.insert(what);
  }
  if (ncache < kCache) cache[ncache++] = what;
}
static thread_local int g_stats_filled = 0;
void note_range_stats_filled(int filled) { g_stats_filled = filled; }
}  // namespace epa

struct EpaTimer {
  hipEvent_t start, stop;
};

extern "C" {

int epa_version(void) { return EPA_VERSION; }
#ifndef EPA_SOURCE_DIGEST
#define EPA_SOURCE_DIGEST "unknown"
#endif
const char* epa_source_digest(void) { return EPA_SOURCE_DIGEST; }
const char* epa_last_error(void) { return epa::g_err; }
int epa_last_range_stats_filled(void) { return epa::g_stats_filled; }
const char* epa_launch_seen(void) {
  std::lock_guard<std::mutex> lk(epa::g_seen_mu);
  epa::g_seen_joined.clear();
  for (const auto& k : epa::g_seen) epa::g_seen_joined += k + ";";
  return epa::g_seen_joined.c_str();
}
const char* epa_launch_trace(int mode) {
  if (mode == 1 || mode == 0) {  // start afresh / stop
    epa::g_trace_on.store(mode == 1, std::memory_order_relaxed);
    epa::g_trace_len = 0;
    epa::g_trace[0] = 0;
  }
  return epa::g_trace;
}

int epa_device_count(int* n) {
  EPA_CHECK_ARG(n != nullptr, "epa_device_count: n is NULL");
  EPA_CHECK_HIP(hipGetDeviceCount(n));
  return EPA_OK;
}
int epa_set_device(int dev) {
  EPA_CHECK_HIP(hipSetDevice(dev));
  return EPA_OK;
}
int epa_device_name(int dev, char* buf, size_t len) {
  EPA_CHECK_ARG(buf != nullptr && len > 0, "epa_device_name: empty buffer");
  hipDeviceProp_t prop;
  EPA_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return EPA_OK;
}
int epa_malloc(void** ptr, size_t bytes) {
  EPA_CHECK_ARG(ptr != nullptr, "epa_malloc: ptr is NULL");
  hipError_t e = hipMalloc(ptr, bytes);
  if (e == hipErrorOutOfMemory) {
    epa::set_error("epa_malloc: out of device memory (%zu bytes)", bytes);
    return EPA_ENOMEM;
  }
  EPA_CHECK_HIP(e);
  return EPA_OK;
}
int epa_free(void* ptr) {
  EPA_CHECK_HIP(hipFree(ptr));
  return EPA_OK;
}
int epa_memset(void* ptr, int value, size_t bytes, epa_stream_t stream) {
  EPA_CHECK_HIP(hipMemsetAsync(ptr, value, bytes, (hipStream_t)stream));
  return EPA_OK;
}
int epa_memcpy_h2d(void* dst, const void* src_host, size_t bytes, epa_stream_t stream) {
  EPA_CHECK_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return EPA_OK;
}
int epa_memcpy_d2h(void* dst_host, const void* src, size_t bytes, epa_stream_t stream) {
  EPA_CHECK_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return EPA_OK;
}
int epa_stream_synchronize(epa_stream_t stream) {
  EPA_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return EPA_OK;
}

int epa_timer_create(void** timer) {
  EPA_CHECK_ARG(timer != nullptr, "epa_timer_create: timer is NULL");
  EpaTimer* t = new EpaTimer;
  hipError_t e = hipEventCreate(&t->start);
  if (e == hipSuccess) e = hipEventCreate(&t->stop);
  if (e != hipSuccess) {
    delete t;
    EPA_CHECK_HIP(e);
  }
  *timer = t;
  return EPA_OK;
}
int epa_timer_destroy(void* timer) {
  if (!timer) return EPA_OK;
  EpaTimer* t = (EpaTimer*)timer;
  (void)hipEventDestroy(t->start);
  (void)hipEventDestroy(t->stop);
  delete t;
  return EPA_OK;
}
int epa_timer_start(void* timer, epa_stream_t stream) {
  EPA_CHECK_ARG(timer != nullptr, "epa_timer_start: timer is NULL");
  EPA_CHECK_HIP(hipEventRecord(((EpaTimer*)timer)->start, (hipStream_t)stream));
  return EPA_OK;
}
int epa_timer_stop(void* timer, epa_stream_t stream) {
  EPA_CHECK_ARG(timer != nullptr, "epa_timer_stop: timer is NULL");
  EPA_CHECK_HIP(hipEventRecord(((EpaTimer*)timer)->stop, (hipStream_t)stream));
  return EPA_OK;
}
int epa_timer_elapsed_ms(void* timer, float* ms) {
  EPA_CHECK_ARG(timer != nullptr && ms != nullptr, "epa_timer_elapsed_ms: NULL argument");
  EpaTimer* t = (EpaTimer*)timer;
  EPA_CHECK_HIP(hipEventSynchronize(t->stop));
  EPA_CHECK_HIP(hipEventElapsedTime(ms, t->start, t->stop));
  return EPA_OK;
}

}  // extern "C"
