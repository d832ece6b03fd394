// Library-level entry points of libtsg_hip.so: version and the thread-local error channel.
#include "tsg_common.h"

#include <atomic>
#include <map>
#include <mutex>
#include <utility>

namespace tsg {
namespace {
thread_local char g_err[512] = "";
std::atomic<unsigned*> g_error_sink{nullptr};
std::atomic<unsigned*> g_error_word[64];       // one device word PER DEVICE (ABI revision 5; ADVICE r3: with one global word a launch on
                                                // device 0 reported into device 1's memory after a second device had registered)
constexpr int kTimeSlots = 1024;
struct EventPair { hipEvent_t start = nullptr, stop = nullptr; bool used = false; };
EventPair g_time_slots[kTimeSlots];               // (created on first use; one measuring thread per process)
thread_local int g_armed_slot = -1;
int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  return dev;
}
}

ErrSink error_sink() {
  return ErrSink{g_error_sink.load(std::memory_order_relaxed), g_error_word[current_device()].load(std::memory_order_relaxed)};
}

bool take_launch_events(hipEvent_t* start, hipEvent_t* stop) {
  const int slot = g_armed_slot;
  if (slot < 0) return false;
  g_armed_slot = -1;
  EventPair& p = g_time_slots[slot];
  if (!p.start && (hipEventCreate(&p.start) != hipSuccess || hipEventCreate(&p.stop) != hipSuccess)) return false;
  p.used = true;
  *start = p.start; *stop = p.stop;
  return true;
}

hipError_t ensure_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> told;      // (device, kernel) -> largest size set so far
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = told[{dev, kernel}];
  if (bytes <= have) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
  if (e == hipSuccess) have = bytes;
  return e;
}

int device_cu_count() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cus[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  return set_error(static_cast<int>(e), "%s: launch failed: %s", what, hipGetErrorString(e));
}
}  // namespace tsg

// One process-wide error sink for every kernel with a bounded wait (persistent LSTM hand-offs, the K1 backward's
// cross-workgroup exchange): a host-readable word set to 1 when a wait expired.  tsg_lstm_error_sink is the original name.
extern "C" int tsg_error_sink(void* p) { tsg::g_error_sink.store(static_cast<unsigned*>(p), std::memory_order_relaxed); return 0; }
extern "C" int tsg_lstm_error_sink(void* p) { return tsg_error_sink(p); }
// The same report into DEVICE memory (a 4-byte word the caller owns and clears): what a device-side guard -- the optimizer's
// found_inf input -- can read without the host, e.g. between the two graphs of a replayed train step.
// Registered for the CURRENT device (hipGetDevice at the time of the call); launches on a device report into that device's word.
extern "C" int tsg_error_word(void* p) {
  tsg::g_error_word[tsg::current_device()].store(static_cast<unsigned*>(p), std::memory_order_relaxed);
  return 0;
}
extern "C" int tsg_version(void) { return TSG_VERSION; }
extern "C" const char* tsg_last_error(void) { return tsg::g_err; }
extern "C" int tsg_time_next_launch(int slot) {
  if (slot == -1) { tsg::g_armed_slot = -1; return 0; }      // disarm
  if (slot < 0 || slot >= tsg::kTimeSlots) return tsg::set_error(TSG_E_SHAPE, "tsg_time_next_launch: slot %d outside [0, %d)", slot, tsg::kTimeSlots);
  tsg::g_armed_slot = slot;
  tsg::g_time_slots[slot].used = false;
  return 0;
}
extern "C" int tsg_timed_launch_us(int slot, float* us) {
  if (!us) return tsg::set_error(TSG_E_NULL, "tsg_timed_launch_us: us is NULL");
  if (slot < 0 || slot >= tsg::kTimeSlots || !tsg::g_time_slots[slot].used)
    return tsg::set_error(TSG_E_SHAPE, "tsg_timed_launch_us: slot %d was not used by a launch", slot);
  hipError_t e = hipEventSynchronize(tsg::g_time_slots[slot].stop);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, tsg::g_time_slots[slot].start, tsg::g_time_slots[slot].stop);
  if (e != hipSuccess) return tsg::set_error((int)e, "tsg_timed_launch_us: %s", hipGetErrorString(e));
  *us = ms * 1e3f;
  return 0;
}
