// Library-level entry points of libtsg_hip.so: version and the thread-local error channel.
#include "tsg_common.h"

#include <atomic>

namespace tsg {
namespace {
thread_local char g_err[512] = "";
std::atomic<unsigned*> g_error_sink{nullptr};
}

unsigned* error_sink() { return g_error_sink.load(std::memory_order_relaxed); }

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
extern "C" int tsg_version(void) { return TSG_VERSION; }
extern "C" const char* tsg_last_error(void) { return tsg::g_err; }
