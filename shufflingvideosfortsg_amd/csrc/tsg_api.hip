// Library-level entry points of libtsg_hip.so: version and the thread-local error channel.
#include "tsg_common.h"

namespace tsg {
namespace {
thread_local char g_err[512] = "";
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

extern "C" int tsg_version(void) { return TSG_VERSION; }
extern "C" const char* tsg_last_error(void) { return tsg::g_err; }
