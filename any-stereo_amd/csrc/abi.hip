// Error reporting + version/device queries of the C ABI (include/anystereo_hip.h).
#include <stdarg.h>

#include "common.h"

namespace as {

char* err_buf() {
  static thread_local char buf[512] = "ok";
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace as

extern "C" {

const char* as_last_error_string(void) { return as::err_buf(); }

int as_abi_version(void) { return 1; }

int as_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

}  // extern "C"
