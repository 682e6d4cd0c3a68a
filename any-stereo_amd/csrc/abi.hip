// Error reporting + version/device queries of the C ABI (include/anystereo_hip.h).
#include <stdarg.h>
#include <stdlib.h>

#include <mutex>
#include <set>
#include <unordered_set>
#include <utility>

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

static int g_precision = -1;
int precision_mode() {
  if (g_precision < 0) {
    const char* e = getenv("ANYSTEREO_PRECISION");
    g_precision = (e && (e[0] == 'f') && e[1] == 'p' && e[2] == '3') ? 0 : 1;  // "fp32" -> 0, default split
  }
  return g_precision;
}

static int g_fast16 = 0;
int fast16_mode() { return g_fast16; }

__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    *reinterpret_cast<float4*>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (long long k = i; k < n; ++k) p[k] = 0.f;
  }
}

int zero_fill(float* p, long long n, hipStream_t s) {
  if (n <= 0) return AS_OK;
  if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) return fail(AS_ERR_BAD_ARG, "zero_fill: buffer not 16-B aligned");
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)cdiv64(cdiv64(n, 4), 256)), dim3(256), 0, s, p, n);
  return check_launch("zero_fill");
}

void lds_opt_in(const void* kernel) {
  // keyed on (device, kernel): the attribute is applied to the CURRENT device's copy of the function, so a process that drives
  // a second GPU must opt that device's copy in as well.  A kernel is marked done only when the runtime accepted the attribute;
  // a refusal is recorded (as_last_error_string) and retried at the next launch, which then fails in check_launch with the cause.
  static std::mutex m;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    dev = -1;
  }
  std::lock_guard<std::mutex> lock(m);
  if (done.count({dev, kernel})) return;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) {
    done.insert({dev, kernel});
  } else {
    (void)hipGetLastError();
    (void)fail(AS_ERR_LAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB) on device %d: %s", dev, hipGetErrorString(e));
  }
}

}  // namespace as

extern "C" {

int as_set_fast16(int on) {
  as::g_fast16 = on ? 1 : 0;
  return AS_OK;
}
int as_get_fast16(void) { return as::g_fast16; }

int as_set_precision(int mode) {
  if (mode != 0 && mode != 1) return as::fail(AS_ERR_BAD_ARG, "set_precision: mode %d (0 = fp32 MFMA, 1 = 3 x fp16 split)", mode);
  as::g_precision = mode;
  return AS_OK;
}
int as_get_precision(void) { return as::precision_mode(); }

const char* as_last_error_string(void) { return as::err_buf(); }

int as_abi_version(void) { return 37; }

int as_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

}  // extern "C"
