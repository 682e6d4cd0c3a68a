// Shared helpers for the gfx950 kernels of libanystereo_hip.so (MI355X only: wave64, no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/anystereo_hip.h"

namespace as {

constexpr int kWave = 64;

// Thread-local description of the last failure (as_last_error_string()).
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AS_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return AS_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Matrix-core arithmetic mode of the GEMM-shaped kernels (as_set_precision): 0 = exact fp32 MFMA,
// 1 = split-precision 3 x fp16 MFMA (~2^-22 relative per product, needs |x| < 65504).
int precision_mode();
inline bool use_split_precision() { return precision_mode() == 1; }

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace as

#define AS_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) return as::fail(code, __VA_ARGS__); \
  } while (0)
