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
// Reduced-precision variant of the split mode (as_set_fast16): plain fp16 operands (the hi parts only), fp32 accumulate —
// ONE MFMA per product instead of three; the counterpart of the reference's autocast path (continuous_IGEVstereo.py:287).
int fast16_mode();

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Split-precision range safety.  MODE.FP16_OVFL (hwreg MODE, bit 23): an overflowed fp16 VALU result — here: the
// v_cvt_f16_f32 of the operand split x = hi + lo/2048 — is clamped to +-65504 instead of becoming inf, so |x| >= 65504
// yields a SATURATED finite operand instead of hi = inf, lo = NaN.  Zero cost per element; set once at kernel entry by
// every kernel that splits.  In-range values convert exactly as before (bit-identical results).
__device__ __forceinline__ void fp16_saturate_mode() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }
// a thread's running max |x| over the values it split -> one atomic per wave that saw an out-of-range (or NaN) operand
__device__ __forceinline__ void note_split_overflow(float amax, unsigned* counter) {
  if (__builtin_amdgcn_ballot_w64(!(amax < 65504.f)) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(counter, 1u);
}
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Dynamic LDS beyond 64 KB needs an opt-in per kernel (hipFuncAttributeMaxDynamicSharedMemorySize).  The attribute belongs to
// the FUNCTION, process-wide — so it is raised ONCE per kernel to the CU's whole 160 KB and never written again: a value that
// follows the launch at hand (the weight-gradient kernel's LDS plan depends on the layer shape) is read by every later launch
// of that kernel, including the kernel nodes of a captured hipGraph replayed after an eager launch with a smaller plan.
void lds_opt_in(const void* kernel);

// Zero-fill as an ordinary KERNEL launch.  hipMemsetAsync becomes a memset NODE under hipGraph capture, and in the captured
// training step (thousands of nodes) replays showed zero-filled buffers cleared late — after a later kernel had written them
// (tools/train_graph_check.py); a kernel node is ordered like every other launch of the stream.
int zero_fill(float* p, long long n, hipStream_t s);

}  // namespace as

#define AS_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) return as::fail(code, __VA_ARGS__); \
  } while (0)
