// How fast does v_mfma_f32_32x32x16_bf16 issue chip-wide?  One block of `waves` waves per CU (256 blocks), each wave runs `iters`
// rounds of 27 MFMAs on 9 accumulator tiles (the wgrad consumer's shape), operands in registers (random bf16 data or zeros).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/experiments/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float acc16;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <int NACC>
__global__ __launch_bounds__(512, 2) void k(const u4* __restrict__ in, float* __restrict__ out, int iters, int active_waves) {
  if ((int)(threadIdx.x >> 6) >= active_waves) return;
  const u4 ua = in[threadIdx.x], ub = in[512 + threadIdx.x], uc = in[1024 + threadIdx.x];
  bf8 a = __builtin_bit_cast(bf8, ua), b0 = __builtin_bit_cast(bf8, ub), b1 = __builtin_bit_cast(bf8, uc);
  acc16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, b1, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  const int iters = 4000;
  std::vector<unsigned> h(1536 * 4);
  u4* din;
  float* dout;
  (void)hipMalloc(&din, h.size() * 4);
  (void)hipMalloc(&dout, 256 * 512 * 4 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    for (auto& v : h) v = mode ? ((0x3C00u + (rand() & 0x3FF)) | ((0x3C00u + (rand() & 0x3FF)) << 16)) : 0u;  // bf16 ~ [0.0078, 0.0156) pairs / zeros
    (void)hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 512, 1024}) {
      for (int waves : {4, 8}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(512), 0, 0, din, dout, 10, waves);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(512), 0, 0, din, dout, iters, waves);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double mfma_per_simd = (double)iters * 27 * (blocks / 256.0) * (waves / 4.0);
        printf("%s data, %4d blocks, %d active waves per block: %.3f ms, %.2f ns per MFMA per SIMD (32 cycles at 2.4 GHz = 13.3 ns), %.0f TFLOP/s\n",
               mode ? "random" : "zero", blocks, waves, ms, ms * 1e6 / mfma_per_simd, blocks * (double)waves * iters * 27 * 32768 / (ms * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
