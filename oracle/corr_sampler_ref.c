/* TEST INFRASTRUCTURE (see oracle/__init__.py) — plain-C restatement of the reference's only native
 * code, the `corr_sampler` CUDA extension, loop for loop:
 *   forward  = sampler_forward_kernel   sampler/sampler_kernel.cu:19-60
 *   backward = sampler_backward_kernel  sampler/sampler_kernel.cu:63-104
 * (one iteration of the (n,y,x) loops below = one CUDA thread; outputs are zero-initialised by the
 * caller exactly as sampler_cuda_forward/backward allocate zeros, :122-124,:148).
 * The extension itself cannot be built here (no nvcc; it needs the torch C++ API), so this file plus
 * the pure-PyTorch oracle/ops.py are cross-checked against each other and against the reference's
 * Python lookup (tests/test_oracle_golden.py, tests/test_host_cpu.py).
 * Build: oracle/Makefile -> oracle/_build/libcorr_sampler_ref.so
 */
#include <math.h>
#include <stdint.h>

#define IDX4(a, b, c, d, B_, C_, D_) ((((int64_t)(a) * (B_) + (b)) * (C_) + (c)) * (D_) + (d))

void ref_corr_sampler_forward_f32(const float* volume, const float* coords, float* corr, int N, int H1, int W1, int W2,
                                  int r) {
  const int rd = 2 * r + 1;
  for (int n = 0; n < N; ++n)
    for (int y = 0; y < H1; ++y)
      for (int x = 0; x < W1; ++x) {
        const float x0 = coords[IDX4(n, 0, y, x, 2, H1, W1)];
        const float dx = x0 - floorf(x0);
        for (int i = 0; i < rd + 1; ++i) {
          const int x1 = (int)floorf(x0) - r + i;
          if (x1 >= 0 && x1 < W2) {
            const float s = volume[IDX4(n, y, x, x1, H1, W1, W2)];
            if (i > 0) corr[IDX4(n, i - 1, y, x, rd, H1, W1)] += s * dx;
            if (i < rd) corr[IDX4(n, i, y, x, rd, H1, W1)] += s * (1.0f - dx);
          }
        }
      }
}

void ref_corr_sampler_backward_f32(const float* coords, const float* corr_grad, float* volume_grad, int N, int H1, int W1,
                                   int W2, int r) {
  const int rd = 2 * r + 1;
  for (int n = 0; n < N; ++n)
    for (int y = 0; y < H1; ++y)
      for (int x = 0; x < W1; ++x) {
        const float x0 = coords[IDX4(n, 0, y, x, 2, H1, W1)];
        const float dx = x0 - floorf(x0);
        for (int i = 0; i < rd + 1; ++i) {
          const int x1 = (int)floorf(x0) - r + i;
          if (x1 >= 0 && x1 < W2) {
            float g = 0.0f;
            if (i > 0) g += corr_grad[IDX4(n, i - 1, y, x, rd, H1, W1)] * dx;
            if (i < rd) g += corr_grad[IDX4(n, i, y, x, rd, H1, W1)] * (1.0f - dx);
            volume_grad[IDX4(n, y, x, x1, H1, W1, W2)] += g;
          }
        }
      }
}
