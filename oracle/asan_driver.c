/* TEST INFRASTRUCTURE: runs the plain-C corr_sampler oracle (corr_sampler_ref.c) under AddressSanitizer + UBSan on the edge
 * cases the reference's kernel guards (sampler_kernel.cu:45-59,:63-104): windows fully / partly outside the row, W2 = 1,
 * radius larger than the row, negative and huge coordinates.  Exit code 0 = no sanitizer report and the transpose identity
 * <fwd(v), g> == <v, bwd(g)> holds.  GPU sanitizers are not available on the pool; this covers the CPU build. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

void ref_corr_sampler_forward_f32(const float* volume, const float* coords, float* corr, int N, int H1, int W1, int W2, int r);
void ref_corr_sampler_backward_f32(const float* coords, const float* corr_grad, float* volume_grad, int N, int H1, int W1, int W2, int r);

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

static int run(int N, int H1, int W1, int W2, int r, float lo, float hi) {
  const size_t nv = (size_t)N * H1 * W1 * W2, nc = (size_t)N * 2 * H1 * W1, no = (size_t)N * (2 * r + 1) * H1 * W1;
  float *v = malloc(nv * 4), *c = malloc(nc * 4), *o = calloc(no, 4), *g = malloc(no * 4), *vg = calloc(nv, 4); /* outputs zero-filled by the caller, as sampler_cuda_forward/backward do */
  unsigned s = 12345u + (unsigned)(W2 * 131 + r);
  for (size_t i = 0; i < nv; ++i) v[i] = frand(&s);
  for (size_t i = 0; i < nc; ++i) c[i] = lo + (hi - lo) * (frand(&s) + 0.5f);
  for (size_t i = 0; i < no; ++i) g[i] = frand(&s);
  c[0] = 0.0f; if (nc > 1) c[1] = (float)(W2 - 1); if (nc > 2) c[2] = -1e9f; if (nc > 3) c[3] = 1e9f;
  ref_corr_sampler_forward_f32(v, c, o, N, H1, W1, W2, r);
  ref_corr_sampler_backward_f32(c, g, vg, N, H1, W1, W2, r);
  double a = 0, b = 0;
  for (size_t i = 0; i < no; ++i) a += (double)o[i] * g[i];
  for (size_t i = 0; i < nv; ++i) b += (double)v[i] * vg[i];
  const int ok = fabs(a - b) <= 1e-4 * (1.0 + fabs(a));
  if (!ok) fprintf(stderr, "transpose identity failed: N=%d H1=%d W1=%d W2=%d r=%d: %g vs %g\n", N, H1, W1, W2, r, a, b);
  free(v); free(c); free(o); free(g); free(vg);
  return ok;
}

int main(void) {
  int ok = 1;
  ok &= run(2, 3, 17, 11, 4, -5.f, 16.f);
  ok &= run(1, 1, 5, 1, 4, -3.f, 3.f);     /* W2 = 1 */
  ok &= run(1, 2, 9, 3, 7, -10.f, 12.f);   /* radius larger than the row */
  ok &= run(1, 2, 33, 17, 0, 0.f, 16.f);   /* radius 0 */
  ok &= run(3, 1, 1, 64, 4, -1e6f, 1e6f);  /* far outside */
  puts(ok ? "corr_sampler_ref: sanitizer run clean" : "corr_sampler_ref: FAILED");
  return ok ? 0 : 1;
}
