// a12-a14, a16-a17: the LIIF-style continuous upsampler's non-GEMM stages (the MLP itself runs on
// the fp32-MFMA 1x1 path of conv.hip over the channel-major latent this file produces).
// All HBM/L2-bound gathers and stencils; lanes = consecutive pixels / consecutive queries so every
// store is a coalesced wave store.
#include "common.h"

namespace {

// ---- structure feature: out = cat(x, affinity(x)) -------------------------------------------------
// pass 1: copy x into out[:, :C] and write the clamped L2 norm over channels to ws[b,y,x]
__global__ __launch_bounds__(256) void sf_norm_copy_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           float* __restrict__ ws, int C, long long plane, long long P) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= P) return;
  const long long b = pix / plane, rem = pix - b * plane;
  const float* xp = x + b * C * plane + rem;
  float* op = out + b * (C + 8) * plane + rem;
  float ss = 0.f;
  int c = 0;
  for (; c + 8 <= C; c += 8) {  // 8 independent loads in flight per lane
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = xp[(long long)(c + i) * plane];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      op[(long long)(c + i) * plane] = v[i];
      ss += v[i] * v[i];
    }
  }
  for (; c < C; ++c) {
    const float v = xp[(long long)c * plane];
    op[(long long)c * plane] = v;
    ss += v * v;
  }
  ws[pix] = fmaxf(sqrtf(ss), 1e-12f);  // F.normalize: x / max(||x||_2, eps)
}

// pass 2: aff[j] = max(0, sum_c xhat[c,p] * xhat[c,p+off_j]), 8 neighbours row-major, zero outside
__global__ __launch_bounds__(256) void sf_affinity_kernel(const float* __restrict__ x, const float* __restrict__ ws,
                                                          float* __restrict__ out, int C, int H, int W, long long P) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= P) return;
  const long long plane = (long long)H * W;
  const long long b = pix / plane;
  const int rem = (int)(pix - b * plane);
  const int y = rem / W, xx = rem - y * W;
  const float* xp = x + b * C * plane;
  const float* np = ws + b * plane;
  int noff[8];
  float nn[8];
  bool ok[8];
  {
    int j = 0;
#pragma unroll
    for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
      for (int ox = -1; ox <= 1; ++ox) {
        if (oy == 0 && ox == 0) continue;
        const int yy = y + oy, x2 = xx + ox;
        ok[j] = yy >= 0 && yy < H && x2 >= 0 && x2 < W;
        noff[j] = ok[j] ? yy * W + x2 : rem;
        nn[j] = np[noff[j]];
        ++j;
      }
  }
  const float n0 = np[rem];
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  // sum_c (x_c(p)/n0) (x_c(q)/nq) = (sum_c x_c(p) x_c(q)) / (n0 nq): the 9 divisions per channel of the literal form
  // (liif.py:439-441) leave the loop; the difference is one rounding of the final quotient
#pragma unroll 2
  for (int c = 0; c < C; ++c) {
    const float* xc = xp + (long long)c * plane;
    const float fc = xc[rem];
    float nb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) nb[j] = xc[noff[j]];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(fc, nb[j], acc[j]);
  }
  float* op = out + (b * (C + 8) + C) * plane + rem;
#pragma unroll
  for (int j = 0; j < 8; ++j) op[(long long)j * plane] = ok[j] ? fmaxf(acc[j] / (n0 * nn[j]), 0.f) : 0.f;
}

// grid_sample(mode='nearest', align_corners=False) source index, evaluated with the same fp32
// operation sequence as ATen (no fma contraction): nearbyint(((c + 1) * n - 1) / 2)
__device__ __forceinline__ int nearest_idx(float c, int n) {
  const float u = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(c, 1.f), (float)n), 1.f), 2.f);
  return (int)rintf(u);
}

struct GatherParams {
  const float* feat;
  const float* coord;
  float* latent;
  int B, C, H, W, Q, lat_ctot, lat_coff;
  float lo, hi;               // clamp bounds (float)(-1+1e-6), (float)(1-1e-6)
  float c0y, sy, c0x, sx;     // make_coord: centre(i) = c0 + s*i, c0 = (float)(-1+1/n), s = (float)(2/n)
};

__global__ __launch_bounds__(256) void liif_gather_kernel(GatherParams p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)p.B * p.Q) return;
  const int b = (int)(t / p.Q);
  const int q = (int)(t - (long long)b * p.Q);
  const float cr = p.coord[t * 2 + 0], cc = p.coord[t * 2 + 1];
  const int iy = nearest_idx(fminf(fmaxf(cr, p.lo), p.hi), p.H);
  const int ix = nearest_idx(fminf(fmaxf(cc, p.lo), p.hi), p.W);
  const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;  // always true after the clamp
  const long long plane = (long long)p.H * p.W;
  const float* fp = p.feat + (long long)b * p.C * plane + (ok ? (long long)iy * p.W + ix : 0);
  float* lp = p.latent + ((long long)b * p.lat_ctot + p.lat_coff) * p.Q + q;
  for (int c = 0; c < p.C; ++c) lp[(long long)c * p.Q] = ok ? fp[(long long)c * plane] : 0.f;
  // rel = (coord_unclamped - cell_centre) * (H, W)   (liif.py:127-129)
  const float qy = ok ? __fadd_rn(p.c0y, __fmul_rn(p.sy, (float)iy)) : 0.f;
  const float qx = ok ? __fadd_rn(p.c0x, __fmul_rn(p.sx, (float)ix)) : 0.f;
  lp[(long long)p.C * p.Q] = __fmul_rn(__fsub_rn(cr, qy), (float)p.H);
  lp[(long long)(p.C + 1) * p.Q] = __fmul_rn(__fsub_rn(cc, qx), (float)p.W);
}

// ---- a14 + first layer of a15 fused -----------------------------------------------------------
// The first MLP layer is linear in the latent [q_feat0 | rel0 | q_feat1 | rel1] and q_feat_i is a nearest GATHER
// of a low-resolution map, so  W1·latent(q) = (W1a·sf0)[n0(q)] + (W1b·sf1)[n1(q)] + Wrel·rel(q):  the two big
// products are evaluated once per LOW-resolution pixel (u0 = W1a·sf0, u1 = W1b·sf1, plain 1x1 convs: 2.8 GFLOP
// instead of 30 GFLOP at 960x540) and this kernel only gathers, adds the 4-term relative-coordinate product and the
// bias, applies ReLU and writes the hidden layer [B,C,Q] — the 228-channel latent is never materialised.
// Same function as liif.py:108-137 + the first Linear/ReLU of liif.py:9-25; the summation order differs.
struct Mlp1Params {
  const float* u[2];
  const float* coord;
  const float* wrel;  // [C][2*n_src]: columns (rel_row, rel_col) per source
  const float* bias;  // [C] or null
  float* out;         // [B,C,Q]
  int B, C, Q, n_src;
  int H[2], W[2];
  float lo, hi;
  float c0y[2], sy[2], c0x[2], sx[2];
};

__global__ __launch_bounds__(256) void liif_mlp1_gather_kernel(Mlp1Params p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)p.B * p.Q) return;
  const int b = (int)(t / p.Q);
  const int q = (int)(t - (long long)b * p.Q);
  const float cr = p.coord[t * 2 + 0], cc = p.coord[t * 2 + 1];
  const float crc = fminf(fmaxf(cr, p.lo), p.hi), ccc = fminf(fmaxf(cc, p.lo), p.hi);
  const float* up[2];
  long long plane[2];
  float rel[4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (s < p.n_src) {
      const int iy = nearest_idx(crc, p.H[s]), ix = nearest_idx(ccc, p.W[s]);
      plane[s] = (long long)p.H[s] * p.W[s];
      up[s] = p.u[s] + (long long)b * p.C * plane[s] + (long long)iy * p.W[s] + ix;
      const float qy = __fadd_rn(p.c0y[s], __fmul_rn(p.sy[s], (float)iy));
      const float qx = __fadd_rn(p.c0x[s], __fmul_rn(p.sx[s], (float)ix));
      rel[2 * s] = __fmul_rn(__fsub_rn(cr, qy), (float)p.H[s]);
      rel[2 * s + 1] = __fmul_rn(__fsub_rn(cc, qx), (float)p.W[s]);
    } else {
      up[s] = p.u[0];
      plane[s] = 0;
      rel[2 * s] = rel[2 * s + 1] = 0.f;
    }
  }
  float* op = p.out + (long long)b * p.C * p.Q + q;
  const int nw = 2 * p.n_src;
  for (int c0 = 0; c0 < p.C; c0 += 8) {  // 16 gathers in flight per lane; wrel / bias are wave-uniform (scalar loads)
    float a0[8], a1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i;
      a0[i] = c < p.C ? up[0][(long long)c * plane[0]] : 0.f;
      a1[i] = (c < p.C && p.n_src > 1) ? up[1][(long long)c * plane[1]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i;
      if (c < p.C) {
        float v = a0[i] + a1[i];
        const float* w = p.wrel + c * nw;
        v = fmaf(w[0], rel[0], v);
        v = fmaf(w[1], rel[1], v);
        if (p.n_src > 1) { v = fmaf(w[2], rel[2], v); v = fmaf(w[3], rel[3], v); }
        if (p.bias) v += p.bias[c];
        op[(long long)c * p.Q] = fmaxf(v, 0.f);
      }
    }
  }
}

__global__ __launch_bounds__(256) void softmax_convex_kernel(const float* __restrict__ disp, const float* __restrict__ scale,
                                                             const float* __restrict__ mask, const float* __restrict__ coord,
                                                             float* __restrict__ out, int B, int H, int W, int Q, int logits,
                                                             float lo, float hi) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)B * Q) return;
  const int b = (int)(t / Q);
  const int q = (int)(t - (long long)b * Q);
  const float* mp = mask + (long long)b * 9 * Q + q;
  float l[9];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    l[k] = mp[(long long)k * Q];
    mx = fmaxf(mx, l[k]);
  }
  float s = 1.f;
  if (logits) {
    s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      l[k] = expf(l[k] - mx);
      s += l[k];
    }
  }
  const int iy = nearest_idx(fminf(fmaxf(coord[t * 2 + 0], lo), hi), H);
  const int ix = nearest_idx(fminf(fmaxf(coord[t * 2 + 1], lo), hi), W);
  const float* dp = disp + (long long)b * H * W;
  const float sc = scale ? scale[b] : 1.f;
  const float four = scale ? 4.f : 1.f;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = iy + k / 3 - 1, xx = ix + k % 3 - 1;
    const float d = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? __fmul_rn(__fmul_rn(dp[(long long)yy * W + xx], four), sc) : 0.f;
    acc += d * (logits ? l[k] / s : l[k]);
  }
  out[t] = acc;
}

}  // namespace

extern "C" {

int as_structure_feature(const float* x, float* out, float* ws, int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(x && out && ws, AS_ERR_BAD_ARG, "structure_feature: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "structure_feature: non-positive size");
  const long long plane = (long long)H * W, P = plane * B;
  AS_REQUIRE(plane < 2147483647ll, AS_ERR_BAD_SHAPE, "structure_feature: plane too large");
  const dim3 grid((unsigned)as::cdiv64(P, 256));
  hipLaunchKernelGGL(sf_norm_copy_kernel, grid, dim3(256), 0, as::as_stream(stream), x, out, ws, C, plane, P);
  hipLaunchKernelGGL(sf_affinity_kernel, grid, dim3(256), 0, as::as_stream(stream), x, (const float*)ws, out, C, H, W, P);
  return as::check_launch("structure_feature");
}

int as_liif_gather(const float* feat, const float* coord, float* latent, int B, int C, int H, int W, int Q, int lat_ctot,
                   int lat_coff, void* stream) {
  AS_REQUIRE(feat && coord && latent, AS_ERR_BAD_ARG, "liif_gather: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Q > 0, AS_ERR_BAD_ARG, "liif_gather: non-positive size");
  AS_REQUIRE(lat_coff >= 0 && lat_coff + C + 2 <= lat_ctot, AS_ERR_BAD_SHAPE, "liif_gather: latent channel window [%d,%d) outside %d", lat_coff, lat_coff + C + 2, lat_ctot);
  GatherParams p{};
  p.feat = feat; p.coord = coord; p.latent = latent;
  p.B = B; p.C = C; p.H = H; p.W = W; p.Q = Q; p.lat_ctot = lat_ctot; p.lat_coff = lat_coff;
  p.lo = (float)(-1.0 + 1e-6); p.hi = (float)(1.0 - 1e-6);
  p.c0y = (float)(-1.0 + 1.0 / H); p.sy = (float)(2.0 * (1.0 / H));
  p.c0x = (float)(-1.0 + 1.0 / W); p.sx = (float)(2.0 * (1.0 / W));
  hipLaunchKernelGGL(liif_gather_kernel, dim3((unsigned)as::cdiv64((long long)B * Q, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("liif_gather");
}

int as_liif_gather_mlp1(const float* u0, const float* u1, const float* coord, const float* wrel, const float* bias, float* out,
                        int B, int C, int H0, int W0, int H1, int W1, int Q, void* stream) {
  AS_REQUIRE(u0 && coord && wrel && out, AS_ERR_BAD_ARG, "liif_gather_mlp1: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H0 > 0 && W0 > 0 && Q > 0, AS_ERR_BAD_ARG, "liif_gather_mlp1: non-positive size");
  AS_REQUIRE(!u1 || (H1 > 0 && W1 > 0), AS_ERR_BAD_ARG, "liif_gather_mlp1: second source without a size");
  Mlp1Params p{};
  p.u[0] = u0; p.u[1] = u1; p.coord = coord; p.wrel = wrel; p.bias = bias; p.out = out;
  p.B = B; p.C = C; p.Q = Q; p.n_src = u1 ? 2 : 1;
  p.H[0] = H0; p.W[0] = W0; p.H[1] = u1 ? H1 : 1; p.W[1] = u1 ? W1 : 1;
  p.lo = (float)(-1.0 + 1e-6); p.hi = (float)(1.0 - 1e-6);
  for (int s = 0; s < 2; ++s) {
    p.c0y[s] = (float)(-1.0 + 1.0 / p.H[s]); p.sy[s] = (float)(2.0 * (1.0 / p.H[s]));
    p.c0x[s] = (float)(-1.0 + 1.0 / p.W[s]); p.sx[s] = (float)(2.0 * (1.0 / p.W[s]));
  }
  hipLaunchKernelGGL(liif_mlp1_gather_kernel, dim3((unsigned)as::cdiv64((long long)B * Q, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("liif_gather_mlp1");
}

int as_convex_upsample(const float* disp, const float* scale, const float* mask, const float* coord, float* out,
                       int B, int H, int W, int Q, int mask_is_logits, void* stream) {
  AS_REQUIRE(disp && mask && coord && out, AS_ERR_BAD_ARG, "convex_upsample: null pointer");
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && Q > 0, AS_ERR_BAD_ARG, "convex_upsample: non-positive size");
  hipLaunchKernelGGL(softmax_convex_kernel, dim3((unsigned)as::cdiv64((long long)B * Q, 256)), dim3(256), 0, as::as_stream(stream),
                     disp, scale, mask, coord, out, B, H, W, Q, mask_is_logits, (float)(-1.0 + 1e-6), (float)(1.0 - 1e-6));
  return as::check_launch("convex_upsample");
}

}  // extern "C"
