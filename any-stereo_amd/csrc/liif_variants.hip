// §8 f4: the implicit upsampler's off-by-default options (liif.py:108-176 gathers, :339-370 SpatialEncoding,
// :575-678 the per-source latent; submodule.py:375-399 the four-tap convex sum).  One general latent builder instead of a
// kernel per option: per source it writes [gathered features | encoded relative coordinate | cell] CHANNEL-major into
// the [B, ctot, Q] latent that the 1x1-conv MLP reads.  HBM/L2-bound gathers; lanes = consecutive queries, so every
// store is one coalesced wave store.
#include "common.h"

namespace {

__device__ __forceinline__ int nearest_idx(float c, int n) {  // grid_sample(nearest, align_corners=False), ATen's op order
  const float u = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(c, 1.f), (float)n), 1.f), 2.f);
  return (int)rintf(u);
}

struct LatentParams {
  const float* feat;    // [B,C,H,W]
  const float* coord;   // [B,Q,2] (row, col), unclamped
  const float* emb;     // [n_enc,2] frequency rows of SpatialEncoding, or null
  const float* cell;    // [B,Q,2] or null
  float* latent;        // [B,ctot,Q]
  float* d_feat;        // backward only
  int B, C, H, W, Q, lat_ctot, lat_coff;
  int unfold9;          // features = the 3x3 zero-padded neighbourhood, channel c*9 + ky*3 + kx  (F.unfold, liif.py:655)
  int n_samp;           // 1 = nearest; 4 = the four half-cell shifted nearest samples (liif.py:156-169)
  int n_enc;
  float sh_y[2], sh_x[2];   // (float)(v * r + 1e-6) for v = -1, +1
  float lo, hi;
  float c0y, sy, c0x, sx;   // make_coord: centre(i) = c0 + s*i
};

__device__ __forceinline__ void sample_index(const LatentParams& p, float cr, float cc, int s, int& iy, int& ix) {
  float r = cr, c = cc;
  if (p.n_samp == 4) {  // (vx, vy) in the order (-1,-1), (-1,1), (1,-1), (1,1); vx shifts the ROW coordinate
    r = __fadd_rn(cr, p.sh_y[s >> 1]);
    c = __fadd_rn(cc, p.sh_x[s & 1]);
  }
  iy = nearest_idx(fminf(fmaxf(r, p.lo), p.hi), p.H);
  ix = nearest_idx(fminf(fmaxf(c, p.lo), p.hi), p.W);
  iy = min(max(iy, 0), p.H - 1);  // no-ops after the clamp (kept so that a NaN coordinate cannot index outside the map)
  ix = min(max(ix, 0), p.W - 1);
}

__global__ __launch_bounds__(256) void latent_kernel(LatentParams p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)p.B * p.Q) return;
  const int b = (int)(t / p.Q);
  const int q = (int)(t - (long long)b * p.Q);
  const float cr = p.coord[t * 2 + 0], cc = p.coord[t * 2 + 1];
  const long long plane = (long long)p.H * p.W;
  const float* fb = p.feat + (long long)b * p.C * plane;
  float* lp = p.latent + ((long long)b * p.lat_ctot + p.lat_coff) * p.Q + q;
  float cy[4], cx[4];
  for (int s = 0; s < p.n_samp; ++s) {
    int iy, ix;
    sample_index(p, cr, cc, s, iy, ix);
    cy[s] = __fadd_rn(p.c0y, __fmul_rn(p.sy, (float)iy));
    cx[s] = __fadd_rn(p.c0x, __fmul_rn(p.sx, (float)ix));
    if (!p.unfold9) {
      const float* fp = fb + (long long)iy * p.W + ix;
      int c = 0;
      for (; c + 4 <= p.C; c += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = fp[(long long)(c + u) * plane];
#pragma unroll
        for (int u = 0; u < 4; ++u) lp[(long long)(c + u) * p.Q] = v[u];
      }
      for (; c < p.C; ++c) lp[(long long)c * p.Q] = fp[(long long)c * plane];
      lp += (long long)p.C * p.Q;
    } else {
      int off[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int yy = iy + k / 3 - 1, xx = ix + k % 3 - 1;
        off[k] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? yy * p.W + xx : -1;
      }
      for (int c = 0; c < p.C; ++c) {
        const float* fc = fb + (long long)c * plane;
#pragma unroll
        for (int k = 0; k < 9; ++k) lp[(long long)(c * 9 + k) * p.Q] = off[k] >= 0 ? fc[off[k]] : 0.f;
      }
      lp += (long long)p.C * 9 * p.Q;
    }
  }
  // relative coordinate to the sample's cell centre; four samples: to the mean of the (-1,-1) and (1,1) centres (liif.py:170-175)
  float qy = cy[0], qx = cx[0];
  if (p.n_samp == 4) {
    qy = __fdiv_rn(__fadd_rn(cy[0], cy[3]), 2.f);
    qx = __fdiv_rn(__fadd_rn(cx[0], cx[3]), 2.f);
  }
  const float ry = __fmul_rn(__fsub_rn(cr, qy), (float)p.H);
  const float rx = __fmul_rn(__fsub_rn(cc, qx), (float)p.W);
  lp[0] = ry;
  lp[p.Q] = rx;
  lp += 2ll * p.Q;
  if (p.n_enc > 0) {  // cat(x, sin(x emb^T), cos(x emb^T))  (liif.py:362-367)
    for (int j = 0; j < p.n_enc; ++j) {
      const float y = fmaf(rx, p.emb[2 * j + 1], __fmul_rn(ry, p.emb[2 * j]));
      lp[(long long)j * p.Q] = sinf(y);
      lp[(long long)(p.n_enc + j) * p.Q] = cosf(y);
    }
    lp += 2ll * p.n_enc * p.Q;
  }
  if (p.cell) {
    lp[0] = p.cell[t * 2 + 0];
    lp[p.Q] = p.cell[t * 2 + 1];
  }
}

// backward of the FEATURE part of latent_kernel: d_feat[b, c, sample pixel (+ tap)] += d_latent[b, coff + ..., q].
// Queries of one pixel are usually neighbours in the launch order (a row-major query grid, or the sorted training
// order), so runs of equal target index inside a wave are summed with shuffles first and only the run head issues
// the atomic (same scheme as liif_gather_bwd_kernel, csrc/backward.hip).
__global__ __launch_bounds__(256) void latent_bwd_kernel(LatentParams p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = t < (long long)p.B * p.Q;
  const long long tt = ok ? t : 0;
  const int b = (int)(tt / p.Q);
  const int q = (int)(tt - (long long)b * p.Q);
  const float cr = p.coord[tt * 2 + 0], cc = p.coord[tt * 2 + 1];
  const long long plane = (long long)p.H * p.W;
  float* fb = p.d_feat + (long long)b * p.C * plane;
  const float* lp = p.latent + ((long long)b * p.lat_ctot + p.lat_coff) * p.Q + q;
  const int lane = threadIdx.x & 63;
  for (int s = 0; s < p.n_samp; ++s) {
    int iy, ix;
    sample_index(p, cr, cc, s, iy, ix);
    const int key = ok ? (b * (int)plane + iy * p.W + ix) : -1;
    // runs = maximal stretches of consecutive lanes with equal keys; lane + 2^k belongs to this lane's run iff no run starts
    // in (lane, lane + 2^k]
    const int prev = __shfl_up(key, 1);
    const bool starts = lane == 0 || prev != key;
    const unsigned long long heads = __ballot(starts);
    const unsigned long long after = (heads >> lane) >> 1;
    unsigned same = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (lane + (1 << k) < 64 && (after & ((1ull << (1 << k)) - 1ull)) == 0ull) same |= 1u << k;
    const bool head = starts;
    const int nch = p.unfold9 ? p.C * 9 : p.C;
    for (int ch = 0; ch < nch; ++ch) {
      float v = ok ? lp[(long long)ch * p.Q] : 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float o = __shfl_down(v, 1 << k);
        v += ((same >> k) & 1u) ? o : 0.f;
      }
      if (head && ok) {
        if (!p.unfold9) {
          atomicAdd(fb + (long long)ch * plane + (long long)iy * p.W + ix, v);
        } else {
          const int c = ch / 9, k9 = ch - c * 9;
          const int yy = iy + k9 / 3 - 1, xx = ix + k9 % 3 - 1;
          if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) atomicAdd(fb + (long long)c * plane + (long long)yy * p.W + xx, v);
        }
      }
    }
    lp += (long long)nch * p.Q;
  }
}

__global__ __launch_bounds__(256) void convex_quater_kernel(const float* __restrict__ disp, const float* __restrict__ scale,
                                                            const float* __restrict__ mask, const float* __restrict__ coord,
                                                            float* __restrict__ out, int B, int H, int W, int Q, int logits,
                                                            float lo, float hi, float shy0, float shy1, float shx0, float shx1) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)B * Q) return;
  const int b = (int)(t / Q);
  const int q = (int)(t - (long long)b * Q);
  const float* mp = mask + (long long)b * 4 * Q + q;
  float l[4];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    l[k] = mp[(long long)k * Q];
    mx = fmaxf(mx, l[k]);
  }
  float s = 1.f;
  if (logits) {
    s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      l[k] = expf(l[k] - mx);
      s += l[k];
    }
  }
  const float cr = coord[t * 2 + 0], cc = coord[t * 2 + 1];
  const float* dp = disp + (long long)b * H * W;
  const float sc = scale ? scale[b] : 1.f;
  const float four = scale ? 4.f : 1.f;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float r = __fadd_rn(cr, (k >> 1) ? shy1 : shy0), c = __fadd_rn(cc, (k & 1) ? shx1 : shx0);
    int iy = nearest_idx(fminf(fmaxf(r, lo), hi), H), ix = nearest_idx(fminf(fmaxf(c, lo), hi), W);
    iy = min(max(iy, 0), H - 1);
    ix = min(max(ix, 0), W - 1);
    const float d = __fmul_rn(__fmul_rn(dp[(long long)iy * W + ix], four), sc);
    acc += d * (logits ? l[k] / s : l[k]);
  }
  out[t] = acc;
}

// ---- AffinityFeature backward (liif.py:432-446 under autograd: the 'with_ISU' / 'with_1_4ISU' / 'only_ISU' modes feed the
// affinity of the LIVE feature map, so the loss reaches x through F.normalize and the eight dot products) ------------------
// n(p) = max(||x(p)||, eps), xh = x / n, aff_j(p) = max(0, xh(p)·xh(p+o_j)).  With g_j the incoming gradient masked by
// aff_j > 0 and o_(7-j) = -o_j:   d xh_c(p) = sum_j w_j(p) xh_c(p+o_j),  w_j(p) = g_j(p) + g_(7-j)(p+o_j);
// dx_c(p) = (d xh_c(p) - xh_c(p) * sum_j w_j(p) aff_j(p)) / n(p)   (+ the pass-through gradient of cat(x, aff)).
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ x, float* __restrict__ ws, int C, long long plane,
                                                   long long P) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= P) return;
  const long long b = pix / plane, rem = pix - b * plane;
  const float* xp = x + b * C * plane + rem;
  float ss = 0.f;
  for (int c = 0; c < C; ++c) {
    const float v = xp[(long long)c * plane];
    ss += v * v;
  }
  ws[pix] = fmaxf(sqrtf(ss), 1e-12f);
}

__global__ __launch_bounds__(256) void affinity_bwd_kernel(const float* __restrict__ x, const float* __restrict__ nrm,
                                                           const float* __restrict__ aff, long long aff_bs,
                                                           const float* __restrict__ g_aff, long long g_aff_bs,
                                                           const float* __restrict__ g_x, long long g_x_bs, float* __restrict__ dx,
                                                           int C, int H, int W, long long P) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= P) return;
  const long long plane = (long long)H * W;
  const long long b = pix / plane;
  const int rem = (int)(pix - b * plane);
  const int y = rem / W, xx = rem - y * W;
  const float* ap = aff + b * aff_bs;
  const float* gp = g_aff + b * g_aff_bs;
  const float* np = nrm + b * plane;
  int noff[8];
  float w[8], inv_nn[8];
  float dot = 0.f;
  {
    int j = 0;
#pragma unroll
    for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
      for (int ox = -1; ox <= 1; ++ox) {
        if (oy == 0 && ox == 0) continue;
        const int yy = y + oy, x2 = xx + ox;
        const bool ok = yy >= 0 && yy < H && x2 >= 0 && x2 < W;
        noff[j] = ok ? yy * W + x2 : -1;
        float wj = 0.f, a = 0.f;
        if (ok) {
          a = ap[(long long)j * plane + rem];
          if (a > 0.f) wj = gp[(long long)j * plane + rem];
          const float a2 = ap[(long long)(7 - j) * plane + noff[j]];
          if (a2 > 0.f) wj += gp[(long long)(7 - j) * plane + noff[j]];
        }
        w[j] = wj;
        dot = fmaf(wj, a, dot);
        inv_nn[j] = ok ? 1.f / np[noff[j]] : 0.f;
        ++j;
      }
  }
  const float inv_n0 = 1.f / np[rem];
  const float* xb = x + b * C * plane;
  float* dxp = dx + b * C * plane + rem;
  const float* gx = g_x ? g_x + b * g_x_bs + rem : nullptr;
  for (int c = 0; c < C; ++c) {
    const float* xc = xb + (long long)c * plane;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(w[j] * inv_nn[j], noff[j] >= 0 ? xc[noff[j]] : 0.f, acc);
    float v = (acc - xc[rem] * inv_n0 * dot) * inv_n0;
    if (gx) v += gx[(long long)c * plane];
    dxp[(long long)c * plane] = v;
  }
}

// backward of convex_quater_kernel: d_mask [B,4,Q] (through the softmax when the mask holds logits) and the scatter-add of
// the four samples into d_disp (zero-filled by the entry point)
__global__ __launch_bounds__(256) void convex_quater_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ scale,
                                                                const float* __restrict__ mask, const float* __restrict__ coord,
                                                                const float* __restrict__ dout, float* __restrict__ dmask,
                                                                float* __restrict__ ddisp, int B, int H, int W, int Q, int logits,
                                                                float lo, float hi, float shy0, float shy1, float shx0, float shx1) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)B * Q) return;
  const int b = (int)(t / Q);
  const int q = (int)(t - (long long)b * Q);
  const float* mp = mask + (long long)b * 4 * Q + q;
  float l[4], d[4];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    l[k] = mp[(long long)k * Q];
    mx = fmaxf(mx, l[k]);
  }
  if (logits) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      l[k] = expf(l[k] - mx);
      s += l[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) l[k] /= s;
  }
  const float cr = coord[t * 2 + 0], cc = coord[t * 2 + 1];
  const float* dp = disp + (long long)b * H * W;
  const float mul = scale ? 4.f * scale[b] : 1.f;
  const float g = dout[t];
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float r = __fadd_rn(cr, (k >> 1) ? shy1 : shy0), c = __fadd_rn(cc, (k & 1) ? shx1 : shx0);
    int iy = nearest_idx(fminf(fmaxf(r, lo), hi), H), ix = nearest_idx(fminf(fmaxf(c, lo), hi), W);
    iy = min(max(iy, 0), H - 1);
    ix = min(max(ix, 0), W - 1);
    d[k] = dp[(long long)iy * W + ix] * mul;
    acc = fmaf(d[k], l[k], acc);
    if (ddisp) atomicAdd(ddisp + (long long)b * H * W + (long long)iy * W + ix, g * l[k] * mul);
  }
  float* dm = dmask + (long long)b * 4 * Q + q;
#pragma unroll
  for (int k = 0; k < 4; ++k) dm[(long long)k * Q] = logits ? g * l[k] * (d[k] - acc) : g * d[k];
}

int fill(LatentParams& p, const float* feat, const float* coord, const float* emb, const float* cell, float* latent, int B, int C,
         int H, int W, int Q, int lat_ctot, int lat_coff, int unfold9, int n_samp, int n_enc, const char* what) {
  AS_REQUIRE(feat && coord && latent, AS_ERR_BAD_ARG, "%s: null pointer", what);
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Q > 0, AS_ERR_BAD_ARG, "%s: non-positive size", what);
  AS_REQUIRE(n_samp == 1 || n_samp == 4, AS_ERR_BAD_ARG, "%s: n_samp must be 1 or 4, got %d", what, n_samp);
  AS_REQUIRE(n_enc >= 0 && (n_enc == 0 || emb), AS_ERR_BAD_ARG, "%s: n_enc=%d without a frequency table", what, n_enc);
  AS_REQUIRE((long long)B * H * W < 2147483647ll, AS_ERR_BAD_SHAPE, "%s: map too large", what);
  const int width = (unfold9 ? 9 * C : C) * n_samp + 2 + 2 * n_enc + (cell ? 2 : 0);
  AS_REQUIRE(lat_coff >= 0 && lat_coff + width <= lat_ctot, AS_ERR_BAD_SHAPE, "%s: latent channel window [%d,%d) outside %d", what,
             lat_coff, lat_coff + width, lat_ctot);
  p.feat = feat; p.coord = coord; p.emb = emb; p.cell = cell; p.latent = latent;
  p.B = B; p.C = C; p.H = H; p.W = W; p.Q = Q; p.lat_ctot = lat_ctot; p.lat_coff = lat_coff;
  p.unfold9 = unfold9 ? 1 : 0; p.n_samp = n_samp; p.n_enc = n_enc;
  p.lo = (float)(-1.0 + 1e-6); p.hi = (float)(1.0 - 1e-6);
  p.c0y = (float)(-1.0 + 1.0 / H); p.sy = (float)(2.0 * (1.0 / H));
  p.c0x = (float)(-1.0 + 1.0 / W); p.sx = (float)(2.0 * (1.0 / W));
  // liif.py:149-150,160-161: rx = 2/H/2 shifts the row, ry = 2/W/2 the column; the Python double `v*r + 1e-6` is rounded to
  // fp32 when it is added to the fp32 coordinate tensor
  const double ry_ = 2.0 / H / 2.0, rx_ = 2.0 / W / 2.0;
  p.sh_y[0] = (float)(-1.0 * ry_ + 1e-6); p.sh_y[1] = (float)(1.0 * ry_ + 1e-6);
  p.sh_x[0] = (float)(-1.0 * rx_ + 1e-6); p.sh_x[1] = (float)(1.0 * rx_ + 1e-6);
  return AS_OK;
}

}  // namespace

extern "C" {

int as_liif_latent(const float* feat, const float* coord, const float* emb, const float* cell, float* latent, int B, int C, int H,
                   int W, int Q, int lat_ctot, int lat_coff, int unfold9, int n_samp, int n_enc, void* stream) {
  LatentParams p{};
  const int rc = fill(p, feat, coord, emb, cell, latent, B, C, H, W, Q, lat_ctot, lat_coff, unfold9, n_samp, n_enc, "liif_latent");
  if (rc != AS_OK) return rc;
  hipLaunchKernelGGL(latent_kernel, dim3((unsigned)as::cdiv64((long long)B * Q, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("liif_latent");
}

int as_liif_latent_bwd(const float* d_latent, const float* coord, float* d_feat, int B, int C, int H, int W, int Q, int lat_ctot,
                       int lat_coff, int unfold9, int n_samp, void* stream) {
  LatentParams p{};
  AS_REQUIRE(d_feat, AS_ERR_BAD_ARG, "liif_latent_bwd: null pointer");
  // the forward's window check (feature channels + the two coordinate channels; encoding / cell channels carry no gradient to feat)
  const int rc = fill(p, d_feat, coord, nullptr, nullptr, const_cast<float*>(d_latent), B, C, H, W, Q, lat_ctot, lat_coff,
                      unfold9, n_samp, 0, "liif_latent_bwd");
  if (rc != AS_OK) return rc;
  p.d_feat = d_feat;
  const int zrc = as::zero_fill(d_feat, (long long)B * C * H * W, as::as_stream(stream));
  if (zrc != AS_OK) return zrc;
  hipLaunchKernelGGL(latent_bwd_kernel, dim3((unsigned)as::cdiv64((long long)B * Q, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("liif_latent_bwd");
}

int as_convex_upsample_quater(const float* disp, const float* scale, const float* mask, const float* coord, float* out, int B, int H,
                              int W, int Q, int mask_is_logits, void* stream) {
  AS_REQUIRE(disp && mask && coord && out, AS_ERR_BAD_ARG, "convex_upsample_quater: null pointer");
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && Q > 0, AS_ERR_BAD_ARG, "convex_upsample_quater: non-positive size");
  const double ry = 2.0 / H / 2.0, rx = 2.0 / W / 2.0;  // submodule.py:383-384
  hipLaunchKernelGGL(convex_quater_kernel, dim3((unsigned)as::cdiv64((long long)B * Q, 256)), dim3(256), 0, as::as_stream(stream),
                     disp, scale, mask, coord, out, B, H, W, Q, mask_is_logits, (float)(-1.0 + 1e-6), (float)(1.0 - 1e-6),
                     (float)(-ry + 1e-6), (float)(ry + 1e-6), (float)(-rx + 1e-6), (float)(rx + 1e-6));
  return as::check_launch("convex_upsample_quater");
}

int as_affinity_bwd(const float* x, const float* aff, long long aff_batch_stride, const float* g_aff, long long g_aff_batch_stride,
                    const float* g_x, long long g_x_batch_stride, float* dx, float* ws, int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(x && aff && g_aff && dx && ws, AS_ERR_BAD_ARG, "affinity_bwd: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "affinity_bwd: non-positive size");
  const long long plane = (long long)H * W, P = plane * B;
  AS_REQUIRE(plane < 2147483647ll, AS_ERR_BAD_SHAPE, "affinity_bwd: plane too large");
  AS_REQUIRE(aff_batch_stride >= 8 * plane && g_aff_batch_stride >= 8 * plane && (!g_x || g_x_batch_stride >= C * plane),
             AS_ERR_BAD_SHAPE, "affinity_bwd: batch stride smaller than the tensor");
  const dim3 grid((unsigned)as::cdiv64(P, 256));
  hipLaunchKernelGGL(norm_kernel, grid, dim3(256), 0, as::as_stream(stream), x, ws, C, plane, P);
  hipLaunchKernelGGL(affinity_bwd_kernel, grid, dim3(256), 0, as::as_stream(stream), x, (const float*)ws, aff, aff_batch_stride, g_aff,
                     g_aff_batch_stride, g_x, g_x_batch_stride, dx, C, H, W, P);
  return as::check_launch("affinity_bwd");
}

int as_convex_upsample_quater_bwd(const float* disp, const float* scale, const float* mask, const float* coord, const float* d_out,
                                  float* d_mask, float* d_disp, int B, int H, int W, int Q, int mask_is_logits, void* stream) {
  AS_REQUIRE(disp && mask && coord && d_out && d_mask, AS_ERR_BAD_ARG, "convex_upsample_quater_bwd: null pointer");
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && Q > 0, AS_ERR_BAD_ARG, "convex_upsample_quater_bwd: non-positive size");
  if (d_disp) {
    const int zrc = as::zero_fill(d_disp, (long long)B * H * W, as::as_stream(stream));
    if (zrc != AS_OK) return zrc;
  }
  const double ry = 2.0 / H / 2.0, rx = 2.0 / W / 2.0;
  hipLaunchKernelGGL(convex_quater_bwd_kernel, dim3((unsigned)as::cdiv64((long long)B * Q, 256)), dim3(256), 0, as::as_stream(stream),
                     disp, scale, mask, coord, d_out, d_mask, d_disp, B, H, W, Q, mask_is_logits, (float)(-1.0 + 1e-6),
                     (float)(1.0 - 1e-6), (float)(-ry + 1e-6), (float)(ry + 1e-6), (float)(-rx + 1e-6), (float)(rx + 1e-6));
  return as::check_launch("convex_upsample_quater_bwd");
}

}  // extern "C"
