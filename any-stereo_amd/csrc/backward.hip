// Backward (training, cfg 4) kernels of the HBM-bound hot-path operators — the transposes of volumes.hip / liif.hip.
// They are what torch.autograd derives for the reference's Python (einsum / avg_pool2d / grid_sample / unfold /
// softmax call sites cited per entry point in include/anystereo_hip.h); the dense convolutions' dgrad/wgrad are not
// here (library kernels).  First, correctness-first versions: one thread per output element, coalesced
// on the side that is written, every output element written exactly once (no zero-fill contract) except the two
// scatter kernels, which clear their destination on the same stream and accumulate with float atomics exactly like
// the reference's grid_sample backward does.
#include "common.h"

namespace {

// ---- a2ᵀ: pooled pyramid levels -> level 0 ----------------------------------------------------------------------
// level i+1 = mean of adjacent pairs of level i, trailing odd element dropped (geometry.py:24,28), so
// d level0[r, x] = Σ_i 2^-i · d level_i[r, x >> i]  for the levels whose width still covers x >> i.
struct PyrBwdParams {
  const float* lv[4];
  float* out;
  long long rows;
  int W2, L;
};

__global__ __launch_bounds__(256) void corr_pyramid_bwd_kernel(PyrBwdParams p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.rows * p.W2) return;
  const long long r = t / p.W2;
  const int x = (int)(t - r * p.W2);
  float acc = p.lv[0][t];
  float sc = 0.5f;
  for (int i = 1; i < p.L; ++i, sc *= 0.5f) {
    const int wi = p.W2 >> i, xi = x >> i;
    if (xi < wi) acc += sc * p.lv[i][r * wi + xi];
  }
  p.out[t] = acc;
}

// geo levels are stored [B,H,W,D>>i,G]; the volume is [B,G,D,H,W] (geometry.py:17-25).
struct GeoBwdParams {
  const float* lv[4];
  float* out;
  int B, G, D, H, W, L;
};

__global__ __launch_bounds__(256) void geo_pyramid_bwd_kernel(GeoBwdParams p) {
  const long long plane = (long long)p.H * p.W;
  const long long total = (long long)p.B * p.G * p.D * plane;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long pix = t % plane;
  long long rest = t / plane;
  const int d = (int)(rest % p.D);
  rest /= p.D;
  const int g = (int)(rest % p.G);
  const long long b = rest / p.G;
  const long long P = b * plane + pix;
  float acc = 0.f, sc = 1.f;
  for (int i = 0; i < p.L; ++i, sc *= 0.5f) {
    const int di = p.D >> i, dd = d >> i;
    if (dd < di) acc += sc * p.lv[i][(P * di + dd) * p.G + g];
  }
  p.out[t] = acc;
}

// ---- a4ᵀ: group-wise correlation (submodule.py:253-271) --------------------------------------------------------
// vol[b,g,d,y,x] = 1/cpg Σ_c fl[b,g·cpg+c,y,x]·fr[b,g·cpg+c,y,x-d] (x >= d)
// d_fl[b,ch,y,x] = 1/cpg Σ_{d<=x} dv[b,g,d,y,x]·fr[b,ch,y,x-d];   d_fr[b,ch,y,x] = 1/cpg Σ_{d: x+d<W} dv[b,g,d,y,x+d]·fl[b,ch,y,x+d]
__global__ __launch_bounds__(256) void gwc_bwd_kernel(const float* __restrict__ fl, const float* __restrict__ fr,
                                                     const float* __restrict__ dv, float* __restrict__ dfl,
                                                     float* __restrict__ dfr, int B, int C, int H, int W, int D, int G) {
  const long long plane = (long long)H * W;
  const long long total = (long long)B * C * plane;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int x = (int)(t % W);
  const long long row = t - x;  // offset of (b,ch,y,0)
  const long long pix = t % plane;
  const int ch = (int)((t / plane) % C);
  const long long b = t / plane / C;
  const int cpg = C / G, g = ch / cpg;
  const float* dvp = dv + ((b * G + g) * D) * plane + (pix - x);  // (b,g,0,y,0)
  float al = 0.f, ar = 0.f;
  for (int d = 0; d < D; ++d) {
    const float* dr = dvp + (long long)d * plane;
    if (d <= x) al += dr[x] * fr[row + x - d];
    if (x + d < W) ar += dr[x + d] * fl[row + x + d];
  }
  const float inv = 1.f / (float)cpg;
  dfl[t] = al * inv;
  dfr[t] = ar * inv;
}

// ---- a5ᵀ: (softmax +) disparity regression (continuous_IGEVstereo.py:267-268, submodule.py:321-325) ------------
__global__ __launch_bounds__(256) void dispreg_bwd_kernel(const float* __restrict__ cost, const float* __restrict__ dout,
                                                         float* __restrict__ dcost, int B, int D, int H, int W, int softmax) {
  const long long plane = (long long)H * W;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)B * plane) return;
  const long long b = t / plane, pix = t - b * plane;
  const float* cp = cost + b * D * plane + pix;
  float* gp = dcost + b * D * plane + pix;
  const float g = dout[t];
  if (!softmax) {
    for (int d = 0; d < D; ++d) gp[(long long)d * plane] = g * (float)d;
    return;
  }
  float mx = -INFINITY;
  for (int d = 0; d < D; ++d) mx = fmaxf(mx, cp[(long long)d * plane]);
  float s = 0.f, m1 = 0.f;
  for (int d = 0; d < D; ++d) {
    const float e = expf(cp[(long long)d * plane] - mx);
    s += e;
    m1 += e * (float)d;
  }
  const float mean = m1 / s;
  for (int d = 0; d < D; ++d) {
    const float pd = expf(cp[(long long)d * plane] - mx) / s;
    gp[(long long)d * plane] = g * pd * ((float)d - mean);
  }
}

// ---- a14ᵀ: nearest gather (liif.py:108-137) ---------------------------------------------------------------------
__device__ __forceinline__ int nearest_idx_b(float c, int n) {
  const float u = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(c, 1.f), (float)n), 1.f), 2.f);
  return (int)rintf(u);
}

// Scatter-add of per-query rows into the low-resolution map.  Lanes whose queries fall into the same source pixel are
// first summed inside the wave (segmented suffix sum over runs of equal keys, 6 shuffle steps) and only the head of
// each run issues the atomic: with the queries sorted by source pixel (the training path sorts them once per step,
// models/base.py) this cuts the atomics by the number of queries per pixel (16 at scale 1); on unsorted queries every
// lane is (almost always) its own run and the kernel degenerates to one atomic per element — correct either way.
__global__ __launch_bounds__(256) void liif_gather_bwd_kernel(const float* __restrict__ dlat, const float* __restrict__ coord,
                                                             float* __restrict__ dfeat, int B, int C, int H, int W, int Q,
                                                             int lat_ctot, int lat_coff, float lo, float hi) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool valid = t < (long long)B * Q;  // no early return: every lane takes part in the shuffles
  const long long tt = valid ? t : 0;
  const long long b = tt / Q, q = tt - b * Q;
  const int iy = nearest_idx_b(fminf(fmaxf(coord[tt * 2 + 0], lo), hi), H);
  const int ix = nearest_idx_b(fminf(fmaxf(coord[tt * 2 + 1], lo), hi), W);
  const bool ok = valid && iy >= 0 && iy < H && ix >= 0 && ix < W;
  const long long plane = (long long)H * W;
  const int key = ok ? (int)(b * plane + (long long)iy * W + ix) : -1;
  const int lane = threadIdx.x & 63;
  // runs = maximal stretches of CONSECUTIVE lanes with equal keys (equal keys further apart are separate runs, each with
  // its own atomic): lane + 2^k belongs to this lane's run iff no run starts in (lane, lane + 2^k]
  const int prev = __shfl_up(key, 1);
  const bool starts = lane == 0 || prev != key;
  const unsigned long long heads = __ballot(starts);
  const unsigned long long after = (heads >> lane) >> 1;
  unsigned same = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (lane + (1 << k) < 64 && (after & ((1ull << (1 << k)) - 1ull)) == 0ull) same |= 1u << k;
  const bool head = ok && starts;
  float* fp = dfeat + b * C * plane + (long long)iy * W + ix;
  const float* lp = dlat + (b * lat_ctot + lat_coff) * Q + q;
  // four channels per trip: their (dependent, ~50-cycle) shuffle chains interleave instead of running one after the other
  int c = 0;
  for (; c + 4 <= C; c += 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = ok ? lp[(long long)(c + u) * Q] : 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float o = __shfl_down(v[u], 1 << k);
        v[u] += ((same >> k) & 1u) ? o : 0.f;
      }
    }
    if (head) {
#pragma unroll
      for (int u = 0; u < 4; ++u) atomicAdd(fp + (long long)(c + u) * plane, v[u]);
    }
  }
  for (; c < C; ++c) {
    float v = ok ? lp[(long long)c * Q] : 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float o = __shfl_down(v, 1 << k);
      v += ((same >> k) & 1u) ? o : 0.f;
    }
    if (head) atomicAdd(fp + (long long)c * plane, v);
  }
}


// Deterministic form of the same transpose (ANYSTEREO_DETERMINISTIC=1 / ops.set_deterministic): the queries of a batch element
// sorted by source pixel once per forward (`order`: a STABLE sort, `starts[b][pix]`: first sorted position of pixel pix, one past
// the last pixel at [npix]); one thread per (batch, channel, pixel) walks its pixel's queries in that fixed order — no atomics,
// no zero fill, the same bits on every run.
__global__ __launch_bounds__(256) void liif_gather_bwd_det_kernel(const float* __restrict__ rows, const int* __restrict__ order,
                                                                 const int* __restrict__ starts, float* __restrict__ out, int B, int C,
                                                                 int npix, int Q, int ctot, int coff) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)B * C * npix) return;
  const int pix = (int)(t % npix);
  const long long bc = t / npix;
  const int c = (int)(bc % C);
  const long long b = bc / C;
  const int s = starts[b * (npix + 1) + pix], e = starts[b * (npix + 1) + pix + 1];
  const float* rp = rows + (b * ctot + coff + c) * Q;
  const int* op = order + b * Q;
  float acc = 0.f;
  for (int i = s; i < e; ++i) acc += rp[op[i]];
  out[t] = acc;
}

// relative coordinates of a14 alone: out[b, 2s+{0,1}, q] = (coord - centre of the nearest cell of source s) * (H_s, W_s)
// and the sort key of the training path: (nearest pixel of source 0) * 4 + parity of the nearest pixel of source 1
struct RelParams {
  const float* coord;
  float* rel;   // [B, 2*n_src, Q] or null
  int* key;     // [B, Q] or null
  int B, Q, n_src;
  int H[2], W[2];
  float lo, hi;
  float c0y[2], sy[2], c0x[2], sx[2];
};

__global__ __launch_bounds__(256) void liif_rel_key_kernel(RelParams p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)p.B * p.Q) return;
  const long long b = t / p.Q, q = t - b * p.Q;
  const float cr = p.coord[t * 2 + 0], cc = p.coord[t * 2 + 1];
  int key = 0;
  for (int s = 0; s < p.n_src; ++s) {
    const int iy = nearest_idx_b(fminf(fmaxf(cr, p.lo), p.hi), p.H[s]);
    const int ix = nearest_idx_b(fminf(fmaxf(cc, p.lo), p.hi), p.W[s]);
    if (p.rel) {
      const float qy = __fadd_rn(p.c0y[s], __fmul_rn(p.sy[s], (float)iy));
      const float qx = __fadd_rn(p.c0x[s], __fmul_rn(p.sx[s], (float)ix));
      p.rel[(b * 2 * p.n_src + 2 * s) * p.Q + q] = __fmul_rn(__fsub_rn(cr, qy), (float)p.H[s]);
      p.rel[(b * 2 * p.n_src + 2 * s + 1) * p.Q + q] = __fmul_rn(__fsub_rn(cc, qx), (float)p.W[s]);
    }
    key = s == 0 ? (iy * p.W[0] + ix) * 4 : key + ((iy & 1) << 1) + (ix & 1);
  }
  if (p.key) p.key[t] = key;
}

// ---- a16/a17ᵀ: (softmax +) convex 3x3 combination at the nearest low-res pixel (submodule.py:357-372) -----------
// d_disp is a scatter-add of nine taps per query onto a map ~16x smaller than the query set: lanes whose queries share the
// nearest pixel (consecutive lanes once the training path has sorted the queries) are summed inside the wave first and only the
// head of each run issues the atomics — same scheme as liif_gather_bwd_kernel (9 atomics per RUN instead of per query:
// 140 -> ~45 us per call at cfg 4).
__global__ __launch_bounds__(256) void convex_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ scale,
                                                        const float* __restrict__ mask, const float* __restrict__ coord,
                                                        const float* __restrict__ dout, float* __restrict__ dmask,
                                                        float* __restrict__ ddisp, int B, int H, int W, int Q, int logits,
                                                        float lo, float hi) {
  const long long t0 = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool valid = t0 < (long long)B * Q;  // no early return: every lane takes part in the shuffles
  const long long t = valid ? t0 : 0;
  const long long b = t / Q, q = t - b * Q;
  const float* mp = mask + b * 9 * Q + q;
  float l[9], dk[9];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    l[k] = mp[(long long)k * Q];
    mx = fmaxf(mx, l[k]);
  }
  if (logits) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      l[k] = expf(l[k] - mx);
      s += l[k];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) l[k] /= s;
  }
  const int iy = nearest_idx_b(fminf(fmaxf(coord[t * 2 + 0], lo), hi), H);
  const int ix = nearest_idx_b(fminf(fmaxf(coord[t * 2 + 1], lo), hi), W);
  const float* dp = disp + b * H * W;
  const float mul = scale ? __fmul_rn(4.f, scale[b]) : 1.f;
  const float g = valid ? dout[t] : 0.f;
  // runs of consecutive lanes with the same (batch, nearest pixel): see liif_gather_bwd_kernel
  const int lane = threadIdx.x & 63;
  const int key = valid ? (int)(b * H * W + (long long)iy * W + ix) : -1;
  const int prev = __shfl_up(key, 1);
  const bool starts = lane == 0 || prev != key;
  const unsigned long long heads = __ballot(starts);
  const unsigned long long after = (heads >> lane) >> 1;
  unsigned same = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (lane + (1 << k) < 64 && (after & ((1ull << (1 << k)) - 1ull)) == 0ull) same |= 1u << k;
  const bool head = valid && starts;
  float out = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = iy + k / 3 - 1, xx = ix + k % 3 - 1;
    const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;  // the same for every lane of a run
    dk[k] = in ? dp[(long long)yy * W + xx] * mul : 0.f;
    out += dk[k] * l[k];
    if (ddisp) {
      float v = in ? g * l[k] * mul : 0.f;
#pragma unroll
      for (int s_ = 0; s_ < 6; ++s_) {
        const float o = __shfl_down(v, 1 << s_);
        v += ((same >> s_) & 1u) ? o : 0.f;
      }
      if (head && in) atomicAdd(ddisp + b * H * W + (long long)yy * W + xx, v);
    }
  }
  if (!valid) return;
  float* gm = dmask + b * 9 * Q + q;
#pragma unroll
  for (int k = 0; k < 9; ++k) gm[(long long)k * Q] = logits ? g * l[k] * (dk[k] - out) : g * dk[k];
}

// ---- a7: ConvGRU gate math (update.py:33-41) as two fused pointwise stages with their transposes --------------------
// stage ZR:  z = sigmoid(lin[:, :C] + cz), r = sigmoid(lin[:, C:] + cr), rh = r * h        (lin = convz‖convr output [B,2C,H,W])
// stage Q :  t = tanh(lin + cq), h' = (1 - z) h + z t
// cz, cr, cq are channel windows of one context tensor [B, ctx_ctot, H, W] (continuous_IGEVstereo.py:273).
struct GateParams {
  const float* lin;
  const float* ctx;
  const float* h;
  const float* z;     // stage Q: input
  float* o0;          // ZR: z        Q: h'
  float* o1;          // ZR: r        Q: t
  float* o2;          // ZR: r*h
  long long plane, total;  // H*W, B*C*H*W
  int C, ctx_ctot, ctx_coff;
};

__global__ __launch_bounds__(256) void gru_gates_zr_kernel(GateParams p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.total) return;
  const long long pix = t % p.plane;
  const long long bc = t / p.plane;
  const int c = (int)(bc % p.C);
  const long long b = bc / p.C;
  const float* lin = p.lin + (b * 2 * p.C) * p.plane + pix;
  const float* cx = p.ctx + (b * p.ctx_ctot + p.ctx_coff) * p.plane + pix;
  const float z = 1.f / (1.f + expf(-(lin[(long long)c * p.plane] + cx[(long long)c * p.plane])));
  const float r = 1.f / (1.f + expf(-(lin[(long long)(p.C + c) * p.plane] + cx[(long long)(p.C + c) * p.plane])));
  p.o0[t] = z;
  p.o1[t] = r;
  p.o2[t] = r * p.h[t];
}

struct GateBwdParams {
  const float* g0;   // ZR: d z       Q: d h'
  const float* g1;   // ZR: d (r*h)
  const float* z;
  const float* r;    // ZR: r         Q: t
  const float* h;
  float* d_lin;      // ZR: [B,2C,H,W]  Q: [B,C,H,W]     (also the gradient of the context window)
  float* d_h;        // partial d h of this stage
  float* d_z;        // Q only
  float* d_ctx;      // optional accumulator of the context gradient [B, ctx_ctot, H, W]: window [ctx_coff, ..) += d_lin
  long long plane, total;
  int C, ctx_ctot, ctx_coff;
};

__global__ __launch_bounds__(256) void gru_gates_zr_bwd_kernel(GateBwdParams p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.total) return;
  const long long pix = t % p.plane;
  const long long bc = t / p.plane;
  const int c = (int)(bc % p.C);
  const long long b = bc / p.C;
  const float z = p.z[t], r = p.r[t], h = p.h[t];
  const float dz = p.g0 ? p.g0[t] : 0.f, drh = p.g1 ? p.g1[t] : 0.f;
  float* dl = p.d_lin + (b * 2 * p.C) * p.plane + pix;
  const float gz = dz * z * (1.f - z), gr = drh * h * r * (1.f - r);
  dl[(long long)c * p.plane] = gz;
  dl[(long long)(p.C + c) * p.plane] = gr;
  p.d_h[t] = drh * r;
  if (p.d_ctx) {  // the context is the same tensor in every GRU iteration: its gradient is summed here, one element per thread
    float* dc = p.d_ctx + (b * p.ctx_ctot + p.ctx_coff) * p.plane + pix;
    dc[(long long)c * p.plane] += gz;
    dc[(long long)(p.C + c) * p.plane] += gr;
  }
}

__global__ __launch_bounds__(256) void gru_gates_q_kernel(GateParams p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.total) return;
  const long long pix = t % p.plane;
  const long long bc = t / p.plane;
  const int c = (int)(bc % p.C);
  const long long b = bc / p.C;
  const float q = tanhf(p.lin[t] + p.ctx[(b * p.ctx_ctot + p.ctx_coff + c) * p.plane + pix]);
  const float z = p.z[t];
  p.o0[t] = (1.f - z) * p.h[t] + z * q;
  p.o1[t] = q;
}

__global__ __launch_bounds__(256) void gru_gates_q_bwd_kernel(GateBwdParams p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= p.total) return;
  const float g = p.g0[t], z = p.z[t], q = p.r[t], h = p.h[t];
  const float gl = g * z * (1.f - q * q);
  p.d_lin[t] = gl;
  p.d_z[t] = g * (q - h);
  p.d_h[t] = g * (1.f - z);
  if (p.d_ctx) {
    const long long pix = t % p.plane, bc = t / p.plane;
    const int c = (int)(bc % p.C);
    const long long b = bc / p.C;
    p.d_ctx[(b * p.ctx_ctot + p.ctx_coff + c) * p.plane + pix] += gl;
  }
}

// ---- a8^T: pool2x / interp of BasicMultiUpdateBlock (update.py:94-102) ------------------------------------------------
// Gather form (one thread per INPUT element, no atomics, fixed summation order).
// pool2x: output (yo, xo) averages rows 2yo-1 .. 2yo+1, divisor 9: an even input row feeds yo = y/2, an odd one (y-1)/2 and (y+1)/2.
__global__ __launch_bounds__(256) void pool2x_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int H, int W, int Ho,
                                                         int Wo, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  long long t = idx / W;
  const int y = (int)(t % H);
  const float* gp = g + (t / H) * Ho * Wo;
  const int ya = y >> 1, yb = (y & 1) ? ya + 1 : ya, xa = x >> 1, xb = (x & 1) ? xa + 1 : xa;
  float s = 0.f;
  for (int yo = ya; yo <= yb; ++yo) {
    if (yo >= Ho) continue;
    for (int xo = xa; xo <= xb; ++xo)
      if (xo < Wo) s += gp[(long long)yo * Wo + xo];
  }
  dx[idx] = s / 9.f;
}

// ---- depthwise 3x3 (padding 1, stride S), training: MobileNetV2's conv_dw layers (extractor.py:331-342 via timm's block names) ----
// MIOpen serves fp32 grouped convolutions with groups == channels with its naive reference solvers only (forward, data and
// weight gradient: 72 calls, 1.65 ms per cfg-4 step).  Forward = as_dwconv3x3; the data gradient at stride 1 is the forward kernel
// on the flipped taps; at stride 2 it is this gather (one thread per INPUT element: the outputs whose window covers it, fixed
// order, no atomics); the weight gradient is a per-channel reduction of nine products over (batch, output pixels), in two
// deterministic stages (slices of the pixel range per block, then a fixed-order sum over the slices by the caller).
__global__ __launch_bounds__(256) void dwconv3x3_s2_bwd_data_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                                    float* __restrict__ dx, int C, int H, int W, int Ho, int Wo, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  long long t = idx / W;
  const int y = (int)(t % H);
  const long long bc = t / H;
  const int c = (int)(bc % C);
  const float* gp = g + bc * Ho * Wo;
  float s = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = y + 1 - ky;  // = 2 * oy
    if (ty < 0 || (ty & 1) || (ty >> 1) >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = x + 1 - kx;
      if (tx < 0 || (tx & 1) || (tx >> 1) >= Wo) continue;
      s = fmaf(w[c * 9 + ky * 3 + kx], gp[(long long)(ty >> 1) * Wo + (tx >> 1)], s);
    }
  }
  dx[idx] = s;
}

// grid (slices, C): block (slice, c) sums d_out[b,c,oy,ox] * x[b,c,oy*S+ky-1,ox*S+kx-1] over its share of the B*Ho*Wo positions
// -> partial[c][slice][9]
template <int S>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                              float* __restrict__ partial, int B, int C, int H, int W, int Ho, int Wo) {
  const int c = blockIdx.y, slice = blockIdx.x, nslice = gridDim.x;
  const long long npos = (long long)B * Ho * Wo;
  const long long per = (npos + nslice - 1) / nslice;
  const long long lo = per * slice, hi = lo + per < npos ? lo + per : npos;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const int ox = (int)(i % Wo);
    const long long r = i / Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const float gv = g[(((long long)b * C + c) * Ho + oy) * Wo + ox];
    const float* xp = x + ((long long)b * C + c) * H * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * S + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * S + kx - 1;
        if (ix >= 0 && ix < W) acc[ky * 3 + kx] = fmaf(gv, xp[(long long)iy * W + ix], acc[ky * 3 + kx]);
      }
    }
  }
  __shared__ float red[4][9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v = acc[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wave][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) partial[((long long)c * nslice + slice) * 9 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// interp (bilinear, align_corners): the weight of output o on input i along one axis, with the forward kernel's arithmetic
__device__ __forceinline__ float interp_axis_weight(int o, int i, int n_in, float sc) {
  const float f = sc * (float)o;
  const int i0 = min((int)f, n_in - 1), i1 = min(i0 + 1, n_in - 1);
  const float t = f - (float)i0;
  return (i == i0 ? 1.f - t : 0.f) + (i == i1 ? t : 0.f);
}
// candidate outputs of input i: sc * o in (i - 1, i + 1), widened by one on both sides (the weight test above is the exact filter)
__device__ __forceinline__ void interp_axis_range(int i, int n_out, float sc, int& lo, int& hi) {
  if (sc <= 0.f) { lo = 0; hi = n_out - 1; return; }
  lo = max(0, (int)floorf((float)(i - 1) / sc) - 1);
  hi = min(n_out - 1, (int)ceilf((float)(i + 1) / sc) + 1);
}
__global__ __launch_bounds__(256) void interp_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int H, int W, int Ho,
                                                         int Wo, float sy, float sx, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  long long t = idx / W;
  const int y = (int)(t % H);
  const float* gp = g + (t / H) * Ho * Wo;
  int ylo, yhi, xlo, xhi;
  interp_axis_range(y, Ho, sy, ylo, yhi);
  interp_axis_range(x, Wo, sx, xlo, xhi);
  float s = 0.f;
  for (int yo = ylo; yo <= yhi; ++yo) {
    const float wy = interp_axis_weight(yo, y, H, sy);
    if (wy == 0.f) continue;
    float r = 0.f;
    for (int xo = xlo; xo <= xhi; ++xo) {
      const float wx = interp_axis_weight(xo, x, W, sx);
      if (wx != 0.f) r += wx * gp[(long long)yo * Wo + xo];
    }
    s += wy * r;
  }
  dx[idx] = s;
}

}  // namespace

extern "C" {

int as_pool2x_bwd(const float* d_out, float* d_x, int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(d_out && d_x, AS_ERR_BAD_ARG, "pool2x_bwd: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "pool2x_bwd: non-positive size");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)B * C * H * W;
  hipLaunchKernelGGL(pool2x_bwd_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream), d_out, d_x, H, W, Ho, Wo, total);
  return as::check_launch("pool2x_bwd");
}

int as_dwconv3x3_s2_bwd_data(const float* d_out, const float* weight, float* d_x, int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(d_out && weight && d_x, AS_ERR_BAD_ARG, "dwconv3x3_s2_bwd_data: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "dwconv3x3_s2_bwd_data: non-positive size");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)B * C * H * W;
  AS_REQUIRE(as::cdiv64(total, 256) < 2147483647ll, AS_ERR_BAD_SHAPE, "dwconv3x3_s2_bwd_data: grid too large");
  hipLaunchKernelGGL(dwconv3x3_s2_bwd_data_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream), d_out, weight, d_x,
                     C, H, W, Ho, Wo, total);
  return as::check_launch("dwconv3x3_s2_bwd_data");
}

int as_dwconv3x3_wgrad_slices(int B, int C, int H, int W, int stride) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return -1;
  const long long npos = (long long)B * ((H - 1) / stride + 1) * ((W - 1) / stride + 1);
  long long n = npos / 4096;  // >= 16 positions per thread and block
  const long long fill = 1024 / C;  // about four blocks per CU over all channels
  if (n > fill) n = fill;
  if (n < 1) n = 1;
  return (int)(n > 64 ? 64 : n);
}

int as_dwconv3x3_wgrad(const float* x, const float* d_out, float* partial, int slices, int B, int C, int H, int W, int stride, void* stream) {
  AS_REQUIRE(x && d_out && partial, AS_ERR_BAD_ARG, "dwconv3x3_wgrad: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), AS_ERR_BAD_ARG, "dwconv3x3_wgrad: bad size / stride");
  AS_REQUIRE(slices >= 1 && slices <= 65535 && C <= 65535, AS_ERR_BAD_SHAPE, "dwconv3x3_wgrad: slices=%d C=%d", slices, C);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const dim3 grid((unsigned)slices, (unsigned)C);
  if (stride == 1) hipLaunchKernelGGL(dwconv3x3_wgrad_kernel<1>, grid, dim3(256), 0, as::as_stream(stream), x, d_out, partial, B, C, H, W, Ho, Wo);
  else hipLaunchKernelGGL(dwconv3x3_wgrad_kernel<2>, grid, dim3(256), 0, as::as_stream(stream), x, d_out, partial, B, C, H, W, Ho, Wo);
  return as::check_launch("dwconv3x3_wgrad");
}

int as_interp_bilinear_ac_bwd(const float* d_out, float* d_x, int B, int C, int H, int W, int Ho, int Wo, void* stream) {
  AS_REQUIRE(d_out && d_x, AS_ERR_BAD_ARG, "interp_bwd: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, AS_ERR_BAD_ARG, "interp_bwd: non-positive size");
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const long long total = (long long)B * C * H * W;
  hipLaunchKernelGGL(interp_bwd_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream), d_out, d_x, H, W, Ho, Wo, sy, sx, total);
  return as::check_launch("interp_bilinear_ac_bwd");
}

int as_gru_gates_zr(const float* lin, const float* ctx, int ctx_ctot, int ctx_coff, const float* h, float* z, float* r, float* rh,
                    int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(lin && ctx && h && z && r && rh, AS_ERR_BAD_ARG, "gru_gates_zr: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "gru_gates_zr: non-positive size");
  AS_REQUIRE(ctx_coff >= 0 && ctx_coff + 2 * C <= ctx_ctot, AS_ERR_BAD_SHAPE, "gru_gates_zr: context window [%d,%d) outside %d", ctx_coff, ctx_coff + 2 * C, ctx_ctot);
  GateParams p{};
  p.lin = lin; p.ctx = ctx; p.h = h; p.o0 = z; p.o1 = r; p.o2 = rh;
  p.plane = (long long)H * W; p.total = p.plane * B * C; p.C = C; p.ctx_ctot = ctx_ctot; p.ctx_coff = ctx_coff;
  hipLaunchKernelGGL(gru_gates_zr_kernel, dim3((unsigned)as::cdiv64(p.total, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("gru_gates_zr");
}

static int gates_zr_bwd(const float* d_z, const float* d_rh, const float* z, const float* r, const float* h, float* d_lin, float* d_h,
                        float* d_ctx, int ctx_ctot, int ctx_coff, int B, int C, int H, int W, void* stream);
int as_gru_gates_zr_bwd(const float* d_z, const float* d_rh, const float* z, const float* r, const float* h, float* d_lin,
                        float* d_h, int B, int C, int H, int W, void* stream) {
  return gates_zr_bwd(d_z, d_rh, z, r, h, d_lin, d_h, nullptr, 0, 0, B, C, H, W, stream);
}
int as_gru_gates_zr_bwd_ctx(const float* d_z, const float* d_rh, const float* z, const float* r, const float* h, float* d_lin,
                            float* d_h, float* d_ctx, int ctx_ctot, int ctx_coff, int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(d_ctx && ctx_coff >= 0 && ctx_coff + 2 * C <= ctx_ctot, AS_ERR_BAD_ARG, "gru_gates_zr_bwd_ctx: context window [%d,%d) outside %d", ctx_coff, ctx_coff + 2 * C, ctx_ctot);
  return gates_zr_bwd(d_z, d_rh, z, r, h, d_lin, d_h, d_ctx, ctx_ctot, ctx_coff, B, C, H, W, stream);
}
static int gates_zr_bwd(const float* d_z, const float* d_rh, const float* z, const float* r, const float* h, float* d_lin, float* d_h,
                        float* d_ctx, int ctx_ctot, int ctx_coff, int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(z && r && h && d_lin && d_h, AS_ERR_BAD_ARG, "gru_gates_zr_bwd: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "gru_gates_zr_bwd: non-positive size");
  GateBwdParams p{};
  p.d_ctx = d_ctx; p.ctx_ctot = ctx_ctot; p.ctx_coff = ctx_coff;
  p.g0 = d_z; p.g1 = d_rh; p.z = z; p.r = r; p.h = h; p.d_lin = d_lin; p.d_h = d_h;
  p.plane = (long long)H * W; p.total = p.plane * B * C; p.C = C;
  hipLaunchKernelGGL(gru_gates_zr_bwd_kernel, dim3((unsigned)as::cdiv64(p.total, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("gru_gates_zr_bwd");
}

int as_gru_gates_q(const float* lin, const float* ctx, int ctx_ctot, int ctx_coff, const float* z, const float* h, float* out,
                   float* t, int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(lin && ctx && z && h && out && t, AS_ERR_BAD_ARG, "gru_gates_q: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "gru_gates_q: non-positive size");
  AS_REQUIRE(ctx_coff >= 0 && ctx_coff + C <= ctx_ctot, AS_ERR_BAD_SHAPE, "gru_gates_q: context window [%d,%d) outside %d", ctx_coff, ctx_coff + C, ctx_ctot);
  GateParams p{};
  p.lin = lin; p.ctx = ctx; p.h = h; p.z = z; p.o0 = out; p.o1 = t;
  p.plane = (long long)H * W; p.total = p.plane * B * C; p.C = C; p.ctx_ctot = ctx_ctot; p.ctx_coff = ctx_coff;
  hipLaunchKernelGGL(gru_gates_q_kernel, dim3((unsigned)as::cdiv64(p.total, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("gru_gates_q");
}

static int gates_q_bwd(const float* d_out, const float* z, const float* t, const float* h, float* d_lin, float* d_z, float* d_h,
                       float* d_ctx, int ctx_ctot, int ctx_coff, int B, int C, int H, int W, void* stream);
int as_gru_gates_q_bwd(const float* d_out, const float* z, const float* t, const float* h, float* d_lin, float* d_z, float* d_h,
                       int B, int C, int H, int W, void* stream) {
  return gates_q_bwd(d_out, z, t, h, d_lin, d_z, d_h, nullptr, 0, 0, B, C, H, W, stream);
}
int as_gru_gates_q_bwd_ctx(const float* d_out, const float* z, const float* t, const float* h, float* d_lin, float* d_z, float* d_h,
                           float* d_ctx, int ctx_ctot, int ctx_coff, int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(d_ctx && ctx_coff >= 0 && ctx_coff + C <= ctx_ctot, AS_ERR_BAD_ARG, "gru_gates_q_bwd_ctx: context window [%d,%d) outside %d", ctx_coff, ctx_coff + C, ctx_ctot);
  return gates_q_bwd(d_out, z, t, h, d_lin, d_z, d_h, d_ctx, ctx_ctot, ctx_coff, B, C, H, W, stream);
}
static int gates_q_bwd(const float* d_out, const float* z, const float* t, const float* h, float* d_lin, float* d_z, float* d_h,
                       float* d_ctx, int ctx_ctot, int ctx_coff, int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(d_out && z && t && h && d_lin && d_z && d_h, AS_ERR_BAD_ARG, "gru_gates_q_bwd: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "gru_gates_q_bwd: non-positive size");
  GateBwdParams p{};
  p.d_ctx = d_ctx; p.ctx_ctot = ctx_ctot; p.ctx_coff = ctx_coff;
  p.g0 = d_out; p.z = z; p.r = t; p.h = h; p.d_lin = d_lin; p.d_z = d_z; p.d_h = d_h;
  p.plane = (long long)H * W; p.total = p.plane * B * C; p.C = C;
  hipLaunchKernelGGL(gru_gates_q_bwd_kernel, dim3((unsigned)as::cdiv64(p.total, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("gru_gates_q_bwd");
}

int as_corr_pyramid_bwd(const float* const* d_levels, float* d_corr0, long long rows, int W2, int L, void* stream) {
  AS_REQUIRE(d_levels && d_corr0, AS_ERR_BAD_ARG, "corr_pyramid_bwd: null pointer");
  AS_REQUIRE(rows > 0 && W2 > 0 && L >= 1 && L <= 4, AS_ERR_BAD_ARG, "corr_pyramid_bwd: bad size (rows %lld, W2 %d, L %d)", rows, W2, L);
  PyrBwdParams p{};
  for (int i = 0; i < L; ++i) {
    AS_REQUIRE(d_levels[i] || (W2 >> i) == 0, AS_ERR_BAD_ARG, "corr_pyramid_bwd: level %d is null", i);
    p.lv[i] = d_levels[i];
  }
  p.out = d_corr0; p.rows = rows; p.W2 = W2; p.L = L;
  hipLaunchKernelGGL(corr_pyramid_bwd_kernel, dim3((unsigned)as::cdiv64(rows * W2, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("corr_pyramid_bwd");
}

int as_geo_pyramid_bwd(const float* const* d_levels, float* d_gev, int B, int G, int D, int H, int W, int L, void* stream) {
  AS_REQUIRE(d_levels && d_gev, AS_ERR_BAD_ARG, "geo_pyramid_bwd: null pointer");
  AS_REQUIRE(B > 0 && G > 0 && D > 0 && H > 0 && W > 0 && L >= 1 && L <= 4, AS_ERR_BAD_ARG, "geo_pyramid_bwd: bad size");
  GeoBwdParams p{};
  for (int i = 0; i < L; ++i) {
    AS_REQUIRE(d_levels[i] || (D >> i) == 0, AS_ERR_BAD_ARG, "geo_pyramid_bwd: level %d is null", i);
    p.lv[i] = d_levels[i];
  }
  p.out = d_gev; p.B = B; p.G = G; p.D = D; p.H = H; p.W = W; p.L = L;
  const long long total = (long long)B * G * D * H * W;
  hipLaunchKernelGGL(geo_pyramid_bwd_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("geo_pyramid_bwd");
}

int as_gwc_volume_bwd(const float* fl, const float* fr, const float* d_vol, float* d_fl, float* d_fr, int B, int C, int H,
                      int W, int D, int G, void* stream) {
  AS_REQUIRE(fl && fr && d_vol && d_fl && d_fr, AS_ERR_BAD_ARG, "gwc_volume_bwd: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && D > 0 && G > 0, AS_ERR_BAD_ARG, "gwc_volume_bwd: non-positive size");
  AS_REQUIRE(C % G == 0, AS_ERR_BAD_SHAPE, "gwc_volume_bwd: C=%d not divisible by G=%d", C, G);
  const long long total = (long long)B * C * H * W;
  hipLaunchKernelGGL(gwc_bwd_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream), fl, fr, d_vol,
                     d_fl, d_fr, B, C, H, W, D, G);
  return as::check_launch("gwc_volume_bwd");
}

int as_disparity_regression_bwd(const float* cost, const float* d_out, float* d_cost, int B, int D, int H, int W,
                                int apply_softmax, void* stream) {
  AS_REQUIRE(cost && d_out && d_cost, AS_ERR_BAD_ARG, "disparity_regression_bwd: null pointer");
  AS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "disparity_regression_bwd: non-positive size");
  hipLaunchKernelGGL(dispreg_bwd_kernel, dim3((unsigned)as::cdiv64((long long)B * H * W, 256)), dim3(256), 0, as::as_stream(stream),
                     cost, d_out, d_cost, B, D, H, W, apply_softmax);
  return as::check_launch("disparity_regression_bwd");
}

int as_liif_gather_bwd(const float* d_latent, const float* coord, float* d_feat, int B, int C, int H, int W, int Q,
                       int lat_ctot, int lat_coff, void* stream) {
  AS_REQUIRE(d_latent && coord && d_feat, AS_ERR_BAD_ARG, "liif_gather_bwd: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Q > 0, AS_ERR_BAD_ARG, "liif_gather_bwd: non-positive size");
  AS_REQUIRE(lat_coff >= 0 && lat_coff + C <= lat_ctot, AS_ERR_BAD_SHAPE, "liif_gather_bwd: latent channel window outside %d", lat_ctot);
  AS_REQUIRE((long long)B * H * W < 2147483647ll, AS_ERR_BAD_SHAPE, "liif_gather_bwd: B*H*W too large for a 32-bit run key");
  const int zrc = as::zero_fill(d_feat, (long long)B * C * H * W, as::as_stream(stream));
  if (zrc != AS_OK) return zrc;
  hipLaunchKernelGGL(liif_gather_bwd_kernel, dim3((unsigned)as::cdiv64((long long)B * Q, 256)), dim3(256), 0, as::as_stream(stream),
                     d_latent, coord, d_feat, B, C, H, W, Q, lat_ctot, lat_coff, (float)(-1.0 + 1e-6), (float)(1.0 - 1e-6));
  return as::check_launch("liif_gather_bwd");
}

int as_liif_gather_bwd_det(const float* d_rows, const int* order, const int* starts, float* out, int B, int C, int npix, int Q,
                           int ctot, int coff, void* stream) {
  AS_REQUIRE(d_rows && order && starts && out, AS_ERR_BAD_ARG, "liif_gather_bwd_det: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && npix > 0 && Q > 0 && coff >= 0 && coff + C <= ctot, AS_ERR_BAD_SHAPE, "liif_gather_bwd_det: bad size / channel window");
  const long long total = (long long)B * C * npix;
  AS_REQUIRE(total < 2147483647ll * 256, AS_ERR_BAD_SHAPE, "liif_gather_bwd_det: grid too large");
  hipLaunchKernelGGL(liif_gather_bwd_det_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream),
                     d_rows, order, starts, out, B, C, npix, Q, ctot, coff);
  return as::check_launch("liif_gather_bwd_det");
}

int as_liif_rel_key(const float* coord, float* rel, int* key, int B, int Q, int n_src, int H0, int W0, int H1, int W1, void* stream) {
  AS_REQUIRE(coord && (rel || key), AS_ERR_BAD_ARG, "liif_rel_key: null pointer");
  AS_REQUIRE(B > 0 && Q > 0 && (n_src == 1 || n_src == 2) && H0 > 0 && W0 > 0 && (n_src == 1 || (H1 > 0 && W1 > 0)), AS_ERR_BAD_ARG,
             "liif_rel_key: bad size");
  AS_REQUIRE((long long)H0 * W0 * 4 < 2147483647ll, AS_ERR_BAD_SHAPE, "liif_rel_key: source 0 too large for a 32-bit key");
  RelParams p{};
  p.coord = coord; p.rel = rel; p.key = key; p.B = B; p.Q = Q; p.n_src = n_src;
  p.H[0] = H0; p.W[0] = W0; p.H[1] = n_src > 1 ? H1 : 1; p.W[1] = n_src > 1 ? W1 : 1;
  p.lo = (float)(-1.0 + 1e-6); p.hi = (float)(1.0 - 1e-6);
  for (int s = 0; s < 2; ++s) {
    p.c0y[s] = (float)(-1.0 + 1.0 / p.H[s]); p.sy[s] = (float)(2.0 * (1.0 / p.H[s]));
    p.c0x[s] = (float)(-1.0 + 1.0 / p.W[s]); p.sx[s] = (float)(2.0 * (1.0 / p.W[s]));
  }
  hipLaunchKernelGGL(liif_rel_key_kernel, dim3((unsigned)as::cdiv64((long long)B * Q, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("liif_rel_key");
}

int as_convex_upsample_bwd(const float* disp, const float* scale, const float* mask, const float* coord, const float* d_out,
                           float* d_mask, float* d_disp, int B, int H, int W, int Q, int mask_is_logits, void* stream) {
  AS_REQUIRE(disp && mask && coord && d_out && d_mask, AS_ERR_BAD_ARG, "convex_upsample_bwd: null pointer");
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && Q > 0, AS_ERR_BAD_ARG, "convex_upsample_bwd: non-positive size");
  AS_REQUIRE((long long)B * H * W < 2147483647ll, AS_ERR_BAD_SHAPE, "convex_upsample_bwd: B*H*W too large for a 32-bit run key");
  if (d_disp) {
    const int zrc = as::zero_fill(d_disp, (long long)B * H * W, as::as_stream(stream));
    if (zrc != AS_OK) return zrc;
  }
  hipLaunchKernelGGL(convex_bwd_kernel, dim3((unsigned)as::cdiv64((long long)B * Q, 256)), dim3(256), 0, as::as_stream(stream),
                     disp, scale, mask, coord, d_out, d_mask, d_disp, B, H, W, Q, mask_is_logits, (float)(-1.0 + 1e-6),
                     (float)(1.0 - 1e-6));
  return as::check_launch("convex_upsample_bwd");
}

}  // extern "C"
