// §8 f4 — backbone-side one-shot operators that MIOpen serves badly at these shapes (measured at 960x540:
// depthwise 3x3 on [1,32,272,480]: 225 us for 33 MB of traffic; Conv3d 8->8 on [1,8,48,136,240]: 885 us for
// 100 MB / 5.4 GFLOP).  Both are direct convolutions: a lane owns one output position (x fastest, so loads
// and stores are coalesced 256-B rows), weights are wave-uniform and arrive through the scalar cache
// (s_load) as SGPR operands of the FMAs, the zero padding comes from the buffer range check (sentinel
// offset) so there is no branch around a load.  BatchNorm (eval) is folded into weights/bias by the caller;
// the activation is fused.
#include <cstdlib>

#include "common.h"

namespace {

using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case AS_ACT_RELU: return fmaxf(v, 0.f);
    case AS_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case AS_ACT_TANH: return tanhf(v);
    case AS_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
    case AS_ACT_LEAKY: return v >= 0.f ? v : 0.01f * v;
    case AS_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    default: return v;
  }
}

constexpr unsigned kOOB = 0x70000000u;  // lane sentinel: stays out of range after adding any in-tensor scalar offset

// ------------------------------------------------------------------------------------------------
// depthwise 3x3, padding 1, stride S: block = 64 columns x (4 waves x R output rows) of one (b, c) plane.  A wave walks its R rows
// with the input rows it has already read kept in registers (stride 1: 3 loads per output instead of 9; stride 2: 6): the
// MobileNetV2 trunk's half-resolution layers are HBM-bound maps of a few MB per plane set that one-output-per-thread blocks
// (35 000 of them at 270x480) spent mostly launching.
// ------------------------------------------------------------------------------------------------
template <int S, int R>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ res,
                                                        float* __restrict__ out, int C, int H, int W, int Ho, int Wo, int act) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bc = blockIdx.z;
  const int c = bc % C;
  const int oy0 = (blockIdx.y * 4 + wave) * R;
  const int ox = blockIdx.x * 64 + lane;
  if (oy0 >= Ho) return;
  const long long plane = (long long)H * W;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long long)bc * plane), 0, (int)(plane * 4), 0x00020000);
  float wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + t];  // uniform -> scalar loads
  unsigned xo[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int ix = ox * S + k - 1;
    xo[k] = (ox < Wo && ix >= 0 && ix < W) ? (unsigned)(ix * 4) : kOOB;
  }
  const float b0 = bias ? bias[c] : 0.f;
  // input rows of the strip: iy = oy0 * S - 1 + j, j = 0 .. (R - 1) * S + 2; row j feeds output r with kernel row ky = j - r * S
  constexpr int NR = (R - 1) * S + 3;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = b0;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int iy = oy0 * S - 1 + j;
    float v[3];
    const bool in = iy >= 0 && iy < H;  // wave-uniform
    const unsigned so = in ? (unsigned)(iy * W * 4) : 0u;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) v[kx] = bload(rs, in ? xo[kx] : kOOB, so);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ky = j - r * S;  // compile-time after unrolling
      if (ky >= 0 && ky < 3) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[r] = fmaf(v[kx], wk[ky * 3 + kx], acc[r]);
      }
    }
  }
  if (ox < Wo) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int oy = oy0 + r;
      if (oy < Ho) {
        const long long o = ((long long)bc * Ho + oy) * Wo + ox;
        float vv = act_apply(acc[r], act);
        if (res) vv += res[o];
        out[o] = vv;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3, padding 1, stride S convolution of a FEW input channels (the image stems: Cin = 3) to Cout = groups x 8 channels, with
// bias (folded BatchNorm) and activation: HBM-bound (27 MAC per output and channel), one thread per output pixel and 8 output
// channels, the 9 x 8 weights of an input channel as scalar operands.  wpack [Cin][9][Cout].
// ------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void conv3x3_few_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                          const float* __restrict__ bias, float* __restrict__ out, int Cin,
                                                          int Cout, int H, int W, int Ho, int Wo, int act) {
  constexpr int CT = 8;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int groups = Cout / CT;
  const int g = blockIdx.z % groups, b = blockIdx.z / groups;
  const int oy = blockIdx.y * 4 + wave;
  const int ox = blockIdx.x * 64 + lane;
  if (oy >= Ho) return;
  const unsigned row_b = (unsigned)W * 4u, plane_b = (unsigned)H * row_b;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long long)b * Cin * (plane_b / 4)), 0,
                                                                      (int)((long long)Cin * plane_b), 0x00020000);
  unsigned xo[3][3], rowoff[3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * S + ky - 1;
    const bool rowok = iy >= 0 && iy < H;  // wave-uniform
    rowoff[ky] = rowok ? (unsigned)iy * row_b : 0u;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * S + kx - 1;
      xo[ky][kx] = (rowok && ox < Wo && ix >= 0 && ix < W) ? (unsigned)(ix * 4) : kOOB;
    }
  }
  float acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) acc[j] = bias ? bias[g * CT + j] : 0.f;
  const float* wt = wp + g * CT;
  unsigned cio = 0;
  for (int ci = 0; ci < Cin; ++ci, cio += plane_b, wt += 9 * Cout) {
    float v[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) v[ky][kx] = bload(rs, xo[ky][kx], cio + rowoff[ky]);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[j] = fmaf(v[ky][kx], wt[(ky * 3 + kx) * Cout + j], acc[j]);
  }
  if (ox < Wo) {
    const long long oplane = (long long)Ho * Wo;
    float* o = out + ((long long)b * Cout + g * CT) * oplane + (long long)oy * Wo + ox;
#pragma unroll
    for (int j = 0; j < CT; ++j) o[j * oplane] = act_apply(acc[j], act);
  }
}

// value of the lane one below / one above in the wave (DPP wave shifts; lane 0 / lane 63 receive 0)
__device__ __forceinline__ float wave_shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
constexpr int kC3dCols = 62;  // outputs per wave of the shift-based kernels (64 loaded columns - 2 halo lanes)

// ------------------------------------------------------------------------------------------------
// Conv3d 3x3x3, padding 1, stride S (all three dims), Cin arbitrary, CT output channels per thread
// (Cout = groups x CT, group = blockIdx.z % groups).  wpack [Cin][27][Cout].
// Block = 64 x-positions x 4 y-rows of one (b, z, group).
// ------------------------------------------------------------------------------------------------
template <int S, int CT, int CO, int NR = 1>
__global__ __launch_bounds__(256) void conv3d_k3_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                        const float* __restrict__ bias, const float* __restrict__ gate, float* __restrict__ out,
                                                        int Cin, int Cout_rt, int D, int H, int W, int Do, int Ho, int Wo, int act) {
  // CO: Cout at compile time (0 = run time).  With a known weight row stride the 9 x CT scalar weights of a slab are
  // loads at immediate offsets from ONE running pointer; with a run-time stride each of the nine rows costs a 64-bit
  // scalar address computation, and the scalar unit (one instruction per SIMD every four cycles) ended up as busy as
  // the vector FMA pipe (~40 scalar against 42 vector instructions per slab).
  // NR (stride 1): output rows per thread.  NR = 2 reads 4 input rows for 2 x 9 x CT FMAs: the slab's weights (288 B of
  // scalar-cache traffic per wave) and two of the rows serve both outputs.
  static_assert(NR == 1 || S == 1, "two rows per thread: stride 1 only");
  const int Cout = CO ? CO : Cout_rt;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int groups = Cout / CT;
  int id = blockIdx.z;
  const int g = id % groups;
  id /= groups;
  const int oz = id % Do;
  const int b = id / Do;
  const int oy = (blockIdx.y * 4 + wave) * NR;
  // stride 1: a wave covers kC3dCols = 62 outputs; lane l loads input column (first output - 1 + l) ONCE per row and hands it
  // to its neighbours with two wave shifts (DPP) — one load per row instead of three overlapping ones
  const int ox = (S == 1) ? blockIdx.x * kC3dCols + lane - 1 : blockIdx.x * 64 + lane;
  if (oy >= Ho) return;
  const unsigned row_b = (unsigned)W * 4u, plane_b = (unsigned)H * row_b, vol_b = (unsigned)D * plane_b;  // Cin*vol_b < kOOB (entry)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long long)b * Cin * (vol_b / 4)), 0,
                                                                      (int)((long long)Cin * vol_b), 0x00020000);
  // per input row: the lane's column offsets (sentinel for rows / columns outside the volume) and the row's byte offset —
  // all loop invariant
  constexpr int NROW = 2 + NR;
  unsigned xo[NROW][3], rowoff[NROW];
#pragma unroll
  for (int r = 0; r < NROW; ++r) {
    const int iy = oy * S + r - 1;
    const bool rowok = iy >= 0 && iy < H;  // wave-uniform
    rowoff[r] = rowok ? (unsigned)iy * row_b : 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ix = (S == 1) ? ox : ox * S + k - 1;  // S == 1 uses entry [r][1] only: this lane's own column
      const bool ok = rowok && ix >= 0 && ix < W && (S == 1 || ox < Wo);
      xo[r][k] = ok ? (unsigned)(ix * 4) : kOOB;
    }
  }
  const bool writes = (S == 1) ? (lane >= 1 && lane <= kC3dCols && ox < Wo) : (ox < Wo);
  float acc[NR][CT];
#pragma unroll
  for (int n = 0; n < NR; ++n)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[n][j] = bias ? bias[g * CT + j] : 0.f;
  const float* wt = wp + g * CT;  // running: + 9*Cout per (ci, kz) slab
  unsigned cio = 0;                // running: ci * vol_b
  for (int ci = 0; ci < Cin; ++ci, cio += vol_b) {
#pragma unroll
    for (int kz = 0; kz < 3; ++kz, wt += 9 * Cout) {
      const int iz = oz * S + kz - 1;
      if (iz < 0 || iz >= D) continue;  // block-uniform
      const unsigned zo = cio + (unsigned)iz * plane_b;
      float v[NROW][3];
#pragma unroll
      for (int r = 0; r < NROW; ++r) {
        if constexpr (S == 1) {
          const float c = bload(rs, xo[r][1], zo + rowoff[r]);
          v[r][1] = c;
          v[r][0] = wave_shr1(c);  // column ox - 1 from lane - 1
          v[r][2] = wave_shl1(c);  // column ox + 1 from lane + 1
        } else {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) v[r][kx] = bload(rs, xo[r][kx], zo + rowoff[r]);
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            const float w = wt[(ky * 3 + kx) * Cout + j];
#pragma unroll
            for (int n = 0; n < NR; ++n) acc[n][j] = fmaf(v[ky + n][kx], w, acc[n][j]);
          }
    }
  }
  if (writes) {
    const long long ovol = (long long)Do * Ho * Wo;
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      if (oy + n >= Ho) break;
      float* o = out + ((long long)b * Cout + g * CT) * ovol + ((long long)oz * Ho + oy + n) * Wo + ox;
      if (gate) {  // FeatureAtt's channel gate [B, Cout, Ho, Wo], the same for every depth slice (submodule.py:328-341)
        const long long oplane = (long long)Ho * Wo;
        const float* gp = gate + ((long long)b * Cout + g * CT) * oplane + (long long)(oy + n) * Wo + ox;
#pragma unroll
        for (int j = 0; j < CT; ++j) o[j * ovol] = act_apply(acc[n][j], act) * gp[j * oplane];
      } else {
#pragma unroll
        for (int j = 0; j < CT; ++j) o[j * ovol] = act_apply(acc[n][j], act);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose3d kernel 4, stride 2, padding 1 (all dims): out[o] = sum_{i,k : o = 2 i - 1 + k} in[i] w[k].
// Per dimension an output position takes exactly two taps: k = (o+1)%2 + {0, 2} from i = (o + 1 - k) / 2.
// A lane owns the output PAIR x = 2 px, 2 px + 1 (they share the inputs px-1, px, px+1), the block a fixed
// (z, y), so the tap set is wave-uniform and the weights are SGPR operands: 3 loads feed 4*CT FMAs.
// wpack [Cin][4][4][4][Cout] (= weight [Cin,Cout,4,4,4] with Cout moved last).
// ------------------------------------------------------------------------------------------------
template <int CT, int CO = 0>
__global__ __launch_bounds__(256) void deconv3d_k4s2_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int Cin, int Cout_rt, int D, int H, int W, int act) {
  const int Cout = CO ? CO : Cout_rt;  // compile-time weight row stride: see conv3d_k3_kernel
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const int groups = Cout / CT;
  int id = blockIdx.z;
  const int g = id % groups;
  id /= groups;
  const int oz = id % Do;
  const int b = id / Do;
  const int oy = blockIdx.y * 4 + wave;
  // input column; outputs 2 px and 2 px + 1.  Like conv3d_k3_kernel<1>: 62 columns per wave, one load per row, the two
  // neighbours by wave shifts
  const int px = blockIdx.x * kC3dCols + lane - 1;
  if (oy >= Ho) return;
  const long long plane = (long long)H * W;
  const long long vol = (long long)D * plane;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long long)b * Cin * vol), 0, (int)((long long)Cin * vol * 4), 0x00020000);
  const unsigned xc = (px >= 0 && px < W) ? (unsigned)(px * 4) : kOOB;
  float a0[CT], a1[CT];  // outputs 2 px (even) and 2 px + 1 (odd)
#pragma unroll
  for (int j = 0; j < CT; ++j) { a0[j] = bias ? bias[g * CT + j] : 0.f; a1[j] = a0[j]; }
  const int kz0 = (oz + 1) & 1, ky0 = (oy + 1) & 1;
  const float* wg = wp + g * CT;
  for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
    for (int tz = 0; tz < 2; ++tz) {
      const int kz = kz0 + 2 * tz;
      const int iz = (oz + 1 - kz) >> 1;  // exact: same parity
      if (iz < 0 || iz >= D) continue;
#pragma unroll
      for (int ty = 0; ty < 2; ++ty) {
        const int ky = ky0 + 2 * ty;
        const int iy = (oy + 1 - ky) >> 1;
        if (iy < 0 || iy >= H) continue;
        const unsigned so = (unsigned)((((long long)ci * D + iz) * H + iy) * W * 4);
        const float v0 = bload(rs, xc, so);
        const float vm = wave_shr1(v0), vp = wave_shl1(v0);
        const float* wt = wg + (long long)(((ci * 4 + kz) * 4 + ky) * 4) * Cout;  // [kx][co]
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          // even output 2 px: (ix = px, kx = 1), (ix = px - 1, kx = 3);  odd 2 px + 1: (ix = px + 1, kx = 0), (ix = px, kx = 2)
          a0[j] = fmaf(v0, wt[1 * Cout + j], a0[j]);
          a0[j] = fmaf(vm, wt[3 * Cout + j], a0[j]);
          a1[j] = fmaf(vp, wt[0 * Cout + j], a1[j]);
          a1[j] = fmaf(v0, wt[2 * Cout + j], a1[j]);
        }
      }
    }
  }
  if (lane >= 1 && lane <= kC3dCols && px < W) {
    const long long ovol = (long long)Do * Ho * Wo;
    float* o = out + ((long long)b * Cout + g * CT) * ovol + ((long long)oz * Ho + oy) * Wo + 2 * px;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      f32x2 v;
      v.x = act_apply(a0[j], act);
      v.y = act_apply(a1[j], act);
      *reinterpret_cast<f32x2*>(o + j * ovol) = v;  // Wo even: 8-B aligned
    }
  }
}

// ------------------------------------------------------------------------------------------------
// InstanceNorm2d (affine = False, biased variance over H*W, submodule.py BasicConv_IN) + activation.
// pass 1: each (plane, segment) block reduces sum and sum of squares in fp64 (no cancellation issue) into
// ws[plane][seg][2]; pass 2 re-derives mean / rstd from the kInSeg partials of its plane and applies them.
// ------------------------------------------------------------------------------------------------
constexpr int kInSeg = 8;

__global__ __launch_bounds__(256) void in_stats_kernel(const float* __restrict__ x, double* __restrict__ ws, long long HW) {
  const long long plane = blockIdx.y;
  const int seg = blockIdx.x;
  const long long per = (HW + kInSeg - 1) / kInSeg;
  const long long lo = seg * per, hi = min(HW, lo + per);
  const float* xp = x + plane * HW;
  double s = 0.0, ss = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256 * 4) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i + k * 256 < hi) ? xp[i + k * 256] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s += (double)v[k]; ss += (double)v[k] * (double)v[k]; }
  }
  __shared__ double red[2][256];
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ws[(plane * kInSeg + seg) * 2 + 0] = red[0][0];
    ws[(plane * kInSeg + seg) * 2 + 1] = red[1][0];
  }
}

__global__ __launch_bounds__(256) void in_apply_kernel(const float* __restrict__ x, const double* __restrict__ ws,
                                                       const float* __restrict__ res, float* __restrict__ out, long long HW,
                                                       float eps, int act) {
  const long long plane = blockIdx.y;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  double s = 0.0, ss = 0.0;
#pragma unroll
  for (int k = 0; k < kInSeg; ++k) { s += ws[(plane * kInSeg + k) * 2]; ss += ws[(plane * kInSeg + k) * 2 + 1]; }
  const double mean = s / (double)HW;
  const double var = fmax(ss / (double)HW - mean * mean, 0.0);
  const float m = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (i < HW) {
    float v = act_apply((x[plane * HW + i] - m) * rstd, act);
    if (res) v = fmaxf(v + res[plane * HW + i], 0.f);  // residual tail relu(res + act(IN(x))) (extractor.py:56-62)
    out[plane * HW + i] = v;
  }
}

// LayerNorm2d: per-pixel normalisation over the C channels of an NCHW tensor, affine, + activation
// (submodule.py:148-187: mu = mean_c x, var = mean_c (x-mu)^2, y = w (x-mu)/sqrt(var+eps) + b).  Lane = pixel.
template <int CMAX>
__global__ __launch_bounds__(256) void layernorm2d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ out,
                                                          int C, long long HW, long long P, float eps, int act) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= P) return;
  const long long bi = pix / HW, rem = pix - bi * HW;
  const float* xp = x + bi * C * HW + rem;
  float v[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) v[c] = c < C ? xp[(long long)c * HW] : 0.f;
  float mu = 0.f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) mu += v[c];
  mu /= (float)C;
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) { const float d = c < C ? v[c] - mu : 0.f; var += d * d; }
  var /= (float)C;
  const float rstd = 1.f / sqrtf(var + eps);
  float* op = out + bi * C * HW + rem;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) op[(long long)c * HW] = act_apply(w[c] * ((v[c] - mu) * rstd) + b[c], act);
}

}  // namespace

extern "C" {

int as_dwconv3x3(const float* x, const float* weight, const float* bias, const float* residual, float* out,
                 int B, int C, int H, int W, int stride, int act, void* stream) {
  AS_REQUIRE(x && weight && out, AS_ERR_BAD_ARG, "dwconv3x3: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "dwconv3x3: non-positive size");
  AS_REQUIRE(stride == 1 || stride == 2, AS_ERR_BAD_ARG, "dwconv3x3: stride=%d (supported: 1, 2)", stride);
  AS_REQUIRE(act >= AS_ACT_NONE && act <= AS_ACT_LEAKY, AS_ERR_BAD_ARG, "dwconv3x3: act=%d", act);
  AS_REQUIRE((long long)H * W * 4 < (long long)kOOB, AS_ERR_BAD_SHAPE, "dwconv3x3: plane too large");
  AS_REQUIRE((long long)B * C <= 65535, AS_ERR_BAD_SHAPE, "dwconv3x3: B*C=%lld exceeds the grid limit", (long long)B * C);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  // rows per wave: long strips on big maps (fewer, longer blocks), one row on the small ones (enough blocks to fill the chip)
  const long long rows_total = (long long)B * C * Ho * as::cdiv(Wo, 64);
  static const int rows_force = getenv("AS_DW_ROWS") ? atoi(getenv("AS_DW_ROWS")) : 0;  // 1 | 4 | 8 (diagnostics / A-B)
  const int R = rows_force ? rows_force : (rows_total >= 8ll * 4 * 4 * 256 ? 8 : (rows_total >= 4ll * 4 * 2 * 256 ? 4 : 1));
  const dim3 grid((unsigned)as::cdiv(Wo, 64), (unsigned)as::cdiv(Ho, 4 * R), (unsigned)(B * C));
  hipStream_t st = as::as_stream(stream);
#define AS_DW(S_, R_) hipLaunchKernelGGL((dwconv3x3_kernel<S_, R_>), grid, dim3(256), 0, st, x, weight, bias, residual, out, C, H, W, Ho, Wo, act)
  if (stride == 1) { if (R == 8) AS_DW(1, 8); else if (R == 4) AS_DW(1, 4); else AS_DW(1, 1); }
  else { if (R == 8) AS_DW(2, 8); else if (R == 4) AS_DW(2, 4); else AS_DW(2, 1); }
#undef AS_DW
  return as::check_launch("dwconv3x3");
}

int as_conv3x3_few(const float* x, const float* wpack, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                   int stride, int act, void* stream) {
  AS_REQUIRE(x && wpack && out, AS_ERR_BAD_ARG, "conv3x3_few: null pointer");
  AS_REQUIRE(B > 0 && Cin > 0 && Cin <= 8 && Cout > 0 && Cout % 8 == 0 && H > 0 && W > 0, AS_ERR_BAD_ARG,
             "conv3x3_few: Cin=%d (1..8), Cout=%d (multiple of 8)", Cin, Cout);
  AS_REQUIRE(stride == 1 || stride == 2, AS_ERR_BAD_ARG, "conv3x3_few: stride=%d (supported: 1, 2)", stride);
  AS_REQUIRE(act >= AS_ACT_NONE && act <= AS_ACT_LEAKY, AS_ERR_BAD_ARG, "conv3x3_few: act=%d", act);
  AS_REQUIRE((long long)Cin * H * W * 4 < (long long)kOOB, AS_ERR_BAD_SHAPE, "conv3x3_few: input too large");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long long gz = (long long)B * (Cout / 8);
  AS_REQUIRE(gz <= 65535, AS_ERR_BAD_SHAPE, "conv3x3_few: B*groups=%lld exceeds the grid limit", gz);
  const dim3 grid((unsigned)as::cdiv(Wo, 64), (unsigned)as::cdiv(Ho, 4), (unsigned)gz);
  hipStream_t s = as::as_stream(stream);
  if (stride == 1) hipLaunchKernelGGL(conv3x3_few_kernel<1>, grid, dim3(256), 0, s, x, wpack, bias, out, Cin, Cout, H, W, Ho, Wo, act);
  else hipLaunchKernelGGL(conv3x3_few_kernel<2>, grid, dim3(256), 0, s, x, wpack, bias, out, Cin, Cout, H, W, Ho, Wo, act);
  return as::check_launch("conv3x3_few");
}

int as_conv3d_k3_gated(const float* x, const float* wpack, const float* bias, const float* gate, float* out,
                       int B, int Cin, int Cout, int D, int H, int W, int stride, int act, void* stream);

int as_conv3d_k3(const float* x, const float* wpack, const float* bias, float* out,
                 int B, int Cin, int Cout, int D, int H, int W, int stride, int act, void* stream) {
  return as_conv3d_k3_gated(x, wpack, bias, nullptr, out, B, Cin, Cout, D, H, W, stride, act, stream);
}

int as_conv3d_k3_gated(const float* x, const float* wpack, const float* bias, const float* gate, float* out,
                       int B, int Cin, int Cout, int D, int H, int W, int stride, int act, void* stream) {
  AS_REQUIRE(x && wpack && out, AS_ERR_BAD_ARG, "conv3d_k3: null pointer");
  AS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "conv3d_k3: non-positive size");
  AS_REQUIRE(stride == 1 || stride == 2, AS_ERR_BAD_ARG, "conv3d_k3: stride=%d (supported: 1, 2)", stride);
  AS_REQUIRE(act >= AS_ACT_NONE && act <= AS_ACT_LEAKY, AS_ERR_BAD_ARG, "conv3d_k3: act=%d", act);
  AS_REQUIRE((long long)Cin * D * H * W * 4 < (long long)kOOB, AS_ERR_BAD_SHAPE, "conv3d_k3: input exceeds 1.75 GiB per batch element");
  const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int ct = (Cout % 8 == 0) ? 8 : 1;
  const long long gz = (long long)B * Do * (Cout / ct);
  AS_REQUIRE(gz <= 65535, AS_ERR_BAD_SHAPE, "conv3d_k3: B*Do*groups=%lld exceeds the grid limit", gz);
  const dim3 grid((unsigned)as::cdiv(Wo, stride == 1 ? kC3dCols : 64), (unsigned)as::cdiv(Ho, 4), (unsigned)gz);
  hipStream_t s = as::as_stream(stream);
#define AS_C3D(S_, CT_, CO_, NR_) hipLaunchKernelGGL((conv3d_k3_kernel<S_, CT_, CO_, NR_>), dim3(grid.x, (unsigned)as::cdiv(Ho, 4 * NR_), grid.z), dim3(256), 0, s, x, wpack, bias, gate, out, Cin, Cout, D, H, W, Do, Ho, Wo, act)
#define AS_C3D_CO(S_, NR_)                                     \
  switch (ct == 8 ? Cout : -1) {                               \
    case 8: AS_C3D(S_, 8, 8, NR_); break;                      \
    case 16: AS_C3D(S_, 8, 16, NR_); break;                    \
    case 32: AS_C3D(S_, 8, 32, NR_); break;                    \
    case 48: AS_C3D(S_, 8, 48, NR_); break;                    \
    case -1: AS_C3D(S_, 1, 0, NR_); break;                     \
    default: AS_C3D(S_, 8, 0, NR_); break;                     \
  }
  if (stride == 1) { AS_C3D_CO(1, 2) } else { AS_C3D_CO(2, 1) }
#undef AS_C3D_CO
#undef AS_C3D
  return as::check_launch("conv3d_k3");
}

int as_deconv3d_k4s2(const float* x, const float* wpack, const float* bias, float* out,
                     int B, int Cin, int Cout, int D, int H, int W, int act, void* stream) {
  AS_REQUIRE(x && wpack && out, AS_ERR_BAD_ARG, "deconv3d_k4s2: null pointer");
  AS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "deconv3d_k4s2: non-positive size");
  AS_REQUIRE(act >= AS_ACT_NONE && act <= AS_ACT_LEAKY, AS_ERR_BAD_ARG, "deconv3d_k4s2: act=%d", act);
  AS_REQUIRE((long long)Cin * D * H * W * 4 < (long long)kOOB, AS_ERR_BAD_SHAPE, "deconv3d_k4s2: input exceeds 1.75 GiB per batch element");
  AS_REQUIRE((reinterpret_cast<uintptr_t>(out) & 7) == 0, AS_ERR_BAD_ARG, "deconv3d_k4s2: out not 8-B aligned");
  const int ct = (Cout % 8 == 0) ? 8 : 1;
  const long long gz = (long long)B * 2 * D * (Cout / ct);
  AS_REQUIRE(gz <= 65535, AS_ERR_BAD_SHAPE, "deconv3d_k4s2: B*Do*groups=%lld exceeds the grid limit", gz);
  const dim3 grid((unsigned)as::cdiv(W, kC3dCols), (unsigned)as::cdiv(2 * H, 4), (unsigned)gz);
  hipStream_t s = as::as_stream(stream);
#define AS_D3D(CT_, CO_) hipLaunchKernelGGL((deconv3d_k4s2_kernel<CT_, CO_>), grid, dim3(256), 0, s, x, wpack, bias, out, Cin, Cout, D, H, W, act)
  switch (ct == 8 ? Cout : -1) {
    case 8: AS_D3D(8, 8); break;
    case 16: AS_D3D(8, 16); break;
    case 32: AS_D3D(8, 32); break;
    case -1: AS_D3D(1, 0); break;
    default: AS_D3D(8, 0); break;
  }
#undef AS_D3D
  return as::check_launch("deconv3d_k4s2");
}

int64_t as_instance_norm_ws_bytes(int planes) { return planes > 0 ? (int64_t)planes * kInSeg * 2 * (int64_t)sizeof(double) : 0; }

int as_instance_norm_act(const float* x, const float* residual, float* out, void* ws, int planes, int64_t HW, float eps, int act,
                         void* stream) {
  AS_REQUIRE(x && out && ws, AS_ERR_BAD_ARG, "instance_norm: null pointer");
  AS_REQUIRE(planes > 0 && planes <= 65535 && HW > 0, AS_ERR_BAD_ARG, "instance_norm: planes=%d HW=%lld", planes, (long long)HW);
  AS_REQUIRE(act >= AS_ACT_NONE && act <= AS_ACT_GELU, AS_ERR_BAD_ARG, "instance_norm: act=%d", act);
  AS_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 7) == 0, AS_ERR_BAD_ARG, "instance_norm: ws not 8-B aligned");
  hipStream_t s = as::as_stream(stream);
  hipLaunchKernelGGL(in_stats_kernel, dim3(kInSeg, (unsigned)planes), dim3(256), 0, s, x, (double*)ws, (long long)HW);
  hipLaunchKernelGGL(in_apply_kernel, dim3((unsigned)as::cdiv64(HW, 256), (unsigned)planes), dim3(256), 0, s, x, (const double*)ws, residual, out,
                     (long long)HW, eps, act);
  return as::check_launch("instance_norm_act");
}

int as_layernorm2d_act(const float* x, const float* weight, const float* bias, float* out, int B, int C, int H, int W, float eps,
                       int act, void* stream) {
  AS_REQUIRE(x && weight && bias && out, AS_ERR_BAD_ARG, "layernorm2d: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && C <= 64 && H > 0 && W > 0, AS_ERR_BAD_ARG, "layernorm2d: B=%d C=%d (C <= 64) H=%d W=%d", B, C, H, W);
  AS_REQUIRE(act >= AS_ACT_NONE && act <= AS_ACT_GELU, AS_ERR_BAD_ARG, "layernorm2d: act=%d", act);
  const long long HW = (long long)H * W, P = HW * B;
  const dim3 grid((unsigned)as::cdiv64(P, 256));
  hipStream_t s = as::as_stream(stream);
  if (C <= 32) hipLaunchKernelGGL(layernorm2d_kernel<32>, grid, dim3(256), 0, s, x, weight, bias, out, C, HW, P, eps, act);
  else hipLaunchKernelGGL(layernorm2d_kernel<64>, grid, dim3(256), 0, s, x, weight, bias, out, C, HW, P, eps, act);
  return as::check_launch("layernorm2d_act");
}

}  // extern "C"
