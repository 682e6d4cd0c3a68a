// a3: fused multi-level geometry/correlation pyramid lookup (forward + backward), and
// a18: the `corr_sampler` forward/backward contract as first-class HIP.
//
// HBM-bound gathers.  Mapping: one lane per 1/4-res pixel (consecutive lanes = consecutive x, so
// every store of the NCHW output is a 256-B coalesced wave store); blockIdx.y enumerates the
// independent (level, channel-quad | corr) tasks of a pixel so that a 960x540 frame (32 640
// pixels) still launches ~3000 waves.  A pixel's (2r+2)-tap window is loaded into registers
// once (one float4 per tap covers 4 geo channels in the [B,H,W,D,G] layout) and reused by
// all 2r+1 interpolated taps: compulsory traffic only, no zero-init pass on the output.
#include "common.h"

namespace {

struct LookupParams {
  const float* geo[AS_MAX_LEVELS];
  const float* corr[AS_MAX_LEVELS];
  float* dgeo[AS_MAX_LEVELS];
  float* dcorr[AS_MAX_LEVELS];
  const float* disp;
  const float* dout;
  float* out;
  int B, H, W, W2, D, G, L, radius;
  int HW, CH;
  int accum;  // lookup_bwd_kernel: add the windows to the gradients instead of storing them (one buffer for all GRU iterations)
  long long P;
  int geo_bytes[AS_MAX_LEVELS];
  int corr_bytes[AS_MAX_LEVELS];
};

using f32x4 = __attribute__((ext_vector_type(4))) float;

// position of tap k (kk = k - R) given the level-scaled base; returns window slot selection.
//   xk = xbase + kk (one fp32 add, as the reference's `dx + disp/2**i`, geometry.py:43,52)
//   result = (1-t)*w[k] + t*w[k+1], except when the add rounded xk up to the next integer
//   (then floor(xk) = i0+kk+1, t = 0 and the sample is exactly w[k+1]).
__device__ __forceinline__ void tap_weights(float xbase, int i0, int kk, float& t, bool& bump) {
  const float xk = xbase + (float)kk;
  const float fk = floorf(xk);
  t = xk - fk;
  bump = ((int)fk != i0 + kk);
}

template <int R>
__global__ __launch_bounds__(256) void lookup_fwd_kernel(LookupParams p) {
  constexpr int K = 2 * R + 1;
  constexpr int NW = 2 * R + 2;
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= p.P) return;
  const int gq = p.G >> 2;
  const int tpl = gq + 1;
  const int level = blockIdx.y / tpl;
  const int sub = blockIdx.y - level * tpl;
  const int b = (int)(pix / p.HW);
  const int rem = (int)(pix - (long long)b * p.HW);
  const int x = rem % p.W;
  const float ds = ldexpf(p.disp[pix], -level);  // disp / 2**i (exact)
  float* outp = p.out + ((long long)b * p.CH + (long long)level * K * (p.G + 1)) * p.HW + rem;

  if (sub < gq) {
    const int Dl = p.D >> level;
    const int i0 = (int)floorf(ds);
    // raw buffer loads with a sentinel offset for taps outside [0, Dl): the hardware range check
    // returns 0, so all 2r+2 window loads issue back to back with no branch / wait in between
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.geo[level], 0, p.geo_bytes[level], 0x00020000);
    const unsigned rowoff = (unsigned)(((pix * Dl) * p.G + 4 * sub) * 4);
    const unsigned dstride = (unsigned)(p.G * 4);
    f32x4 w[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int dd = i0 - R + j;
      const unsigned off = (dd >= 0 && dd < Dl) ? rowoff + (unsigned)dd * dstride : 0x7FFFFFF0u;
      w[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
    }
    float* o = outp + (long long)(4 * sub) * K * p.HW;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float t;
      bool bump;
      tap_weights(ds, i0, k - R, t, bump);
      const float a = bump ? 0.f : 1.f - t, c = bump ? 1.f : t;
      const f32x4 v = a * w[k] + c * w[k + 1];
      o[(long long)(0 * K + k) * p.HW] = v.x;
      o[(long long)(1 * K + k) * p.HW] = v.y;
      o[(long long)(2 * K + k) * p.HW] = v.z;
      o[(long long)(3 * K + k) * p.HW] = v.w;
    }
  } else {
    const int Wl = p.W2 >> level;
    const float xs = ldexpf((float)x, -level);
    const float xb = xs - ds;  // coords/2**i - disp/2**i
    const int i0 = (int)floorf(xb);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.corr[level], 0, p.corr_bytes[level], 0x00020000);
    const unsigned rowoff = (unsigned)(pix * Wl * 4);
    float w[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int dd = i0 - R + j;
      const unsigned off = (dd >= 0 && dd < Wl) ? rowoff + (unsigned)dd * 4u : 0x7FFFFFF0u;
      w[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
    }
    float* o = outp + (long long)p.G * K * p.HW;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float t;
      bool bump;
      tap_weights(xb, i0, k - R, t, bump);
      const float a = bump ? 0.f : 1.f - t, c = bump ? 1.f : t;
      o[(long long)k * p.HW] = a * w[k] + c * w[k + 1];
    }
  }
}

// ---- cooperative, LDS-staged forward (the shipped path for r=4, G in {0, 8}) -----------------------
// PMC on the lane-per-pixel kernel above shows compulsory traffic (19 MB fetched, 20.6 MB written at
// 960x540) but 64 distinct cache lines per wave load: it is request-bound, not byte-bound.  Here a block
// owns 64 consecutive pixels.  Work items are (pixel, tap, channel quad) in MEMORY order, so consecutive
// lanes read consecutive 16-B pieces of a pixel's contiguous window run (288 B for 9 taps x 8 channels)
// — a wave load touches ~7 runs instead of 64 — and each item reads taps t and t+1 (the second is an
// L1 hit of its neighbour's first).  Results are transposed through an LDS tile [channel][pixel] and
// leave as full 256-B coalesced NCHW rows.  Interpolation arithmetic is identical to the kernel above.
template <int R, int G>
__global__ __launch_bounds__(256) void lookup_fwd_coop_kernel(LookupParams p) {
  constexpr int K = 2 * R + 1;
  constexpr int NQ = G / 4;
  constexpr int PX = 64;
  constexpr int TS = PX + 1;  // tile row stride (floats)
  extern __shared__ float sm[];
  float* sdisp = sm;            // [PX]
  float* tile = sm + PX;        // [CH][TS]
  const int tid = threadIdx.x;
  const long long pix0 = (long long)blockIdx.x * PX;
  if (tid < PX) sdisp[tid] = (pix0 + tid < p.P) ? p.disp[pix0 + tid] : 0.f;
  __syncthreads();
  const unsigned kOOB = 0x7FFFFFF0u;
  constexpr int LV = 2;  // levels gathered per pass: every window load of both levels (and both volumes) is in
                         // flight before the first one is consumed — one memory round trip per pass, not four
  constexpr int GITEMS = PX * K * (NQ > 0 ? NQ : 1);
  constexpr int GNIT = NQ > 0 ? (GITEMS + 255) / 256 : 1;
  constexpr int CITEMS = PX * K;
  constexpr int CNIT = (CITEMS + 255) / 256;

  for (int level0 = 0; level0 < p.L; level0 += LV) {
    f32x4 g0[LV][GNIT], g1[LV][GNIT];
    float c0[LV][CNIT], c1[LV][CNIT];
#pragma unroll
    for (int lv = 0; lv < LV; ++lv) {
      const int level = level0 + lv;
      if (level < p.L) {
        if constexpr (G > 0) {
          const int Dl = p.D >> level;
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.geo[level], 0, p.geo_bytes[level], 0x00020000);
#pragma unroll
          for (int it = 0; it < GNIT; ++it) {
            const int idx = tid + it * 256;
            const int px = idx / (K * NQ), rem = idx - px * (K * NQ);
            const int tap = rem / NQ, q = rem - tap * NQ;
            const bool live = idx < GITEMS && pix0 + px < p.P;
            const float ds = ldexpf(sdisp[px < PX ? px : 0], -level);
            const int dd = (int)floorf(ds) - R + tap;
            const unsigned rowoff = (unsigned)((((pix0 + px) * Dl) * G + 4 * q) * 4);
            const unsigned o0 = (live && dd >= 0 && dd < Dl) ? rowoff + (unsigned)dd * (G * 4) : kOOB;
            const unsigned o1 = (live && dd + 1 >= 0 && dd + 1 < Dl) ? rowoff + (unsigned)(dd + 1) * (G * 4) : kOOB;
            g0[lv][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o0, 0, 0));
            g1[lv][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o1, 0, 0));
          }
        }
        const int Wl = p.W2 >> level;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)p.corr[level], 0, p.corr_bytes[level], 0x00020000);
#pragma unroll
        for (int it = 0; it < CNIT; ++it) {
          const int idx = tid + it * 256;
          const int px = idx / K, tap = idx - px * K;
          const bool live = idx < CITEMS && pix0 + px < p.P;
          const long long pix = pix0 + px;
          const int x = (int)((pix % p.HW) % p.W);
          const float ds = ldexpf(sdisp[px < PX ? px : 0], -level);
          const float xb = ldexpf((float)x, -level) - ds;
          const int dd = (int)floorf(xb) - R + tap;
          const unsigned rowoff = (unsigned)(pix * Wl * 4);
          const unsigned o0 = (live && dd >= 0 && dd < Wl) ? rowoff + (unsigned)dd * 4u : kOOB;
          const unsigned o1 = (live && dd + 1 >= 0 && dd + 1 < Wl) ? rowoff + (unsigned)(dd + 1) * 4u : kOOB;
          c0[lv][it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)o0, 0, 0));
          c1[lv][it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)o1, 0, 0));
        }
      }
    }
#pragma unroll
    for (int lv = 0; lv < LV; ++lv) {
      const int level = level0 + lv;
      if (level < p.L) {
        const int chbase = level * K * (G + 1);
        if constexpr (G > 0) {
#pragma unroll
          for (int it = 0; it < GNIT; ++it) {
            const int idx = tid + it * 256;
            if (idx < GITEMS) {
              const int px = idx / (K * NQ), rem = idx - px * (K * NQ);
              const int tap = rem / NQ, q = rem - tap * NQ;
              const float ds = ldexpf(sdisp[px], -level);
              float tt;
              bool bm;
              tap_weights(ds, (int)floorf(ds), tap - R, tt, bm);
              const float a = bm ? 0.f : 1.f - tt, c = bm ? 1.f : tt;
              const f32x4 v = a * g0[lv][it] + c * g1[lv][it];
              float* t = tile + (chbase + (4 * q) * K + tap) * TS + px;
              t[0] = v.x;
              t[K * TS] = v.y;
              t[2 * K * TS] = v.z;
              t[3 * K * TS] = v.w;
            }
          }
        }
#pragma unroll
        for (int it = 0; it < CNIT; ++it) {
          const int idx = tid + it * 256;
          if (idx < CITEMS) {
            const int px = idx / K, tap = idx - px * K;
            const int x = (int)(((pix0 + px) % p.HW) % p.W);
            const float ds = ldexpf(sdisp[px], -level);
            const float xb = ldexpf((float)x, -level) - ds;
            float tt;
            bool bm;
            tap_weights(xb, (int)floorf(xb), tap - R, tt, bm);
            const float a = bm ? 0.f : 1.f - tt, c = bm ? 1.f : tt;
            tile[(chbase + G * K + tap) * TS + px] = a * c0[lv][it] + c * c1[lv][it];
          }
        }
      }
    }
  }
  __syncthreads();
  // coalesced NCHW rows: lane = pixel, 4 channel rows per pass
  const int px = tid & 63;
  const long long pix = pix0 + px;
  if (pix < p.P) {
    const int b = (int)(pix / p.HW);
    const int rem = (int)(pix - (long long)b * p.HW);
    const int ch0 = tid >> 6;
    float* o = p.out + ((long long)b * p.CH + ch0) * p.HW + rem;
    const float* t = tile + ch0 * TS + px;
    const long long step = 4ll * p.HW;
    for (int ch = ch0; ch < p.CH; ch += 4, o += step, t += 4 * TS) *o = *t;
  }
}

// ---- quad-per-pixel forward (the shipped path for r = 4 with (G, L) = (8, 2) or (0, 4)) -------------------
// Ablation of the cooperative kernel above at 960x540 (14.9 us): launch + disparity 3.5 us, gather phase 6.2 us of
// which 5.2 us remain with every load disabled — the (pixel, tap, quad) work items cost ~150 VALU instructions each
// in index arithmetic (64-bit row offsets, three divisions, tap weights evaluated twice) — and 5.2 us in a store
// loop that exposed one LDS latency per channel.  Here four lanes share a pixel: lane (px, sub) owns ONE window —
// (level, channel quad) of the geometry volume, i.e. 10 consecutive taps x 4 channels = ten 16-B loads 32 B apart
// (the sibling quad's lane reads the interleaved other half of the same 320-B run), plus, for quad 0, the 10-tap
// correlation window of that level — so floor/frac and the row offset are computed once per lane, all loads are in
// flight together, and the 36 (+9) results go to the LDS tile [channel][pixel]; the tile leaves as 256-B NCHW rows,
// eight LDS reads in flight per lane.  Interpolation arithmetic is identical to the kernels above.
// gather + interpolation of a block's 64 pixels into the LDS tile [CH][65] (shared by the plain and the fused kernel)
template <int G>
__device__ __forceinline__ void quad_fill_tile(const LookupParams& p, float* tile, int bid, bool have_d = false, float d_in = 0.f) {
  constexpr int R = 4, K = 9, NW = 10;
  constexpr int PX = 64, TS = PX + 1;
  const int tid = threadIdx.x;
  const int px = tid >> 2, sub = tid & 3;
  const long long pix0 = (long long)bid * PX;
  const long long pix = pix0 + px;
  const bool live = pix < p.P;
  const int level = G ? (sub >> 1) : sub;
  const int q = G ? (sub & 1) : 0;
  const unsigned kOOB = 0x7FFFFFF0u;
  const float d0 = have_d ? d_in : (pix < p.P ? p.disp[pix] : 0.f);
  const float ds = ldexpf(d0, -level);  // disp / 2**level (exact)
  f32x4 gw[NW];
  float cw[NW];
  float xb = 0.f;
  int ci0 = 0;
  if (G) {
    const int Dl = p.D >> level;
    const int i0 = (int)floorf(ds);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.geo[level], 0, p.geo_bytes[level], 0x00020000);
    const unsigned rowoff = ((unsigned)pix * (unsigned)Dl * G + 4u * q) * 4u;  // < 2^31 (host check)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int dd = i0 - R + j;
      const unsigned off = (live && dd >= 0 && dd < Dl) ? rowoff + (unsigned)dd * (G * 4) : kOOB;
      gw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
    }
  }
  const bool do_corr = (q == 0);
  {
    const int Wl = p.W2 >> level;
    const int x = (int)((pix % p.HW) % p.W);
    xb = ldexpf((float)x, -level) - ds;  // coords/2**i - disp/2**i
    ci0 = (int)floorf(xb);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)p.corr[level], 0, p.corr_bytes[level], 0x00020000);
    const unsigned rowoff = (unsigned)pix * (unsigned)Wl * 4u;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int dd = ci0 - R + j;
      const unsigned off = (live && do_corr && dd >= 0 && dd < Wl) ? rowoff + (unsigned)dd * 4u : kOOB;
      cw[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)off, 0, 0));
    }
  }
  const int chbase = level * K * (G + 1);
  if (G) {
    const int i0 = (int)floorf(ds);
    float* t = tile + (chbase + (4 * q) * K) * TS + px;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float tt;
      bool bump;
      tap_weights(ds, i0, k - R, tt, bump);
      const float a = bump ? 0.f : 1.f - tt, c = bump ? 1.f : tt;
      const f32x4 v = a * gw[k] + c * gw[k + 1];
      t[(0 * K + k) * TS] = v.x;
      t[(1 * K + k) * TS] = v.y;
      t[(2 * K + k) * TS] = v.z;
      t[(3 * K + k) * TS] = v.w;
    }
  }
  if (do_corr) {
    float* t = tile + (chbase + G * K) * TS + px;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float tt;
      bool bump;
      tap_weights(xb, ci0, k - R, tt, bump);
      const float a = bump ? 0.f : 1.f - tt, c = bump ? 1.f : tt;
      t[k * TS] = a * cw[k] + c * cw[k + 1];
    }
  }
}

template <int G>
__global__ __launch_bounds__(256) void lookup_fwd_quad_kernel(LookupParams p) {
  constexpr int PX = 64, TS = PX + 1;
  extern __shared__ float tile[];  // [CH][TS]
  const int tid = threadIdx.x;
  const long long pix0 = (long long)blockIdx.x * PX;
  quad_fill_tile<G>(p, tile, blockIdx.x);
  __syncthreads();
  // coalesced NCHW rows: lane = pixel, wave w takes channels w, w+4, ...; 8 LDS reads in flight per lane
  const int lx = tid & 63;
  const long long opix = pix0 + lx;
  if (opix < p.P) {
    const int b = (int)(opix / p.HW);
    const int rem = (int)(opix - (long long)b * p.HW);
    const int ch0 = tid >> 6;
    float* o = p.out + ((long long)b * p.CH + ch0) * p.HW + rem;
    const float* t = tile + ch0 * TS + lx;
    const long long step = 4ll * p.HW;
    int ch = ch0;
    for (; ch + 28 < p.CH; ch += 32, o += 8 * step, t += 32 * TS) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = t[i * 4 * TS];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i * step] = v[i];
    }
    for (; ch < p.CH; ch += 4, o += step, t += 4 * TS) *o = *t;
  }
}

// ---- lookup fused with the motion encoder's first correlation conv (update.py:84-85: relu(convc1(corr))) -------------------
// The [B, L*9*(G+1), h, w] lookup result (21 MB at 960x540, written and re-read every GRU iteration) never leaves the CU:
// the quad kernel's LDS tile [CH][64 px] is the B operand of a 1x1 convolution CH -> 64 on the matrix cores (split
// precision: 3 x v_mfma_f32_32x32x16_f16 per product), bias + ReLU in the epilogue, and the result goes out as a blocked
// split-fp16 link tensor (conv.hip, `out_bs`) — what the next convolution's loaders stage by LDS-DMA — and / or as fp32 NCHW.
// Wave w: output channels [32 (w&1), +32) x pixels [32 (w>>1), +32); its 22 weight fragments (16 B per lane, packed by
// frag_pack_kernel) are fetched at kernel entry, so they are in flight during the gather.
using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4v = __attribute__((ext_vector_type(4))) _Float16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

struct FusedParams {
  const _Float16* wimg;  // [ks][mt(2)][hi|lo][lane][8]
  const float* bias;     // [64] or null
  _Float16* out_bs;      // blocked split-fp16 tensor [B][2][cb_tot][H][W][8] or null
  int cb_tot, cb_off;    // 8-channel blocks of that tensor, first block of this result
  float* out_f32;        // [B][64][H][W] or null
  int relu;
};

__device__ unsigned g_split_overflow_lookup;

template <int G>
__device__ __forceinline__ void lookup_convc1_body(const LookupParams& p, const FusedParams& f, float* tile, int bid, bool have_d, float d_in) {
  constexpr int PX = 64, TS = PX + 1;
  constexpr int CH = (G ? 2 : 4) * 9 * (G + 1);
  constexpr int KS = (CH + 15) / 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave & 1, nt = wave >> 1;
  const int c = lane & 31, half = lane >> 5;
  half8 ah[KS], al[KS];
  {
    const half8* wi = reinterpret_cast<const half8*>(f.wimg);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      ah[ks] = wi[((ks * 2 + mt) * 2) * 64 + lane];
      al[ks] = wi[((ks * 2 + mt) * 2 + 1) * 64 + lane];
    }
  }
  for (int i = tid; i < (KS * 16 - CH) * TS; i += 256) tile[CH * TS + i] = 0.f;  // padded channels of the last k-step
  quad_fill_tile<G>(p, tile, bid, have_d, d_in);
  __syncthreads();
  f32x16 acc_h, acc_x;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc_h[i] = f.bias ? f.bias[32 * mt + (i & 3) + 8 * (i >> 2) + 4 * half] : 0.f;
    acc_x[i] = 0.f;
  }
  float amax = 0.f;
  const float* tp = tile + 8 * half * TS + 32 * nt + c;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tp[(16 * ks + j) * TS];
    half8 bh, bl;
#pragma unroll
    for (int j = 0; j < 8; j += 2) amax = fmaxf(amax, fmaxf(fabsf(v[j]), fabsf(v[j + 1])));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = __builtin_amdgcn_fmed3f(v[j], -65504.f, 65504.f);
      const _Float16 hk = (_Float16)x;
      bh[j] = hk;
      bl[j] = (_Float16)((x - (float)hk) * 2048.f);
    }
    acc_h = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, acc_h, 0, 0, 0);
    acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, acc_x, 0, 0, 0);
    acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, acc_x, 0, 0, 0);
  }
  const long long pix = (long long)bid * PX + 32 * nt + c;
  if (pix < p.P) {
    const int b = (int)(pix / p.HW);
    const int rem = (int)(pix - (long long)b * p.HW);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float x = acc_h[4 * qd + k] + acc_x[4 * qd + k] * (1.f / 2048.f);
        o[k] = f.relu ? fmaxf(x, 0.f) : x;
      }
      const int ch = 32 * mt + 8 * qd + 4 * half;  // this lane's four channels of block (4 mt + qd)
      if (f.out_f32) {
        float* op = f.out_f32 + ((long long)b * 64 + ch) * p.HW + rem;
#pragma unroll
        for (int k = 0; k < 4; ++k) op[(long long)k * p.HW] = o[k];
      }
      if (f.out_bs) {
        half4v hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          amax = fmaxf(amax, fabsf(o[k]));
          const float x = __builtin_amdgcn_fmed3f(o[k], -65504.f, 65504.f);
          const _Float16 hk = (_Float16)x;
          hi[k] = hk;
          lo[k] = (_Float16)((x - (float)hk) * 2048.f);
        }
        const long long blk = f.cb_off + 4 * mt + qd;
        const long long e_hi = ((((long long)b * 2 + 0) * f.cb_tot + blk) * p.HW + rem) * 8 + 4 * half;
        const long long e_lo = ((((long long)b * 2 + 1) * f.cb_tot + blk) * p.HW + rem) * 8 + 4 * half;
        *reinterpret_cast<u32x2*>(f.out_bs + e_hi) = __builtin_bit_cast(u32x2, hi);
        *reinterpret_cast<u32x2*>(f.out_bs + e_lo) = __builtin_bit_cast(u32x2, lo);
      }
    }
  }
  if (__builtin_amdgcn_ballot_w64(!(amax < 65504.f)) != 0ull && lane == 0) atomicAdd(&g_split_overflow_lookup, 1u);
}


template <int G>
__global__ __launch_bounds__(256, 2) void lookup_convc1_kernel(LookupParams p, FusedParams f) {
  extern __shared__ float tile[];  // [KS*16][65]
  lookup_convc1_body<G>(p, f, tile, blockIdx.x, false, 0.f);
}

// ---- register-direct form of the same fusion (IGEV geometry: G = 8, L = 2, r = 4) ------------------------------------------------
// The MFMA's K order is free (the weight image is packed to match), so it is chosen such that every B fragment is made of values
// the lane interpolated ITSELF — no LDS tile, no transpose, no barrier between gather and matrix product:
//   wave = 32 pixels; lane (n = l & 31, h = l >> 5) owns pixel n, geometry channels 4h .. 4h+3 of BOTH levels, and the
//   correlation window of level h.  Its loads: 2 levels x 10 taps x one 16-B unit (lanes n and n+32 read the two halves of the same
//   32-B tap record: every wave load is 64 live lanes on 32 lines, descriptor wave-uniform) + 10 dwords of correlation.
//   k-step ks < 9  = tap ks:      B[j] = level (j >> 2), channel 4h + (j & 3), tap ks          (8 values = two float4 lerps)
//   k-step 9, 10   = correlation: B[j] = level h, tap 8 (ks - 9) + j                            (tap 8 alone in k-step 10)
// A = convc1's weights, permuted likewise by frag_pack_direct_kernel, resident in LDS as fragments (44 KB, one LDS-DMA copy per
// block instead of 22 KB of global fragment loads per WAVE: the old form pulled 180 KB of weights + 92 KB of windows through a
// CU's L1 per 128 pixels, this one 44 + 92).  Each wave computes all 64 output channels of its 32 pixels (two 32-row tiles
// share every B fragment): 11 k-steps x 2 tiles x 3 MFMAs.  Interpolation arithmetic = quad_fill_tile's (same tap_weights).
constexpr int kDirectKS = 11;
constexpr int kDirectImgBytes = kDirectKS * 2 * 2 * 1024;  // [ks][tile][hi|lo][lane][8] fp16

__global__ __launch_bounds__(256) void frag_pack_direct_kernel(const float* w, _Float16* img) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= kDirectKS * 2 * 2 * 512) return;
  const int blk = idx >> 9, lane = (idx >> 3) & 63, j = idx & 7;
  const int hl = blk & 1, t = (blk >> 1) & 1, ks = blk >> 2;
  const int h = lane >> 5, co = 32 * t + (lane & 31);
  int chan = -1;
  if (ks < 9) chan = (j >> 2) * 81 + (4 * h + (j & 3)) * 9 + ks;
  else if (ks == 9) chan = h * 81 + 72 + j;
  else if (j == 0) chan = h * 81 + 72 + 8;
  const float v = chan >= 0 ? w[(long long)co * 162 + chan] : 0.f;
  const float x = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  const _Float16 hk = (_Float16)x;
  img[idx] = hl == 0 ? hk : (_Float16)((x - (float)hk) * 2048.f);
}

typedef __attribute__((address_space(3))) void lk_lds_void;

template <int NWAVE>
__global__ __launch_bounds__(NWAVE * 64, 2) void lookup_convc1_direct_kernel(LookupParams p, FusedParams f, const _Float16* __restrict__ wimg) {
  constexpr int R = 4, K = 9, NW = 10, G = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];  // kDirectImgBytes
  as::fp16_saturate_mode();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  // weight image -> LDS by LDS-DMA (1 KB per wave instruction), in flight during the gather
  {
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wimg, 0, kDirectImgBytes, 0x00020000);
    constexpr int PIECES = kDirectImgBytes / 1024;  // 44
#pragma unroll
    for (int g = 0; g < (PIECES + NWAVE - 1) / NWAVE; ++g) {
      const int piece = g * NWAVE + wave;
      const unsigned vo_ = (unsigned)lane * 16u, so_ = (unsigned)piece * 1024u;
      if (piece < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lk_lds_void*)(wlds + piece * 1024), 16, vo_, so_, 0, 0);
    }
  }
  const long long pix = ((long long)blockIdx.x * NWAVE + wave) * 32 + n;
  const bool live = pix < p.P;
  const unsigned kOOB = 0x7FFFFFF0u;
  const float d0 = live ? p.disp[pix] : 0.f;
  f32x4 gw[2][NW];
  float cw[NW];
  float dsl[2];
  int i0l[2];
#pragma unroll
  for (int lv = 0; lv < 2; ++lv) {
    const int Dl = p.D >> lv;
    dsl[lv] = ldexpf(d0, -lv);  // disp / 2**level (exact)
    i0l[lv] = (int)floorf(dsl[lv]);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.geo[lv], 0, p.geo_bytes[lv], 0x00020000);
    const unsigned rowoff = ((unsigned)pix * (unsigned)Dl * G + 4u * h) * 4u;  // < 2^31 (host check)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int dd = i0l[lv] - R + j;
      const unsigned off = (live && dd >= 0 && dd < Dl) ? rowoff + (unsigned)dd * (G * 4) : kOOB;
      gw[lv][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
    }
  }
  // correlation window of level h.  Every lane issues the loads of BOTH levels through wave-uniform descriptors, with the
  // out-of-range sentinel for the level it does not own (no fetch, reads 0), and adds the two (x + 0 = x): a per-lane choice of
  // descriptor would make the compiler wrap each load in a waterfall loop.
  const int x = (int)((pix % p.HW) % p.W);
  const float xb = ldexpf((float)x, -h) - (h ? dsl[1] : dsl[0]);  // coords/2**i - disp/2**i
  const int ci0 = (int)floorf(xb);
  {
    const int Wl = p.W2 >> h;
    const unsigned rowoff = (unsigned)pix * (unsigned)Wl * 4u;
    float c01[2][NW];
#pragma unroll
    for (int lv = 0; lv < 2; ++lv) {
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)p.corr[lv], 0, p.corr_bytes[lv], 0x00020000);
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int dd = ci0 - R + j;
        const unsigned off = (live && h == lv && dd >= 0 && dd < Wl) ? rowoff + (unsigned)dd * 4u : kOOB;
        c01[lv][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (int)off, 0, 0));
      }
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) cw[j] = c01[0][j] + c01[1][j];
  }
  f32x16 acc_h[2], acc_x[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc_h[t][i] = f.bias ? f.bias[32 * t + (i & 3) + 8 * (i >> 2) + 4 * h] : 0.f;
      acc_x[t][i] = 0.f;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // the weight image has landed (every wave waited for its own pieces)
  float amax = 0.f;
  const half8* wfr = reinterpret_cast<const half8*>(wlds) + lane;
#ifdef AS_LK_TRAFFIC_ONLY
  // Diagnostic build only (tools/variant.sh lookup.hip lk_traffic -DAS_LK_TRAFFIC_ONLY; never defined in the product build): the
  // kernel's loads and stores on the same grid and nothing else — no interpolation, no operand split, no matrix instruction, one
  // LDS read so the weight image's DMA is still waited for.  Every loaded value reaches a stored one through a sum (results are
  // wrong); its time is the floor any arithmetic of this kernel sits on.
  {
    f32x4 s4 = gw[0][0];
#pragma unroll
    for (int lv = 0; lv < 2; ++lv)
#pragma unroll
      for (int j = 0; j < NW; ++j)
        if (lv || j) s4 += gw[lv][j];
    float s1 = s4.x + s4.y + s4.z + s4.w;
#pragma unroll
    for (int j = 0; j < NW; ++j) s1 += cw[j];
    const half8 a0 = wfr[0];
    s1 += (float)a0[0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc_h[t][i] += s1;
  }
#else
#pragma unroll
  for (int ks = 0; ks < kDirectKS; ++ks) {
    float v[8];
    if (ks < K) {
#pragma unroll
      for (int lv = 0; lv < 2; ++lv) {
        float tt;
        bool bump;
        tap_weights(dsl[lv], i0l[lv], ks - R, tt, bump);
        const float a = bump ? 0.f : 1.f - tt, c = bump ? 1.f : tt;
        const f32x4 r = a * gw[lv][ks] + c * gw[lv][ks + 1];
        v[4 * lv + 0] = r.x; v[4 * lv + 1] = r.y; v[4 * lv + 2] = r.z; v[4 * lv + 3] = r.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * (ks - K) + j;  // compile-time
        if (k < K) {
          float tt;
          bool bump;
          tap_weights(xb, ci0, k - R, tt, bump);
          const float a = bump ? 0.f : 1.f - tt, c = bump ? 1.f : tt;
          v[j] = a * cw[k] + c * cw[k + 1];
        } else {
          v[j] = 0.f;
        }
      }
    }
    half8 bh, bl;
#pragma unroll
    for (int j = 0; j < 8; j += 2) amax = fmaxf(amax, fmaxf(fabsf(v[j]), fabsf(v[j + 1])));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const _Float16 hk = (_Float16)v[j];  // MODE.FP16_OVFL: |v| >= 65504 saturates (counted below), never inf / NaN
      bh[j] = hk;
      bl[j] = (_Float16)((v[j] - (float)hk) * 2048.f);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const half8 ah = wfr[((ks * 2 + t) * 2 + 0) * 64];
      const half8 al = wfr[((ks * 2 + t) * 2 + 1) * 64];
      acc_h[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc_h[t], 0, 0, 0);
      acc_x[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc_x[t], 0, 0, 0);
      acc_x[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc_x[t], 0, 0, 0);
    }
  }
#endif
  if (live) {
    const int b = (int)(pix / p.HW);
    const int rem = (int)(pix - (long long)b * p.HW);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xo = acc_h[t][4 * qd + k] + acc_x[t][4 * qd + k] * (1.f / 2048.f);
          o[k] = f.relu ? fmaxf(xo, 0.f) : xo;
        }
        const int ch = 32 * t + 8 * qd + 4 * h;  // this lane's four channels of block (4 t + qd)
        if (f.out_f32) {
          float* op = f.out_f32 + ((long long)b * 64 + ch) * p.HW + rem;
#pragma unroll
          for (int k = 0; k < 4; ++k) op[(long long)k * p.HW] = o[k];
        }
        if (f.out_bs) {
          half4v hi, lo;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            amax = fmaxf(amax, fabsf(o[k]));
            const _Float16 hk = (_Float16)o[k];
            hi[k] = hk;
            lo[k] = (_Float16)((o[k] - (float)hk) * 2048.f);
          }
          const long long blk = f.cb_off + 4 * t + qd;
          const long long e_hi = ((((long long)b * 2 + 0) * f.cb_tot + blk) * p.HW + rem) * 8 + 4 * h;
          const long long e_lo = ((((long long)b * 2 + 1) * f.cb_tot + blk) * p.HW + rem) * 8 + 4 * h;
          *reinterpret_cast<u32x2*>(f.out_bs + e_hi) = __builtin_bit_cast(u32x2, hi);
          *reinterpret_cast<u32x2*>(f.out_bs + e_lo) = __builtin_bit_cast(u32x2, lo);
        }
      }
  }
  if (__builtin_amdgcn_ballot_w64(!(amax < 65504.f)) != 0ull && lane == 0) atomicAdd(&g_split_overflow_lookup, 1u);
}

// ---- the front of a GRU iteration in ONE launch -------------------------------------------------------------------------------
// After the disparity head's first conv (as_conv2d, AS_EPI_RELU_TAPS) three small dependent launches sat on the loop's critical
// chain: as_tap_shift_sum (disp += delta), the fused lookup + convc1, and the 7x7 conv of the disparity branch (update.py:87).
// Both consumers need only the new disparity, so they run side by side here, each block computing the disparity of the pixels
// it needs straight from the tap planes (same summation order as tap_shift_sum_kernel: bit-identical disparity):
//   blocks [0, n_lookup):  64 pixels each: disp -> (stored for the next iteration) -> lookup -> convc1 + ReLU -> blocked result
//   blocks [n_lookup, ..): 16x16 pixels x 32 channels: disp on the 22x22 halo patch -> relu(conv7x7 + bias) -> blocked result,
//                          plus the disparity pass-through channel of the motion features (update.py:91)
struct FrontParams {
  const float* taps;      // [B][groups*9][H][W] per-tap channel reductions of the head's conv2 (AS_EPI_RELU_TAPS)
  int groups;
  const float* head_bias; // [1] or null
  const float* disp_old;  // [B][1][H][W]
  float* disp_new;        // [B][1][H][W]
  const float* w7;        // tap-major [49][CP7] weights of the 7x7 conv (zero padded columns)
  const float* b7;        // [64] or null
  int CP7;
  _Float16* d1_bs;        // blocked split-fp16 result of the 7x7 conv [B][2][8][H][W][8]
  _Float16* copy_bs;      // blocked tensor that receives the new disparity in channel copy_coff (or null)
  int copy_ctot, copy_coff;
  int n_lookup, n_conv7, tiles_x, tiles_y;
};

// disp_old + (sum over groups and taps of the shifted tap planes + bias): tap_shift_sum_kernel's arithmetic for one pixel
__device__ __forceinline__ float front_disp(const FrontParams& q, __amdgpu_buffer_rsrc_t rs, unsigned boff, long long plane, int H, int W,
                                            int y, int x, float addend) {
  // rs covers the WHOLE tap tensor (wave-uniform descriptor: a per-lane one makes hipcc wrap every load in a waterfall
  // loop); boff = this pixel's batch offset in bytes
  float acc = 0.f;
  unsigned toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    toff[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? boff + (unsigned)(((long long)t * plane + (long long)yy * W + xx) * 4) : 0x7FFFFFF0u;
  }
  const unsigned gstep = (unsigned)(9 * plane * 4);
  for (int g0 = 0; g0 < q.groups; g0 += 4) {
    float v[4][9];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int t = 0; t < 9; ++t)
        v[g][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            rs, (int)((toff[t] == 0x7FFFFFF0u || g0 + g >= q.groups) ? 0x7FFFFFF0u : toff[t] + (unsigned)(g0 + g) * gstep), 0, 0));
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc += v[g][t];
  }
  const float delta = acc + (q.head_bias ? q.head_bias[0] : 0.f);
  return addend + delta;
}

template <int G>
__global__ __launch_bounds__(256, 2) void loop_front_kernel(LookupParams p, FusedParams f, FrontParams q, const float* __restrict__ w7,
                                                                 const float* __restrict__ b7) {
  extern __shared__ float tile[];
  const long long plane = p.HW;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)q.taps, 0, (int)((long long)p.B * q.groups * 9 * plane * 4), 0x00020000);
  if ((int)blockIdx.x < q.n_lookup) {
    // ---- role 1: disparity of this block's 64 pixels (the four lanes of a pixel compute the same value), lookup + convc1 ----
    const int px = threadIdx.x >> 2;
    const int lb = (int)blockIdx.x;   // the (longer) lookup blocks come first: 510 + 270 blocks on 768 slots (3 per CU) leave a
                                      // dozen of the SHORT 7x7 blocks for the second round instead of a dozen long ones
    const long long pix = (long long)lb * 64 + px;
    float d = 0.f;
    if (pix < p.P) {
      const int b = (int)(pix / plane);
      const int rem = (int)(pix - (long long)b * plane);
      d = front_disp(q, rs, (unsigned)((long long)b * q.groups * 9 * plane * 4), plane, p.H, p.W, rem / p.W, rem % p.W, q.disp_old[pix]);
      if ((threadIdx.x & 3) == 0) q.disp_new[pix] = d;
    }
    lookup_convc1_body<G>(p, f, tile, lb, true, d);
    return;
  }
  // ---- role 2: 7x7 conv of the new disparity, 16x16 pixels x 16 output channels per block ----
  constexpr int CO = 32;
  float* patch = tile;  // 22 x 22
  // w7 / b7 are separate noalias kernel arguments: as members of q the compiler cannot prove the stores of this kernel leave them
  // alone and fetches them with per-lane vector loads (1568 per thread) instead of scalar loads
  as::fp16_saturate_mode();
  const int bid = blockIdx.x - q.n_lookup;
  const int groups = 64 / CO;
  const int cg = bid % groups;
  const int t_ = bid / groups;
  const int tx = t_ % q.tiles_x, ty = (t_ / q.tiles_x) % q.tiles_y, b = t_ / (q.tiles_x * q.tiles_y);
  const int c0 = cg * CO;
  const int x0 = tx * 16, y0 = ty * 16;
  const unsigned boff = (unsigned)((long long)b * q.groups * 9 * plane * 4);
  for (int idx = threadIdx.x; idx < 22 * 22; idx += 256) {
    const int py = idx / 22, pxx = idx - py * 22;
    const int gy = y0 - 3 + py, gx = x0 - 3 + pxx;
    float v = 0.f;  // zero padding of the convolution outside the image
    if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) v = front_disp(q, rs, boff, plane, p.H, p.W, gy, gx, q.disp_old[(long long)b * plane + (long long)gy * p.W + gx]);
    patch[idx] = v;
  }
  __syncthreads();
  const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
  const int gy = y0 + ly, gx = x0 + lx;
  const bool inside = gy < p.H && gx < p.W;
  const long long pixo = (long long)gy * p.W + gx;
  float amax = 0.f;
  // 8 output channels (one blocked 16-B unit) per pass: one kernel row's 7 x 8 weights are 56 scalar registers, loaded by
  // seven s_load_dwordx8 (a wider channel group would exceed the scalar file and reload weights inside the tap loop)
#pragma unroll 1
  for (int k = 0; k < CO / 8; ++k) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
      const float* pr = patch + (ly + ky) * 22 + lx;
      const float* __restrict__ wr = w7 + (ky * 7) * q.CP7 + c0 + 8 * k;  // wave-uniform (scalar loads)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const float v = pr[kx];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, wr[kx * q.CP7 + j], acc[j]);
      }
    }
    if (inside) {
      half8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = fmaxf(acc[j] + (b7 ? b7[c0 + 8 * k + j] : 0.f), 0.f);
        amax = fmaxf(amax, v);
        const _Float16 hj = (_Float16)v;
        hi[j] = hj;
        lo[j] = (_Float16)((v - (float)hj) * 2048.f);
      }
      _Float16* rec = q.d1_bs + (((long long)b * 2 * 8 + (c0 >> 3) + k) * plane + pixo) * 8;
      *reinterpret_cast<half8*>(rec) = hi;
      *reinterpret_cast<half8*>(rec + 8ll * plane * 8) = lo;
    }
  }
  if (inside) {
    if (q.copy_bs && cg == 0) {  // the new disparity as channel copy_coff of the motion features (update.py:91)
      const float v = patch[(ly + 3) * 22 + lx + 3];
      const int c8 = (q.copy_ctot + 7) >> 3;
      _Float16* rec = q.copy_bs + (((long long)b * 2 * c8 + (q.copy_coff >> 3)) * plane + pixo) * 8 + (q.copy_coff & 7);
      const _Float16 hk = (_Float16)v;
      rec[0] = hk;
      rec[(long long)c8 * plane * 8] = (_Float16)((v - (float)hk) * 2048.f);
      amax = fmaxf(amax, fabsf(v));
    }
  }
  as::note_split_overflow(amax, &g_split_overflow_lookup);
}

// [Cout][ldw] fp32 weight columns [koff, koff+K) -> split-fp16 MFMA fragments [ks][tile of 32 rows][hi|lo][lane][8]:
// lane (r = l&31, h = l>>5) element j = W[32 tile + r][koff + 16 ks + 8 h + j] (zero beyond K)
__global__ __launch_bounds__(256) void frag_pack_kernel(const float* w, int cout, int ldw, int koff, int K, int ksteps, _Float16* img) {
  const int tiles = cout / 32;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= ksteps * tiles * 2 * 512) return;
  const int blk = idx >> 9, lane = (idx >> 3) & 63, j = idx & 7;
  const int hl = blk & 1, t = (blk >> 1) % tiles, ks = (blk >> 1) / tiles;
  const int k = 16 * ks + 8 * (lane >> 5) + j;
  const float v = k < K ? w[(long long)(32 * t + (lane & 31)) * ldw + koff + k] : 0.f;
  const float x = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  const _Float16 hk = (_Float16)x;
  img[idx] = hl == 0 ? hk : (_Float16)((x - (float)hk) * 2048.f);
}

// Backward w.r.t. the volumes: the transpose of the above.  Each (pixel, task) owns a private
// window of its pixel's row, so the accumulated window is written with plain stores into the
// caller-zeroed gradient (no atomics; same property as sampler_kernel.cu:63-104).
template <int R>
__global__ __launch_bounds__(256) void lookup_bwd_kernel(LookupParams p) {
  constexpr int K = 2 * R + 1;
  constexpr int NW = 2 * R + 2;
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= p.P) return;
  const int gq = p.G >> 2;
  const int tpl = gq + 1;
  const int level = blockIdx.y / tpl;
  const int sub = blockIdx.y - level * tpl;
  const int b = (int)(pix / p.HW);
  const int rem = (int)(pix - (long long)b * p.HW);
  const int x = rem % p.W;
  const float ds = ldexpf(p.disp[pix], -level);
  const float* gp = p.dout + ((long long)b * p.CH + (long long)level * K * (p.G + 1)) * p.HW + rem;

  if (sub < gq) {
    const int Dl = p.D >> level;
    const int i0 = (int)floorf(ds);
    float4 w[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) w[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* g = gp + (long long)(4 * sub) * K * p.HW;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float t;
      bool bump;
      tap_weights(ds, i0, k - R, t, bump);
      const float a = bump ? 0.f : 1.f - t, c = bump ? 1.f : t;
      const float gx = g[(long long)(0 * K + k) * p.HW], gy = g[(long long)(1 * K + k) * p.HW];
      const float gz = g[(long long)(2 * K + k) * p.HW], gw = g[(long long)(3 * K + k) * p.HW];
      w[k].x += a * gx; w[k].y += a * gy; w[k].z += a * gz; w[k].w += a * gw;
      w[k + 1].x += c * gx; w[k + 1].y += c * gy; w[k + 1].z += c * gz; w[k + 1].w += c * gw;
    }
    float4* row = reinterpret_cast<float4*>(p.dgeo[level] + (pix * Dl) * p.G + 4 * sub);
    const int gstride = p.G >> 2;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int dd = i0 - R + j;
      if (dd >= 0 && dd < Dl) {
        if (p.accum) {
          const float4 o = row[(long long)dd * gstride];
          w[j].x += o.x; w[j].y += o.y; w[j].z += o.z; w[j].w += o.w;
        }
        row[(long long)dd * gstride] = w[j];
      }
    }
  } else {
    const int Wl = p.W2 >> level;
    const float xb = ldexpf((float)x, -level) - ds;
    const int i0 = (int)floorf(xb);
    float w[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) w[j] = 0.f;
    const float* g = gp + (long long)p.G * K * p.HW;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float t;
      bool bump;
      tap_weights(xb, i0, k - R, t, bump);
      const float a = bump ? 0.f : 1.f - t, c = bump ? 1.f : t;
      const float gv = g[(long long)k * p.HW];
      w[k] += a * gv;
      w[k + 1] += c * gv;
    }
    float* row = p.dcorr[level] + pix * Wl;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int dd = i0 - R + j;
      if (dd >= 0 && dd < Wl) row[dd] = p.accum ? row[dd] + w[j] : w[j];
    }
  }
}

// ---- corr_sampler (a18) ------------------------------------------------------------------------
// One lane per (n,y,x); runtime radius; the window is streamed with a running `prev` value so
// each volume element is loaded once:  out[k] = s[k]*(1-dx) + s[k+1]*dx  (sampler_kernel.cu:45-59,
// products and sums rounded in the volume dtype exactly as `scalar_t(dx)` arithmetic does).
template <typename T>
__global__ __launch_bounds__(256) void sampler_fwd_kernel(const T* __restrict__ vol, const float* __restrict__ coords,
                                                          T* __restrict__ out, int H1, int W1, int W2, int r,
                                                          int cch, long long P) {
#pragma clang fp contract(off)
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= P) return;
  const int hw = H1 * W1;
  const int n = (int)(pix / hw);
  const int rem = (int)(pix - (long long)n * hw);
  const float x0 = coords[(long long)n * cch * hw + rem];
  const float fl = floorf(x0);
  const T dx = (T)(x0 - fl);
  const T omdx = (T)(1.0f - (x0 - fl));
  const int i0 = (int)fl - r;
  const T* row = vol + pix * W2;
  const int rd = 2 * r + 1;
  T* o = out + (long long)n * rd * hw + rem;
  T prev = (T)0;
  for (int i = 0; i <= rd; ++i) {
    const int x1 = i0 + i;
    const T s = (x1 >= 0 && x1 < W2) ? row[x1] : (T)0;
    if (i > 0) o[(long long)(i - 1) * hw] = prev * omdx + s * dx;
    prev = s;
  }
}

// One wave per pixel row of the gradient volume, lanes along x2: every store is a coalesced 256-B run (the first
// version gave a lane a whole row — 64 cache lines per store instruction, 31 us for the 31 MB volume at 960x540).
// The whole row is written (zeros outside the window), so volume_grad needs no memset and no atomics
// (sampler_kernel.cu:63-104 zero-fills and scatters with atomicAdd).
template <typename T>
__global__ __launch_bounds__(256) void sampler_bwd_kernel(const float* __restrict__ coords, const T* __restrict__ cg,
                                                          T* __restrict__ vg, int H1, int W1, int W2, int r, int cch,
                                                          long long P) {
#pragma clang fp contract(off)
  constexpr int PPW = 8;  // pixels per wave
  const int lane = threadIdx.x & 63;
  const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int hw = H1 * W1;
  const int rd = 2 * r + 1;
  for (int k = 0; k < PPW; ++k) {
    const long long pix = wave_id * PPW + k;
    if (pix >= P) return;
    const int n = (int)(pix / hw);
    const int rem = (int)(pix - (long long)n * hw);
    const float x0 = coords[(long long)n * cch * hw + rem];
    const float fl = floorf(x0);
    const T dx = (T)(x0 - fl);
    const T omdx = (T)(1.0f - (x0 - fl));
    const int i0 = (int)fl - r;
    T* row = vg + pix * W2;
    const T* g = cg + (long long)n * rd * hw + rem;
    for (int x1 = lane; x1 < W2; x1 += 64) {
      const int i = x1 - i0;  // window slot 0..rd
      // sampler_kernel.cu:95-101: g = corr_grad[i-1]*dx (i>0)  +  corr_grad[i]*(1-dx) (i<rd)
      T acc = (T)0;
      if (i > 0 && i <= rd) acc = acc + g[(long long)(i - 1) * hw] * dx;
      if (i >= 0 && i < rd) acc = acc + g[(long long)i * hw] * omdx;
      row[x1] = acc;
    }
  }
}

int fill_common(LookupParams& p, int B, int H, int W, int W2, int D, int G, int L, int radius) {
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && W2 > 0, AS_ERR_BAD_ARG, "lookup: non-positive size B=%d H=%d W=%d W2=%d", B, H, W, W2);
  AS_REQUIRE(L >= 1 && L <= AS_MAX_LEVELS, AS_ERR_BAD_ARG, "lookup: L=%d outside [1,%d]", L, AS_MAX_LEVELS);
  AS_REQUIRE(G >= 0 && (G % 4) == 0, AS_ERR_BAD_ARG, "lookup: G=%d must be a multiple of 4 (float4 channel quads)", G);
  AS_REQUIRE(G == 0 || D > 0, AS_ERR_BAD_ARG, "lookup: D=%d", D);
  AS_REQUIRE(radius >= 1 && radius <= 4, AS_ERR_BAD_ARG, "lookup: radius=%d outside the supported [1,4]", radius);
  AS_REQUIRE((W2 >> (L - 1)) >= 1 && (G == 0 || (D >> (L - 1)) >= 1), AS_ERR_BAD_SHAPE,
             "lookup: level %d of the pyramid is empty (W2=%d D=%d)", L - 1, W2, D);
  AS_REQUIRE((long long)B * H * W * (long long)(2 * radius + 1) * L * (G + 1) < (1ll << 40), AS_ERR_BAD_SHAPE, "lookup: too large");
  p.B = B; p.H = H; p.W = W; p.W2 = W2; p.D = D; p.G = G; p.L = L; p.radius = radius;
  p.HW = H * W;
  p.CH = L * (2 * radius + 1) * (G + 1);
  p.P = (long long)B * H * W;
  return AS_OK;
}

}  // namespace

#define LAUNCH_LOOKUP(KERNEL, p, stream)                                                     \
  do {                                                                                      \
    dim3 grid((unsigned)as::cdiv64((p).P, 256), (unsigned)((p).L * (((p).G >> 2) + 1)));     \
    switch ((p).radius) {                                                                    \
      case 1: hipLaunchKernelGGL(KERNEL<1>, grid, dim3(256), 0, as::as_stream(stream), p); break; \
      case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(256), 0, as::as_stream(stream), p); break; \
      case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(256), 0, as::as_stream(stream), p); break; \
      default: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), 0, as::as_stream(stream), p); break; \
    }                                                                                       \
  } while (0)

extern "C" {

int as_geo_corr_lookup_fwd(const float* const* geo, const float* const* corr, const float* disp, float* out,
                           int B, int H, int W, int W2, int D, int G, int L, int radius, void* stream) {
  LookupParams p{};
  int rc = fill_common(p, B, H, W, W2, D, G, L, radius);
  if (rc != AS_OK) return rc;
  AS_REQUIRE(corr && disp && out && (G == 0 || geo), AS_ERR_BAD_ARG, "lookup_fwd: null pointer");
  for (int i = 0; i < L; ++i) {
    AS_REQUIRE(corr[i] && (G == 0 || geo[i]), AS_ERR_BAD_ARG, "lookup_fwd: null level %d", i);
    p.corr[i] = corr[i];
    p.geo[i] = G ? geo[i] : nullptr;
    AS_REQUIRE(G == 0 || (reinterpret_cast<uintptr_t>(geo[i]) & 15) == 0, AS_ERR_BAD_ARG, "lookup_fwd: geo[%d] not 16-B aligned", i);
  }
  for (int i = 0; i < L; ++i) {
    const long long cb = p.P * (W2 >> i) * 4, gb = p.P * (long long)(D >> i) * G * 4;
    AS_REQUIRE(cb < 0x7FFFFFF0ll && gb < 0x7FFFFFF0ll, AS_ERR_BAD_SHAPE, "lookup_fwd: pyramid level %d exceeds 2 GiB", i);
    p.corr_bytes[i] = (int)cb;
    p.geo_bytes[i] = (int)gb;
  }
  p.disp = disp;
  p.out = out;
  // quarter-resolution maps up to ~64k pixels are latency-bound (one round of blocks): the quad kernel's short
  // instruction stream wins (12.0 vs 14.3 us at 960x540); beyond that the cooperative kernel's fully coalesced
  // window reads win (54 vs 67 us at Middlebury-F)
  if (radius == 4 && p.P <= 65536 && ((G == 8 && L == 2) || (G == 0 && L == 4))) {
    const size_t lds = (size_t)(p.CH * 65) * sizeof(float);
    const dim3 grid((unsigned)as::cdiv64(p.P, 64));
    if (G == 8) hipLaunchKernelGGL((lookup_fwd_quad_kernel<8>), grid, dim3(256), lds, as::as_stream(stream), p);
    else hipLaunchKernelGGL((lookup_fwd_quad_kernel<0>), grid, dim3(256), lds, as::as_stream(stream), p);
    return as::check_launch("geo_corr_lookup_fwd");
  }
  if (radius == 4 && (G == 8 || G == 0)) {
    const size_t lds = (size_t)(64 + p.CH * 65) * sizeof(float);
    const dim3 grid((unsigned)as::cdiv64(p.P, 64));
    if (G == 8) {
      if (lds > 64 * 1024) as::lds_opt_in((const void*)lookup_fwd_coop_kernel<4, 8>);
      hipLaunchKernelGGL((lookup_fwd_coop_kernel<4, 8>), grid, dim3(256), lds, as::as_stream(stream), p);
    } else {
      hipLaunchKernelGGL((lookup_fwd_coop_kernel<4, 0>), grid, dim3(256), lds, as::as_stream(stream), p);
    }
    return as::check_launch("geo_corr_lookup_fwd");
  }
  LAUNCH_LOOKUP(lookup_fwd_kernel, p, stream);
  return as::check_launch("geo_corr_lookup_fwd");
}

static int lookup_bwd_impl(const float* disp, const float* d_out, float* const* d_geo, float* const* d_corr, int B, int H, int W, int W2,
                           int D, int G, int L, int radius, int accum, void* stream);
int as_geo_corr_lookup_bwd(const float* disp, const float* d_out, float* const* d_geo, float* const* d_corr,
                           int B, int H, int W, int W2, int D, int G, int L, int radius, void* stream) {
  return lookup_bwd_impl(disp, d_out, d_geo, d_corr, B, H, W, W2, D, G, L, radius, 0, stream);
}
int as_geo_corr_lookup_bwd_accum(const float* disp, const float* d_out, float* const* d_geo, float* const* d_corr,
                                 int B, int H, int W, int W2, int D, int G, int L, int radius, void* stream) {
  return lookup_bwd_impl(disp, d_out, d_geo, d_corr, B, H, W, W2, D, G, L, radius, 1, stream);
}
static int lookup_bwd_impl(const float* disp, const float* d_out, float* const* d_geo, float* const* d_corr, int B, int H, int W, int W2,
                           int D, int G, int L, int radius, int accum, void* stream) {
  LookupParams p{};
  p.accum = accum;
  int rc = fill_common(p, B, H, W, W2, D, G, L, radius);
  if (rc != AS_OK) return rc;
  AS_REQUIRE(d_corr && disp && d_out && (G == 0 || d_geo), AS_ERR_BAD_ARG, "lookup_bwd: null pointer");
  for (int i = 0; i < L; ++i) {
    AS_REQUIRE(d_corr[i] && (G == 0 || d_geo[i]), AS_ERR_BAD_ARG, "lookup_bwd: null level %d", i);
    p.dcorr[i] = d_corr[i];
    p.dgeo[i] = G ? d_geo[i] : nullptr;
    AS_REQUIRE(G == 0 || (reinterpret_cast<uintptr_t>(d_geo[i]) & 15) == 0, AS_ERR_BAD_ARG, "lookup_bwd: d_geo[%d] not 16-B aligned", i);
  }
  p.disp = disp;
  p.dout = d_out;
  LAUNCH_LOOKUP(lookup_bwd_kernel, p, stream);
  return as::check_launch("geo_corr_lookup_bwd");
}

static int sampler_check(const void* a, const void* b, const void* c, int N, int H1, int W1, int W2, int radius,
                         int cch, int dtype) {
  AS_REQUIRE(a && b && c, AS_ERR_BAD_ARG, "corr_sampler: null pointer");
  AS_REQUIRE(N > 0 && H1 > 0 && W1 > 0 && W2 > 0, AS_ERR_BAD_ARG, "corr_sampler: non-positive size");
  AS_REQUIRE(radius >= 0 && radius <= 64, AS_ERR_BAD_ARG, "corr_sampler: radius=%d", radius);
  AS_REQUIRE(cch == 1 || cch == 2, AS_ERR_BAD_ARG, "corr_sampler: coords must have 1 or 2 channels, got %d", cch);
  AS_REQUIRE(dtype == AS_F32 || dtype == AS_F16 || dtype == AS_F64, AS_ERR_BAD_ARG, "corr_sampler: dtype code %d", dtype);
  AS_REQUIRE((long long)N * H1 * W1 * (long long)W2 < (1ll << 40), AS_ERR_BAD_SHAPE, "corr_sampler: too large");
  return AS_OK;
}

int as_corr_sampler_fwd(const void* volume, const float* coords, void* out, int N, int H1, int W1, int W2, int radius,
                        int coords_channels, int dtype, void* stream) {
  int rc = sampler_check(volume, coords, out, N, H1, W1, W2, radius, coords_channels, dtype);
  if (rc != AS_OK) return rc;
  const long long P = (long long)N * H1 * W1;
  dim3 grid((unsigned)as::cdiv64(P, 256));
  hipStream_t s = as::as_stream(stream);
  if (dtype == AS_F32)
    hipLaunchKernelGGL(sampler_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)volume, coords, (float*)out, H1, W1, W2, radius, coords_channels, P);
  else if (dtype == AS_F16)
    hipLaunchKernelGGL(sampler_fwd_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)volume, coords, (_Float16*)out, H1, W1, W2, radius, coords_channels, P);
  else
    hipLaunchKernelGGL(sampler_fwd_kernel<double>, grid, dim3(256), 0, s, (const double*)volume, coords, (double*)out, H1, W1, W2, radius, coords_channels, P);
  return as::check_launch("corr_sampler_fwd");
}

int as_corr_sampler_bwd(const float* coords, const void* corr_grad, void* volume_grad, int N, int H1, int W1, int W2,
                        int radius, int coords_channels, int dtype, void* stream) {
  int rc = sampler_check(coords, corr_grad, volume_grad, N, H1, W1, W2, radius, coords_channels, dtype);
  if (rc != AS_OK) return rc;
  const long long P = (long long)N * H1 * W1;
  dim3 grid((unsigned)as::cdiv64(P, 32));  // 4 waves x 8 pixels per block
  hipStream_t s = as::as_stream(stream);
  if (dtype == AS_F32)
    hipLaunchKernelGGL(sampler_bwd_kernel<float>, grid, dim3(256), 0, s, coords, (const float*)corr_grad, (float*)volume_grad, H1, W1, W2, radius, coords_channels, P);
  else if (dtype == AS_F16)
    hipLaunchKernelGGL(sampler_bwd_kernel<_Float16>, grid, dim3(256), 0, s, coords, (const _Float16*)corr_grad, (_Float16*)volume_grad, H1, W1, W2, radius, coords_channels, P);
  else
    hipLaunchKernelGGL(sampler_bwd_kernel<double>, grid, dim3(256), 0, s, coords, (const double*)corr_grad, (double*)volume_grad, H1, W1, W2, radius, coords_channels, P);
  return as::check_launch("corr_sampler_bwd");
}


// fragment image of the LDS-tile kernel, followed (IGEV's 162 columns) by the register-direct kernel's permuted image
static int64_t convc1_tile_image_bytes(int cin) { return (long long)((cin + 15) / 16) * 2 * 2 * 1024; }
int64_t as_lookup_convc1_pack_bytes(int cin) { return convc1_tile_image_bytes(cin) + (cin == 162 ? kDirectImgBytes : 0); }

int as_lookup_convc1_pack(const float* w, int cin, void* image, void* stream) {
  AS_REQUIRE(w && image && cin > 0, AS_ERR_BAD_ARG, "lookup_convc1_pack: bad argument");
  const int ks = (cin + 15) / 16;
  hipLaunchKernelGGL(frag_pack_kernel, dim3(as::cdiv(ks * 2 * 2 * 512, 256)), dim3(256), 0, as::as_stream(stream), w, 64, cin, 0, cin, ks,
                     (_Float16*)image);
  if (cin == 162)
    hipLaunchKernelGGL(frag_pack_direct_kernel, dim3(as::cdiv(kDirectKS * 2 * 2 * 512, 256)), dim3(256), 0, as::as_stream(stream), w,
                       (_Float16*)((unsigned char*)image + convc1_tile_image_bytes(cin)));
  return as::check_launch("lookup_convc1_pack");
}

int as_lookup_convc1_fwd(const float* const* geo, const float* const* corr, const float* disp, const void* wimage, const float* bias,
                         void* out_bs, int out_bs_ctot, int out_bs_coff, float* out_f32, int relu,
                         int B, int H, int W, int W2, int D, int G, int L, int radius, void* stream) {
  LookupParams p{};
  int rc = fill_common(p, B, H, W, W2, D, G, L, radius);
  if (rc != AS_OK) return rc;
  AS_REQUIRE(corr && disp && wimage && (out_bs || out_f32) && (G == 0 || geo), AS_ERR_BAD_ARG, "lookup_convc1: null pointer");
  AS_REQUIRE(radius == 4 && ((G == 8 && L == 2) || (G == 0 && L == 4)), AS_ERR_BAD_ARG,
             "lookup_convc1: built for radius 4 with (G, L) = (8, 2) or (0, 4); got r=%d G=%d L=%d", radius, G, L);
  AS_REQUIRE(!out_bs || (out_bs_coff % 8 == 0 && out_bs_coff + 64 <= (out_bs_ctot + 7) / 8 * 8), AS_ERR_BAD_SHAPE,
             "lookup_convc1: channel window [%d,%d) outside the blocked tensor's %d channels", out_bs_coff, out_bs_coff + 64, out_bs_ctot);
  for (int i = 0; i < L; ++i) {
    AS_REQUIRE(corr[i] && (G == 0 || geo[i]), AS_ERR_BAD_ARG, "lookup_convc1: null level %d", i);
    p.corr[i] = corr[i];
    p.geo[i] = G ? geo[i] : nullptr;
    AS_REQUIRE(G == 0 || (reinterpret_cast<uintptr_t>(geo[i]) & 15) == 0, AS_ERR_BAD_ARG, "lookup_convc1: geo[%d] not 16-B aligned", i);
    const long long cb = p.P * (W2 >> i) * 4, gb = p.P * (long long)(D >> i) * G * 4;
    AS_REQUIRE(cb < 0x7FFFFFF0ll && gb < 0x7FFFFFF0ll, AS_ERR_BAD_SHAPE, "lookup_convc1: pyramid level %d exceeds 2 GiB", i);
    p.corr_bytes[i] = (int)cb;
    p.geo_bytes[i] = (int)gb;
  }
  p.disp = disp;
  FusedParams f{(const _Float16*)wimage, bias, (_Float16*)out_bs, (out_bs_ctot + 7) / 8, out_bs_coff / 8, out_f32, relu};
  // IGEV geometry: the register-direct kernel (AS_LOOKUP_DIRECT=0: the LDS-tile kernel; =2: 64-pixel blocks)
  static const int direct_mode = getenv("AS_LOOKUP_DIRECT") ? atoi(getenv("AS_LOOKUP_DIRECT")) : 1;
  if (G == 8 && L == 2 && direct_mode) {
    const _Float16* dimg = (const _Float16*)((const unsigned char*)wimage + convc1_tile_image_bytes(p.CH));
    if (direct_mode == 2)
      hipLaunchKernelGGL((lookup_convc1_direct_kernel<2>), dim3((unsigned)as::cdiv64(p.P, 64)), dim3(128), kDirectImgBytes, as::as_stream(stream), p, f, dimg);
    else
      hipLaunchKernelGGL((lookup_convc1_direct_kernel<4>), dim3((unsigned)as::cdiv64(p.P, 128)), dim3(256), kDirectImgBytes, as::as_stream(stream), p, f, dimg);
    return as::check_launch("lookup_convc1_fwd(direct)");
  }
  const size_t lds = (size_t)((p.CH + 15) / 16 * 16 * 65) * sizeof(float);
  const dim3 grid((unsigned)as::cdiv64(p.P, 64));
  if (G == 8) hipLaunchKernelGGL((lookup_convc1_kernel<8>), grid, dim3(256), lds, as::as_stream(stream), p, f);
  else hipLaunchKernelGGL((lookup_convc1_kernel<0>), grid, dim3(256), lds, as::as_stream(stream), p, f);
  return as::check_launch("lookup_convc1_fwd");
}

int as_loop_front_fwd(const float* const* geo, const float* const* corr, const float* taps, int groups, const float* head_bias,
                      const float* disp_old, float* disp_new, const void* wimage, const float* bias_c1, void* cor_bs,
                      const float* w7, int cp7, const float* b7, void* d1_bs, void* copy_bs, int copy_ctot, int copy_coff,
                      int B, int H, int W, int W2, int D, int G, int L, int radius, void* stream) {
  LookupParams p{};
  int rc = fill_common(p, B, H, W, W2, D, G, L, radius);
  if (rc != AS_OK) return rc;
  AS_REQUIRE(corr && taps && disp_old && disp_new && wimage && cor_bs && (!w7 || d1_bs) && (G == 0 || geo), AS_ERR_BAD_ARG, "loop_front: null pointer");
  AS_REQUIRE(radius == 4 && ((G == 8 && L == 2) || (G == 0 && L == 4)), AS_ERR_BAD_ARG,
             "loop_front: built for radius 4 with (G, L) = (8, 2) or (0, 4); got r=%d G=%d L=%d", radius, G, L);
  AS_REQUIRE(groups >= 1 && groups <= 64 && (!w7 || cp7 >= 64) && (long long)B * groups * 9 * H * W * 4 < 0x7FFFFFF0ll, AS_ERR_BAD_SHAPE, "loop_front: groups=%d cp7=%d", groups, cp7);
  AS_REQUIRE(!copy_bs || (copy_coff >= 0 && copy_coff < (copy_ctot + 7) / 8 * 8), AS_ERR_BAD_SHAPE, "loop_front: copy channel %d outside %d", copy_coff, copy_ctot);
  for (int i = 0; i < L; ++i) {
    AS_REQUIRE(corr[i] && (G == 0 || geo[i]), AS_ERR_BAD_ARG, "loop_front: null level %d", i);
    p.corr[i] = corr[i];
    p.geo[i] = G ? geo[i] : nullptr;
    AS_REQUIRE(G == 0 || (reinterpret_cast<uintptr_t>(geo[i]) & 15) == 0, AS_ERR_BAD_ARG, "loop_front: geo[%d] not 16-B aligned", i);
    const long long cb = p.P * (W2 >> i) * 4, gb = p.P * (long long)(D >> i) * G * 4;
    AS_REQUIRE(cb < 0x7FFFFFF0ll && gb < 0x7FFFFFF0ll, AS_ERR_BAD_SHAPE, "loop_front: pyramid level %d exceeds 2 GiB", i);
    p.corr_bytes[i] = (int)cb;
    p.geo_bytes[i] = (int)gb;
  }
  p.disp = disp_new;  // unused by the kernel's lookup role (the disparity comes from registers)
  FusedParams f{(const _Float16*)wimage, bias_c1, (_Float16*)cor_bs, 8, 0, nullptr, 1};
  FrontParams q{};
  q.taps = taps; q.groups = groups; q.head_bias = head_bias; q.disp_old = disp_old; q.disp_new = disp_new;
  q.w7 = w7; q.b7 = b7; q.CP7 = cp7; q.d1_bs = (_Float16*)d1_bs; q.copy_bs = (_Float16*)copy_bs; q.copy_ctot = copy_ctot; q.copy_coff = copy_coff;
  q.n_lookup = (int)as::cdiv64(p.P, 64);
  q.tiles_x = as::cdiv(W, 16); q.tiles_y = as::cdiv(H, 16);
  // w7 == NULL: the head's finish + lookup + convc1 only (the 7x7 conv is launched by the caller from disp_new)
  const long long n7 = w7 ? (long long)B * q.tiles_x * q.tiles_y * 2 : 0;  // 32 of the 64 output channels per block
  q.n_conv7 = (int)n7;
  const size_t lds = (size_t)((p.CH + 15) / 16 * 16 * 65) * sizeof(float);
  const dim3 grid((unsigned)(q.n_lookup + n7));
  if (G == 8) hipLaunchKernelGGL((loop_front_kernel<8>), grid, dim3(256), lds, as::as_stream(stream), p, f, q, q.w7, q.b7);
  else hipLaunchKernelGGL((loop_front_kernel<0>), grid, dim3(256), lds, as::as_stream(stream), p, f, q, q.w7, q.b7);
  return as::check_launch("loop_front_fwd");
}

unsigned as_lookup_split_overflow(int reset) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_split_overflow_lookup), sizeof(v)) != hipSuccess) return 0xFFFFFFFFu;
  if (reset && v) {
    const unsigned z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_split_overflow_lookup), &z, sizeof(z));
  }
  return v;
}

}  // extern "C"
