// a1+a2: all-pairs correlation with the pooled pyramid fused into the epilogue (fp32 MFMA),
// a2:    geometry-encoding-volume pyramid (transpose to the disparity-major lookup layout + pooling),
// a4:    group-wise correlation volume,  a5: softmax + disparity regression.
//
// All four are one-shot (once per stereo pair) and HBM-bound: each input element is read once per
// consumer tile (re-reads are L1/L2 hits) and every output element is written exactly once.
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ------------------------------------------------------------------------------------------------
// corr[b,y,x1,x2] = sum_c f1[b,c,y,x1] f2[b,c,y,x2]  — one (b,y) row is a (W1 x C)·(C x W2) GEMM.
// v_mfma_f32_32x32x2_f32: A[i=x1][k=c] and B[k=c][j=x2] are both "index = lane&31, k = lane>>5", and in
// NCHW a lane's element for both operands is f[b, c, y, x] with x contiguous across lanes, so the
// fragments are loaded straight from global memory as 128-B coalesced runs (no LDS round trip).
// C/D layout: col j = lane&31 = x2 (contiguous in the output row), row i = x1 from the register
// index, so pooled level l (mean of 2^l adjacent x2) is a butterfly over lanes xor 1,2,4 and the
// whole pyramid is written from registers: the level-0 volume is never re-read.
// ------------------------------------------------------------------------------------------------
struct CorrParams {
  const float* f1;
  const float* f2;
  float* lvl[AS_MAX_LEVELS];
  int B, C, H, W1, W2, L;
  int MT, NT;  // 32-wide tiles along x1 / x2
};

constexpr int kKS = 48;  // k-steps (channel pairs) held in registers per pass: C <= 96 in one pass

// Fragments come through raw buffer loads: the descriptor spans one batch element of the feature map,
// so columns beyond W (sentinel offset) and channels beyond C read as 0 by the hardware range check —
// no branches around loads, a whole fragment (48 loads) is in flight at once.  The B fragment of
// N-tile j+1 is fetched while the 48 MFMAs of tile j run (one wave per SIMD at 960x540: latency must
// be hidden inside the wave, not by occupancy).
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}


// Epilogue shared by the three build kernels: write one 32x32 tile of level 0 and its pooled levels
// straight from the MFMA accumulator layout (col = lane&31 = x2, row = (r&3) + 8 (r>>2) + 4 (lane>>5)).
// Stores go through per-(row, x1-tile) buffer descriptors: rows beyond W1 fall outside num_records and
// are dropped by the hardware, columns beyond the level width get the sentinel offset, so there is no
// branch and no 64-bit per-element address arithmetic (that version spent 20 us of a 38 us kernel here).
// Pooling over 2^s adjacent x2 is a DPP butterfly (quad_perm xor 1 / xor 2, row_half_mirror for the third
// step, which pairs quads that already hold their 4-lane means).
struct TileOut {
  __amdgpu_buffer_rsrc_t rs[AS_MAX_LEVELS];
};

__device__ __forceinline__ TileOut make_tile_out(const CorrParams& p, long long rowbase, int mt) {
  TileOut o;
  const int rows = min(32, p.W1 - mt * 32);
#pragma unroll
  for (int s = 0; s < AS_MAX_LEVELS; ++s) {
    if (s < p.L) {
      const int wl = p.W2 >> s;
      o.rs[s] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.lvl[s] + (rowbase + (long long)mt * 32) * wl), 0, rows * wl * 4, 0x00020000);
    } else {
      o.rs[s] = o.rs[0];
    }
  }
  return o;
}

__device__ __forceinline__ float dpp_xor_step(float v, int s) {
  const int iv = __builtin_bit_cast(int, v);
  int o;
  if (s == 1) o = __builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
  else if (s == 2) o = __builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  else if (s == 3) o = __builtin_amdgcn_update_dpp(iv, iv, 0x141, 0xF, 0xF, false);  // row_half_mirror
  else o = __shfl_xor(iv, 1 << (s - 1));
  return __builtin_bit_cast(float, o);
}

__device__ __forceinline__ void store_pyramid_tile(const CorrParams& p, const TileOut& o, const f32x16& acc, int nt, int lane) {
  const unsigned kOOB = 0x70000000u;  // + 31 rows x (W2 <= 2^16) x 4 B never wraps and never is < num_records
  const int l31 = lane & 31, half = lane >> 5;
  const int x2 = nt * 32 + l31;
  unsigned voff[AS_MAX_LEVELS], rstride[AS_MAX_LEVELS];
#pragma unroll
  for (int s = 0; s < AS_MAX_LEVELS; ++s) {
    const int wl = p.W2 >> s;
    const int xs = x2 >> s;
    const bool ok = s < p.L && (l31 & ((1 << s) - 1)) == 0 && xs < wl;
    voff[s] = ok ? (unsigned)((4 * half * wl + xs) * 4) : kOOB;
    rstride[s] = (unsigned)(wl * 4);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const unsigned rl = (unsigned)((r & 3) + 8 * (r >> 2));
    float v = acc[r];
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), o.rs[0], (int)(voff[0] + rl * rstride[0]), 0, 0);
#pragma unroll
    for (int s = 1; s < AS_MAX_LEVELS; ++s) {
      if (s < p.L) {
        v = (v + dpp_xor_step(v, s)) * 0.5f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), o.rs[s], (int)(voff[s] + rl * rstride[s]), 0, 0);
      }
    }
  }
}

template <bool A_RESIDENT>
__global__ __launch_bounds__(256) void corr_build_kernel(CorrParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int row = blockIdx.x;  // b*H + y
  const int b = row / p.H;
  const int y = row - b * p.H;
  const int mt = blockIdx.y * 4 + wave;
  if (mt >= p.MT) return;
  const long long cs1 = (long long)p.H * p.W1;  // channel stride (elements)
  const long long cs2 = (long long)p.H * p.W2;
  const __amdgpu_buffer_rsrc_t r1 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.f1 + (long long)b * p.C * cs1), 0, (int)(p.C * cs1 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r2 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.f2 + (long long)b * p.C * cs2), 0, (int)(p.C * cs2 * 4), 0x00020000);
  const unsigned kstride1 = (unsigned)(2 * cs1 * 4), kstride2 = (unsigned)(2 * cs2 * 4);
  const unsigned kOOB = 0x7FFFFFF0u;
  const int x1 = mt * 32 + l31;
  const unsigned a_off = x1 < p.W1 ? (unsigned)((half * cs1 + (long long)y * p.W1 + x1) * 4) : kOOB;
  const unsigned b_row = (unsigned)((half * cs2 + (long long)y * p.W2) * 4);
  const int ksteps = (p.C + 1) >> 1;
  const int passes = (ksteps + kKS - 1) / kKS;
  const long long rowbase = (long long)row * p.W1;

  const TileOut tout = make_tile_out(p, rowbase, mt);
  float a[kKS], bq[kKS], bn[kKS];
  if (A_RESIDENT) {
#pragma unroll
    for (int k = 0; k < kKS; ++k) a[k] = bload(r1, a_off == kOOB ? kOOB : a_off + k * kstride1);
  }
  // prefetch B of (tile 0, pass 0)
  {
    const int x2 = l31;
    const unsigned bo = x2 < p.W2 ? b_row + x2 * 4 : kOOB;
#pragma unroll
    for (int k = 0; k < kKS; ++k) bq[k] = bload(r2, bo == kOOB ? kOOB : bo + k * kstride2);
  }
  for (int nt = 0; nt < p.NT; ++nt) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ps = 0; ps < passes; ++ps) {
      // next (tile, pass) in iteration order
      int nnt = nt, nps = ps + 1;
      if (nps == passes) { nps = 0; nnt = nt + 1; }
      if (nnt < p.NT) {
        const int x2 = nnt * 32 + l31;
        const unsigned bo = x2 < p.W2 ? b_row + x2 * 4 + (unsigned)nps * kKS * kstride2 : kOOB;
#pragma unroll
        for (int k = 0; k < kKS; ++k) bn[k] = bload(r2, bo == kOOB ? kOOB : bo + k * kstride2);
      }
      if (!A_RESIDENT) {
        const unsigned ao = a_off == kOOB ? kOOB : a_off + (unsigned)ps * kKS * kstride1;
#pragma unroll
        for (int k = 0; k < kKS; ++k) a[k] = bload(r1, ao == kOOB ? kOOB : ao + k * kstride1);
      }
#pragma unroll
      for (int k = 0; k < kKS; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], bq[k], acc, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < kKS; ++k) bq[k] = bn[k];
    }
    store_pyramid_tile(p, tout, acc, nt, lane);
  }
}

// ---- split-precision variant: 3 x fp16 MFMA (32x32x16) per 16 channels ----------------------------
// An fp32 operand v is split as v = hi + lo/2048 with hi = fp16(v), lo = fp16((v - hi) * 2048): 22
// significand bits, the lo part rescaled so it never falls into fp16's subnormal range.  Then
//   a*b ~= hi_a*hi_b + (hi_a*lo_b + lo_a*hi_b)/2048      (dropped lo*lo term and split residual: ~2^-22 relative)
// accumulated in fp32 in two accumulators.  That is 3 MFMAs at 16x the fp32-MFMA rate: the all-pairs
// product stops being matrix-core bound (fp32 MFMA needs >= 11 us at 960x540 even at 100 % utilisation)
// and becomes the HBM-bound stream it should be.  Requires |v| < 65504 (fp16 range): feature maps are
// O(1).  Operand k-layout of v_mfma_f32_32x32x16_f16: lane (r = l&31, h = l>>5) holds k = 8h..8h+7.
using half8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int kKS16 = 6;  // 16-channel k-steps per pass (96 channels)

__device__ unsigned g_split_overflow_vol;  // as_volumes_split_overflow

// x = hi + lo/2048; with fp16_saturate_mode() (common.h) an operand outside fp16's range saturates instead of becoming
// inf / NaN; `amax` collects max |x| for the out-of-range count
__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& lo, float& amax) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) amax = fmaxf(amax, fmaxf(fabsf(v[j]), fabsf(v[j + 1])));
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 h = (_Float16)v[j];
    hi[j] = h;
    lo[j] = (_Float16)((v[j] - (float)h) * 2048.f);
  }
}

template <bool A_RESIDENT>
__global__ __launch_bounds__(256, 2) void corr_build_f16x3_kernel(CorrParams p) {
  as::fp16_saturate_mode();
  float ovf_amax = 0.f;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int row = blockIdx.x;  // b*H + y
  const int b = row / p.H;
  const int y = row - b * p.H;
  const int mt = blockIdx.y * 4 + wave;
  if (mt >= p.MT) return;
  const long long cs1 = (long long)p.H * p.W1;
  const long long cs2 = (long long)p.H * p.W2;
  const __amdgpu_buffer_rsrc_t r1 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.f1 + (long long)b * p.C * cs1), 0, (int)(p.C * cs1 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r2 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.f2 + (long long)b * p.C * cs2), 0, (int)(p.C * cs2 * 4), 0x00020000);
  const unsigned kOOB = 0x7FFFFFF0u;
  const int x1 = mt * 32 + l31;
  // channel of element j in k-step s: 16 s + 8 half + j
  const unsigned a_off = x1 < p.W1 ? (unsigned)((8 * half * cs1 + (long long)y * p.W1 + x1) * 4) : kOOB;
  const unsigned b_row = (unsigned)((8 * half * cs2 + (long long)y * p.W2) * 4);
  const unsigned cstr1 = (unsigned)(cs1 * 4), cstr2 = (unsigned)(cs2 * 4);
  const int ksteps = (p.C + 15) >> 4;
  const int passes = (ksteps + kKS16 - 1) / kKS16;
  const long long rowbase = (long long)row * p.W1;

  const TileOut tout = make_tile_out(p, rowbase, mt);
  half8 ahi[kKS16], alo[kKS16];
  auto load_a = [&](int ps) {
#pragma unroll
    for (int s = 0; s < kKS16; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = bload(r1, a_off == kOOB ? kOOB : a_off + (unsigned)((ps * kKS16 + s) * 16 + j) * cstr1);
      split8(v, ahi[s], alo[s], ovf_amax);
    }
  };
  if (A_RESIDENT) load_a(0);

  // N tiles are split over blockIdx.z and B is streamed one 16-channel k-step at a time (8 loads ->
  // split -> 3 MFMAs): ~110 VGPRs, so 4 waves per SIMD hide the load latency by switching waves.
  const int ntw = (p.NT + (int)gridDim.z - 1) / (int)gridDim.z;
  const int nt_lo = blockIdx.z * ntw, nt_hi = min(p.NT, nt_lo + ntw);
  for (int nt = nt_lo; nt < nt_hi; ++nt) {
    f32x16 acc_hh, acc_x;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_hh[r] = 0.f; acc_x[r] = 0.f; }
    const int x2 = nt * 32 + l31;
    const unsigned bo = x2 < p.W2 ? b_row + x2 * 4 : kOOB;
    for (int ps = 0; ps < passes; ++ps) {
      if (!A_RESIDENT) load_a(ps);
#pragma unroll
      for (int s = 0; s < kKS16; ++s) {
        float bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          bv[j] = bload(r2, bo == kOOB ? kOOB : bo + (unsigned)((ps * kKS16 + s) * 16 + j) * cstr2);
        half8 bhi, blo;
        split8(bv, bhi, blo, ovf_amax);
        acc_hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[s], bhi, acc_hh, 0, 0, 0);
        acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[s], blo, acc_x, 0, 0, 0);
        acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[s], bhi, acc_x, 0, 0, 0);
      }
    }
    f32x16 res;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = acc_hh[r] + acc_x[r] * (1.f / 2048.f);
    store_pyramid_tile(p, tout, res, nt, lane);
  }
  as::note_split_overflow(ovf_amax, &g_split_overflow_vol);
}

// ---- split precision, LDS-staged (C <= 96): the default at IGEV sizes -----------------------------
// Block = (row, 4 x1-tiles, 4 x2-tiles).  The 128-wide f2 slab of the row is fetched ONCE per block by all
// 256 threads with every load in flight at once (48 dwords per thread), split to fp16 hi/lo once, and
// parked in LDS in MFMA-fragment order [kstep][x2][16 ch] (a lane's 8 consecutive channels = one
// ds_read_b128, conflict-free).  Each wave then keeps its x1 tile (A, pre-split) in registers and sweeps
// the 4 x2 tiles: 72 MFMAs with no global load in the loop.  The streaming kernel above paid a full
// memory round trip per 16-channel k-step per wave and re-split f2 once per x1 tile; this one has one
// round trip per block and ~2 blocks resident per CU (48 KB LDS), which is what an HBM-bound one-shot
// kernel of this size (72 MB, ~10 us at peak) needs.
constexpr int kNB = 4;  // x2 tiles per block

__global__ __launch_bounds__(256, 2) void corr_build_lds_kernel(CorrParams p) {
  as::fp16_saturate_mode();
  float ovf_amax = 0.f;
  extern __shared__ __attribute__((aligned(16))) unsigned char corr_smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int row = blockIdx.x;  // b*H + y
  const int b = row / p.H;
  const int y = row - b * p.H;
  const int mt = blockIdx.y * 4 + wave;
  const int nb0 = blockIdx.z * (kNB * 32);
  const long long cs1 = (long long)p.H * p.W1;
  const long long cs2 = (long long)p.H * p.W2;
  const __amdgpu_buffer_rsrc_t r1 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.f1 + (long long)b * p.C * cs1), 0, (int)(p.C * cs1 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r2 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.f2 + (long long)b * p.C * cs2), 0, (int)(p.C * cs2 * 4), 0x00020000);
  const unsigned kOOB = 0x7FFFFFF0u;
  const unsigned cstr1 = (unsigned)(cs1 * 4), cstr2 = (unsigned)(cs2 * 4);
  const int ksteps = (p.C + 15) >> 4;  // <= kKS16
  const unsigned lo_base = (unsigned)ksteps * (kNB * 32) * 32;  // bytes of the hi image

  // Invalid lanes (columns beyond W, idle waves) carry the sentinel and just add the channel offset to it:
  // whatever that wraps to is either outside num_records (reads 0) or some in-range element of the same
  // tensor, and it only ever feeds output rows/columns that the epilogue drops — no select per load.
  // issue everything: this wave's A fragment (channels 16 s + 8 half + j of column x1) ...
  const int x1 = mt * 32 + l31;
  const unsigned a_off = (mt < p.MT && x1 < p.W1) ? (unsigned)((8 * half * cs1 + (long long)y * p.W1 + x1) * 4) : kOOB;
  float av[kKS16][8];
#pragma unroll
  for (int s = 0; s < kKS16; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) av[s][j] = bload(r1, a_off + (unsigned)(s * 16 + j) * cstr1);
  // ... and this thread's share of the f2 slab: item i = (8-channel group g = (tid>>7) + 2 i, column tid&127)
  const int xx = tid & 127;
  const int x2s = nb0 + xx;
  const unsigned b_off = x2s < p.W2 ? (unsigned)(((long long)y * p.W2 + x2s) * 4) : kOOB;
  float bv[kKS16][8];
#pragma unroll
  for (int i = 0; i < kKS16; ++i) {
    const int g = (tid >> 7) + 2 * i;
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[i][j] = bload(r2, b_off + (unsigned)(g * 8 + j) * cstr2);
  }
#pragma unroll
  for (int i = 0; i < kKS16; ++i) {
    if (i < ksteps) {
      half8 hi, lo;
      split8(bv[i], hi, lo, ovf_amax);
      const int g = (tid >> 7) + 2 * i;
      const unsigned o = (unsigned)(((g >> 1) * (kNB * 32) + xx) * 32 + (g & 1) * 16);
      *reinterpret_cast<half8*>(corr_smem + o) = hi;
      *reinterpret_cast<half8*>(corr_smem + lo_base + o) = lo;
    }
  }
  half8 ahi[kKS16], alo[kKS16];
#pragma unroll
  for (int s = 0; s < kKS16; ++s) split8(av[s], ahi[s], alo[s], ovf_amax);
  __syncthreads();
  if (mt >= p.MT) return;

  const long long rowbase = (long long)row * p.W1;
  const TileOut tout = make_tile_out(p, rowbase, mt);
  for (int t = 0; t < kNB; ++t) {
    const int nt = blockIdx.z * kNB + t;
    if (nt >= p.NT) break;
    f32x16 acc_hh, acc_x;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_hh[r] = 0.f; acc_x[r] = 0.f; }
    const unsigned fo = (unsigned)((t * 32 + l31) * 32 + half * 16);
#pragma unroll
    for (int s = 0; s < kKS16; ++s) {
      if (s == 0 || s < ksteps) {
        const half8 bhi = *reinterpret_cast<const half8*>(corr_smem + fo + (unsigned)s * (kNB * 32 * 32));
        const half8 blo = *reinterpret_cast<const half8*>(corr_smem + lo_base + fo + (unsigned)s * (kNB * 32 * 32));
        acc_hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[s], bhi, acc_hh, 0, 0, 0);
        acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[s], blo, acc_x, 0, 0, 0);
        acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[s], bhi, acc_x, 0, 0, 0);
      }
    }
    f32x16 res;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = acc_hh[r] + acc_x[r] * (1.f / 2048.f);
    store_pyramid_tile(p, tout, res, nt, lane);
  }
  as::note_split_overflow(ovf_amax, &g_split_overflow_vol);
}

// ------------------------------------------------------------------------------------------------
// gev [B,G,D,H,W] -> level i [B,H,W,D>>i,G].  32 consecutive pixels of a row per block; the
// (g,d) x pixel tile is transposed through LDS so that both the read (x-contiguous) and the write
// (one pixel's D*G run = 1.5 KB contiguous) are coalesced.
// ------------------------------------------------------------------------------------------------
struct GeoParams {
  const float* gev;
  float* lvl[AS_MAX_LEVELS];
  int B, G, D, H, W, L;
};

// mean over the 2^LV raw disparities [d<<LV, (d+1)<<LV) of channel g at pixel px, as repeated pair means
// ((a+b)/2 of (a+b)/2 ...) — bitwise what repeated avg_pool2d computes.  tile rows are (disparity, channel):
// row r = dd*G + g at pitch 33 floats, so 64 consecutive output elements (g fastest, then d) read 64 distinct banks.
template <int LV>
__device__ __forceinline__ float geo_pooled(const float* tile, int g, int d, int G, int px) {
  constexpr int N = 1 << LV;
  float v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = tile[(((d << LV) + i) * G + g) * 33 + px];
#pragma unroll
  for (int s2 = N; s2 > 1; s2 >>= 1)
#pragma unroll
    for (int i = 0; i < (s2 >> 1); ++i) v[i] = (v[2 * i] + v[2 * i + 1]) * 0.5f;
  return v[0];
}

// TG / TD: compile-time G / D (8 / 48 for IGEV: index arithmetic by constants) or 0 = runtime values.
template <int TG, int TD>
__global__ __launch_bounds__(256) void geo_pyramid_kernel(GeoParams p) {
  extern __shared__ float tile[];  // [G*D][33]
  const int G = TG ? TG : p.G, D = TD ? TD : p.D;
  const int xt = blockIdx.x;
  const int row = blockIdx.y;  // b*H + y
  const int b = row / p.H;
  const int y = row - b * p.H;
  const int x0 = xt * 32;
  const int GD = G * D;
  const long long plane = (long long)p.H * p.W;
  // staging: unconditional buffer loads, 16 in flight per thread (the `cond ? load : 0` form costs a branch and a
  // full memory round trip per element); columns beyond W read 0 through the sentinel
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.gev + (long long)b * GD * plane), 0, (int)((long long)GD * plane * 4), 0x00020000);
  const unsigned plane_u = (unsigned)plane;
  const int px_l = threadIdx.x & 31, gd_l = threadIdx.x >> 5;  // 8 (g,d) rows x 32 pixels per pass
  const unsigned base = (x0 + px_l < p.W) ? ((unsigned)(y * p.W + x0 + px_l)) * 4u : 0x70000000u;
  constexpr int NB = 16;
#pragma unroll 1
  for (int r0 = 0; r0 < GD; r0 += 8 * NB) {
    float v[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) v[i] = bload(rs, base + (unsigned)(r0 + i * 8 + gd_l) * plane_u * 4u);  // rows >= GD: out of range -> 0
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int gd = r0 + i * 8 + gd_l;  // source row g*D + d -> tile row d*G + g
      const int g = gd / D, d = gd - g * D;
      if (gd < GD) tile[(d * G + g) * 33 + px_l] = v[i];
    }
  }
  __syncthreads();
  const int npx = min(32, p.W - x0);
  const long long pix0 = (long long)row * p.W + x0;
#define AS_GEO_LEVEL(LV)                                                          \
  if ((LV) < p.L) {                                                               \
    const int per = (D >> (LV)) * G;                                              \
    float* dst = p.lvl[LV] + pix0 * per;                                          \
    for (int idx = threadIdx.x; idx < npx * per; idx += 256) {                    \
      const int px = idx / per;                                                   \
      const int rem = idx - px * per;                                             \
      const int d = rem / G, g = rem - d * G;                                     \
      dst[idx] = geo_pooled<LV>(tile, g, d, G, px);                               \
    }                                                                             \
  }
  AS_GEO_LEVEL(0) AS_GEO_LEVEL(1) AS_GEO_LEVEL(2) AS_GEO_LEVEL(3)
#undef AS_GEO_LEVEL
  static_assert(AS_MAX_LEVELS == 4, "geo_pyramid: one AS_GEO_LEVEL per level");
}

// ------------------------------------------------------------------------------------------------
// a4  vol[b,g,d,y,x] = mean_{c in group g} fl[b,c,y,x] * fr[b,c,y,x-d]  (0 for x<d).
// Block = 64 consecutive x of one row; fl tile and fr tile(+D-1 halo) staged in LDS once, each
// of the 4 waves owns a quarter of the disparities and keeps its accumulators in registers.
// ------------------------------------------------------------------------------------------------
// Rewritten for round 1b: the first version staged its tiles with `cond ? load : 0` (a branch and a full
// memory round trip per element, 66 per thread) and ran at 7 % of the HBM roofline.  Now: every fr element of
// the block's (row, 64-column) slab (+D-1 halo, all C channels) is fetched with ONE batch of unconditional
// buffer loads (zero padding from the range check) and parked in LDS; fl comes straight from global into
// registers (a lane's own column, 256-B coalesced rows); each wave owns G/4 groups and keeps all D
// accumulators of a lane in registers; the sliding window fr[c][x-d] is read four disparities at a time
// with ds_read_b128 (neighbouring lanes hit the same addresses: LDS broadcast, no bank conflict).
// ------------------------------------------------------------------------------------------------
constexpr int kGwcNit = 48;  // staging rounds: C * (64 + DMAX) <= 48 * 256 elements per block

// CG: channels per group; kGwcD = DMAX: disparities per lane held in registers (48 | 64, D <= DMAX)
template <int CG, int kGwcD>
__global__ __launch_bounds__(256, 3) void gwc_kernel(const float* __restrict__ fl, const float* __restrict__ fr,
                                                  float* __restrict__ out, int B, int C, int H, int W, int D, int G) {
  extern __shared__ __attribute__((aligned(16))) float gwc_sm[];  // [C][FWP]: local index j <-> column x0 - (kGwcD - 1) - 1 + j ... see below
  constexpr int FW = 64 + kGwcD;          // 112 columns: x0 - 48 .. x0 + 63 (one spare on the left keeps reads 16-B aligned)
  constexpr int FWP = FW + 4;             // row pitch (floats): 116 -> rows start 16-B aligned, banks skewed by 20
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.y;
  const int b = row / H;
  const int y = row - b * H;
  const int x0 = blockIdx.x * 64;
  const long long plane = (long long)H * W;
  const unsigned kOOB = 0x7FFFFFF0u;
  const __amdgpu_buffer_rsrc_t rr =
      __builtin_amdgcn_make_buffer_rsrc((void*)(fr + (long long)b * C * plane), 0, (int)((long long)C * plane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rl =
      __builtin_amdgcn_make_buffer_rsrc((void*)(fl + (long long)b * C * plane), 0, (int)((long long)C * plane * 4), 0x00020000);
  // ---- stage fr: item = (channel c, column j), j = 0..FW-1 <-> x = x0 - kGwcD + j; batches of 16 loads in flight ----
  const int items = C * FW;
  const unsigned plane_u = (unsigned)plane, row_u = (unsigned)(y * W);
  constexpr int NB = 24;  // two batches cover C = 96
#pragma unroll 1
  for (int i0 = 0; i0 < kGwcNit; i0 += NB) {
    if (i0 * 256 >= items) break;
    float v[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + (i0 + i) * 256;
      const int c = idx / FW, j = idx - c * FW;
      const int xs = x0 - kGwcD + j;
      const unsigned o = ((unsigned)c * plane_u + row_u + (unsigned)xs) * 4u;  // < 2^31 (host check); garbage when invalid
      v[i] = bload(rr, (idx < items && xs >= 0 && xs < W) ? o : kOOB);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + (i0 + i) * 256;
      if (idx < items) {
        const int c = idx / FW, j = idx - c * FW;
        gwc_sm[c * FWP + j] = v[i];
      }
    }
  }
  __syncthreads();
  // ---- this lane's column: wave w owns groups [w G/4, (w+1) G/4); the D accumulators in two halves of DH ----
  const int gpw = G >> 2;
  const int x = x0 + lane;
  const unsigned lo = x < W ? (row_u + (unsigned)x) * 4u : 0x70000000u;
  const unsigned pl4 = plane_u * 4u;
  const float inv = 1.0f / (float)CG;
  constexpr int DH = kGwcD / 2;
#pragma unroll 1
  for (int gi = 0; gi < gpw; ++gi) {
    const int g = wave * gpw + gi;
    float a[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) a[c] = bload(rl, lo + (unsigned)(g * CG + c) * pl4);  // x >= W: sentinel + offset, any in-range garbage is never stored
#pragma unroll 1
    for (int hd = 0; hd < 2; ++hd) {
      float acc[DH];
#pragma unroll
      for (int d = 0; d < DH; ++d) acc[d] = 0.f;
      // fr[c][x - d] lives at local column kGwcD + lane - d
      const float* r0 = gwc_sm + (g * CG) * FWP + lane + kGwcD - hd * DH;
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        const float* r = r0 + c * FWP;
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = fmaf(a[c], r[-d], acc[d]);
        __builtin_amdgcn_sched_barrier(0);  // one channel's window at a time
      }
      // stores through a descriptor over this batch element's [G,D,H,W] volume: columns beyond W carry the sentinel,
      // disparities beyond D fall outside num_records of the group window
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(out + (((long long)b * G + g) * D) * plane), 0, (int)((long long)D * plane * 4), 0x00020000);
      const unsigned so = x < W ? lo + (unsigned)(hd * DH) * pl4 : 0x70000000u;
#pragma unroll
      for (int d = 0; d < DH; ++d)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc[d] * inv), ro, (int)(so + (unsigned)d * pl4), 0, 0);
    }
  }
}

// a5  out[b,0,y,x] = sum_d d * softmax_d(cost[b,:,y,x])
__global__ __launch_bounds__(256) void softmax_dispreg_kernel(const float* __restrict__ cost, float* __restrict__ out,
                                                              int D, long long plane, long long P, int softmax) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= P) return;
  const long long b = pix / plane;
  const long long rem = pix - b * plane;
  const float* c = cost + b * D * plane + rem;
  if (!softmax) {
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc += c[(long long)d * plane] * (float)d;
    out[pix] = acc;
    return;
  }
  float m = -INFINITY;
  for (int d = 0; d < D; ++d) m = fmaxf(m, c[(long long)d * plane]);
  float s = 0.f, sd = 0.f;
  for (int d = 0; d < D; ++d) {
    const float e = expf(c[(long long)d * plane] - m);
    s += e;
    sd += e * (float)d;
  }
  out[pix] = sd / s;
}

}  // namespace

extern "C" {

int as_corr_build_pyramid(const float* f1, const float* f2, float* const* levels, int B, int C, int H, int W1, int W2,
                          int L, void* stream) {
  AS_REQUIRE(f1 && f2 && levels, AS_ERR_BAD_ARG, "corr_build: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W1 > 0 && W2 > 0, AS_ERR_BAD_ARG, "corr_build: non-positive size");
  AS_REQUIRE(L >= 1 && L <= AS_MAX_LEVELS, AS_ERR_BAD_ARG, "corr_build: L=%d outside [1,%d]", L, AS_MAX_LEVELS);
  AS_REQUIRE((W2 >> (L - 1)) >= 1, AS_ERR_BAD_SHAPE, "corr_build: W2=%d too small for %d levels", W2, L);
  AS_REQUIRE(W1 <= 65536 && W2 <= 65536, AS_ERR_BAD_SHAPE, "corr_build: W1=%d / W2=%d beyond 65536", W1, W2);
  AS_REQUIRE((long long)B * H < 2147483647ll && (long long)B * H * W1 * (long long)W2 < (1ll << 40), AS_ERR_BAD_SHAPE, "corr_build: too large");
  CorrParams p{};
  p.f1 = f1; p.f2 = f2; p.B = B; p.C = C; p.H = H; p.W1 = W1; p.W2 = W2; p.L = L;
  p.MT = as::cdiv(W1, 32);
  p.NT = as::cdiv(W2, 32);
  for (int i = 0; i < L; ++i) {
    AS_REQUIRE(levels[i], AS_ERR_BAD_ARG, "corr_build: null level %d", i);
    p.lvl[i] = levels[i];
  }
  AS_REQUIRE((long long)C * H * W1 * 4 < 0x7FFFFFF0ll && (long long)C * H * W2 * 4 < 0x7FFFFFF0ll, AS_ERR_BAD_SHAPE,
             "corr_build: a feature map exceeds 2 GiB per batch element");
  dim3 grid((unsigned)(B * H), (unsigned)as::cdiv(p.MT, 4));
  if (as::use_split_precision()) {
    // enough waves to keep ~4 per SIMD: split the N tiles over grid.z while B*H*MT waves are few
    int nsplit = 1;
    while (nsplit < p.NT && (long long)B * H * p.MT * nsplit < 4096) nsplit *= 2;
    const dim3 g3(grid.x, grid.y, (unsigned)nsplit);
    const int ks16 = (C + 15) / 16;
    if (ks16 <= kKS16) {
      const dim3 gl(grid.x, grid.y, (unsigned)as::cdiv(p.NT, kNB));
      hipLaunchKernelGGL(corr_build_lds_kernel, gl, dim3(256), (size_t)ks16 * (kNB * 32) * 32 * 2, as::as_stream(stream), p);
    } else if (ks16 <= kKS16) hipLaunchKernelGGL(corr_build_f16x3_kernel<true>, g3, dim3(256), 0, as::as_stream(stream), p);
    else hipLaunchKernelGGL(corr_build_f16x3_kernel<false>, g3, dim3(256), 0, as::as_stream(stream), p);
  } else if ((C + 1) / 2 <= kKS) hipLaunchKernelGGL(corr_build_kernel<true>, grid, dim3(256), 0, as::as_stream(stream), p);
  else hipLaunchKernelGGL(corr_build_kernel<false>, grid, dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("corr_build_pyramid");
}

int as_geo_pyramid(const float* gev, float* const* levels, int B, int G, int D, int H, int W, int L, void* stream) {
  AS_REQUIRE(gev && levels, AS_ERR_BAD_ARG, "geo_pyramid: null pointer");
  AS_REQUIRE(B > 0 && G > 0 && D > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "geo_pyramid: non-positive size");
  AS_REQUIRE(L >= 1 && L <= AS_MAX_LEVELS && (D >> (L - 1)) >= 1, AS_ERR_BAD_ARG, "geo_pyramid: L=%d D=%d", L, D);
  const size_t lds = (size_t)G * D * 33 * sizeof(float);
  AS_REQUIRE(lds <= 160 * 1024, AS_ERR_BAD_SHAPE, "geo_pyramid: G*D=%d needs %zu B of LDS (> 160 KiB)", G * D, lds);
  AS_REQUIRE((long long)B * H <= 65535, AS_ERR_BAD_SHAPE, "geo_pyramid: B*H=%lld exceeds grid.y", (long long)B * H);
  GeoParams p{};
  p.gev = gev; p.B = B; p.G = G; p.D = D; p.H = H; p.W = W; p.L = L;
  for (int i = 0; i < L; ++i) {
    AS_REQUIRE(levels[i], AS_ERR_BAD_ARG, "geo_pyramid: null level %d", i);
    p.lvl[i] = levels[i];
  }
  AS_REQUIRE((long long)G * D * H * W * 4 < 0x70000000ll, AS_ERR_BAD_SHAPE, "geo_pyramid: volume exceeds 1.75 GiB per batch element");
  dim3 grid((unsigned)as::cdiv(W, 32), (unsigned)(B * H));
  if (G == 8 && D == 48) {
    hipLaunchKernelGGL((geo_pyramid_kernel<8, 48>), grid, dim3(256), lds, as::as_stream(stream), p);
  } else {
    if (lds > 64 * 1024) as::lds_opt_in((const void*)geo_pyramid_kernel<0, 0>);
    hipLaunchKernelGGL((geo_pyramid_kernel<0, 0>), grid, dim3(256), lds, as::as_stream(stream), p);
  }
  return as::check_launch("geo_pyramid");
}

int as_gwc_volume_fwd(const float* fl, const float* fr, float* out, int B, int C, int H, int W, int D, int G, void* stream) {
  AS_REQUIRE(fl && fr && out, AS_ERR_BAD_ARG, "gwc: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && D > 0 && G > 0, AS_ERR_BAD_ARG, "gwc: non-positive size");
  AS_REQUIRE(C % G == 0, AS_ERR_BAD_SHAPE, "gwc: C=%d not divisible by G=%d", C, G);
  AS_REQUIRE((long long)B * H <= 65535, AS_ERR_BAD_SHAPE, "gwc: B*H=%lld exceeds grid.y", (long long)B * H);
  AS_REQUIRE(D >= 1 && D <= 64, AS_ERR_BAD_SHAPE, "gwc: D=%d (supported: 1..64)", D);
  AS_REQUIRE(G % 4 == 0 && C % G == 0 && (C / G == 12 || C / G == 8 || C / G == 4), AS_ERR_BAD_SHAPE,
             "gwc: C=%d G=%d (supported: G %% 4 == 0, C/G in {4, 8, 12})", C, G);
  const int dmax = D <= 48 ? 48 : 64;
  AS_REQUIRE((long long)C * (64 + dmax) <= kGwcNit * 256, AS_ERR_BAD_SHAPE, "gwc: C=%d too large for one LDS slab", C);
  AS_REQUIRE((long long)C * H * W * 4 < 0x70000000ll && (long long)64 * H * W * 4 < 0x0FFFFFFFll, AS_ERR_BAD_SHAPE,
             "gwc: a feature map / disparity slab too large for 32-bit buffer offsets");
  const size_t lds = (size_t)C * (64 + dmax + 4) * sizeof(float);
  dim3 grid((unsigned)as::cdiv(W, 64), (unsigned)(B * H));
  const int cg = C / G;
  hipStream_t s = as::as_stream(stream);
#define AS_GWC(CG_, DM_)                                                                                         \
  {                                                                                                              \
    if (lds > 64 * 1024) as::lds_opt_in((const void*)gwc_kernel<CG_, DM_>);                                                                     \
    hipLaunchKernelGGL((gwc_kernel<CG_, DM_>), grid, dim3(256), lds, s, fl, fr, out, B, C, H, W, D, G);           \
  }
  if (dmax == 48) { if (cg == 12) AS_GWC(12, 48) else if (cg == 8) AS_GWC(8, 48) else AS_GWC(4, 48) }
  else { if (cg == 12) AS_GWC(12, 64) else if (cg == 8) AS_GWC(8, 64) else AS_GWC(4, 64) }
#undef AS_GWC
  return as::check_launch("gwc_volume_fwd");
}

int as_disparity_regression(const float* cost, float* out, int B, int D, int H, int W, int apply_softmax, void* stream) {
  AS_REQUIRE(cost && out, AS_ERR_BAD_ARG, "dispreg: null pointer");
  AS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "dispreg: non-positive size");
  const long long plane = (long long)H * W, P = plane * B;
  hipLaunchKernelGGL(softmax_dispreg_kernel, dim3((unsigned)as::cdiv64(P, 256)), dim3(256), 0, as::as_stream(stream), cost, out, D, plane, P, apply_softmax);
  return as::check_launch("disparity_regression");
}

unsigned as_volumes_split_overflow(int reset) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_split_overflow_vol), sizeof(v)) != hipSuccess) return 0xFFFFFFFFu;
  if (reset && v) {
    const unsigned z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_split_overflow_vol), &z, sizeof(z));
  }
  return v;
}

}  // extern "C"
