// a12-a17 as the inference pipeline runs them: the LIIF-style continuous upsampler with no per-query intermediate in HBM.
//
//   sf_norm_kernel + sf_affinity_tile_kernel   aff8 = AffinityFeature(cat(srcs))            (liif.py:432-446), no copy of x
//   liif_lowres_cl_kernel                      u = W1[:, cols] . cat(srcs, aff8) per LOW-resolution pixel, stored
//                                              channels-last [B][H*W][128] (a query's 128-vector = 512 contiguous bytes)
//   liif_tail_kernel                           per tile of 32 queries: nearest gather of u0 / u1 + relative-coordinate
//                                              term + ReLU -> 128 -> 64 -> 64 -> 9 MLP on the matrix cores (weights resident
//                                              in LDS) -> softmax -> 9-tap convex combination of the 3x3 disparity
//                                              neighbourhood -> out [B,1,Q]
// Reference: liif.py:108-137 (gather, rel coord), :9-25 + :644-678 (MLP), continuous_IGEVstereo.py:204-214 (x4*scale,
// softmax), submodule.py:357-372 (convex upsampling, in-place clamp of the caller's coordinates).
//
// Arithmetic = the split-precision scheme of conv.hip / volumes.hip: x = hi + lo/2048 with fp16 hi/lo, three
// v_mfma_f32_32x32x16_f16 per product (hi.hi, hi.lo, lo.hi), fp32 accumulation: ~2^-22 relative per product.
//
// Operand chaining (no LDS / no lane movement between the layers): a 32x32 fp32 accumulator tile X has its column
// (= query) on the lane and its rows (= channels) in the 16 registers, row(i, half) = (i&3) + 8(i>>2) + 4 half.  Registers
// 8s..8s+7 of tile t, converted to fp16, ARE the B fragment of k-step ks = 2t+s of the next layer when that layer's weights
// are packed with the same k order:  element j of lane half h  <->  channel 16 ks + 8(j>>2) + 4h + (j&3).
// The first layer's gathered operand uses the same order, so a lane's 8 channels of a k-step are two 16-B groups
// [16ks+4h, +4) and [16ks+8+4h, +4) of the channels-last rows.
#include <algorithm>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kHid1 = 128, kHid2 = 64, kHid3 = 64, kOut = 9;
constexpr int kFragBlocks = 64;                      // 1-KiB fragment blocks of the weight image (see liif_tail_pack_kernel)
constexpr int kBiasFloats = kHid2 + kHid3 + 32;       // b2 | b3 | b4 (padded to 32 rows)
constexpr int kImageBytes = kFragBlocks * 1024 + kBiasFloats * 4;
constexpr int kBlkRel = 0, kBlkW2 = 8, kBlkW3 = 40, kBlkW4 = 56;
constexpr float kF16Max = 65504.f;

// x = hi + lo/2048.  |x| >= 65504 (outside fp16) is clamped to +-65504 instead of becoming inf / NaN; the caller counts it.
// `amax` tracks max |x| over everything a thread splits (one v_max3 per two elements): the kernel tests it once at its end.
// v_cvt_pkrtz_f16_f32 converts two values per instruction; rounding toward zero never produces inf from a finite value, so
// |x| >= 65504 saturates to +-65504 with no extra clamp (hi = rtz(x) leaves a residual < 1 ulp(fp16), lo = rtz(residual *
// 2048): ~2^-20 relative per operand instead of round-to-nearest's 2^-22 — far inside this path's tolerance).
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using u32x4v = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& lo, float& amax) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) amax = fmaxf(amax, fmaxf(fabsf(v[j]), fabsf(v[j + 1])));
  u32x4v ph, pl;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(v[j], v[j + 1]));
    const float d0 = (v[j] - (float)h2[0]) * 2048.f, d1 = (v[j + 1] - (float)h2[1]) * 2048.f;
    ph[j >> 1] = __builtin_bit_cast(unsigned, h2);
    pl[j >> 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d0, d1));
  }
  hi = __builtin_bit_cast(half8, ph);
  lo = __builtin_bit_cast(half8, pl);
}

// Non-negative inputs (post-ReLU activations): the running maximum is an INTEGER max of the bit patterns (one v_max3_i32 per
// two elements, no |x| modifiers, no canonicalisation; a NaN's pattern is larger than any finite value's, so it is flagged).
__device__ __forceinline__ void split8_pos(const float (&v)[8], half8& hi, half8& lo, int& imax) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) imax = max(max(imax, __builtin_bit_cast(int, v[j])), __builtin_bit_cast(int, v[j + 1]));
  u32x4v ph, pl;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const half2v h2 = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(v[j], v[j + 1]));
    const float d0 = (v[j] - (float)h2[0]) * 2048.f, d1 = (v[j + 1] - (float)h2[1]) * 2048.f;
    ph[j >> 1] = __builtin_bit_cast(unsigned, h2);
    pl[j >> 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d0, d1));
  }
  hi = __builtin_bit_cast(half8, ph);
  lo = __builtin_bit_cast(half8, pl);
}

// relu(x) for a non-NaN x as an integer max (negative floats are negative integers): no canonicalising v_max in front
__device__ __forceinline__ float relu_bits(float x) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }

__device__ unsigned g_split_overflow_liif;  // queries / pixels whose operands left the fp16 range (as_split_overflow_count)

__device__ __forceinline__ void note_overflow(float amax) {  // amax >= 65504 (or NaN): some operand was saturated
  if (__builtin_amdgcn_ballot_w64(!(amax < kF16Max)) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&g_split_overflow_liif, 1u);
}

// grid_sample(mode='nearest', align_corners=False) source index with ATen's fp32 operation sequence (no fma contraction)
__device__ __forceinline__ int nearest_idx(float c, int n) {
  const float u = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(c, 1.f), (float)n), 1.f), 2.f);
  return (int)rintf(u);
}

__device__ __forceinline__ int acc_row(int i, int half) { return (i & 3) + 8 * (i >> 2) + 4 * half; }

// ---------------------------------------------------------------------------------------------------------------------
// affinity: per channel group the 8 neighbour dot products + the squared norm on an LDS tile, then one finishing pass
// ---------------------------------------------------------------------------------------------------------------------
struct SfParams {
  const float* src[3];
  int c[3];
  int n_src;
  int n_groups;   // channel groups (gridDim.z = B * n_groups): partial sums per group, so small maps still fill the chip
  int n_chunks;   // 8-channel chunks over all sources
  float* part;    // [B][n_groups][9][H][W]: 8 dots, then sum of squares
  float* aff;     // [B,8,H,W]
  int B, H, W;
};

constexpr int kTW = 32, kTH = 8, kPW = kTW + 2, kPH = kTH + 2, kPos = kPW * kPH;  // 34 x 10 halo tile

__global__ __launch_bounds__(256) void sf_partial_kernel(SfParams p) {
  __shared__ __attribute__((aligned(16))) float4 tile[2][2 * kPos];  // double buffered: [channel quad of the chunk][position]
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
  const int b = blockIdx.z / p.n_groups, grp = blockIdx.z - b * p.n_groups;
  const long long plane = (long long)p.H * p.W;
  // chunk range of this group
  const int per = p.n_chunks / p.n_groups, extra = p.n_chunks - per * p.n_groups;
  const int ch_lo = grp * per + min(grp, extra), ch_hi = ch_lo + per + (grp < extra ? 1 : 0);
  float acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = 0.f;
  // this thread's staging items (680 items over 256 threads: up to 3), fixed across chunks
  int it_pos[3], it_quad[3];
  long long it_off[3];
  bool it_ok[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int it = threadIdx.x + 256 * k;
    const int quad = it / kPos, pos = it - quad * kPos;
    const int py = pos / kPW, px = pos - py * kPW;
    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
    it_pos[k] = it;
    it_quad[k] = it < 2 * kPos ? quad : 0;
    it_ok[k] = it < 2 * kPos && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
    it_off[k] = it_ok[k] ? (long long)yy * p.W + xx : 0;   // out-of-image items read pixel 0 and are zeroed
  }
  // staging split in two: the loads of chunk c+1 are issued before chunk c's arithmetic and committed to the other LDS
  // buffer after it (T14: issue early, write late); loads are unconditional (clamped address) and zeroed by selects
  const float* __restrict__ sA = p.src[0];
  const float* __restrict__ sB = p.src[1];
  const float* __restrict__ sC = p.src[2];
  float4 st[3];
  int st_cl[3];
  auto load_chunk = [&](int chunk) {
    int s = 0, c0 = chunk * 8;
    if (p.n_src > 1 && c0 >= p.c[0]) { c0 -= p.c[0]; s = 1; }
    if (s == 1 && p.n_src > 2 && c0 >= p.c[1]) { c0 -= p.c[1]; s = 2; }
    const float* __restrict__ xs = (s == 0 ? sA : (s == 1 ? sB : sC)) + (long long)b * p.c[s] * plane;
    const long long last = (long long)(p.c[s] - 1) * plane;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int cq = c0 + 4 * it_quad[k];
      const int cl = p.c[s] - cq;
      const float* __restrict__ g = xs + it_off[k];
      float4 v;
      v.x = g[min((long long)cq * plane, last)];
      v.y = g[min((long long)(cq + 1) * plane, last)];
      v.z = g[min((long long)(cq + 2) * plane, last)];
      v.w = g[min((long long)(cq + 3) * plane, last)];
      st[k] = v;      // raw: the zeroing selects run at commit time, so nothing waits for these loads before the arithmetic
      st_cl[k] = cl;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float4 v = st[k];
      const bool ok = it_ok[k];
      const int cl = st_cl[k];
      v.x = (ok && cl > 0) ? v.x : 0.f;
      v.y = (ok && cl > 1) ? v.y : 0.f;
      v.z = (ok && cl > 2) ? v.z : 0.f;
      v.w = (ok && cl > 3) ? v.w : 0.f;
      if (it_pos[k] < 2 * kPos) tile[buf][it_pos[k]] = v;
    }
  };
  if (ch_lo < ch_hi) { load_chunk(ch_lo); commit(0); }
  for (int chunk = ch_lo; chunk < ch_hi; ++chunk) {
    const int buf = (chunk - ch_lo) & 1;
    __syncthreads();                                  // chunk's tile visible; the other buffer's readers are done
    const bool more = chunk + 1 < ch_hi;
    if (more) load_chunk(chunk + 1);                  // in flight under this chunk's arithmetic
    const float4* tl = tile[buf];
    const int ctr = (ty + 1) * kPW + tx + 1;
    const float4 a0 = tl[ctr], a1 = tl[kPos + ctr];
    int j = 0;
#pragma unroll
    for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
      for (int ox = -1; ox <= 1; ++ox) {
        if (oy == 0 && ox == 0) continue;
        const int q = ctr + oy * kPW + ox;
        const float4 n0 = tl[q], n1 = tl[kPos + q];
        float t = acc[j];
        t = fmaf(a0.x, n0.x, t); t = fmaf(a0.y, n0.y, t); t = fmaf(a0.z, n0.z, t); t = fmaf(a0.w, n0.w, t);
        t = fmaf(a1.x, n1.x, t); t = fmaf(a1.y, n1.y, t); t = fmaf(a1.z, n1.z, t); t = fmaf(a1.w, n1.w, t);
        acc[j] = t;
        ++j;
      }
    float ss = acc[8];
    ss = fmaf(a0.x, a0.x, ss); ss = fmaf(a0.y, a0.y, ss); ss = fmaf(a0.z, a0.z, ss); ss = fmaf(a0.w, a0.w, ss);
    ss = fmaf(a1.x, a1.x, ss); ss = fmaf(a1.y, a1.y, ss); ss = fmaf(a1.z, a1.z, ss); ss = fmaf(a1.w, a1.w, ss);
    acc[8] = ss;
    if (more) commit(buf ^ 1);
  }
  const int y = y0 + ty, x = x0 + tx;
  if (y >= p.H || x >= p.W) return;
  float* op = p.part + ((long long)(b * p.n_groups + grp) * 9) * plane + (long long)y * p.W + x;
#pragma unroll
  for (int j = 0; j < 9; ++j) op[(long long)j * plane] = acc[j];
}

__global__ __launch_bounds__(256) void sf_finish_kernel(SfParams p) {
  const long long plane = (long long)p.H * p.W;
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= plane * p.B) return;
  const long long b = pix / plane;
  const int rem = (int)(pix - b * plane);
  const int y = rem / p.W, x = rem - y * p.W;
  const float* __restrict__ pp = p.part + b * p.n_groups * 9 * plane;
  // every load of the pixel is issued before the first use (a per-neighbour loop over the groups serialises ~70 round trips):
  // up to 4 channel groups x (9 squared norms + 8 dots), addresses clamped, dead groups zeroed by selects
  int nrem[9];
  bool ok[9];
  {
    int j = 0;
#pragma unroll
    for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
      for (int ox = -1; ox <= 1; ++ox) {
        const int yy = y + oy, xx = x + ox;
        ok[j] = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        nrem[j] = ok[j] ? yy * p.W + xx : rem;
        ++j;
      }
  }
  float ssv[9], dot[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { ssv[j] = 0.f; dot[j] = 0.f; }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int ge = min(g, p.n_groups - 1);
    const float* __restrict__ pg = pp + (long long)ge * 9 * plane;
    float a[9], d[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) a[j] = pg[8 * plane + nrem[j]];
#pragma unroll
    for (int j = 0; j < 9; ++j) d[j] = j == 4 ? 0.f : pg[(long long)(j < 4 ? j : j - 1) * plane + rem];
    const bool live = g < p.n_groups;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      ssv[j] += live ? a[j] : 0.f;
      dot[j] += live ? d[j] : 0.f;
    }
  }
  const float n0 = fmaxf(sqrtf(ssv[4]), 1e-12f);   // F.normalize: x / max(||x||_2, eps)
  float* op = p.aff + b * 8 * plane + rem;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    if (j == 4) continue;
    // sum_c (x_c(p)/n0)(x_c(q)/nq) = (sum_c x_c(p) x_c(q)) / (n0 nq): one rounding of the final quotient instead of one
    // division per channel (liif.py:439-441); clamp(min=0); zero-padded neighbours contribute 0
    const float nq = fmaxf(sqrtf(ssv[j]), 1e-12f);
    op[(long long)(j < 4 ? j : j - 1) * plane] = ok[j] ? fmaxf(dot[j] / (n0 * nq), 0.f) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// first MLP layer at LOW resolution, channels-last result
// ---------------------------------------------------------------------------------------------------------------------
struct LowresParams {
  const float* src[3];  // NCHW fp32 pieces of the concatenated input, channel counts multiples of 8 except the last
  int c[3];
  int n_src;
  const _Float16* wimg; // as_liif_lowres_pack: [ks][nt][hi|lo][lane][8] fp16 fragments of W[:, koff : koff+K]
  int K, ksteps;
  float* out;           // [B][P][128]
  int B;
  long long P, total;   // pixels per image, B*P
  int tiles, iters;     // 32-pixel tiles; groups of 4 tiles per block
};
constexpr int kLowresKs = 12;  // up to 192 input channels

struct LowresPackParams {
  const float* w;  // [128][ldw]
  int ldw, koff, K, ksteps;
  _Float16* img;
};

__global__ __launch_bounds__(256) void liif_lowres_pack_kernel(LowresPackParams p) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.ksteps * 8 * 512) return;
  const int blk = idx >> 9, lane = (idx >> 3) & 63, j = idx & 7;
  const int hl = blk & 1, nt = (blk >> 1) & 3, ks = blk >> 3;
  const int c = lane & 31, half = lane >> 5;
  const int k = 16 * ks + 8 * half + j;
  const float v = k < p.K ? p.w[(long long)(32 * nt + c) * p.ldw + p.koff + k] : 0.f;
  const float x = __builtin_amdgcn_fmed3f(v, -kF16Max, kF16Max);
  const _Float16 hk = (_Float16)x;
  p.img[idx] = hl == 0 ? hk : (_Float16)((x - (float)hk) * 2048.f);
}

struct F8 { float v[8]; };

// the 8 activations of lane half `half` for k-step ks (channels [16 ks + 8 half, +8) of the concatenated input)
__device__ __forceinline__ F8 lowres_issue(const LowresParams& p, const float* __restrict__ src0, const float* __restrict__ src1,
                                           const float* __restrict__ src2, int c01, int ks, int half, long long b, long long pp) {
  const int kb = 16 * ks;
  const bool s0 = kb < p.c[0], s1 = kb < c01;
  const float* __restrict__ sp = s0 ? src0 : (s1 ? src1 : src2);
  const int cs = s0 ? p.c[0] : (s1 ? p.c[1] : p.c[2]);
  const int chb = kb - (s0 ? 0 : (s1 ? p.c[0] : c01));
  const int hoff = (chb + 16 <= cs) ? 8 * half : 0;
  const float* __restrict__ g = sp + (b * cs + chb + hoff) * p.P + pp;
  F8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r.v[j] = g[(long long)j * p.P];
  return r;
}

__device__ __forceinline__ bool lowres_live(const LowresParams& p, int c01, int ks, int half) {
  const int kb = 16 * ks;
  const bool s0 = kb < p.c[0], s1 = kb < c01;
  const int cs = s0 ? p.c[0] : (s1 ? p.c[1] : p.c[2]);
  const int chb = kb - (s0 ? 0 : (s1 ? p.c[0] : c01));
  return kb + 8 * half < p.K && chb + 8 * half < cs;
}

// block = 8 waves: wave w computes output channels [64 blockIdx.y + 32 (w&1), +32) of pixel tile (4 * group + (w>>1)); the
// block's half of the weight fragments sits in LDS (KS x 4 KB: three blocks per CU at K = 184) for `iters` tile groups
template <int KS>
__global__ __launch_bounds__(512) void liif_lowres_cl_kernel(LowresParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt0 = 2 * blockIdx.y;
  {
    // image order [ks][nt][hi|lo][lane]: this block needs nt0, nt0+1 -> per k-step one contiguous 4 KB run
    const uint4* src = reinterpret_cast<const uint4*>(p.wimg);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int i = threadIdx.x; i < KS * 256; i += 512) dst[i] = src[(i >> 8) * 512 + nt0 * 128 + (i & 255)];
  }
  __syncthreads();
  const half8* W0 = reinterpret_cast<const half8*>(smem);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntl = wave & 1, sub = wave >> 1;
  const int c = lane & 31, half = lane >> 5;
  const int c01 = p.c[0] + p.c[1];
  // no-alias views: without them every load waits for the previous tile's output stores (may-alias ordering)
  const float* __restrict__ src0 = p.src[0];
  const float* __restrict__ src1 = p.src[1];
  const float* __restrict__ src2 = p.src[2];
  float* __restrict__ outp = p.out;
  float amax = 0.f;
  for (int it = 0; it < p.iters; ++it) {
    const int tile = (blockIdx.x * p.iters + it) * 4 + sub;
    if (tile >= p.tiles) break;
    int opaque = 0;
    asm volatile("" : "+v"(opaque));   // keep the LDS fragment reads inside the loop (see liif_tail_kernel)
    const half8* W = W0 + opaque;
    long long pix = (long long)tile * 32 + c;
    if (pix >= p.total) pix = p.total - 1;
    const long long b = pix / p.P, pp = pix - b * p.P;
    f32x16 acc_h, acc_x;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc_h[i] = 0.f; acc_x[i] = 0.f; }
    // a 16-channel k-step lies inside ONE source (every source but the last holds a multiple of 16 channels), so the source
    // is wave-uniform.  Loads are unconditional: a half wave whose 8 channels lie beyond the source (the 8-channel affinity
    // block fills half a k-step) re-reads the other half's channels and is zeroed by a select — a `cond ? load : 0` form
    // compiles to a branch and a full wait per load.  Two k-steps of loads are in flight ahead of the MFMAs; the scheduling
    // barriers keep the compiler from hoisting all 96 address computations to the top (which spills).
    // DEPTH k-steps of loads in flight ahead of the MFMAs (a ring of register sets with compile-time indices): the kernel is
    // latency-bound (a tile is 36 MFMAs), so the ring covers an HBM round trip with the first k-steps' arithmetic
    constexpr int DEPTH = KS < 6 ? KS : 6;
    F8 ring[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring[d] = lowres_issue(p, src0, src1, src2, c01, d, half, b, pp);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const F8 cur = ring[ks % DEPTH];
      float v[8];
      const bool ok = lowres_live(p, c01, ks, half);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ok ? cur.v[j] : 0.f;
      __builtin_amdgcn_sched_barrier(0);
      if (ks + DEPTH < KS) ring[ks % DEPTH] = lowres_issue(p, src0, src1, src2, c01, ks + DEPTH, half, b, pp);
      __builtin_amdgcn_sched_barrier(0);
      half8 ah, al;
      split8(v, ah, al, amax);
      const half8 bh = W[((ks * 2 + ntl) * 2) * 64 + lane], bl = W[((ks * 2 + ntl) * 2 + 1) * 64 + lane];
      acc_h = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc_h, 0, 0, 0);
      acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc_x, 0, 0, 0);
      acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc_x, 0, 0, 0);
    }
    // C[row = pixel][col = output channel]: one register = 32 consecutive channels of one pixel (128 B per half wave)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long long opix = (long long)tile * 32 + acc_row(i, half);
      if (opix < p.total) outp[opix * kHid1 + 32 * (nt0 + ntl) + c] = acc_h[i] + acc_x[i] * (1.f / 2048.f);
    }
  }
  note_overflow(amax);
}

// ---------------------------------------------------------------------------------------------------------------------
// weight image of the tail kernel (built once per weight version)
// ---------------------------------------------------------------------------------------------------------------------
struct PackParams {
  const float* wrel;  // [128][2*n_src]
  const float* b1;    // [128] or null
  const float* w2;    // [64][128]
  const float* b2;    // [64] or null
  const float* w3;    // [64][64]
  const float* b3;
  const float* w4;    // [9][64]
  const float* b4;
  int n_src;
  _Float16* image;    // kImageBytes
};

__device__ __forceinline__ int perm_k(int ks, int half, int j) { return 16 * ks + 8 * (j >> 2) + 4 * half + (j & 3); }

__global__ __launch_bounds__(256) void liif_tail_pack_kernel(PackParams p) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // one fp16 element of the fragment blocks
  if (idx < kFragBlocks * 512) {
    const int blk = idx >> 9, lane = (idx >> 3) & 63, j = idx & 7;
    const int r = lane & 31, half = lane >> 5;
    float v = 0.f;
    int hl;
    if (blk < kBlkW2) {  // relative-coordinate / bias product: A[channel 32 mt + r][k = 8 half + j], k<4: wrel, k==4: b1
      const int mt = blk >> 1, k = 8 * half + j, ch = 32 * mt + r;
      hl = blk & 1;
      if (k < 2 * p.n_src) v = p.wrel[ch * 2 * p.n_src + k];
      else if (k == 4) v = p.b1 ? p.b1[ch] : 0.f;
    } else if (blk < kBlkW3) {
      const int t = blk - kBlkW2;
      hl = t & 1;
      const int mt = (t >> 1) & 1, ks = t >> 2;
      v = p.w2[(32 * mt + r) * kHid1 + perm_k(ks, half, j)];
    } else if (blk < kBlkW4) {
      const int t = blk - kBlkW3;
      hl = t & 1;
      const int mt = (t >> 1) & 1, ks = t >> 2;
      v = p.w3[(32 * mt + r) * kHid2 + perm_k(ks, half, j)];
    } else {
      const int t = blk - kBlkW4;
      hl = t & 1;
      const int ks = t >> 1;
      v = r < kOut ? p.w4[r * kHid3 + perm_k(ks, half, j)] : 0.f;
    }
    const float x = __builtin_amdgcn_fmed3f(v, -kF16Max, kF16Max);
    const _Float16 hk = (_Float16)x;
    p.image[idx] = hl == 0 ? hk : (_Float16)((x - (float)hk) * 2048.f);
  } else if (idx < kFragBlocks * 512 + kBiasFloats) {
    const int i = idx - kFragBlocks * 512;
    float* bias = reinterpret_cast<float*>(p.image + kFragBlocks * 512);
    float v;
    if (i < kHid2) v = p.b2 ? p.b2[i] : 0.f;
    else if (i < kHid2 + kHid3) v = p.b3 ? p.b3[i - kHid2] : 0.f;
    else v = (i - kHid2 - kHid3 < kOut && p.b4) ? p.b4[i - kHid2 - kHid3] : 0.f;
    bias[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// the per-query tail
// ---------------------------------------------------------------------------------------------------------------------
struct TailParams {
  const float* u[2];       // channels-last first-layer maps [B][H_s*W_s][128]
  float* coord;            // [B][Q][2] (row, col); clamped in place when clamp_inplace
  const _Float16* image;   // liif_tail_pack_kernel
  const float* disp;       // [B][Hd*Wd]
  const float* scale;      // [B] or null: disparity x 4 x scale_b
  float* out;              // [B][Q]
  float* logits;           // optional [B][9][Q]
  int B, Q, n_src, clamp_inplace;
  int H[2], W[2], Hd, Wd;
  long long total;
  int tiles, tpw;
  float lo, hi;
  float c0y[2], sy[2], c0x[2], sx[2];
  // DIRECT1: u[1] holds the second input's RAW channels-last rows [B][H1*W1][kDirectCP] (structure feature, zero padded) and
  // image1 the fragments of W1[:, its columns] (as_liif_lowres_pack layout): its first-layer product is taken per query
  const _Float16* image1;
  // > 0: u[1] holds B1 batch elements and query batch b reads element b % B1 (training: the loop-invariant second input is
  // shared by the n evaluations (iteration, sample) of one batched call, models/base.py::_upsample_batched)
  int B1;
  // Query ORDER hint (device memory, may be null): the row length R of a raster-ordered query grid (as_liif_query_rows).  With a
  // sensible R the waves of a block take the SAME 32-column tile of consecutive query rows (a 4- or 8-row x 32-column patch per
  // block step) instead of 32-query runs of one row each: the rows of the low-resolution tables a patch shares (one 1/4-resolution
  // row per 4 query rows at scale 1) are fetched once per block step instead of once per block that happens to pass by.  Any
  // value is a valid order — every query is processed exactly once either way — so the hint cannot change a result.
  const int* row_len;
};
constexpr int kDirectCP = 48;                       // row pitch (floats) of a direct source: 3 k-steps
constexpr int kDirectKS = kDirectCP / 16;
constexpr int kImage1Bytes = kDirectKS * 8 * 1024;  // [ks][mt(4)][hi|lo][lane][8]

#define AS_MFMA3(AH, AL, BH, BL, ACCH, ACCX)                               \
  ACCH = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACCH, 0, 0, 0);    \
  ACCX = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, ACCX, 0, 0, 0);    \
  ACCX = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, ACCX, 0, 0, 0);

// per-tile state that is prepared one tile ahead (coordinates, source rows, first gather group, disparity neighbours)
struct TileCtx {
  long long t;
  bool valid;
  const float4* __restrict__ up0;
  const float4* __restrict__ up1;
  half8 xh, xl;          // B operand of the relative-coordinate / bias product
  f32x16 ga;             // channels [0, 32) of the gathered source-0 row, in accumulator order (C operand of the T product)
  float4 g1[4];          // the same 16-B groups of the source-1 row
  half8 dh[3], dl[3];    // DIRECT1: the source-1 row (48 raw channels) split into the B operands of its three k-steps
  float dn[5];           // this lane's disparity neighbours (x 4 x scale): k = 4 half + i (i < 4), k = 8 (i = 4, half 0)
  int pix[2];            // nearest pixel (row * W + column) of this lane's query in source 0 / 1 (the training backward scatters there)
  int bsrc1;             // batch element of source 1 this query reads (b % B1)
  float rel[4];          // the relative coordinates as fp32 (the backward's wrel gradient)
};

template <int NSRC, bool DIRECT1 = false>
__device__ __forceinline__ void tile_prepare(const TailParams& p, const float* __restrict__ u0p, const float* __restrict__ u1p,
                                             const float* __restrict__ dispp, long long t, bool valid, float cr, float cc,
                                             int half, float& amax, TileCtx& x) {
  x.t = t;
  x.valid = valid;
  const int b = (int)(t / p.Q);
  const float crc = fminf(fmaxf(cr, p.lo), p.hi), ccc = fminf(fmaxf(cc, p.lo), p.hi);
  if (p.clamp_inplace && valid && half == 0) {   // the reference clamps the caller's tensor in place (submodule.py:366)
    p.coord[t * 2 + 0] = crc;
    p.coord[t * 2 + 1] = ccc;
  }
  float rel[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NSRC; ++s) {
    const int iy = nearest_idx(crc, p.H[s]), ix = nearest_idx(ccc, p.W[s]);
    const int pitch = (DIRECT1 && s == 1) ? kDirectCP : kHid1;
    const int bs = (s == 1 && p.B1 > 0) ? b % p.B1 : b;
    const float4* __restrict__ up =
        reinterpret_cast<const float4*>((s ? u1p : u0p) + (((long long)bs * p.H[s] + iy) * p.W[s] + ix) * pitch) + ((DIRECT1 && s == 1) ? 2 * half : half);
    if (s == 0) x.up0 = up; else x.up1 = up;
    x.pix[s] = iy * p.W[s] + ix;
    if (s == 1) x.bsrc1 = bs;
    const float qy = __fadd_rn(p.c0y[s], __fmul_rn(p.sy[s], (float)iy));
    const float qx = __fadd_rn(p.c0x[s], __fmul_rn(p.sx[s], (float)ix));
    rel[2 * s] = __fmul_rn(__fsub_rn(cr, qy), (float)p.H[s]);          // (coord_unclamped - cell centre) * (H, W)
    rel[2 * s + 1] = __fmul_rn(__fsub_rn(cc, qx), (float)p.W[s]);
  }
  if (NSRC == 1) { x.up1 = x.up0; x.pix[1] = 0; x.bsrc1 = b; }
#pragma unroll
  for (int k = 0; k < 4; ++k) x.rel[k] = rel[k];
  if (dispp) {  // kernel-uniform (the training forward asks for the logits only)
    const int iy = nearest_idx(crc, p.Hd), ix = nearest_idx(ccc, p.Wd);
    const float* __restrict__ dp = dispp + (long long)b * p.Hd * p.Wd;
    const float sc = p.scale ? p.scale[b] : 1.f;
    const float four = p.scale ? 4.f : 1.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int k = i < 4 ? 4 * half + i : 8;
      const int yy = iy + k / 3 - 1, xx = ix + k % 3 - 1;
      const bool in = yy >= 0 && yy < p.Hd && xx >= 0 && xx < p.Wd;
      const float d = dp[(long long)min(max(yy, 0), p.Hd - 1) * p.Wd + min(max(xx, 0), p.Wd - 1)];   // unconditional load
      x.dn[i] = in ? __fmul_rn(__fmul_rn(d, four), sc) : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 5; ++i) x.dn[i] = 0.f;
  }
  // B operand of the relative-coordinate product: k = 0..3 rel, k = 4 the constant 1 (bias row); other k = 0
  float xv[8] = {rel[0], rel[1], rel[2], rel[3], 1.f, 0.f, 0.f, 0.f};
  if (half) {
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = 0.f;
  }
  split8(xv, x.xh, x.xl, amax);
}

// registers 4q..4q+3 of an accumulator tile <- one 16-B group of a channels-last row
__device__ __forceinline__ void put4(f32x16& a, int q, const float4 v) {
  a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
}

template <int NSRC, bool DIRECT1 = false>
__device__ __forceinline__ void tile_first_gather(TileCtx& x, float& amax) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    put4(x.ga, i, x.up0[2 * i]);
    x.g1[i] = (NSRC > 1 && !DIRECT1) ? x.up1[2 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if constexpr (DIRECT1) {
    // lane half h supplies k = 8 h .. 8 h + 7 of a k-step = row floats [16 ks + 8 h, +8) = float4 #(4 ks + 2 h), #(4 ks + 2 h + 1)
    // (up1 already points at float4 #2h)
    float4 r[kDirectKS][2];
#pragma unroll
    for (int ks = 0; ks < kDirectKS; ++ks) { r[ks][0] = x.up1[4 * ks]; r[ks][1] = x.up1[4 * ks + 1]; }
#pragma unroll
    for (int ks = 0; ks < kDirectKS; ++ks) {
      const float v[8] = {r[ks][0].x, r[ks][0].y, r[ks][0].z, r[ks][0].w, r[ks][1].x, r[ks][1].y, r[ks][1].z, r[ks][1].w};
      split8(v, x.dh[ks], x.dl[ks], amax);
    }
  }
}

// DIRECT1 (two inputs): the second input's first-layer rows are not precomputed at its resolution — at 1/2 resolution that table is
// 4 x the 1/4-resolution one (66.8 MB at 960x540, gathered ~2.3 x) — the tail gathers its 40 RAW channels (192-B rows, 25 MB)
// and takes the product with W1's columns here: three more k-steps per 32-channel tile into the accumulators the relative-
// coordinate product already owns (no VALU add of a gathered row).  Its fragments (24 KB) sit behind the main image, so the
// block is 8 waves sharing one 90 KB copy instead of two 4-wave blocks with 66 KB each.
template <int NSRC, bool DIRECT1 = false>
__global__ __launch_bounds__(DIRECT1 ? 512 : 256, DIRECT1 ? 1 : 2) void liif_tail_kernel(TailParams p) {
  constexpr int NT = DIRECT1 ? 512 : 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  {
    const uint4* src = reinterpret_cast<const uint4*>(p.image);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int i = threadIdx.x; i < kImageBytes / 16; i += NT) dst[i] = src[i];
    if constexpr (DIRECT1) {
      const uint4* src1 = reinterpret_cast<const uint4*>(p.image1);
      uint4* dst1 = reinterpret_cast<uint4*>(smem + ((kImageBytes + 15) / 16) * 16);
      for (int i = threadIdx.x; i < kImage1Bytes / 16; i += NT) dst1[i] = src1[i];
    }
  }
  __syncthreads();
  const half8* W0 = reinterpret_cast<const half8*>(smem);
  const half8* W1d0 = reinterpret_cast<const half8*>(smem + ((kImageBytes + 15) / 16) * 16);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, half = lane >> 5;
  // blocks b and b+8 share an XCD (its L2): give every XCD one contiguous range of queries so that the low-resolution
  // rows a query neighbourhood shares are fetched into ONE L2
  const int nb = gridDim.x;
  const int mapped = (nb & 7) == 0 ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const long long gw = (long long)mapped * (NT / 64) + wave;
  float amax = 0.f;
  int imax = 0;
  // no-alias views (otherwise a tile's gathers wait for the previous tile's stores)
  const float* __restrict__ u0p = p.u[0];
  const float* __restrict__ u1p = p.u[1];
  const float* __restrict__ dispp = p.disp;
  const float* __restrict__ coordp = p.coord;
  float* __restrict__ outp = p.out;
  float* __restrict__ logitsp = p.logits;

  // order of the queries over (block, wave, step): raster patches when the caller's hint describes a grid, 32-query runs else.
  // All of it is wave-uniform 32-bit scalar arithmetic: one division per wave, then increments.
  constexpr int G = NT / 64;  // waves per block = query rows per patch
  int R = 0, rows = 0, ct_n = 1, rg = 0, ct = 0;
  if (p.row_len) R = __builtin_amdgcn_readfirstlane(*p.row_len);
  const bool patch = R >= 32 && R <= (1 << 20) && p.total < (1ll << 31) && (int)p.total % R == 0;
  long long tile0;
  int ntile;
  if (patch) {
    rows = (int)p.total / R;
    ct_n = (R + 31) >> 5;
    const int steps_total = ((rows + G - 1) / G) * ct_n;           // patches of G rows x 32 columns, row-major
    const int per = (steps_total + nb - 1) / nb;                   // consecutive patches per block
    const int s0 = mapped * per;
    if (s0 >= steps_total) return;
    ntile = min(per, steps_total - s0);
    rg = s0 / ct_n;
    ct = s0 - rg * ct_n;
    tile0 = 0;
  } else {
    tile0 = gw * p.tpw;
    if (tile0 >= p.tiles) return;
    ntile = (int)min((long long)p.tpw, (long long)p.tiles - tile0);
  }
  for (int ti = 0; ti < ntile; ++ti) {
    // the weight image is re-read from LDS for every tile: an opaque zero offset keeps the compiler from hoisting the 56
    // fragment reads (224 registers) out of the tile loop
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    const half8* W = W0 + opaque;
    const half8* W1d = W1d0 + opaque;
    const float* bias = reinterpret_cast<const float*>(smem + kFragBlocks * 1024) + opaque;
    TileCtx cur;
    {
      long long t;
      bool valid;
      if (patch) {
        const int row = rg * G + wave, col = ct * 32 + c;
        valid = row < rows && col < R;
        t = (long long)row * R + col;
        if (++ct == ct_n) { ct = 0; ++rg; }
      } else {
        t = (tile0 + ti) * 32 + c;
        valid = t < p.total;
      }
      if (!valid) t = p.total - 1;
      tile_prepare<NSRC, DIRECT1>(p, u0p, u1p, dispp, t, valid, coordp[t * 2], coordp[t * 2 + 1], half, amax, cur);
      tile_first_gather<NSRC, DIRECT1>(cur, amax);
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 a2h[2], a2x[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) a2h[m][i] = bias[32 * m + acc_row(i, half)];
    // First layer.  A 32x32 accumulator tile holds, for this lane's query, channels 32 t4 + (i&3) + 8(i>>2) + 4 half in
    // register i — exactly the 16-B groups #(2i) (+ half) of the channels-last rows.  So the gathered source-0 row is loaded
    // straight into the C operand of the relative-coordinate product (T = Wrel.rel + b1 + u0[n0]); only the source-1 row is
    // added on the vector ALU:  h1 = relu(T_h + T_x/2048 + u1[n1]).  The next 32-channel group of source 0 is requested at
    // the start of a stage, the source-1 groups of k-step ks+2 as soon as those of ks are consumed; the scheduling barriers
    // pin that order (left alone the compiler hoists all 32 loads to the top of the tile, which spills).
    f32x16 ga = cur.ga, gn = zero16;
    float4 g1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) g1[i] = DIRECT1 ? make_float4(0.f, 0.f, 0.f, 0.f) : cur.g1[i];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      if (t4 < 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) put4(gn, i, cur.up0[2 * (4 * (t4 + 1) + i)]);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x16 th, tx;
      {
        const half8 ah = W[(kBlkRel + 2 * t4) * 64 + lane], al = W[(kBlkRel + 2 * t4 + 1) * 64 + lane];
        th = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur.xh, ga, 0, 0, 0);
        tx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur.xl, zero16, 0, 0, 0);
        tx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, cur.xh, tx, 0, 0, 0);
        if constexpr (DIRECT1) {
#pragma unroll
          for (int kd = 0; kd < kDirectKS; ++kd) {  // + W1[:, source-1 columns] . (its raw row): image1 [ks][mt][hi|lo][lane]
            const half8 dah = W1d[((kd * 4 + t4) * 2) * 64 + lane], dal = W1d[((kd * 4 + t4) * 2 + 1) * 64 + lane];
            AS_MFMA3(dah, dal, cur.dh[kd], cur.dl[kd], th, tx)
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int ks = 2 * t4 + s;
        const float4 qa = g1[2 * s], qb = g1[2 * s + 1];
        const float q8[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = relu_bits(DIRECT1 ? fmaf(tx[8 * s + j], 1.f / 2048.f, th[8 * s + j]) : fmaf(tx[8 * s + j], 1.f / 2048.f, th[8 * s + j]) + q8[j]);
        __builtin_amdgcn_sched_barrier(0);
        if (t4 < 3 && NSRC > 1 && !DIRECT1) {  // this k-step's source-1 registers are free: fetch k-step ks + 2
#pragma unroll
          for (int i = 0; i < 2; ++i) g1[2 * s + i] = cur.up1[2 * (2 * (ks + 2) + i)];
        }
        __builtin_amdgcn_sched_barrier(0);
        half8 bh, bl;
        split8_pos(v, bh, bl, imax);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int blk = kBlkW2 + (ks * 2 + m) * 2;
          const half8 ah = W[blk * 64 + lane], al = W[(blk + 1) * 64 + lane];
          a2h[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a2h[m], 0, 0, 0);
          a2x[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ks == 0 ? zero16 : a2x[m], 0, 0, 0);
          a2x[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, a2x[m], 0, 0, 0);
        }
      }
      if (t4 < 3) ga = gn;
    }
    // layer 3: 64 -> 64
    f32x16 a3h[2], a3x[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) a3h[m][i] = bias[kHid2 + 32 * m + acc_row(i, half)];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int mt = ks >> 1, s = ks & 1;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = relu_bits(fmaf(a2x[mt][8 * s + j], 1.f / 2048.f, a2h[mt][8 * s + j]));
      half8 bh, bl;
      split8_pos(v, bh, bl, imax);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int blk = kBlkW3 + (ks * 2 + m) * 2;
        const half8 ah = W[blk * 64 + lane], al = W[(blk + 1) * 64 + lane];
        a3h[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a3h[m], 0, 0, 0);
        a3x[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ks == 0 ? zero16 : a3x[m], 0, 0, 0);
        a3x[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, a3x[m], 0, 0, 0);
      }
    }
    // layer 4: 64 -> 9 (rows 9..31 of the tile carry zero weights)
    f32x16 a4h, a4x;
#pragma unroll
    for (int i = 0; i < 16; ++i) a4h[i] = bias[kHid2 + kHid3 + acc_row(i, half)];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int mt = ks >> 1, s = ks & 1;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = relu_bits(fmaf(a3x[mt][8 * s + j], 1.f / 2048.f, a3h[mt][8 * s + j]));
      half8 bh, bl;
      split8_pos(v, bh, bl, imax);
      const int blk = kBlkW4 + ks * 2;
      const half8 ah = W[blk * 64 + lane], al = W[(blk + 1) * 64 + lane];
      a4h = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a4h, 0, 0, 0);
      a4x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ks == 0 ? zero16 : a4x, 0, 0, 0);
      a4x = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, a4x, 0, 0, 0);
    }
    // logits: half 0 holds rows 0..3 (registers 0..3) and row 8 (register 4), half 1 rows 4..7 (registers 0..3)
    float lg[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) lg[i] = a4h[i] + a4x[i] * (1.f / 2048.f);
    if (logitsp && cur.valid) {
      const int b = (int)(cur.t / p.Q);
      const int q = (int)(cur.t - (long long)b * p.Q);
      float* lp = logitsp + ((long long)b * kOut) * p.Q + q;
#pragma unroll
      for (int i = 0; i < 4; ++i) lp[(long long)(4 * half + i) * p.Q] = lg[i];
      if (!half) lp[8ll * p.Q] = lg[4];
    }
    float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
    if (!half) mx = fmaxf(mx, lg[4]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float ssum = 0.f, dsum = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const float e = (i < 4 || !half) ? expf(lg[i] - mx) : 0.f;  // row 8 lives in half 0 only
      ssum += e;
      dsum = fmaf(e, cur.dn[i], dsum);
    }
    ssum += __shfl_xor(ssum, 32);
    dsum += __shfl_xor(dsum, 32);
    if (outp && cur.valid && !half) outp[cur.t] = dsum / ssum;
  }
  note_overflow(fmaxf(amax, __builtin_bit_cast(float, imax)));
}

// ---------------------------------------------------------------------------------------------------------------------
// Training: backward of the per-query MLP (liif.py:9-25 under autograd) without its activations in HBM on the way there.
// The training forward is liif_tail_kernel with logits only (no activation is saved); liif_mlp_bwd_kernel RECOMPUTES a tile's
// h1 / h2 / h3 with the forward's own instruction sequence (same values, bit for bit — the ReLU masks are the forward's),
// then runs the data-gradient chain on the matrix cores with the same operand chaining, transposed weights:
//     d3 = [h3 > 0] . W4^T dlogits     d2 = [h2 > 0] . W3^T d3     d1 = [h1 > 0] . W2^T d2
// and emits what the step's batched weight-gradient launches and the first layer's scatter-add read: h1, h2, h3 (post-ReLU),
// d3, d2, d1 (gradients w.r.t. the pre-activations), each [B][C][Q] fp32, written once.  Against the layer-by-layer form
// (four 1x1 convolutions + three ReLU-backward passes + three data-gradient convolutions over [B,128|64,Q] tensors) the
// per-query activations cross HBM once instead of ~5 times.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kTBlk4 = 0, kTBlk3 = 4, kTBlk2 = 20, kTBlocks = 52;   // W4^T (2 m x hi|lo), W3^T (4 ks x 2 m x 2), W2^T (4 ks x 4 m x 2)
constexpr int kImageTBytes = kTBlocks * 1024;

struct PackTParams {
  const float* w2;    // [64][128]
  const float* w3;    // [64][64]
  const float* w4;    // [9][64]
  _Float16* image;    // kImageTBytes
};

// A fragments of the TRANSPOSED layers: row = input channel of the layer, k = its output channel in the chaining order
__global__ __launch_bounds__(256) void liif_mlp_bwd_pack_kernel(PackTParams p) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= kTBlocks * 512) return;
  const int blk = idx >> 9, lane = (idx >> 3) & 63, j = idx & 7;
  const int r = lane & 31, half = lane >> 5;
  float v;
  int hl;
  if (blk < kTBlk3) {
    const int m = blk >> 1, rr = perm_k(0, half, j);
    hl = blk & 1;
    v = rr < kOut ? p.w4[rr * kHid3 + 32 * m + r] : 0.f;
  } else if (blk < kTBlk2) {
    const int t = blk - kTBlk3;
    hl = t & 1;
    const int m = (t >> 1) & 1, ks = t >> 2;
    v = p.w3[perm_k(ks, half, j) * kHid2 + 32 * m + r];
  } else {
    const int t = blk - kTBlk2;
    hl = t & 1;
    const int m = (t >> 1) & 3, ks = t >> 3;
    v = p.w2[perm_k(ks, half, j) * kHid1 + 32 * m + r];
  }
  const float x = __builtin_amdgcn_fmed3f(v, -kF16Max, kF16Max);
  const _Float16 hk = (_Float16)x;
  p.image[idx] = hl == 0 ? hk : (_Float16)((x - (float)hk) * 2048.f);
}

struct MlpBwdParams {
  TailParams t;            // sources, coordinates, forward image, sizes (disp / out / logits unused)
  const _Float16* imageT;  // liif_mlp_bwd_pack_kernel
  const float* dlogits;    // [B][9][Q]
  float* h1;               // [B][128][Q]
  float* h2;               // [B][64][Q]
  float* h3;               // [B][64][Q]
  float* d3;               // [B][64][Q]
  float* d2;               // [B][64][Q]
  float* d1;               // [B][128][Q], or null with the first layer's consumers fused (du0 / du1 / dwrel below)
  // fused first-layer backward: d1 is scattered straight into the gradients of the two first-layer maps (NCHW, zero-filled by the
  // launcher; source 1 holds B1 batch elements: every evaluation of the shared input adds into element b % B1) and reduced against
  // the relative coordinates for the wrel columns' gradient — d1 itself never reaches memory
  float* du0;              // [B][128][H0][W0]
  float* du1;              // [B1][128][H1][W1]
  float* dwrel;            // [128][4], atomically accumulated
};
constexpr int kStageRow = 33;                                   // padded row of a wave's 32 x 32 staging tile (conflict-free column reads)
constexpr int kStageFloats = 32 * kStageRow + 4 * 32;           // + the tile's relative coordinates [4][32]

// Diagnostic builds only (tools/variant.sh liif_fused.hip <name> -DAS_ABL_BWD_NO_HD [-DAS_ABL_BWD_NO_ATOMICS]): timing of
// liif_mlp_bwd_kernel without its activation / gradient stores (h1 h2 h3 d3 d2 [d1]) and without the first layer's scatter atomics
// (results are wrong; tools/r06_bwd_ablation.sh).  Never defined in the product build.
#ifdef AS_ABL_BWD_NO_HD
#define AS_BWD_HD_STORE(V, R, VO, SO, AUX) ((void)(V))
#else
#define AS_BWD_HD_STORE(V, R, VO, SO, AUX) __builtin_amdgcn_raw_buffer_store_b32(V, R, VO, SO, AUX)
#endif
#ifdef AS_ABL_BWD_NO_ATOMICS
#define AS_BWD_ATOMIC(V, R, VO, SO, AUX) ((void)(V))
#else
#define AS_BWD_ATOMIC(V, R, VO, SO, AUX) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(V, R, VO, SO, AUX)
#endif
template <bool FUSE1>
__global__ __launch_bounds__(512) void liif_mlp_bwd_kernel(MlpBwdParams P) {
  constexpr int NT = 512;
  const TailParams& p = P.t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kOffT = ((kImageBytes + 15) / 16) * 16;
  {
    const uint4* src = reinterpret_cast<const uint4*>(p.image);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int i = threadIdx.x; i < kImageBytes / 16; i += NT) dst[i] = src[i];
    const uint4* srcT = reinterpret_cast<const uint4*>(P.imageT);
    uint4* dstT = reinterpret_cast<uint4*>(smem + kOffT);
    for (int i = threadIdx.x; i < kImageTBytes / 16; i += NT) dstT[i] = srcT[i];
  }
  __syncthreads();
  const half8* W0 = reinterpret_cast<const half8*>(smem);
  const half8* WT0 = reinterpret_cast<const half8*>(smem + kOffT);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, half = lane >> 5;
  const int nb = gridDim.x;
  const int mapped = (nb & 7) == 0 ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const long long gw = (long long)mapped * (NT / 64) + wave;
  float amax = 0.f;
  int imax = 0;
  const float* __restrict__ u0p = p.u[0];
  const float* __restrict__ u1p = p.u[1];
  const float* __restrict__ coordp = p.coord;
  const long long Q = p.Q;
  // results / logit gradients through buffer descriptors: ONE per-lane byte offset per tensor width (batch element, lane half's
  // 4-channel shift, query), the channel row of a register as the instruction's SCALAR offset — no per-store address registers;
  // lanes past the last query carry an out-of-range offset (the access is dropped by the range check, which ignores soffset)
  const unsigned n128 = (unsigned)((long long)p.B * kHid1 * Q * 4), n64 = (unsigned)((long long)p.B * kHid2 * Q * 4);
  const __amdgpu_buffer_rsrc_t r_dl = __builtin_amdgcn_make_buffer_rsrc((void*)P.dlogits, 0, (int)((long long)p.B * kOut * Q * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_h1 = __builtin_amdgcn_make_buffer_rsrc((void*)P.h1, 0, (int)n128, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_h2 = __builtin_amdgcn_make_buffer_rsrc((void*)P.h2, 0, (int)n64, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_h3 = __builtin_amdgcn_make_buffer_rsrc((void*)P.h3, 0, (int)n64, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_d3 = __builtin_amdgcn_make_buffer_rsrc((void*)P.d3, 0, (int)n64, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_d2 = __builtin_amdgcn_make_buffer_rsrc((void*)P.d2, 0, (int)n64, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_d1 = __builtin_amdgcn_make_buffer_rsrc((void*)P.d1, 0, (int)n128, 0x00020000);
  const unsigned q4 = (unsigned)(Q * 4);  // byte pitch of a channel row
#define AS_ROW(I) (((I) & 3) + 8 * ((I) >> 2))  /* acc_row without the lane half (that part sits in the per-lane offset) */
  constexpr bool fuse1 = FUSE1;  // the first layer's consumers inside this kernel (P.du0 / du1 / dwrel) or d1 written out
  const unsigned hw0 = (unsigned)(p.H[0] * p.W[0]), hw1 = (unsigned)(p.H[1] * p.W[1]);
  const int b1n = p.B1 > 0 ? p.B1 : p.B;
  const __amdgpu_buffer_rsrc_t r_u0 = __builtin_amdgcn_make_buffer_rsrc((void*)P.du0, 0, fuse1 ? (int)((long long)p.B * kHid1 * hw0 * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_u1 = __builtin_amdgcn_make_buffer_rsrc((void*)P.du1, 0, fuse1 ? (int)((long long)b1n * kHid1 * hw1 * 4) : 0, 0x00020000);
  float* stage = reinterpret_cast<float*>(smem + kOffT + kImageTBytes) + wave * kStageFloats;  // this wave's 32 x 33 tile + rel [4][32]
  float accw[4][4];  // lane (channel r of m-tile m, query half): sum over this wave's queries of d1[32 m + r][q] * rel_k(q)
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int k = 0; k < 4; ++k) accw[m][k] = 0.f;

  const long long tile0 = gw * p.tpw;
  if (tile0 >= p.tiles) return;
  const int ntile = (int)min((long long)p.tpw, (long long)p.tiles - tile0);
  for (int ti = 0; ti < ntile; ++ti) {
    int opaque = 0;
    asm volatile("" : "+v"(opaque));  // keeps the fragment reads inside the tile loop (see liif_tail_kernel)
    const half8* W = W0 + opaque;
    const half8* WT = WT0 + opaque;
    const float* bias = reinterpret_cast<const float*>(smem + kFragBlocks * 1024) + opaque;
    TileCtx cur;
    {
      long long t = (tile0 + ti) * 32 + c;
      const bool valid = t < p.total;
      if (!valid) t = p.total - 1;
      tile_prepare<2, false>(p, u0p, u1p, nullptr, t, valid, coordp[t * 2], coordp[t * 2 + 1], half, amax, cur);
      tile_first_gather<2, false>(cur, amax);
    }
    const int b = (int)(cur.t / Q);
    const long long qq = cur.t - (long long)b * Q;   // this lane's query inside its batch element
    const unsigned vo128 = cur.valid ? (unsigned)((((long long)b * kHid1 + 4 * half) * Q + qq) * 4) : 0xFFFFFFF0u;
    const unsigned vo64 = cur.valid ? (unsigned)((((long long)b * kHid2 + 4 * half) * Q + qq) * 4) : 0xFFFFFFF0u;
    const unsigned vo9 = cur.valid ? (unsigned)((((long long)b * kOut + 4 * half) * Q + qq) * 4) : 0xFFFFFFF0u;
    // fused scatter-add: runs of consecutive queries (lanes of one 32-lane half; the halves hold different channels of the same
    // queries) that fall into the same source pixel are pre-summed with a segmented suffix sum and only the head of a run issues
    // the atomic — the scheme of liif_gather_bwd_kernel (the training path sorts the queries by source pixel, ~16 per pixel)
    unsigned same0 = 0u, same1 = 0u, vo_u0 = 0u, vo_u1 = 0u;
    bool head0 = false, head1 = false;
    if constexpr (fuse1) {
      const int key0 = b * (int)hw0 + cur.pix[0], key1 = cur.bsrc1 * (int)hw1 + cur.pix[1];
      const int prev0 = __shfl_up(key0, 1, 32), prev1 = __shfl_up(key1, 1, 32);
      head0 = c == 0 || prev0 != key0;
      head1 = c == 0 || prev1 != key1;
      const unsigned long long hb0 = __ballot(head0), hb1 = __ballot(head1);
      const unsigned after0 = ((unsigned)(half ? (hb0 >> 32) : hb0) >> c) >> 1, after1 = ((unsigned)(half ? (hb1 >> 32) : hb1) >> c) >> 1;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const unsigned msk = (1u << (1 << k)) - 1u;
        if (c + (1 << k) < 32 && (after0 & msk) == 0u) same0 |= 1u << k;
        if (c + (1 << k) < 32 && (after1 & msk) == 0u) same1 |= 1u << k;
      }
      vo_u0 = head0 ? (unsigned)((((long long)b * kHid1 + 4 * half) * hw0 + cur.pix[0]) * 4) : 0xFFFFFFF0u;
      vo_u1 = head1 ? (unsigned)((((long long)cur.bsrc1 * kHid1 + 4 * half) * hw1 + cur.pix[1]) * 4) : 0xFFFFFFF0u;
      if (half == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) stage[32 * kStageRow + k * 32 + c] = cur.valid ? cur.rel[k] : 0.f;
      }
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ================= forward recompute: liif_tail_kernel's sequence, layer for layer =================
    unsigned m1[4] = {0u, 0u, 0u, 0u}, m2[2] = {0u, 0u}, m3[2] = {0u, 0u};  // bit i = [activation in accumulator register i > 0]
    f32x16 a2h[2], a2x[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) a2h[m][i] = bias[32 * m + acc_row(i, half)];
    f32x16 ga = cur.ga, gn = zero16;
    float4 g1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) g1[i] = cur.g1[i];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      if (t4 < 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) put4(gn, i, cur.up0[2 * (4 * (t4 + 1) + i)]);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x16 th, tx;
      {
        const half8 ah = W[(kBlkRel + 2 * t4) * 64 + lane], al = W[(kBlkRel + 2 * t4 + 1) * 64 + lane];
        th = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur.xh, ga, 0, 0, 0);
        tx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur.xl, zero16, 0, 0, 0);
        tx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, cur.xh, tx, 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int ks = 2 * t4 + s;
        const float4 qa = g1[2 * s], qb = g1[2 * s + 1];
        const float q8[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = relu_bits(fmaf(tx[8 * s + j], 1.f / 2048.f, th[8 * s + j]) + q8[j]);
        __builtin_amdgcn_sched_barrier(0);
        if (t4 < 3) {
#pragma unroll
          for (int i = 0; i < 2; ++i) g1[2 * s + i] = cur.up1[2 * (2 * (ks + 2) + i)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          m1[t4] |= (v[j] > 0.f ? 1u : 0u) << (8 * s + j);
          AS_BWD_HD_STORE(__builtin_bit_cast(unsigned, v[j]), r_h1, (int)vo128, (int)((32 * t4 + AS_ROW(8 * s + j)) * q4), 0);
        }
        half8 bh, bl;
        split8_pos(v, bh, bl, imax);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int blk = kBlkW2 + (ks * 2 + m) * 2;
          const half8 ah = W[blk * 64 + lane], al = W[(blk + 1) * 64 + lane];
          a2h[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a2h[m], 0, 0, 0);
          a2x[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ks == 0 ? zero16 : a2x[m], 0, 0, 0);
          a2x[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, a2x[m], 0, 0, 0);
        }
      }
      if (t4 < 3) ga = gn;
    }
    f32x16 a3h[2], a3x[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) a3h[m][i] = bias[kHid2 + 32 * m + acc_row(i, half)];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int mt = ks >> 1, s = ks & 1;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = relu_bits(fmaf(a2x[mt][8 * s + j], 1.f / 2048.f, a2h[mt][8 * s + j]));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        m2[mt] |= (v[j] > 0.f ? 1u : 0u) << (8 * s + j);
        AS_BWD_HD_STORE(__builtin_bit_cast(unsigned, v[j]), r_h2, (int)vo64, (int)((32 * mt + AS_ROW(8 * s + j)) * q4), 0);
      }
      half8 bh, bl;
      split8_pos(v, bh, bl, imax);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int blk = kBlkW3 + (ks * 2 + m) * 2;
        const half8 ah = W[blk * 64 + lane], al = W[(blk + 1) * 64 + lane];
        a3h[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a3h[m], 0, 0, 0);
        a3x[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ks == 0 ? zero16 : a3x[m], 0, 0, 0);
        a3x[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, a3x[m], 0, 0, 0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float v = relu_bits(fmaf(a3x[mt][i], 1.f / 2048.f, a3h[mt][i]));
        m3[mt] |= (v > 0.f ? 1u : 0u) << i;
        AS_BWD_HD_STORE(__builtin_bit_cast(unsigned, v), r_h3, (int)vo64, (int)((32 * mt + AS_ROW(i)) * q4), 0);
      }
    // ================= data-gradient chain =================
    // B operand of the first product: the 9 logit gradients of this lane's query in the chaining order of one k-step
    half8 bh, bl;
    {
      float dl[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // rows 8 (j>>2) + 4 half + (j&3): half 0 holds rows 0..3 and 8, half 1 rows 4..7; everything else is zero padding
        const bool live = j < 4 || (j == 4 && half == 0);
        dl[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_dl, live ? (int)vo9 : (int)0xFFFFFFF0u, (int)((8 * (j >> 2) + (j & 3)) * q4), 0));
      }
      split8(dl, bh, bl, amax);
    }
    // d3 = [h3 > 0] . W4^T dlogits
    half8 f3h[4], f3l[4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const half8 ah = WT[(kTBlk4 + 2 * m) * 64 + lane], al = WT[(kTBlk4 + 2 * m + 1) * 64 + lane];
      f32x16 gh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, zero16, 0, 0, 0);
      f32x16 gx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, zero16, 0, 0, 0);
      gx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, gx, 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = 8 * s + j;
          v[j] = ((m3[m] >> i) & 1u) ? fmaf(gx[i], 1.f / 2048.f, gh[i]) : 0.f;
          AS_BWD_HD_STORE(__builtin_bit_cast(unsigned, v[j]), r_d3, (int)vo64, (int)((32 * m + AS_ROW(i)) * q4), 0);
        }
        split8(v, f3h[2 * m + s], f3l[2 * m + s], amax);
      }
    }
    // d2 = [h2 > 0] . W3^T d3
    half8 f2h[4], f2l[4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x16 gh = zero16, gx = zero16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int blk = kTBlk3 + (ks * 2 + m) * 2;
        const half8 ah = WT[blk * 64 + lane], al = WT[(blk + 1) * 64 + lane];
        AS_MFMA3(ah, al, f3h[ks], f3l[ks], gh, gx)
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = 8 * s + j;
          v[j] = ((m2[m] >> i) & 1u) ? fmaf(gx[i], 1.f / 2048.f, gh[i]) : 0.f;
          AS_BWD_HD_STORE(__builtin_bit_cast(unsigned, v[j]), r_d2, (int)vo64, (int)((32 * m + AS_ROW(i)) * q4), 0);
        }
        split8(v, f2h[2 * m + s], f2l[2 * m + s], amax);
      }
    }
    // d1 = [h1 > 0] . W2^T d2
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      f32x16 gh = zero16, gx = zero16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int blk = kTBlk2 + (ks * 4 + m) * 2;
        const half8 ah = WT[blk * 64 + lane], al = WT[(blk + 1) * 64 + lane];
        AS_MFMA3(ah, al, f2h[ks], f2l[ks], gh, gx)
      }
      if constexpr (!fuse1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = ((m1[m] >> i) & 1u) ? fmaf(gx[i], 1.f / 2048.f, gh[i]) : 0.f;
          AS_BWD_HD_STORE(__builtin_bit_cast(unsigned, v), r_d1, (int)vo128, (int)((32 * m + AS_ROW(i)) * q4), 0);
        }
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous m-tile's column reads are done before the tile is overwritten
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = (cur.valid && ((m1[m] >> i) & 1u)) ? fmaf(gx[i], 1.f / 2048.f, gh[i]) : 0.f;
          stage[(AS_ROW(i) + 4 * half) * kStageRow + c] = v;  // [channel row of the m-tile][query]
          float s0 = v, s1 = v;
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const float o0 = __shfl_down(s0, 1 << k, 32), o1 = __shfl_down(s1, 1 << k, 32);
            s0 += ((same0 >> k) & 1u) ? o0 : 0.f;
            s1 += ((same1 >> k) & 1u) ? o1 : 0.f;
          }
          // lanes that are not the head of a run carry an out-of-range offset: the atomic is dropped by the range check (a branch
          // around each of the 128 atomics of a tile costs 259 spilled registers)
          AS_BWD_ATOMIC(s0, r_u0, (int)vo_u0, (int)((32 * m + AS_ROW(i)) * hw0 * 4u), 0);
          AS_BWD_ATOMIC(s1, r_u1, (int)vo_u1, (int)((32 * m + AS_ROW(i)) * hw1 * 4u), 0);
          if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // two registers' shuffle chains in flight, not sixteen (register pressure)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the tile (and the tile's relative coordinates) have landed in LDS
        // wrel gradient: lane (channel c of the m-tile, query half) walks its 16 queries of the row
        const float* rowp = stage + c * kStageRow + 16 * half;
        const float* relp = stage + 32 * kStageRow + 16 * half;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float dv = rowp[j];
#pragma unroll
          for (int k = 0; k < 4; ++k) accw[m][k] = fmaf(dv, relp[k * 32 + j], accw[m][k]);
        }
      }
    }
  }
#undef AS_ROW
  if constexpr (fuse1) {  // the two query halves of a channel, then one atomic per (channel, column) and wave
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float tot = accw[m][k] + __shfl_xor(accw[m][k], 32);
        if (half == 0) atomicAdd(P.dwrel + (32 * m + c) * 4 + k, tot);
      }
  }
  note_overflow(fmaxf(amax, __builtin_bit_cast(float, imax)));
}

// channels-last copy of a (<= 48-channel) structure feature: out[b][p][0..kDirectCP) = cat(srcs)[b, :, p], zero padded — the rows the
// DIRECT1 tail gathers.  Thread = (pixel, 4-channel group): four coalesced plane reads, one 16-B store.
struct RowsParams {
  const float* src[3];
  int c[3];
  int n_src, C;
  float* out;
  long long P, total;  // pixels per image, B * P
};

__global__ __launch_bounds__(256) void liif_rows_cl_kernel(RowsParams p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  constexpr int NG = kDirectCP / 4;
  if (idx >= p.total * NG) return;
  const int g = (int)(idx / p.total);          // group-major: consecutive threads = consecutive pixels (coalesced reads)
  const long long pix = idx - (long long)g * p.total;
  const long long b = pix / p.P, pp = pix - b * p.P;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = 4 * g + j;
    float x = 0.f;
    if (ch < p.C) {
      int s = 0, cl = ch;
      if (cl >= p.c[0]) { cl -= p.c[0]; s = 1; if (cl >= p.c[1]) { cl -= p.c[1]; s = 2; } }
      x = p.src[s][(b * p.c[s] + cl) * p.P + pp];
    }
    v[j] = x;
  }
  *reinterpret_cast<float4*>(p.out + pix * kDirectCP + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
}

// The row length of a raster-ordered query grid, found on the device (no host synchronisation; capturable): the first index at
// which the row coordinate changes, accepted when the column coordinate restarts there and the row repeats with that period.
// 0 = no such structure (random / sorted training queries).  One block; scans at most the first 2^16 queries.
__global__ __launch_bounds__(256) void liif_query_rows_kernel(const float* __restrict__ coord, long long total, int* __restrict__ out) {
  __shared__ int first;
  if (threadIdx.x == 0) first = 0x7FFFFFFF;
  __syncthreads();
  const float r0 = coord[0];
  const long long lim = total < 65536 ? total : 65536;
  for (long long i = 1 + threadIdx.x; i < lim; i += 256)
    if (coord[2 * i] != r0) { atomicMin(&first, (int)i); break; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int R = first == 0x7FFFFFFF ? 0 : first;
    if (R > 0) {
      const bool restarts = coord[2ll * R + 1] == coord[1];                                   // column of row 1 = column of row 0
      const bool periodic = 2ll * R >= total || coord[2ll * (2ll * R)] != coord[2ll * R];     // row 2 differs from row 1
      const bool whole = 2ll * R - 1 >= total || coord[2ll * (2ll * R - 1)] == coord[2ll * R];  // row 1 lasts R queries
      if (!(restarts && periodic && whole)) R = 0;
    }
    *out = R;
  }
}

}  // namespace

extern "C" {

int64_t as_liif_affinity_ws_bytes(int B, int H, int W, int n_chunks) {
  const long long tiles = (long long)as::cdiv(W, kTW) * as::cdiv(H, kTH) * B;
  const int ng = (int)std::max<long long>(1, std::min<long long>(std::min(4, n_chunks), as::cdiv64(512, tiles)));
  return (long long)B * ng * 9 * H * W * 4;
}

int as_liif_affinity(const float* const* srcs, const int* channels, int n_src, float* aff, float* ws, int B, int H, int W, void* stream) {
  AS_REQUIRE(srcs && channels && aff && ws, AS_ERR_BAD_ARG, "liif_affinity: null pointer");
  AS_REQUIRE(n_src >= 1 && n_src <= 3 && B > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "liif_affinity: bad sizes");
  SfParams p{};
  int ctot = 0;
  for (int s = 0; s < n_src; ++s) {
    AS_REQUIRE(srcs[s] && channels[s] > 0, AS_ERR_BAD_ARG, "liif_affinity: source %d is empty", s);
    AS_REQUIRE(s == n_src - 1 || channels[s] % 8 == 0, AS_ERR_BAD_SHAPE, "liif_affinity: source %d has %d channels (multiple of 8 needed before the last source)", s, channels[s]);
    p.src[s] = srcs[s];
    p.c[s] = channels[s];
    ctot += channels[s];
  }
  AS_REQUIRE((long long)H * W < 2147483647ll, AS_ERR_BAD_SHAPE, "liif_affinity: plane too large");
  p.n_src = n_src; p.part = ws; p.aff = aff; p.B = B; p.H = H; p.W = W;
  p.n_chunks = (ctot + 7) / 8;
  p.n_groups = (int)(as_liif_affinity_ws_bytes(B, H, W, p.n_chunks) / ((long long)B * 9 * H * W * 4));
  AS_REQUIRE((long long)B * p.n_groups < 65536, AS_ERR_BAD_SHAPE, "liif_affinity: batch too large");
  const long long P = (long long)B * H * W;
  hipLaunchKernelGGL(sf_partial_kernel, dim3(as::cdiv(W, kTW), as::cdiv(H, kTH), B * p.n_groups), dim3(256), 0, as::as_stream(stream), p);
  hipLaunchKernelGGL(sf_finish_kernel, dim3((unsigned)as::cdiv64(P, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("liif_affinity");
}

static int ks_bucket(int ksteps) { return ksteps <= 3 ? 3 : (ksteps <= 6 ? 6 : kLowresKs); }  // kernel instantiations

int64_t as_liif_lowres_pack_bytes(int K) { return (long long)ks_bucket((K + 15) / 16) * 8 * 1024; }

int as_liif_lowres_pack(const float* w, int ldw, int koff, int K, void* image, void* stream) {
  AS_REQUIRE(w && image, AS_ERR_BAD_ARG, "liif_lowres_pack: null pointer");
  AS_REQUIRE(K > 0 && K <= 16 * kLowresKs && koff >= 0 && koff + K <= ldw, AS_ERR_BAD_SHAPE, "liif_lowres_pack: %d input channels (max %d), columns [%d,%d) of %d", K, 16 * kLowresKs, koff, koff + K, ldw);
  LowresPackParams p{w, ldw, koff, K, ks_bucket((K + 15) / 16), (_Float16*)image};  // k-steps beyond K: zero fragments
  hipLaunchKernelGGL(liif_lowres_pack_kernel, dim3(as::cdiv(p.ksteps * 8 * 512, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("liif_lowres_pack");
}

int as_liif_lowres_cl(const float* const* srcs, const int* channels, int n_src, const void* wimage, float* out,
                      int B, int H, int W, void* stream) {
  AS_REQUIRE(srcs && channels && wimage && out, AS_ERR_BAD_ARG, "liif_lowres_cl: null pointer");
  AS_REQUIRE(n_src >= 1 && n_src <= 3 && B > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "liif_lowres_cl: bad sizes");
  LowresParams p{};
  int K = 0;
  for (int s = 0; s < n_src; ++s) {
    AS_REQUIRE(srcs[s] && channels[s] > 0, AS_ERR_BAD_ARG, "liif_lowres_cl: source %d is empty", s);
    AS_REQUIRE((s == n_src - 1 ? channels[s] % 8 : channels[s] % 16) == 0, AS_ERR_BAD_SHAPE, "liif_lowres_cl: source %d has %d channels (multiples of 16, the last source of 8, needed)", s, channels[s]);
    p.src[s] = srcs[s];
    p.c[s] = channels[s];
    K += channels[s];
  }
  for (int s = n_src; s < 3; ++s) { p.src[s] = srcs[n_src - 1]; p.c[s] = channels[n_src - 1]; }  // never selected for a live channel
  AS_REQUIRE(K <= 16 * kLowresKs, AS_ERR_BAD_SHAPE, "liif_lowres_cl: %d input channels (max %d)", K, 16 * kLowresKs);
  p.n_src = n_src; p.wimg = (const _Float16*)wimage; p.K = K; p.ksteps = (K + 15) / 16; p.out = out; p.B = B;
  p.P = (long long)H * W; p.total = p.P * B;
  const long long tiles = as::cdiv64(p.total, 32);
  AS_REQUIRE(tiles < 2147483647ll, AS_ERR_BAD_SHAPE, "liif_lowres_cl: too many pixels");
  p.tiles = (int)tiles;
  const int kb = ks_bucket(p.ksteps);
  const long long groups = as::cdiv64(tiles, 4);                     // 4 pixel tiles per block iteration
  const long long resident = 256ll * std::min(4, 160 / (kb * 4)) / 2;  // co-resident blocks per grid.y slice (LDS: kb x 4 KB)
  p.iters = (int)std::max<long long>(1, as::cdiv64(groups, resident));
  const dim3 grid((unsigned)as::cdiv64(groups, p.iters), 2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(liif_lowres_cl_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 4 * 1024);
    attr_set = true;
  }
  const hipStream_t st = as::as_stream(stream);
  if (kb == 3) hipLaunchKernelGGL(liif_lowres_cl_kernel<3>, grid, dim3(512), 3 * 4 * 1024, st, p);
  else if (kb == 6) hipLaunchKernelGGL(liif_lowres_cl_kernel<6>, grid, dim3(512), 6 * 4 * 1024, st, p);
  else hipLaunchKernelGGL(liif_lowres_cl_kernel<12>, grid, dim3(512), 12 * 4 * 1024, st, p);
  return as::check_launch("liif_lowres_cl");
}

int64_t as_liif_tail_image_bytes(void) { return kImageBytes; }

int as_liif_tail_pack(const float* wrel, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                      const float* w4, const float* b4, int n_src, void* image, void* stream) {
  AS_REQUIRE(wrel && w2 && w3 && w4 && image, AS_ERR_BAD_ARG, "liif_tail_pack: null pointer");
  AS_REQUIRE(n_src == 1 || n_src == 2, AS_ERR_BAD_ARG, "liif_tail_pack: n_src must be 1 or 2");
  PackParams p{wrel, b1, w2, b2, w3, b3, w4, b4, n_src, (_Float16*)image};
  const int n = kFragBlocks * 512 + kBiasFloats;
  hipLaunchKernelGGL(liif_tail_pack_kernel, dim3(as::cdiv(n, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("liif_tail_pack");
}

int as_liif_rows_pitch(void) { return kDirectCP; }

int as_liif_rows_cl(const float* const* srcs, const int* channels, int n_src, float* out, int B, int H, int W, void* stream) {
  AS_REQUIRE(srcs && channels && out, AS_ERR_BAD_ARG, "liif_rows_cl: null pointer");
  AS_REQUIRE(n_src >= 1 && n_src <= 3 && B > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "liif_rows_cl: bad sizes");
  RowsParams p{};
  int C = 0;
  for (int s = 0; s < n_src; ++s) {
    AS_REQUIRE(srcs[s] && channels[s] > 0, AS_ERR_BAD_ARG, "liif_rows_cl: source %d is empty", s);
    p.src[s] = srcs[s];
    p.c[s] = channels[s];
    C += channels[s];
  }
  for (int s = n_src; s < 3; ++s) { p.src[s] = srcs[n_src - 1]; p.c[s] = 1 << 30; }  // never reached
  AS_REQUIRE(C <= kDirectCP, AS_ERR_BAD_SHAPE, "liif_rows_cl: %d channels (max %d)", C, kDirectCP);
  p.n_src = n_src; p.C = C; p.out = out; p.P = (long long)H * W; p.total = p.P * B;
  const long long n = p.total * (kDirectCP / 4);
  AS_REQUIRE(n < (1ll << 40), AS_ERR_BAD_SHAPE, "liif_rows_cl: too many pixels");
  hipLaunchKernelGGL(liif_rows_cl_kernel, dim3((unsigned)as::cdiv64(n, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("liif_rows_cl");
}

static int liif_tail_impl(const float* u0, const float* u1, float* coord, const void* image, const void* image1, const float* disp,
                          const float* scale, float* out, float* logits, int B, int Q, int H0, int W0, int H1, int W1, int Hd, int Wd,
                          int clamp_inplace, const int* row_len, void* stream);

int as_liif_tail(const float* u0, const float* u1, float* coord, const void* image, const float* disp, const float* scale,
                 float* out, float* logits, int B, int Q, int H0, int W0, int H1, int W1, int Hd, int Wd, int clamp_inplace,
                 const int* row_len, void* stream) {
  return liif_tail_impl(u0, u1, coord, image, nullptr, disp, scale, out, logits, B, Q, H0, W0, H1, W1, Hd, Wd, clamp_inplace, row_len, stream);
}

int as_liif_tail_direct(const float* u0, const float* rows1, float* coord, const void* image, const void* image1, const float* disp,
                        const float* scale, float* out, float* logits, int B, int Q, int H0, int W0, int H1, int W1, int Hd, int Wd,
                        int clamp_inplace, const int* row_len, void* stream) {
  AS_REQUIRE(rows1 && image1, AS_ERR_BAD_ARG, "liif_tail_direct: null rows / fragment image of the second input");
  return liif_tail_impl(u0, rows1, coord, image, image1, disp, scale, out, logits, B, Q, H0, W0, H1, W1, Hd, Wd, clamp_inplace, row_len, stream);
}

int as_liif_query_rows(const float* coord, int B, int Q, int* row_len, void* stream) {
  AS_REQUIRE(coord && row_len, AS_ERR_BAD_ARG, "liif_query_rows: null pointer");
  AS_REQUIRE(B > 0 && Q > 0, AS_ERR_BAD_ARG, "liif_query_rows: non-positive size");
  hipLaunchKernelGGL(liif_query_rows_kernel, dim3(1), dim3(256), 0, as::as_stream(stream), coord, (long long)B * Q, row_len);
  return as::check_launch("liif_query_rows");
}

static int liif_tail_impl(const float* u0, const float* u1, float* coord, const void* image, const void* image1, const float* disp,
                          const float* scale, float* out, float* logits, int B, int Q, int H0, int W0, int H1, int W1, int Hd, int Wd,
                          int clamp_inplace, const int* row_len, void* stream) {
  AS_REQUIRE(u0 && coord && image && disp && out, AS_ERR_BAD_ARG, "liif_tail: null pointer");
  AS_REQUIRE(B > 0 && Q > 0 && H0 > 0 && W0 > 0 && Hd > 0 && Wd > 0, AS_ERR_BAD_ARG, "liif_tail: non-positive size");
  AS_REQUIRE(!u1 || (H1 > 0 && W1 > 0), AS_ERR_BAD_ARG, "liif_tail: second source without a size");
  TailParams p{};
  p.u[0] = u0; p.u[1] = u1; p.coord = coord; p.image = (const _Float16*)image; p.disp = disp; p.scale = scale;
  p.out = out; p.logits = logits; p.B = B; p.Q = Q; p.n_src = u1 ? 2 : 1; p.clamp_inplace = clamp_inplace;
  p.H[0] = H0; p.W[0] = W0; p.H[1] = u1 ? H1 : 1; p.W[1] = u1 ? W1 : 1; p.Hd = Hd; p.Wd = Wd;
  p.total = (long long)B * Q;
  p.lo = (float)(-1.0 + 1e-6); p.hi = (float)(1.0 - 1e-6);
  for (int s = 0; s < 2; ++s) {
    p.c0y[s] = (float)(-1.0 + 1.0 / p.H[s]); p.sy[s] = (float)(2.0 * (1.0 / p.H[s]));
    p.c0x[s] = (float)(-1.0 + 1.0 / p.W[s]); p.sx[s] = (float)(2.0 * (1.0 / p.W[s]));
  }
  const long long tiles = as::cdiv64(p.total, 32);
  AS_REQUIRE(tiles < 2147483647ll, AS_ERR_BAD_SHAPE, "liif_tail: too many queries");
  p.tiles = (int)tiles;
  p.image1 = (const _Float16*)image1;
  p.row_len = row_len;
  // 2 blocks of 4 waves per CU (LDS: 2 x 66 KB weight images) — or one block of 8 waves around one 90 KB image pair (direct
  // second input); a wave walks `tpw` consecutive tiles
  const long long waves = 256ll * 2 * 4;
  const int wpb = image1 ? 8 : 4;
  p.tpw = (int)std::max<long long>(1, as::cdiv64(tiles, waves));
  long long blocks = as::cdiv64(tiles, (long long)wpb * p.tpw);
  blocks = (blocks + 7) / 8 * 8;  // multiple of 8: the XCD-aware block order is a bijection
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(liif_tail_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kImageBytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(liif_tail_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kImageBytes);
    attr_set = true;
  }
  if (image1) {
    AS_REQUIRE(p.n_src == 2, AS_ERR_BAD_ARG, "liif_tail_direct: needs the second input");
    constexpr int lds = ((kImageBytes + 15) / 16) * 16 + kImage1Bytes;
    as::lds_opt_in(reinterpret_cast<const void*>(liif_tail_kernel<2, true>));
    hipLaunchKernelGGL((liif_tail_kernel<2, true>), dim3((unsigned)blocks), dim3(512), lds, as::as_stream(stream), p);
  } else if (p.n_src == 2)
    hipLaunchKernelGGL(liif_tail_kernel<2>, dim3((unsigned)blocks), dim3(256), kImageBytes, as::as_stream(stream), p);
  else
    hipLaunchKernelGGL(liif_tail_kernel<1>, dim3((unsigned)blocks), dim3(256), kImageBytes, as::as_stream(stream), p);
  return as::check_launch("liif_tail");
}

int64_t as_liif_mlp_bwd_image_bytes(void) { return kImageTBytes; }

int as_liif_mlp_bwd_pack(const float* w2, const float* w3, const float* w4, void* imageT, void* stream) {
  AS_REQUIRE(w2 && w3 && w4 && imageT, AS_ERR_BAD_ARG, "liif_mlp_bwd_pack: null pointer");
  PackTParams p{w2, w3, w4, (_Float16*)imageT};
  hipLaunchKernelGGL(liif_mlp_bwd_pack_kernel, dim3(as::cdiv(kTBlocks * 512, 256)), dim3(256), 0, as::as_stream(stream), p);
  return as::check_launch("liif_mlp_bwd_pack");
}

static int mlp_params(TailParams& p, const float* u0, const float* u1, const float* coord, const void* image, int B, int B1, int Q,
                      int H0, int W0, int H1, int W1, const char* what) {
  AS_REQUIRE(u0 && coord && image, AS_ERR_BAD_ARG, "%s: null pointer", what);
  AS_REQUIRE(B > 0 && Q > 0 && H0 > 0 && W0 > 0, AS_ERR_BAD_ARG, "%s: non-positive size", what);
  AS_REQUIRE(!u1 || (H1 > 0 && W1 > 0), AS_ERR_BAD_ARG, "%s: second source without a size", what);
  AS_REQUIRE(B1 >= 0 && (B1 == 0 || (u1 && B % B1 == 0)), AS_ERR_BAD_SHAPE, "%s: B1=%d does not divide B=%d", what, B1, B);
  p.u[0] = u0; p.u[1] = u1; p.coord = const_cast<float*>(coord); p.image = (const _Float16*)image;
  p.B = B; p.Q = Q; p.n_src = u1 ? 2 : 1; p.clamp_inplace = 0; p.B1 = B1;
  p.H[0] = H0; p.W[0] = W0; p.H[1] = u1 ? H1 : 1; p.W[1] = u1 ? W1 : 1; p.Hd = 1; p.Wd = 1;
  p.total = (long long)B * Q;
  p.lo = (float)(-1.0 + 1e-6); p.hi = (float)(1.0 - 1e-6);
  for (int s = 0; s < 2; ++s) {
    p.c0y[s] = (float)(-1.0 + 1.0 / p.H[s]); p.sy[s] = (float)(2.0 * (1.0 / p.H[s]));
    p.c0x[s] = (float)(-1.0 + 1.0 / p.W[s]); p.sx[s] = (float)(2.0 * (1.0 / p.W[s]));
  }
  const long long tiles = as::cdiv64(p.total, 32);
  AS_REQUIRE(tiles < 2147483647ll, AS_ERR_BAD_SHAPE, "%s: too many queries", what);
  p.tiles = (int)tiles;
  return AS_OK;
}

int as_liif_mlp_fwd(const float* u0, const float* u1, const float* coord, const void* image, float* logits, int B, int B1, int Q,
                    int H0, int W0, int H1, int W1, void* stream) {
  AS_REQUIRE(logits, AS_ERR_BAD_ARG, "liif_mlp_fwd: null logits");
  TailParams p{};
  const int rc = mlp_params(p, u0, u1, coord, image, B, B1, Q, H0, W0, H1, W1, "liif_mlp_fwd");
  if (rc != AS_OK) return rc;
  p.logits = logits;
  const long long waves = 256ll * 2 * 4;
  p.tpw = (int)std::max<long long>(1, as::cdiv64(p.tiles, waves));
  long long blocks = as::cdiv64(p.tiles, 4ll * p.tpw);
  blocks = (blocks + 7) / 8 * 8;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(liif_tail_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kImageBytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(liif_tail_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kImageBytes);
    attr_set = true;
  }
  if (p.n_src == 2) hipLaunchKernelGGL(liif_tail_kernel<2>, dim3((unsigned)blocks), dim3(256), kImageBytes, as::as_stream(stream), p);
  else hipLaunchKernelGGL(liif_tail_kernel<1>, dim3((unsigned)blocks), dim3(256), kImageBytes, as::as_stream(stream), p);
  return as::check_launch("liif_mlp_fwd");
}

int as_liif_mlp_bwd(const float* u0, const float* u1, const float* coord, const void* image, const void* imageT, const float* dlogits,
                    float* h1, float* h2, float* h3, float* d3, float* d2, float* d1, float* du0, float* du1, float* dwrel, int B, int B1,
                    int Q, int H0, int W0, int H1, int W1, void* stream) {
  AS_REQUIRE(imageT && dlogits && h1 && h2 && h3 && d3 && d2, AS_ERR_BAD_ARG, "liif_mlp_bwd: null pointer");
  AS_REQUIRE((d1 != nullptr) != (du0 != nullptr), AS_ERR_BAD_ARG, "liif_mlp_bwd: either d1 or the fused first-layer outputs (du0, du1, dwrel)");
  AS_REQUIRE(!du0 || (du1 && dwrel), AS_ERR_BAD_ARG, "liif_mlp_bwd: du0, du1 and dwrel go together");
  AS_REQUIRE(u1, AS_ERR_BAD_ARG, "liif_mlp_bwd: built for the two-input upsampler");
  MlpBwdParams P{};
  const int rc = mlp_params(P.t, u0, u1, coord, image, B, B1, Q, H0, W0, H1, W1, "liif_mlp_bwd");
  if (rc != AS_OK) return rc;
  AS_REQUIRE((long long)B * kHid1 * Q * 4 < 0x7FFFFFF0ll, AS_ERR_BAD_SHAPE, "liif_mlp_bwd: [B,128,Q] exceeds 2 GiB (32-bit buffer offsets); split the batch");
  P.imageT = (const _Float16*)imageT; P.dlogits = dlogits;
  P.h1 = h1; P.h2 = h2; P.h3 = h3; P.d3 = d3; P.d2 = d2; P.d1 = d1;
  P.du0 = du0; P.du1 = du1; P.dwrel = dwrel;
  if (du0) {
    const int b1n = B1 > 0 ? B1 : B;
    AS_REQUIRE((long long)B * kHid1 * H0 * W0 * 4 < 0x7FFFFFF0ll && (long long)b1n * kHid1 * H1 * W1 * 4 < 0x7FFFFFF0ll &&
               (long long)B * H0 * W0 < 2147483647ll && (long long)b1n * H1 * W1 < 2147483647ll, AS_ERR_BAD_SHAPE,
               "liif_mlp_bwd: a first-layer map exceeds the 32-bit offsets of the fused scatter");
    int zrc = as::zero_fill(du0, (long long)B * kHid1 * H0 * W0, as::as_stream(stream));
    if (zrc == AS_OK) zrc = as::zero_fill(du1, (long long)b1n * kHid1 * H1 * W1, as::as_stream(stream));
    if (zrc == AS_OK) zrc = as::zero_fill(dwrel, kHid1 * 4, as::as_stream(stream));
    if (zrc != AS_OK) return zrc;
  }
  // one block of 8 waves per CU around the forward + transposed weight images (118 KB); a wave walks `tpw` consecutive tiles
  const long long waves = 256ll * 8;
  P.t.tpw = (int)std::max<long long>(1, as::cdiv64(P.t.tiles, waves));
  long long blocks = as::cdiv64(P.t.tiles, 8ll * P.t.tpw);
  blocks = (blocks + 7) / 8 * 8;
  constexpr int lds = ((kImageBytes + 15) / 16) * 16 + kImageTBytes + 8 * kStageFloats * 4;  // + one staging tile per wave
  static_assert(lds <= 160 * 1024, "liif_mlp_bwd: LDS budget");
  if (du0) {
    as::lds_opt_in(reinterpret_cast<const void*>(liif_mlp_bwd_kernel<true>));
    hipLaunchKernelGGL(liif_mlp_bwd_kernel<true>, dim3((unsigned)blocks), dim3(512), lds, as::as_stream(stream), P);
  } else {
    as::lds_opt_in(reinterpret_cast<const void*>(liif_mlp_bwd_kernel<false>));
    hipLaunchKernelGGL(liif_mlp_bwd_kernel<false>, dim3((unsigned)blocks), dim3(512), lds, as::as_stream(stream), P);
  }
  return as::check_launch("liif_mlp_bwd");
}

unsigned as_liif_split_overflow(int reset) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_split_overflow_liif), sizeof(v)) != hipSuccess) return 0xFFFFFFFFu;
  if (reset && v) {
    const unsigned z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_split_overflow_liif), &z, sizeof(z));
  }
  return v;
}

}  // extern "C"
