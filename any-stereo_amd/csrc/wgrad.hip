// Weight gradient of the update block's stride-1 "same" convolutions (training, cfg 4): dW[co][ci][ky][kx] =
//   sum_{b,y,x} dy[b][co][y][x] * x[b][ci][y+ky-P][x+kx-P]   (+ db[co] = sum dy), what autograd derives for the nn.Conv2d layers of
// update.py:16-92 applied once per GRU iteration (train_continuous_IGEV.py:214-239).  The Python side stacks the (input, output
// gradient) pairs of all iterations of a step along the batch axis (grad.WeightAnchor), so one launch reduces over
// K = iters * B * H * W pixels.
//
// GEMM view: M = Cout (rows of dy), N = Cin * KS * KS, K = pixels.  v_mfma_f32_32x32x16_bf16 with both operands split into
// bf16 hi + bf16 lo (x = hi + lo exactly to 16 significand bits; hi*hi + hi*lo + lo*hi in ONE fp32 accumulator: ~2^-16 relative
// per product, fp32 range — gradients of 1e-9 need no scaling, unlike the fp16 split of the forward kernels).
//
// Block = 512 threads: waves 0-3 CONSUMERS (wave w owns output-gradient rows co0+32w.. and all TAPS*NT column tiles: 144 fp32
// accumulators for 3x3; nothing but ds_read + MFMA in their loop), waves 4-7 LOADERS.  K-chunk = one image row segment of
// XS = 16*KST <= 80 pixels, chunks ordered (image, segment, row) so that consecutive chunks of a block share two x rows:
//   * dy (128 rows x XS): fetched by the loaders as coalesced 16-B groups (every load of a chunk in flight before the first
//     conversion), split, parked hi / lo in a double-buffered LDS image [co][pitch];
//   * x (32*NT channels x KS rows x XS+halo): a ring of KS+1 row slots in LDS — a chunk stages only its ONE new row
//     (y+1) while the consumers read rows y-2 .. y of the previous chunk; a new (image, segment) column restages all KS rows
//     behind one extra barrier.  Rows outside the image are staged as zeros, columns outside it too, so the flat k index needs
//     no masks.  The three kx taps of a row are the same eight pixels shifted by one element: one aligned 16-byte read + the two
//     neighbouring dwords, and five v_alignbit build the kx = 0 and kx = 2 fragments.  Row pitches = 8 * odd elements: the 16
//     lanes of a read phase hit 16 distinct 16-byte bank groups.
//   First version (dy fragments fetched per lane straight from global memory, all KS x rows restaged per chunk): 1.76 ms for
//   the gru04 z|r layer — the per-lane 32-B fetches touch 64 cache lines per wave-load and ~150 KB per chunk went through the
//   CU's L1 fill path (~21 B/clk) against 4640 clk of MFMA work; this form moves 51 KB per chunk.
// Split-K: the grid is (K ranges) x (output tiles); each block writes its partial tile to a workspace and
// wgrad_finish_kernel sums the ranges in a fixed order (deterministic, no atomics).  The bias gradient rides along as one
// extra column tile of ones in the blocks of the first channel tile.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float acc16;
typedef __attribute__((ext_vector_type(4))) float f4;

constexpr int kWgradMaxTensors = 32;
struct WgradParams {
  // the B images may be spread over up to 32 tensors of `per` images each (the GRU iterations of a training step: no stacking copy)
  const float* xs[kWgradMaxTensors];   // each [per][Cin][H][W]
  const float* dys[kWgradMaxTensors];  // each [per][Cout][H][W]
  int per;
  float* ws;        // [nsplit][T][Cout][Cin] partial weight gradients, T = KS*KS
  float* wsb;       // [nsplit][Cout] partial bias gradients (null: none)
  int B, Cin, Cout, H, W;
  int nseg, XS, KST;   // segments per image row, segment width = 16*KST
  int pitch_x, pitch_g;  // LDS row pitches (bf16 elements, 8 * odd)
  int n_co, n_ci;      // output tiles: 128 rows x 32*NT channels
  int nsplit;
  long long chunks;  // B*nseg*H, ordered (image, segment, row)
  int dbg;           // timing ablations (AS_WGRAD_DBG): 1 = loaders only meet the barriers, 2 = consumers only meet the barriers
};

typedef __attribute__((ext_vector_type(4))) unsigned int u4;
typedef __attribute__((ext_vector_type(2))) unsigned int u2;
struct B3 { bf8 k[3]; };  // the fragments of the taps kx = 0, 1, 2 of one row

// D = elements e0..e7 (aligned), L = (e-2, e-1), R = (e8, e9): kx = 1 is D, kx = 0 / 2 are D shifted by one element
__device__ __forceinline__ B3 tap_shifts(u4 D, unsigned L, unsigned R) {
  const unsigned sm = __builtin_amdgcn_alignbit(D[0], L, 16), s0 = __builtin_amdgcn_alignbit(D[1], D[0], 16),
                 s1 = __builtin_amdgcn_alignbit(D[2], D[1], 16), s2 = __builtin_amdgcn_alignbit(D[3], D[2], 16),
                 s3 = __builtin_amdgcn_alignbit(R, D[3], 16);
  B3 o;
  const u4 a = {sm, s0, s1, s2}, c = {s0, s1, s2, s3};
  o.k[0] = __builtin_bit_cast(bf8, a);
  o.k[1] = __builtin_bit_cast(bf8, D);
  o.k[2] = __builtin_bit_cast(bf8, c);
  return o;
}

// x = hi + lo, both bf16 (round to nearest even): pairs -> packed dwords.  One packed conversion per pair and part (written
// element by element the compiler emits a v_cvt_pk_bf16_f32 per ELEMENT).
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = cvt_pk_bf16(a, b);
  lo = cvt_pk_bf16(a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xFFFF0000u));
}

struct Pos { int b, seg, y; };  // a K-chunk: image b, row segment seg, row y (y innermost: consecutive chunks share two x rows)
__device__ __forceinline__ Pos pos_of(const WgradParams& p, long long c) {  // one division per block, then pos_next
  Pos q;
  const long long bs = c / p.H;
  q.y = (int)(c - bs * p.H);
  q.b = (int)(bs / p.nseg);
  q.seg = (int)(bs - (long long)q.b * p.nseg);
  return q;
}
__device__ __forceinline__ void pos_next(const WgradParams& p, Pos& q) {
  if (++q.y == p.H) {
    q.y = 0;
    if (++q.seg == p.nseg) { q.seg = 0; ++q.b; }
  }
}

// Four consecutive pixels of one row: fetch (zeros outside [0, W) / for invalid rows) ...
template <bool VEC>
__device__ __forceinline__ f4 load4(const float* __restrict__ src, long long row, int col, int W, bool row_ok) {
  f4 v;
  if (VEC) {  // W % 4 == 0 and col % 4 == 0: the group lies inside the row or outside it as a whole
    const bool ok = row_ok && col >= 0 && col < W;
    const f4 t = *reinterpret_cast<const f4*>(src + (ok ? row + col : 0));
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ok ? t[j] : 0.f;
  } else {
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = src[(row_ok && col + j >= 0 && col + j < W) ? row + col + j : 0];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (row_ok && col + j >= 0 && col + j < W) ? t[j] : 0.f;
  }
  return v;
}
// ... and park as bf16 hi / lo pairs (8-byte aligned destinations)
__device__ __forceinline__ void store4(f4 v, unsigned short* dst_hi, unsigned short* dst_lo) {
  unsigned h0, l0, h1, l1;
  split_pair(v[0], v[1], h0, l0);
  split_pair(v[2], v[3], h1, l1);
  const u2 hi = {h0, h1}, lo = {l0, l1};
  *reinterpret_cast<u2*>(dst_hi) = hi;
  *reinterpret_cast<u2*>(dst_lo) = lo;
}

template <int KS, int NT, bool VEC>
__global__ __launch_bounds__(512, 2) void wgrad_kernel(WgradParams p) {
  constexpr int TAPS = KS * KS, PAD = KS / 2, NB = TAPS * NT, CIB = 32 * NT;
  constexpr int NS = KS + 1;                     // ring slots of the x image: rows y-PAD .. y+PAD in use, row y+PAD+1 being staged
  constexpr bool RING = KS == 3;                 // 1x1: no row is shared between chunks — two slots alternating by chunk parity
  constexpr int PADL = KS == 3 ? 8 : 0;          // stored element s = col - x0 + PADL: the centre tap's 8-pixel groups are 16-byte aligned
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // image b -> (tensor b / per, image b % per): wave-uniform, read from the kernel arguments
#define WG_IMG(ARR, B_, C_) ((ARR)[(B_) / p.per] + (long long)((B_) % p.per) * (C_) * plane_hw)
  const int px = p.pitch_x, pg = p.pitch_g;
  const int ximg = CIB * NS * px;                // bf16 elements of the hi (or lo) x ring   [slot][ci][px]: consecutive channels one (odd) pitch apart
  const int gimg = 128 * pg;                     // bf16 elements of one hi (or lo) dy image [co][pg]
  unsigned short* const ldx = reinterpret_cast<unsigned short*>(smem);  // [hi, lo][NS][CIB][px]
  unsigned short* const ldg = ldx + 2 * ximg;                           // [buf 2][hi, lo][128][pg]
  const int tid = threadIdx.x;
  const int tiles = p.n_co * p.n_ci;
  const int split = blockIdx.x / tiles, tile = blockIdx.x - split * tiles;
  const int tco = tile / p.n_ci, tci = tile - tco * p.n_ci;
  const int co0 = tco * 128, ci0 = tci * CIB;
  const long long c_lo = p.chunks * split / p.nsplit, c_hi = p.chunks * (split + 1) / p.nsplit;

  if (tid >= 256) {
    // ---------------- loaders ----------------
    // Global loads run TWO chunks ahead of the LDS images in two register sets (A, B): the fetches of chunk c+2 are issued
    // when chunk c is parked, so a load has a whole chunk period to land (one chunk ahead left one exposed round trip per chunk:
    // 6.3 us per chunk against 2.2 us of MFMA work).
    const int lt = tid - 256;
    const int g4 = p.XS >> 2;                                  // 4-pixel groups of a dy row segment
    const int xg4 = KS == 3 ? g4 + 2 : g4;                     // ... of an x row: cols x0-4 .. x0+XS+3 (KS = 3), s = 4g' + (KS == 3 ? 4 : 0)
    constexpr int NG = 10;                                     // dy groups per loader thread: 128 rows x XS/4 <= 256 * NG (XS <= 80)
    constexpr int NX = (CIB * 22 + 255) / 256;                 // x groups per loader thread and row: CIB x (XS/4 + 2) <= 256 * NX
    const int n_g = 128 * g4, n_x = CIB * xg4;
    struct Regs { f4 g[NG]; f4 x[NX]; };
    // Everything about a thread's work items that does not change from chunk to chunk is computed ONCE: the loaders share
    // their SIMDs' vector issue with the consumers' MFMAs (6 slots per MFMA), and a division + 64-bit row arithmetic per item
    // and chunk (~100 VALU instructions) made the LOADERS' instruction stream the bound of the kernel (1.5 ms for gru04 z|r).
    const int plane_hw = p.H * p.W;
    constexpr unsigned OOB = 0x80000000u;  // a byte offset no image reaches (images < 2 GiB): the range check returns zeros
    // VEC: byte offsets from the image's first element, OOB for items that do not exist; !VEC: element offsets + flags
    unsigned g_off[NG], x_off[NX];
    int g_col[NG], g_lds[NG], x_col[NX], x_lds[NX];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int it = lt + 256 * j, rr = it / g4, g = it - rr * g4;
      const bool ok = it < n_g && co0 + rr < p.Cout;
      g_off[j] = ok ? (unsigned)((co0 + rr) * plane_hw + 4 * g) * (VEC ? 4u : 1u) : OOB;
      g_col[j] = 4 * g;
      g_lds[j] = rr * pg + 4 * g;
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int it = lt + 256 * j, cl = it / xg4, g = it - cl * xg4;
      const bool ok = it < n_x && ci0 + cl < p.Cin;
      x_col[j] = 4 * g - (KS == 3 ? 4 : 0);
      x_off[j] = ok ? (unsigned)((ci0 + cl) * plane_hw + 4 * g) * (VEC ? 4u : 1u) : OOB;  // + (row * W + x0 - 4) per chunk
      x_lds[j] = cl * px + 4 * g + (KS == 3 ? PADL - 4 : 0);
    }
    const int img_g = (int)((long long)p.Cout * plane_hw * 4), img_x = (int)((long long)p.Cin * plane_hw * 4);  // bytes per image

#define WG_LOAD_G(SET, Q)                                                                                   \
  {                                                                                                         \
    const int x0 = (Q).seg * p.XS;                                                                          \
    if (VEC) { /* range-checked 16-B buffer loads: items outside the tensor / the row read zeros, no selects */ \
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)WG_IMG(p.dys, (Q).b, p.Cout), 0, img_g, 0x00020000); \
      const unsigned add = (unsigned)((Q).y * p.W + x0) * 4u;                                               \
      _Pragma("unroll") for (int j = 0; j < NG; ++j) {                                                      \
        const unsigned off = (x0 + g_col[j] < p.W) ? g_off[j] + add : OOB;                                  \
        (SET).g[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));     \
      }                                                                                                     \
    } else {                                                                                                \
      const float* gb = WG_IMG(p.dys, (Q).b, p.Cout);                                         \
      _Pragma("unroll") for (int j = 0; j < NG; ++j)                                                        \
        (SET).g[j] = load4<false>(gb, (long long)(g_off[j] & ~OOB) - g_col[j] + (Q).y * p.W, x0 + g_col[j], p.W, g_off[j] != OOB); \
    }                                                                                                       \
  }
#define WG_STORE_G(SET, CN)                                                                                 \
  {                                                                                                         \
    unsigned short* const gdst = ldg + (int)(((CN) - c_lo) & 1) * 2 * gimg;                                 \
    _Pragma("unroll") for (int j = 0; j < NG; ++j)                                                          \
      if (lt + 256 * j < n_g) store4((SET).g[j], gdst + g_lds[j], gdst + gimg + g_lds[j]);                  \
  }
#define WG_LOAD_X(SET, Q, YY)                                                                               \
  {                                                                                                         \
    const int x0 = (Q).seg * p.XS;                                                                          \
    const bool row_in = (YY) >= 0 && (YY) < p.H;                                                            \
    if (VEC) {                                                                                              \
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)WG_IMG(p.xs, (Q).b, p.Cin), 0, row_in ? img_x : 0, 0x00020000); \
      const unsigned add = (unsigned)((YY) * p.W + x0 - (KS == 3 ? 4 : 0)) * 4u;                            \
      _Pragma("unroll") for (int j = 0; j < NX; ++j) {                                                      \
        const int col = x0 + x_col[j];                                                                      \
        const unsigned off = (col >= 0 && col < p.W) ? x_off[j] + add : OOB;                                \
        (SET).x[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));     \
      }                                                                                                     \
    } else {                                                                                                \
      const float* xb = WG_IMG(p.xs, (Q).b, p.Cin);                                           \
      _Pragma("unroll") for (int j = 0; j < NX; ++j)                                                        \
        (SET).x[j] = load4<false>(xb, (long long)(x_off[j] & ~OOB) - (x_col[j] + (KS == 3 ? 4 : 0)) + (row_in ? (YY) : 0) * p.W, x0 + x_col[j], p.W, \
                                  x_off[j] != OOB && row_in);                                               \
    }                                                                                                       \
  }
#define WG_STORE_X(SET, SLOT)                                                                               \
  {                                                                                                         \
    unsigned short* const xdst = ldx + (SLOT) * (CIB * px);                                                 \
    _Pragma("unroll") for (int j = 0; j < NX; ++j)                                                          \
      if (lt + 256 * j < n_x) store4((SET).x[j], xdst + x_lds[j], xdst + ximg + x_lds[j]);                  \
  }
// all KS rows of chunk Q (a new column): fetched and parked row by row (once per image column, not worth registers)
#define WG_STAGE_COLUMN(SET, Q)                                                                             \
  for (int yy = (Q).y - PAD; yy <= (Q).y + PAD; ++yy) {                                                     \
    WG_LOAD_X(SET, Q, yy)                                                                                   \
    WG_STORE_X(SET, (yy + NS) % NS)                                                                         \
  }
// park chunk CN (position Q, fetched into SET two iterations ago), then fetch chunk CN + 2 into SET, then meet the consumers
#define WG_ITER(SET, CN, Q)                                                                                 \
  {                                                                                                         \
    const bool live = (CN) < c_hi, fresh = RING && live && (Q).y == 0;                                      \
    if (live && p.dbg != 1) WG_STORE_G(SET, CN)                                                             \
    if (fresh) {                                                                                            \
      __syncthreads(); /* the consumers are done with the previous column's rows: any slot may be rewritten */ \
      WG_STAGE_COLUMN(SET, Q)                                                                               \
    } else if (live && p.dbg != 1) {                                                                        \
      WG_STORE_X(SET, RING ? ((Q).y + PAD + NS) % NS : (int)(((CN) - c_lo) & 1))                            \
    }                                                                                                       \
    pos_next(p, Q);                                                                                         \
    pos_next(p, Q);                                                                                         \
    if ((CN) + 2 < c_hi && p.dbg != 1) {                                                                    \
      WG_LOAD_G(SET, Q)                                                                                     \
      if (!RING || (Q).y != 0) WG_LOAD_X(SET, Q, (Q).y + PAD)                                               \
    }                                                                                                       \
    __syncthreads();                                                                                        \
  }

    Regs A, B;
    Pos qa = pos_of(p, c_lo), qb;
    if (c_lo < c_hi) {  // prologue: chunk c_lo in full
      WG_LOAD_G(A, qa)
      WG_STORE_G(A, c_lo)
      if (RING) {
        WG_STAGE_COLUMN(A, qa)
      } else {  // 1x1: no rows shared between chunks, the x image simply alternates between two slots
        WG_LOAD_X(A, qa, qa.y)
        WG_STORE_X(A, 0)
      }
    }
    pos_next(p, qa);  // qa = chunk c_lo + 1 (set A), qb = chunk c_lo + 2 (set B)
    qb = qa;
    pos_next(p, qb);
    if (c_lo + 1 < c_hi) {
      WG_LOAD_G(A, qa)
      if (!RING || qa.y != 0) WG_LOAD_X(A, qa, qa.y + PAD)
    }
    if (c_lo + 2 < c_hi) {
      WG_LOAD_G(B, qb)
      if (!RING || qb.y != 0) WG_LOAD_X(B, qb, qb.y + PAD)
    }
    __syncthreads();  // chunk c_lo staged
    for (long long cn = c_lo + 1; cn <= c_hi; cn += 2) {
      WG_ITER(A, cn, qa)
      if (cn + 1 <= c_hi) WG_ITER(B, cn + 1, qb)
    }
#undef WG_ITER
#undef WG_STAGE_COLUMN
#undef WG_STORE_X
#undef WG_LOAD_X
#undef WG_STORE_G
#undef WG_LOAD_G
    return;
  }

  // ---------------- consumers ----------------
  const int lane = tid & 63, r = lane & 31, h = lane >> 5, w = tid >> 6;
  const bool do_bias = p.wsb != nullptr && tci == 0;
  acc16 acc[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  acc16 accb;
#pragma unroll
  for (int i = 0; i < 16; ++i) accb[i] = 0.f;
  const u4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const bf8 ones = __builtin_bit_cast(bf8, ones_u);
  const int a_e = (32 * w + r) * pg + 8 * h;       // this lane's dy element (k-step 0)
  const int b_e = r * px + PADL + 8 * h;           // this lane's x element inside a column tile (slot 0, k-step 0, centre tap)
  Pos q = pos_of(p, c_lo);
  __syncthreads();  // chunk c_lo staged
  struct Raw { u4 dh, dl; unsigned lh, rh, ll, rl; };  // one x row of a k-step as read from LDS: hi / lo groups + their neighbour dwords
  for (long long cc = c_lo; cc < c_hi; ++cc) {
    const unsigned short* ga = ldg + (int)((cc - c_lo) & 1) * 2 * gimg + a_e;
    if (p.dbg == 2) {
    } else if (KS == 3) {
      // The LDS reads of row (k-step, ky) + 1 are issued BEFORE the nine MFMAs of row (k-step, ky): with one consumer wave per
      // SIMD nothing else hides the ~150-cycle read latency (reads after the MFMAs: 1.54 ms for gru04 z|r, 35 % of the MFMA rate).
      const unsigned short* xr[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) xr[ky] = ldx + b_e + ((q.y + ky - PAD + NS) % NS) * (CIB * px);
#define WG_READ(R, KY, KS_)                                                                      \
  {                                                                                              \
    const unsigned short* xe = xr[KY] + (KS_) * 16;                                              \
    (R).dh = *reinterpret_cast<const u4*>(xe);                                                   \
    (R).dl = *reinterpret_cast<const u4*>(xe + ximg);                                            \
    (R).lh = *reinterpret_cast<const unsigned*>(xe - 2);                                         \
    (R).rh = *reinterpret_cast<const unsigned*>(xe + 8);                                         \
    (R).ll = *reinterpret_cast<const unsigned*>(xe + ximg - 2);                                  \
    (R).rl = *reinterpret_cast<const unsigned*>(xe + ximg + 8);                                  \
  }
#define WG_MMA(R, KY)                                                                            \
  {                                                                                              \
    const B3 bh = tap_shifts((R).dh, (R).lh, (R).rh), bl = tap_shifts((R).dl, (R).ll, (R).rl);   \
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) acc[(KY) * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bh.k[kx], acc[(KY) * 3 + kx], 0, 0, 0); \
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) acc[(KY) * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bl.k[kx], acc[(KY) * 3 + kx], 0, 0, 0); \
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) acc[(KY) * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bh.k[kx], acc[(KY) * 3 + kx], 0, 0, 0); \
  }
      Raw R0, R1, R2;
      u4 ah_n = *reinterpret_cast<const u4*>(ga), al_n = *reinterpret_cast<const u4*>(ga + gimg);
      WG_READ(R0, 0, 0)
      for (int ks = 0; ks < p.KST; ++ks) {
        const bf8 a_hi = __builtin_bit_cast(bf8, ah_n), a_lo = __builtin_bit_cast(bf8, al_n);
        const bool more = ks + 1 < p.KST;
        WG_READ(R1, 1, ks)
        __builtin_amdgcn_sched_barrier(0);  // keep the reads AHEAD of the MFMAs that do not need them (the scheduler sinks them
        WG_MMA(R0, 0)                        // next to their first use and waits on them at once: 61 instead of 32 cycles per MFMA)
        __builtin_amdgcn_sched_barrier(0);
        WG_READ(R2, 2, ks)
        __builtin_amdgcn_sched_barrier(0);
        WG_MMA(R1, 1)
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
          WG_READ(R0, 0, ks + 1)
          ah_n = *reinterpret_cast<const u4*>(ga + (ks + 1) * 16);
          al_n = *reinterpret_cast<const u4*>(ga + gimg + (ks + 1) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        WG_MMA(R2, 2)
        if (do_bias) {
          accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, ones, accb, 0, 0, 0);
          accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, ones, accb, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef WG_MMA
#undef WG_READ
    } else {
      const unsigned short* xr0 = ldx + b_e + (int)((cc - c_lo) & 1) * (CIB * px);
      for (int ks = 0; ks < p.KST; ++ks) {
        const bf8 a_hi = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(ga + ks * 16));
        const bf8 a_lo = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(ga + gimg + ks * 16));
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          const unsigned short* xe = xr0 + (ct * 32) * px + ks * 16;
          const bf8 b_hi = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(xe));
          const bf8 b_lo = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(xe + ximg));
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[ct], 0, 0, 0);
        }
        if (do_bias) {
          accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, ones, accb, 0, 0, 0);
          accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, ones, accb, 0, 0, 0);
        }
      }
    }
    pos_next(p, q);
    __syncthreads();  // this chunk's images are free; the next chunk's dy and its new x row are staged
    if (RING && cc + 1 < c_hi && q.y == 0) __syncthreads();  // new column: the loaders restage all KS rows after the barrier above
  }

  // ---------------- partial tile -> workspace [split][tap][Cout][Cin] ----------------
  const long long plane = (long long)p.Cout * p.Cin;
  float* __restrict__ wsp = p.ws + (long long)split * TAPS * plane;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const int ci = ci0 + ct * 32 + r;
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int coe = co0 + 32 * w + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (coe < p.Cout && ci < p.Cin) wsp[tp * plane + (long long)coe * p.Cin + ci] = acc[ct * TAPS + tp][i];
      }
    }
  }
  if (do_bias && r == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int coe = co0 + 32 * w + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (coe < p.Cout) p.wsb[(long long)split * p.Cout + coe] = accb[i];
    }
  }
}

// dW[co][ci][t] = sum_s ws[s][t][co][ci] (fixed order); db[co] = sum_s wsb[s][co].  One thread per (tap, co, ci): the reads of a
// wave are 256 contiguous bytes of one workspace plane (one thread per (co, ci) walking all taps and ranges left 16 blocks doing
// 2304 dependent loads each on the 64 -> 64 layers: 0.4 of their 0.68 ms).
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, float* __restrict__ dw,
                                                           float* __restrict__ db, int Cout, int Cin, int T, int nsplit) {
  const long long plane = (long long)Cout * Cin;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx < plane * T) {
    const int t = (int)(idx / plane);
    const long long i = idx - (long long)t * plane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four loads in flight; the sum order is fixed by (k mod 4, k)
    int k = 0;
    for (; k + 3 < nsplit; k += 4) {
      s0 += ws[((long long)k * T + t) * plane + i];
      s1 += ws[((long long)(k + 1) * T + t) * plane + i];
      s2 += ws[((long long)(k + 2) * T + t) * plane + i];
      s3 += ws[((long long)(k + 3) * T + t) * plane + i];
    }
    for (; k < nsplit; ++k) s0 += ws[((long long)k * T + t) * plane + i];
    dw[i * T + t] = (s0 + s1) + (s2 + s3);
  }
  if (db && wsb && idx < Cout) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += wsb[(long long)k * Cout + idx];
    db[idx] = s;
  }
}

struct WgradPlan {
  int nseg, XS, KST, pitch_x, pitch_g, n_co, n_ci, nsplit, NT;
  long long chunks;
  size_t lds;
  long long ws_floats;
};

bool wgrad_plan(int B, int Cin, int Cout, int H, int W, int KS, WgradPlan& q) {
  if (!(KS == 1 || KS == 3) || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return false;
  q.NT = KS == 1 ? 2 : 1;
  q.nseg = as::cdiv(W, 80);  // row segments of <= 80 pixels: both LDS images (x ring + double-buffered dy) fit 160 KB
  q.XS = as::cdiv(as::cdiv(W, q.nseg), 16) * 16;
  q.KST = q.XS / 16;
  auto pitch = [](int need) {  // multiple of 8 elements (16-B rows) with an odd number of 16-B units: conflict-free b128 reads
    int v = (need + 7) / 8 * 8;
    if ((v / 8) % 2 == 0) v += 8;
    return v;
  };
  q.pitch_x = pitch(q.XS + (KS == 3 ? 12 : 0));
  q.pitch_g = pitch(q.XS);
  q.n_co = as::cdiv(Cout, 128);
  q.n_ci = as::cdiv(Cin, 32 * q.NT);
  q.chunks = (long long)B * H * q.nseg;
  const int tiles = q.n_co * q.n_ci;
  long long ns = (512 + tiles / 2) / tiles;  // about two rounds of blocks on 256 CUs
  if (ns > q.chunks / 4) ns = q.chunks / 4;   // at least four chunks per block
  if (ns < 1) ns = 1;
  if (ns > 128) ns = 128;  // the workspace (and the finish pass over it) grows with the range count
  q.nsplit = (int)ns;
  q.lds = ((size_t)2 * (32 * q.NT) * (KS + 1) * q.pitch_x + (size_t)2 * 2 * 128 * q.pitch_g) * sizeof(unsigned short);
  q.ws_floats = (long long)q.nsplit * (KS * KS) * Cout * Cin + (long long)q.nsplit * Cout;
  return true;
}

template <int KS, int NT>
int wgrad_launch(const WgradParams& p, const WgradPlan& q, bool vec, hipStream_t s) {
  const dim3 grid((unsigned)(q.nsplit * q.n_co * q.n_ci));
  if (vec) {
    as::lds_opt_in((const void*)wgrad_kernel<KS, NT, true>);
    hipLaunchKernelGGL((wgrad_kernel<KS, NT, true>), grid, dim3(512), q.lds, s, p);
  } else {
    as::lds_opt_in((const void*)wgrad_kernel<KS, NT, false>);
    hipLaunchKernelGGL((wgrad_kernel<KS, NT, false>), grid, dim3(512), q.lds, s, p);
  }
  return as::check_launch("conv2d_wgrad");
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight / bias gradient of the motion encoder's 7x7 convolution of the ONE-channel disparity map (update.py:81,87:
// convd1 = Conv2d(1, 64, 7, padding=3)) for all GRU iterations of a step in one launch:
//     dW[co][ky][kx] = sum_{b,y,x} dy[b][co][y][x] . disp[b][y+ky-3][x+kx-3]      db[co] = sum dy[b][co][y][x]
// (the disparity is detached per iteration, continuous_IGEVstereo.py:285: there is no input gradient).  49 x 64 outputs over
// B*H*W pixels: block = (sample, band of 8 rows); the band of the disparity map with its halo sits in LDS; a wave's lane is an
// output channel, the four waves walk the band's pixels interleaved, every lane keeps its 49 tap sums + the bias sum in
// registers (the window reads are LDS broadcasts).  Block partials -> fixed-order sum in the finish kernel (deterministic).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kW7Band = 8, kW7Out = 50;  // rows per block; 49 taps + bias

struct W7Params {
  const float* xs[kWgradMaxTensors];
  const float* dys[kWgradMaxTensors];
  float* part;  // [blocks][64][kW7Out]
  int per, B, Cout, H, W, bands;
};

__global__ __launch_bounds__(256) void conv7x7_c1_wgrad_kernel(W7Params p) {
  extern __shared__ __attribute__((aligned(16))) float w7_lds[];
  const int PW = p.W + 6;
  float* xt = w7_lds;                         // [kW7Band + 6][PW]
  float* red = w7_lds + (kW7Band + 6) * PW;   // [4][kW7Out][64]
  const int s = blockIdx.x / p.bands, band = blockIdx.x - s * p.bands;
  const int r0 = band * kW7Band;
  const int rows = min(kW7Band, p.H - r0);
  const float* __restrict__ x = p.xs[s / p.per] + (long long)(s % p.per) * p.H * p.W;
  const float* __restrict__ d = p.dys[s / p.per] + (long long)(s % p.per) * p.Cout * p.H * p.W;
  for (int i = threadIdx.x; i < (kW7Band + 6) * PW; i += 256) {
    const int ry = i / PW, rx = i - ry * PW;
    const int gy = r0 + ry - 3, gx = rx - 3;
    xt[i] = (ry < rows + 6 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? x[(long long)gy * p.W + gx] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = lane < p.Cout;
  const float* __restrict__ dc = d + (long long)(live ? lane : 0) * p.H * p.W + (long long)r0 * p.W;
  float acc[kW7Out];
#pragma unroll
  for (int t = 0; t < kW7Out; ++t) acc[t] = 0.f;
  const int npx = rows * p.W;
  for (int i = wave; i < npx; i += 4) {
    const int r = i / p.W, c = i - r * p.W;
    const float g = live ? dc[i] : 0.f;
    const float* win = xt + r * PW + c;  // window origin: input (row r0 + r - 3, column c - 3)
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) acc[ky * 7 + kx] = fmaf(g, win[ky * PW + kx], acc[ky * 7 + kx]);
    acc[49] += g;
  }
#pragma unroll
  for (int t = 0; t < kW7Out; ++t) red[(wave * kW7Out + t) * 64 + lane] = acc[t];
  __syncthreads();
  for (int i = threadIdx.x; i < kW7Out * 64; i += 256) {
    const int t = i >> 6, co = i & 63;
    const float v = (red[(0 * kW7Out + t) * 64 + co] + red[(1 * kW7Out + t) * 64 + co]) + (red[(2 * kW7Out + t) * 64 + co] + red[(3 * kW7Out + t) * 64 + co]);
    p.part[((long long)blockIdx.x * 64 + co) * kW7Out + t] = v;
  }
}

__global__ __launch_bounds__(256) void conv7x7_c1_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                                       int Cout, int blocks) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Cout * kW7Out) return;
  const int co = i / kW7Out, t = i - co * kW7Out;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 4 <= blocks; k += 4) {
    s0 += part[((long long)k * 64 + co) * kW7Out + t];
    s1 += part[((long long)(k + 1) * 64 + co) * kW7Out + t];
    s2 += part[((long long)(k + 2) * 64 + co) * kW7Out + t];
    s3 += part[((long long)(k + 3) * 64 + co) * kW7Out + t];
  }
  for (; k < blocks; ++k) s0 += part[((long long)k * 64 + co) * kW7Out + t];
  const float v = (s0 + s1) + (s2 + s3);
  if (t < 49) dw[co * 49 + t] = v;
  else if (db) db[co] = v;
}

}  // namespace

extern "C" {

int64_t as_conv7x7_c1_wgrad_ws_bytes(int B, int Cout, int H, int W) {
  if (B <= 0 || Cout <= 0 || Cout > 64 || H <= 0 || W <= 0) return -1;
  return (long long)B * as::cdiv(H, kW7Band) * 64 * kW7Out * 4;
}

int as_conv7x7_c1_wgrad_multi(const float* const* xs, const float* const* dys, int n, int per, float* dw, float* db, int Cout, int H, int W,
                              void* ws, int64_t ws_bytes, void* stream) {
  AS_REQUIRE(xs && dys && dw && ws && n >= 1 && n <= kWgradMaxTensors && per >= 1, AS_ERR_BAD_ARG, "conv7x7_c1_wgrad: null pointer / %d tensors (1..%d)", n, kWgradMaxTensors);
  const int B = n * per;
  const int64_t need = as_conv7x7_c1_wgrad_ws_bytes(B, Cout, H, W);
  AS_REQUIRE(need > 0, AS_ERR_BAD_ARG, "conv7x7_c1_wgrad: B=%d Cout=%d (1..64) H=%d W=%d", B, Cout, H, W);
  AS_REQUIRE(ws_bytes >= need, AS_ERR_BAD_ARG, "conv7x7_c1_wgrad: workspace of %lld bytes, need %lld", (long long)ws_bytes, (long long)need);
  const size_t lds = ((size_t)(kW7Band + 6) * (W + 6) + 4 * kW7Out * 64) * sizeof(float);
  AS_REQUIRE(lds <= 160 * 1024, AS_ERR_BAD_SHAPE, "conv7x7_c1_wgrad: rows of %d pixels do not fit the LDS band", W);
  W7Params p;
  for (int i = 0; i < kWgradMaxTensors; ++i) {
    p.xs[i] = xs[i < n ? i : 0];
    p.dys[i] = dys[i < n ? i : 0];
    AS_REQUIRE(p.xs[i] && p.dys[i], AS_ERR_BAD_ARG, "conv7x7_c1_wgrad: null tensor %d", i);
  }
  p.part = (float*)ws; p.per = per; p.B = B; p.Cout = Cout; p.H = H; p.W = W; p.bands = as::cdiv(H, kW7Band);
  const long long blocks = (long long)B * p.bands;
  AS_REQUIRE(blocks < 2147483647ll, AS_ERR_BAD_SHAPE, "conv7x7_c1_wgrad: grid too large");
  hipStream_t s = as::as_stream(stream);
  as::lds_opt_in((const void*)conv7x7_c1_wgrad_kernel);
  hipLaunchKernelGGL(conv7x7_c1_wgrad_kernel, dim3((unsigned)blocks), dim3(256), lds, s, p);
  hipLaunchKernelGGL(conv7x7_c1_wgrad_finish_kernel, dim3((unsigned)as::cdiv(Cout * kW7Out, 256)), dim3(256), 0, s, (const float*)ws, dw, db, Cout, (int)blocks);
  return as::check_launch("conv7x7_c1_wgrad");
}

int64_t as_conv2d_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W, int KS) {
  WgradPlan q;
  if (!wgrad_plan(B, Cin, Cout, H, W, KS, q)) return -1;
  return q.ws_floats * 4;
}

static int wgrad_run(const float* const* xs, const float* const* dys, int n, int per, float* dw, float* db, int Cin, int Cout, int H, int W,
                     int KS, void* ws, int64_t ws_bytes, void* stream) {
  AS_REQUIRE(xs && dys && dw && ws && n >= 1 && n <= kWgradMaxTensors && per >= 1, AS_ERR_BAD_ARG, "conv2d_wgrad: null pointer / %d tensors (1..%d)", n, kWgradMaxTensors);
  const int B = n * per;
  WgradPlan q;
  AS_REQUIRE(wgrad_plan(B, Cin, Cout, H, W, KS, q), AS_ERR_BAD_ARG, "conv2d_wgrad: KS=%d (1 or 3), B=%d Cin=%d Cout=%d H=%d W=%d", KS, B, Cin, Cout, H, W);
  AS_REQUIRE(ws_bytes >= q.ws_floats * 4, AS_ERR_BAD_ARG, "conv2d_wgrad: workspace of %lld bytes, need %lld", (long long)ws_bytes, (long long)q.ws_floats * 4);
  AS_REQUIRE((long long)Cin * H * W * 4 < 0x7FFFFFF0ll && (long long)Cout * H * W * 4 < 0x7FFFFFF0ll, AS_ERR_BAD_SHAPE, "conv2d_wgrad: an image exceeds 2 GiB");
  AS_REQUIRE(q.XS <= 80 && q.lds <= 160 * 1024, AS_ERR_BAD_SHAPE, "conv2d_wgrad: row segment of %d pixels", q.XS);
  WgradParams p;
  bool vec = (W % 4) == 0;
  for (int i = 0; i < kWgradMaxTensors; ++i) {
    p.xs[i] = xs[i < n ? i : 0];
    p.dys[i] = dys[i < n ? i : 0];
    AS_REQUIRE(p.xs[i] && p.dys[i], AS_ERR_BAD_ARG, "conv2d_wgrad: null tensor %d", i);
    vec = vec && (reinterpret_cast<uintptr_t>(p.xs[i]) % 16) == 0 && (reinterpret_cast<uintptr_t>(p.dys[i]) % 16) == 0;
  }
  p.per = per;
  p.ws = (float*)ws;
  p.wsb = db ? (float*)ws + (long long)q.nsplit * (KS * KS) * Cout * Cin : nullptr;
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
  static const int dbg = getenv("AS_WGRAD_DBG") ? atoi(getenv("AS_WGRAD_DBG")) : 0;
  p.dbg = dbg;
  p.nseg = q.nseg; p.XS = q.XS; p.KST = q.KST; p.pitch_x = q.pitch_x; p.pitch_g = q.pitch_g; p.n_co = q.n_co; p.n_ci = q.n_ci; p.nsplit = q.nsplit; p.chunks = q.chunks;
  hipStream_t s = as::as_stream(stream);
  const int rc = KS == 3 ? wgrad_launch<3, 1>(p, q, vec, s) : wgrad_launch<1, 2>(p, q, vec, s);
  if (rc != AS_OK) return rc;
  const long long n_out = (long long)Cout * Cin * KS * KS;
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)as::cdiv64(n_out > Cout ? n_out : Cout, 256)), dim3(256), 0, s, p.ws, p.wsb, dw, db, Cout,
                     Cin, KS * KS, q.nsplit);
  return as::check_launch("conv2d_wgrad_finish");
}

int as_conv2d_wgrad(const float* x, const float* dy, float* dw, float* db, int B, int Cin, int Cout, int H, int W, int KS, void* ws,
                    int64_t ws_bytes, void* stream) {
  return wgrad_run(&x, &dy, 1, B, dw, db, Cin, Cout, H, W, KS, ws, ws_bytes, stream);
}

int as_conv2d_wgrad_multi(const float* const* xs, const float* const* dys, int n, int per, float* dw, float* db, int Cin, int Cout, int H,
                          int W, int KS, void* ws, int64_t ws_bytes, void* stream) {
  return wgrad_run(xs, dys, n, per, dw, db, Cin, Cout, H, W, KS, ws, ws_bytes, stream);
}

}  // extern "C"
