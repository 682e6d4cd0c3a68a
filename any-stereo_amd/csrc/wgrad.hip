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
// accumulators for 3x3), waves 4-7 LOADERS.  K-chunk = one image row segment of XS = 16*KST pixels:
//   * the dy operand of a wave is private to it (no reuse across waves), so it never touches LDS: each lane fetches its 8
//     consecutive pixels straight from global memory two k-steps ahead (three rotating register sets) and splits them in
//     registers;
//   * the x operand (32*NT channels x KS rows x XS+2 pixels) is shared by all four consumers: the loaders fetch it, split it
//     and park hi / lo in a double-buffered LDS image, one barrier per chunk.  The three kx taps of a row are the same eight
//     pixels shifted by one element: one aligned 16-byte read + the two neighbouring dwords, and five v_alignbit build the
//     kx = 0 and kx = 2 fragments (16-byte reads at 2-byte-aligned addresses ran the loop 6x below the matrix-core rate).
//     Row pitch = 8 * odd elements: the 16 lanes of a read phase hit 16 distinct 16-byte bank groups.
// Split-K: the grid is (K ranges) x (output tiles); each block writes its partial tile to a workspace and
// wgrad_finish_kernel sums the ranges in a fixed order (deterministic, no atomics).  The bias gradient rides along as one
// extra column tile of ones in the blocks of the first channel tile.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float acc16;
typedef __attribute__((ext_vector_type(4))) float f4;

struct WgradParams {
  const float* x;   // [B][Cin][H][W]
  const float* dy;  // [B][Cout][H][W]
  float* ws;        // [nsplit][T][Cout][Cin] partial weight gradients, T = KS*KS
  float* wsb;       // [nsplit][Cout] partial bias gradients (null: none)
  int B, Cin, Cout, H, W;
  int nseg, XS, KST;  // segments per image row, segment width = 16*KST
  int pitch;          // LDS row pitch of the x image (bf16 elements)
  int n_co, n_ci;     // output tiles: 128 rows x 32*NT channels
  int nsplit;
  long long chunks;  // B*H*nseg
};

struct G8 { float v[8]; };
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
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

// x = hi + lo, both bf16 (round to nearest even): pairs -> packed dwords
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  bf2 h;
  h[0] = (__bf16)a;
  h[1] = (__bf16)b;
  __builtin_memcpy(&hi, &h, 4);
  const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xFFFF0000u);
  bf2 l;
  l[0] = (__bf16)ra;
  l[1] = (__bf16)rb;
  __builtin_memcpy(&lo, &l, 4);
}

__device__ __forceinline__ void split8(const G8& g, bf8& hi, bf8& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_pair(g.v[2 * j], g.v[2 * j + 1], h[j], l[j]);
  __builtin_memcpy(&hi, h, 16);
  __builtin_memcpy(&lo, l, 16);
}

struct Pos { int b, y, seg; };  // a K-chunk: image b, row y, row segment seg
__device__ __forceinline__ Pos pos_of(const WgradParams& p, long long c) {  // one division per block, then pos_next
  Pos q;
  const long long ru = c / p.nseg;
  q.seg = (int)(c - ru * p.nseg);
  q.b = (int)(ru / p.H);
  q.y = (int)(ru - (long long)q.b * p.H);
  return q;
}
__device__ __forceinline__ void pos_next(const WgradParams& p, Pos& q) {
  if (++q.seg == p.nseg) {
    q.seg = 0;
    if (++q.y == p.H) { q.y = 0; ++q.b; }
  }
}

// the dy fragment of k-step ks of chunk q for this lane: 8 consecutive pixels of row `co` (zeros outside the row / tile / range)
template <bool VEC>
__device__ __forceinline__ G8 dy_load(const WgradParams& p, const float* __restrict__ dy, const Pos& q, int ks, bool live, int co, int h) {
  G8 g;
  const int col0 = q.seg * p.XS + ks * 16 + 8 * h;
  const long long row = live ? (((long long)q.b * p.Cout + co) * p.H + q.y) * p.W : 0;
  if (VEC) {
    const bool ok = live && col0 < p.W;  // W % 8 == 0: the group is inside the row or outside it as a whole
    const f4* src = reinterpret_cast<const f4*>(dy + (ok ? row + col0 : 0));
    const f4 a = src[0], b = src[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g.v[j] = ok ? a[j] : 0.f;
      g.v[4 + j] = ok ? b[j] : 0.f;
    }
  } else {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = dy[(live && col0 + j < p.W) ? row + col0 + j : 0];
#pragma unroll
    for (int j = 0; j < 8; ++j) g.v[j] = (live && col0 + j < p.W) ? t[j] : 0.f;
  }
  return g;
}

template <int KS, int NT, bool VEC>
__global__ __launch_bounds__(512, 2) void wgrad_kernel(WgradParams p) {
  constexpr int TAPS = KS * KS, PAD = KS / 2, NB = TAPS * NT, CIB = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const float* __restrict__ xg = p.x;
  const float* __restrict__ dyg = p.dy;
  constexpr int PADL = KS == 3 ? 8 : 0;          // stored element s = col - x0 + PADL: the centre tap's 8-pixel groups are 16-byte aligned
  const int pitch = p.pitch;                     // 8 * odd >= XS + PADL + 2
  const int img = CIB * KS * pitch;              // bf16 elements of one (hi or lo) image
  unsigned short* const lds = reinterpret_cast<unsigned short*>(smem);  // [buf 2][hi, lo][CIB][KS][pitch]
  const int tid = threadIdx.x;
  const int tiles = p.n_co * p.n_ci;
  const int split = blockIdx.x / tiles, tile = blockIdx.x - split * tiles;
  const int tco = tile / p.n_ci, tci = tile - tco * p.n_ci;
  const int co0 = tco * 128, ci0 = tci * CIB;
  const long long c_lo = p.chunks * split / p.nsplit, c_hi = p.chunks * (split + 1) / p.nsplit;

  if (tid >= 256) {
    // ---------------- loaders: the x image of chunk c -> LDS buffer ----------------
    const int lt = tid - 256, cl = lt >> 3, qd = lt & 7;
    const int npair = KS == 3 ? p.XS / 2 + 2 : p.XS / 2;  // element pairs s = PADL - 2 + 2m + {0,1} (KS = 3), 2m + {0,1} (KS = 1)
    constexpr int C0 = KS == 3 ? -2 : 0, S0 = KS == 3 ? PADL - 2 : 0;
    constexpr int NIT = 7;  // pair columns per thread: npair <= 8 * NIT (XS <= 96)
    Pos q = pos_of(p, c_lo);
    for (long long cn = c_lo; cn <= c_hi; ++cn) {  // stage chunk cn (the prologue stages c_lo), then meet the consumers
      if (cn < c_hi) {
        const int x0 = q.seg * p.XS;
        unsigned short* const dst = lds + (int)((cn - c_lo) & 1) * 2 * img;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          const int cil = ct * 32 + cl, ci = ci0 + cil;
          const bool ci_ok = ci < p.Cin;
#pragma unroll
          for (int ky = 0; ky < KS; ++ky) {
            const int yy = q.y + ky - PAD;
            const bool row_ok = ci_ok && yy >= 0 && yy < p.H;
            const long long row = row_ok ? (((long long)q.b * p.Cin + ci) * p.H + yy) * p.W : 0;
            float va[NIT], vb[NIT];
#pragma unroll
            for (int i = 0; i < NIT; ++i) {  // every load in flight before the first conversion
              const int ca = x0 + 2 * (qd + 8 * i) + C0, cb = ca + 1;
              va[i] = xg[(row_ok && ca >= 0 && ca < p.W) ? row + ca : 0];
              vb[i] = xg[(row_ok && cb >= 0 && cb < p.W) ? row + cb : 0];
            }
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
              const int m = qd + 8 * i;
              const int ca = x0 + 2 * m + C0, cb = ca + 1;
              const bool oa = row_ok && ca >= 0 && ca < p.W, ob = row_ok && cb >= 0 && cb < p.W;
              if (m < npair) {
                unsigned hi, lo;
                split_pair(oa ? va[i] : 0.f, ob ? vb[i] : 0.f, hi, lo);
                const int e = (cil * KS + ky) * pitch + S0 + 2 * m;
                *reinterpret_cast<unsigned*>(dst + e) = hi;
                *reinterpret_cast<unsigned*>(dst + img + e) = lo;
              }
            }
          }
        }
        pos_next(p, q);
      }
      __syncthreads();
    }
    return;
  }

  // ---------------- consumers ----------------
  const int lane = tid & 63, r = lane & 31, h = lane >> 5, w = tid >> 6;
  const int co = co0 + 32 * w + r;
  const bool co_ok = co < p.Cout;
  const bool do_bias = p.wsb != nullptr && tci == 0;
  acc16 acc[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  acc16 accb;
#pragma unroll
  for (int i = 0; i < 16; ++i) accb[i] = 0.f;
  bf8 ones;
  {
    unsigned o[4] = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    __builtin_memcpy(&ones, o, 16);
  }
  const long long steps = (c_hi - c_lo) * p.KST;
  // (chunk, k-step) of the step being computed and of the step whose dy is fetched (two ahead)
  long long cc = c_lo, cf = c_lo;
  int ks = 0, kf = 0;
  Pos qf = pos_of(p, c_lo);
#define WG_FETCH(TGT)                                                  \
  TGT = dy_load<VEC>(p, dyg, qf, kf, co_ok && cf < c_hi, co, h);       \
  if (++kf == p.KST) { kf = 0; ++cf; pos_next(p, qf); }
  G8 R0, R1, R2;
  WG_FETCH(R0)
  WG_FETCH(R1)
  __syncthreads();  // chunk c_lo staged
  const int lane_e = (r * KS) * pitch + PADL + 8 * h;  // this lane's element inside a column tile's image (tap row 0, k-step 0, centre tap)

#define WG_STEP(CUR, TGT)                                                                                              \
  {                                                                                                                    \
    WG_FETCH(TGT)                                                                                                      \
    bf8 a_hi, a_lo;                                                                                                    \
    split8(CUR, a_hi, a_lo);                                                                                           \
    const unsigned short* xb = lds + (int)((cc - c_lo) & 1) * 2 * img + lane_e + ks * 16;                              \
    _Pragma("unroll") for (int ct = 0; ct < NT; ++ct) {                                                                \
      _Pragma("unroll") for (int ky = 0; ky < KS; ++ky) {                                                              \
        const unsigned short* xe = xb + (ct * 32 * KS + ky) * pitch;                                                   \
        const u4 dh = *reinterpret_cast<const u4*>(xe), dl = *reinterpret_cast<const u4*>(xe + img);                    \
        if (KS == 3) {                                                                                                 \
          const unsigned lh = *reinterpret_cast<const unsigned*>(xe - 2), rh = *reinterpret_cast<const unsigned*>(xe + 8);           \
          const unsigned ll = *reinterpret_cast<const unsigned*>(xe + img - 2), rl = *reinterpret_cast<const unsigned*>(xe + img + 8); \
          const B3 bh = tap_shifts(dh, lh, rh), bl = tap_shifts(dl, ll, rl);                                           \
          const int t = ct * TAPS + ky * KS;                                                                           \
          _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) acc[t + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bh.k[kx], acc[t + kx], 0, 0, 0); \
          _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) acc[t + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bl.k[kx], acc[t + kx], 0, 0, 0); \
          _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) acc[t + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, bh.k[kx], acc[t + kx], 0, 0, 0); \
        } else {                                                                                                       \
          const bf8 b_hi = __builtin_bit_cast(bf8, dh), b_lo = __builtin_bit_cast(bf8, dl);                            \
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[ct], 0, 0, 0);                             \
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[ct], 0, 0, 0);                             \
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[ct], 0, 0, 0);                             \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
    if (do_bias) {                                                                                                     \
      accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, ones, accb, 0, 0, 0);                                       \
      accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, ones, accb, 0, 0, 0);                                       \
    }                                                                                                                  \
    if (++ks == p.KST) {                                                                                               \
      ks = 0;                                                                                                          \
      ++cc;                                                                                                            \
      __syncthreads();                                                                                                 \
    }                                                                                                                  \
  }

  for (long long n = 0; n < steps; n += 3) {
    WG_STEP(R0, R2)
    if (n + 1 < steps) WG_STEP(R1, R0)
    if (n + 2 < steps) WG_STEP(R2, R1)
  }
#undef WG_STEP
#undef WG_FETCH

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

// dW[co][ci][t] = sum_s ws[s][t][co][ci] (fixed order); db[co] = sum_s wsb[s][co]
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, float* __restrict__ dw,
                                                           float* __restrict__ db, int Cout, int Cin, int T, int nsplit) {
  const long long plane = (long long)Cout * Cin;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < plane) {
    for (int t = 0; t < T; ++t) {
      float s = 0.f;
      for (int k = 0; k < nsplit; ++k) s += ws[((long long)k * T + t) * plane + i];
      dw[i * T + t] = s;
    }
  }
  if (db && wsb && i < Cout) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += wsb[(long long)k * Cout + i];
    db[i] = s;
  }
}

struct WgradPlan {
  int nseg, XS, KST, pitch, n_co, n_ci, nsplit, NT;
  long long chunks;
  size_t lds;
  long long ws_floats;
};

bool wgrad_plan(int B, int Cin, int Cout, int H, int W, int KS, WgradPlan& q) {
  if (!(KS == 1 || KS == 3) || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return false;
  q.NT = KS == 1 ? 4 : 1;
  q.nseg = as::cdiv(W, 96);
  q.XS = as::cdiv(as::cdiv(W, q.nseg), 16) * 16;
  q.KST = q.XS / 16;
  q.n_co = as::cdiv(Cout, 128);
  q.n_ci = as::cdiv(Cin, 32 * q.NT);
  q.chunks = (long long)B * H * q.nseg;
  const int tiles = q.n_co * q.n_ci;
  long long ns = (512 + tiles / 2) / tiles;  // about two rounds of blocks on 256 CUs
  if (ns > q.chunks / 4) ns = q.chunks / 4;   // at least four chunks per block
  if (ns < 1) ns = 1;
  if (ns > 256) ns = 256;
  q.nsplit = (int)ns;
  q.pitch = q.XS + (KS == 3 ? 10 : 0);  // >= XS + PADL + 2, rounded up to 8 * odd
  q.pitch = (q.pitch + 7) / 8 * 8;
  if ((q.pitch / 8) % 2 == 0) q.pitch += 8;
  q.lds = (size_t)2 * 2 * (32 * q.NT) * KS * q.pitch * sizeof(unsigned short);
  q.ws_floats = (long long)q.nsplit * (KS * KS) * Cout * Cin + (long long)q.nsplit * Cout;
  return true;
}

template <int KS, int NT>
int wgrad_launch(const WgradParams& p, const WgradPlan& q, bool vec, hipStream_t s) {
  const dim3 grid((unsigned)(q.nsplit * q.n_co * q.n_ci));
  if (vec) {
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<KS, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds);
    hipLaunchKernelGGL((wgrad_kernel<KS, NT, true>), grid, dim3(512), q.lds, s, p);
  } else {
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<KS, NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds);
    hipLaunchKernelGGL((wgrad_kernel<KS, NT, false>), grid, dim3(512), q.lds, s, p);
  }
  return as::check_launch("conv2d_wgrad");
}

}  // namespace

extern "C" {

int64_t as_conv2d_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W, int KS) {
  WgradPlan q;
  if (!wgrad_plan(B, Cin, Cout, H, W, KS, q)) return -1;
  return q.ws_floats * 4;
}

int as_conv2d_wgrad(const float* x, const float* dy, float* dw, float* db, int B, int Cin, int Cout, int H, int W, int KS, void* ws,
                    int64_t ws_bytes, void* stream) {
  AS_REQUIRE(x && dy && dw && ws, AS_ERR_BAD_ARG, "conv2d_wgrad: null pointer");
  WgradPlan q;
  AS_REQUIRE(wgrad_plan(B, Cin, Cout, H, W, KS, q), AS_ERR_BAD_ARG, "conv2d_wgrad: KS=%d (1 or 3), B=%d Cin=%d Cout=%d H=%d W=%d", KS, B, Cin, Cout, H, W);
  AS_REQUIRE(ws_bytes >= q.ws_floats * 4, AS_ERR_BAD_ARG, "conv2d_wgrad: workspace of %lld bytes, need %lld", (long long)ws_bytes, (long long)q.ws_floats * 4);
  AS_REQUIRE((long long)B * Cin * H * W < (1ll << 40) && (long long)B * Cout * H * W < (1ll << 40), AS_ERR_BAD_SHAPE, "conv2d_wgrad: tensor too large");
  AS_REQUIRE(q.XS <= 96 && q.lds <= 160 * 1024, AS_ERR_BAD_SHAPE, "conv2d_wgrad: row segment of %d pixels", q.XS);
  WgradParams p;
  p.x = x; p.dy = dy;
  p.ws = (float*)ws;
  p.wsb = db ? (float*)ws + (long long)q.nsplit * (KS * KS) * Cout * Cin : nullptr;
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
  p.nseg = q.nseg; p.XS = q.XS; p.KST = q.KST; p.pitch = q.pitch; p.n_co = q.n_co; p.n_ci = q.n_ci; p.nsplit = q.nsplit; p.chunks = q.chunks;
  hipStream_t s = as::as_stream(stream);
  const bool vec = (W % 8) == 0 && (reinterpret_cast<uintptr_t>(dy) % 16) == 0;
  const int rc = KS == 3 ? wgrad_launch<3, 1>(p, q, vec, s) : wgrad_launch<1, 4>(p, q, vec, s);
  if (rc != AS_OK) return rc;
  const long long plane = (long long)Cout * Cin;
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)as::cdiv64(plane > Cout ? plane : Cout, 256)), dim3(256), 0, s, p.ws, p.wsb, dw, db, Cout,
                     Cin, KS * KS, q.nsplit);
  return as::check_launch("conv2d_wgrad_finish");
}

}  // extern "C"
