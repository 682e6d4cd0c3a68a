// a6/a7/a9/a15: implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact
// fp32 fma chains => results track the reference's fp32 convs to rounding order), with the GRU
// gate arithmetic fused into the epilogue; plus the two degenerate shapes as direct VALU kernels
// and the pool2x / interp resamplers (a8).
//
// GEMM view:  D[co][pixel] = sum_k Wp[k][co] * X[k][pixel],  k = (tap, channel).
//   MFMA A operand = weights  (A[i=co][k]  : lane&31 -> co,    lane>>5 -> k parity)
//   MFMA B operand = patch    (B[k][j=pix] : lane&31 -> pixel, lane>>5 -> k parity)
//   C/D: col j = lane&31 = pixel  => every epilogue load/store is a 128-B run along x in NCHW.
//
// Block = 4 waves, output tile 64 co x 64 pixels (2x2 MFMA tiles).  All four waves accumulate the
// WHOLE tile over a quarter of K each (intra-block split-K: wave w owns channel pairs w, w+4, ... of
// every chunk), so the serial MFMA chain per wave is K/8 instead of K/2: small feature maps
// (1/8, 1/16 resolution: tens of tiles) finish in a quarter of the time and large ones decompose into
// thousands of equal blocks that balance over the 256 CUs.  The four partial tiles are reduced
// through LDS, each wave then owns ONE 32x32 tile of the epilogue.
// Per channel chunk the block stages an input halo patch [KC][TH+KS-1][TW+KS-1] and the packed
// weight slab [KS*KS*KC][64] in LDS, double buffered: the next chunk is fetched into registers
// while the MFMAs of the current one run, one barrier per chunk.  Inside a chunk k is ordered
// (tap, channel) so every ds_read address is lane_base + compile-time immediate.  The channel concat
// of the reference (torch.cat([h, x...])) is never materialised: a chunk's channels are read from
// whichever source tensor owns them.
#include <stdlib.h>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using half4v = __attribute__((ext_vector_type(4))) _Float16;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

struct ConvParams {
  const float* src[AS_MAX_SRCS];
  int src_c[AS_MAX_SRCS];
  int src_end[AS_MAX_SRCS];  // exclusive prefix end of each source's channel range
  int n_src;
  const float* wpack;
  const float* bias;
  const float* add;
  int add_ctot, add_coff;
  const float* h;
  const float* z;
  float* out;
  float* out2;
  int out_ctot, out_coff;
  int B, H, W, Cin, Cout, Cout_pad, act;  // H, W: OUTPUT plane (= input plane for stride 1)
  int Hi, Wi;                             // input plane (conv_split_kernel with stride 2)
  int tiles_x, tiles_y, n_tiles, chunks;
  // Blocked split-fp16 tensors ("BS8", conv_split_kernel only): [B][2][ceil(C/8)][H][W][8 halves] = a plane set of hi parts and
  // one of lo parts of the operand split, 8 channels of a pixel contiguous — exactly the 16-B units of the kernel's LDS patch
  // image, so a loader item is two fully coalesced 16-B loads and no arithmetic instead of eight dword loads and the split.
  int src_bs[AS_MAX_SRCS];     // 1: src[i] is such a tensor (src_c[i] logical channels)
  _Float16* out_bs;            // optional blocked copy of the result (LINEAR: out, GRU_Q: out, GRU_ZR: out2 = r*h)
  int out_bs_c8tot, out_bs_coff8;  // blocks of the destination tensor per batch element; first block of the window
  int out_bs_ctot;             // logical channels of the destination tensor (slots past them are zero-filled, never another producer's)
  int bs_only;                 // 1: the fp32 copy of that result is not written
  int all_bs;                  // every source (and a dual launch's second one) is blocked: all-DMA operand staging where it fits
  // Dual launch (conv_split_kernel, LINEAR epilogue, one source): a second convolution of the same shape rides in the same grid
  // as extra output-channel tiles [n_tiles/2, n_tiles) with its own source, weights, bias and output channel window — the two
  // 64 -> 64 branch convs of the motion encoder (update.py:86,88) are one launch instead of two on the loop's critical stream.
  int dual;
  const float* src2;
  int src2_bs;
  const float* wpack2;
  const float* bias2;
  int out_coff2, out_bs_coff8_2;
  // ... optionally with its own residual (both convolutions or neither), activation and DENSE outputs of its own (dual_sep: out_b
  // [B,Cout,H,W] / out_bs_b blocked, Cout channels) instead of a channel window of out / out_bs — the two heads of a context-network
  // scale (extractor.py:254-273: same input, same shapes, different weights) as one launch per layer
  const float* h2;
  int act2;
  int dual_sep;
  float* out_b;
  _Float16* out_bs_b;
  // AS_EPI_RELU_TAPS: [Cout][9] weights of a following 3x3, Cout -> 1 convolution whose per-tap channel reductions this
  // conv's epilogue accumulates instead of storing its own result (out = [B][n_tiles * 9][H][W])
  const float* tap_w;
  int xcd_map;  // conv_split_kernel: XCD-aware block order (1: channel tiles of one pixel tile on the same XCD; 2: + a contiguous band of pixel tiles per XCD)
  int fast16;   // conv_split_kernel: skip the two cross-term MFMAs (fp16 operands, fp32 accumulate: as_set_fast16)
  int ksplit;   // split-K factor (conv_split_kernel, EPI = kEpiPartial): blocks per output tile
  float* ws;    // [ksplit][B][Cout_pad][H][W] fp32 partial sums
  int stagger;  // conv_split_kernel: start delay per XCD index in units of 64 clocks (AS_CONV_XCD_STAGGER; 0 = none)
  int lean_offset;  // conv_split_kernel<LEAN>: start delay of the grid's second half in units of 64 clocks (AS_CONV_LEAN_OFFSET)
};

constexpr int kNumCU = 256;   // MI355X
constexpr int kEpiPartial = 3;  // internal: store the raw partial sums of a K slice into p.ws
constexpr int kEpiTaps = AS_EPI_RELU_TAPS;

template <int KS> struct ConvCfg;
template <> struct ConvCfg<3> { static constexpr int KC = 8; };
template <> struct ConvCfg<1> { static constexpr int KC = 32; };

constexpr int kBM = 64;  // pixels per block
constexpr int kBN = 64;  // output channels per block

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case AS_ACT_RELU: return fmaxf(v, 0.f);
    case AS_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case AS_ACT_TANH: return tanhf(v);
    case AS_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
    case AS_ACT_LEAKY: return v >= 0.f ? v : 0.01f * v;
    default: return v;
  }
}

// ---- epilogue of one 32(co) x 32(pixel) accumulator tile -----------------------------------------
// All global operands of the tile (context term, h, z) are fetched up front through raw buffer
// descriptors whose range covers exactly the block's valid output channels: lanes outside the image
// (sentinel offset) and channels >= Cout read 0 and their stores are dropped by the hardware range check,
// so there is not a single branch around a memory operation and one latency is paid per tile, not per
// element.  Bias comes from an LDS copy made by the caller.
__device__ __forceinline__ float as_bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ void as_bstore(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);
}

struct EpiCtx {
  __amdgpu_buffer_rsrc_t r_add, r_out, r_h, r_z, r_bs, r_bsl;
  bool has_add, has_bs, skip_out;
  int act;           // activation of this block's convolution (a dual launch's second one may have its own)
  int cvalid;        // valid output channels of this block's tile
  int cend;          // blocked copy: channels from the tile's first to the destination tensor's logical end (slots past it are zeroed;
                     // slots in [cvalid, cend) belong to another producer and are left untouched)
  unsigned plane4;   // bytes per channel plane
};

// n0: first output channel of the block, bn: channels per block (descriptor windows start at channel n0)
template <int EPI>
__device__ __forceinline__ EpiCtx make_epi_ctx(const ConvParams& p, int b, int n0, int bn, bool second = false) {
  EpiCtx e;
  const long long plane = (long long)p.H * p.W;
  const int cvalid = min(p.Cout - n0, bn);
  const int recs = cvalid > 0 ? (int)((long long)cvalid * plane * 4) : 0;
  e.plane4 = (unsigned)(plane * 4);
  e.act = second ? p.act2 : p.act;
  const bool sep = second && p.dual_sep;  // the second convolution of a dual launch writes dense tensors of its own
  e.has_add = p.add != nullptr;
  const float* addp = p.add ? p.add + ((long long)b * p.add_ctot + p.add_coff + n0) * plane : p.out;
  e.r_add = __builtin_amdgcn_make_buffer_rsrc((void*)addp, 0, p.add ? recs : 0, 0x00020000);
  if (EPI == kEpiPartial) {
    // slab of this block's K slice; every channel of the padded tile is stored (the finish kernel ignores >= Cout)
    const int ks = (int)(blockIdx.x / ((unsigned)p.B * p.tiles_x * p.tiles_y * p.n_tiles));
    float* dst = p.ws + (((long long)ks * p.B + b) * p.Cout_pad + n0) * plane;
    e.r_out = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)((long long)bn * plane * 4), 0x00020000);
    e.r_add = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, 0, 0x00020000);
    e.has_add = false;
    e.r_h = e.r_out;
    e.r_z = e.r_out;
  } else if (EPI == AS_EPI_LINEAR) {
    float* outp = sep ? p.out_b + ((long long)b * p.Cout + n0) * plane
                      : p.out + ((long long)b * p.out_ctot + (second ? p.out_coff2 : p.out_coff) + n0) * plane;
    e.r_out = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, recs, 0x00020000);
    // optional residual (ResidualBlock tail, extractor.py:56-62): out = relu(h + act(...)); 0 records when absent
    const float* hsel = second ? p.h2 : p.h;
    const float* res = hsel ? hsel + ((long long)b * p.Cout + n0) * plane : outp;
    e.r_h = __builtin_amdgcn_make_buffer_rsrc((void*)res, 0, hsel ? recs : 0, 0x00020000);
    e.r_z = e.r_out;
  } else if (EPI == AS_EPI_GRU_ZR) {
    const int ch = p.Cout >> 1;
    const bool is_r = n0 >= ch;            // block-uniform: bn divides ch
    const int c0 = is_r ? n0 - ch : n0;    // channel inside the [B,ch,H,W] outputs
    float* dst = (is_r ? p.out2 : p.out) + ((long long)b * ch + c0) * plane;
    e.r_out = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, recs, 0x00020000);
    e.r_h = __builtin_amdgcn_make_buffer_rsrc((void*)(p.h + ((long long)b * ch + c0) * plane), 0, is_r ? recs : 0, 0x00020000);
    e.r_z = e.r_out;
  } else {
    const long long o = ((long long)b * p.Cout + n0) * plane;
    e.r_out = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + o), 0, recs, 0x00020000);
    e.r_h = __builtin_amdgcn_make_buffer_rsrc((void*)(p.h + o), 0, recs, 0x00020000);
    e.r_z = __builtin_amdgcn_make_buffer_rsrc((void*)(p.z + o), 0, recs, 0x00020000);
  }
  // blocked split-fp16 copy: window of this block's channel tile (n0 and the window offset are multiples of 8)
  e.has_bs = false;
  e.skip_out = false;
  e.cvalid = cvalid;
  e.cend = cvalid;
  e.r_bs = e.r_out;
  e.r_bsl = e.r_out;
  if (EPI != kEpiPartial && (sep ? p.out_bs_b != nullptr : p.out_bs != nullptr)) {
    int c0 = n0;
    bool on = true;
    if (EPI == AS_EPI_GRU_ZR) { const int ch = p.Cout >> 1; on = n0 >= ch; c0 = n0 - ch; }
    if (on) {
      const int nblk = cvalid > 0 ? (cvalid + 7) / 8 : 0;
      const int coff8 = sep ? 0 : (second ? p.out_bs_coff8_2 : p.out_bs_coff8);
      const int c8tot = sep ? (p.Cout + 7) / 8 : p.out_bs_c8tot;
      _Float16* dst = (sep ? p.out_bs_b : p.out_bs) + (((long long)b * 2 * c8tot + coff8 + (c0 >> 3)) * plane) * 8;
      e.r_bs = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)((long long)nblk * plane * 16), 0x00020000);
      e.r_bsl = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + (long long)c8tot * plane * 8), 0, (int)((long long)nblk * plane * 16), 0x00020000);
      e.has_bs = true;
      e.cend = (sep ? p.Cout : p.out_bs_ctot) - (coff8 * 8 + c0);
      e.skip_out = p.bs_only != 0;
    }
  }
  return e;
}

// The epilogue of a block is a sequence of ROUNDS (one per 8 accumulator rows of a 32x32 tile).  Each round
// needs up to three global operands (context term, h, z); the rounds are software pipelined: epi_load(i+1) is
// issued before epi_finish(i) computes and stores, so only the first memory latency is exposed per block.
// col0: channel of accumulator row 0 relative to n0; poff: byte offset of this lane's pixel inside a channel
// plane or the OOB sentinel; bias_s: LDS bias of the block (index = channel - n0).
__device__ unsigned g_split_overflow_conv;  // as_conv_split_overflow

struct EpiRegs {
  unsigned off[8];
  float av[8], hv[8], zv[8];
};

template <int EPI>
__device__ __forceinline__ void epi_load(const EpiCtx& e, int col0, int half, unsigned poff, int g, EpiRegs& R) {
  constexpr unsigned kOOB = 0x7FFFFFF0u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = g * 8 + i;
    const int col = col0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    R.off[i] = poff == kOOB ? kOOB : (unsigned)col * e.plane4 + poff;
    R.av[i] = as_bload(e.r_add, R.off[i]);                         // 0 when there is no add tensor (0 records)
    if (EPI == AS_EPI_GRU_ZR || EPI == AS_EPI_LINEAR) R.hv[i] = as_bload(e.r_h, R.off[i]);  // 0 records: z half / no residual
    if (EPI == AS_EPI_GRU_Q) { R.hv[i] = as_bload(e.r_h, R.off[i]); R.zv[i] = as_bload(e.r_z, R.off[i]); }
  }
}

template <int EPI>
__device__ __forceinline__ void epi_finish(const ConvParams& p, const EpiCtx& e, const f32x16& v, int col0, int half, int g,
                                           const float* bias_s, bool is_r, const EpiRegs& R, float& amax,
                                           unsigned poff = 0x7FFFFFF0u) {
  float ov[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = g * 8 + i;
    const int col = col0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const float x = v[r] + bias_s[col] + R.av[i];
    float o;
    if (EPI == kEpiPartial) {
      o = v[r];
    } else if (EPI == AS_EPI_LINEAR) {
      o = act_apply(x, e.act);
      if (p.h) o = fmaxf(o + R.hv[i], 0.f);
    } else if (EPI == AS_EPI_GRU_ZR) {
      const float gte = 1.f / (1.f + expf(-x));
      o = is_r ? gte * R.hv[i] : gte;
    } else {
      o = (1.f - R.zv[i]) * R.hv[i] + R.zv[i] * tanhf(x);
    }
    if (!e.skip_out) as_bstore(e.r_out, R.off[i], o);
    ov[i] = o;
  }
  if (EPI != kEpiPartial && e.has_bs) {
    // rows g*8 .. g*8+7 of the tile = channels [8 m + 4 half, +4) of blocks m = col0/8 + 2g, 2g+1: this lane's 8-B share of
    // the block's 16-B pixel unit in the hi and in the lo plane set (the lane with the other `half` writes the other share)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      half4v hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float o = ov[m * 4 + k];
        const _Float16 hk = (_Float16)o;
        hi[k] = hk;
        lo[k] = (_Float16)((o - (float)hk) * 2048.f);
      }
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(ov[m * 4]), fabsf(ov[m * 4 + 1]))), fmaxf(fabsf(ov[m * 4 + 2]), fabsf(ov[m * 4 + 3])));
      const unsigned blk = (unsigned)(col0 >> 3) + 2u * g + m;
      const unsigned off = poff == 0x7FFFFFF0u ? poff : blk * (e.plane4 * 4u) + poff * 4u + (unsigned)half * 8u;
      const int crel = (int)blk * 8 + 4 * half;  // first of this lane's four channels, relative to the tile
      if (crel + 4 <= e.cvalid) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), e.r_bs, (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), e.r_bsl, (int)off, 0, 0);
      } else {  // the result's last channels end inside or before this group: 2-B stores
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool own = crel + k < e.cvalid, pad = crel + k >= e.cend;
          if (own || pad) {
            const unsigned ok = off == 0x7FFFFFF0u ? off : off + 2u * k;
            const _Float16 zero = (_Float16)0.f;
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, own ? hi[k] : zero), e.r_bs, (int)ok, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, own ? lo[k] : zero), e.r_bsl, (int)ok, 0, 0);
          }
        }
      }
    }
  }
}

template <int EPI>
__device__ __forceinline__ void epilogue_tile(const ConvParams& p, const EpiCtx& e, const f32x16& v, int col0, int half,
                                              unsigned poff, const float* bias_s, bool is_r) {
  EpiRegs R0, R1;
  float amax = 0.f;  // fp32 path: no blocked split-fp16 output, nothing to track
  epi_load<EPI>(e, col0, half, poff, 0, R0);
  epi_load<EPI>(e, col0, half, poff, 1, R1);
  epi_finish<EPI>(p, e, v, col0, half, 0, bias_s, is_r, R0, amax);
  epi_finish<EPI>(p, e, v, col0, half, 1, bias_s, is_r, R1, amax);
}

// Block epilogue of the split-precision kernel: rounds (q, c, g) over the wave's PTW x 2 accumulator tiles,
// loads of round i+1 in flight while round i is finished.  poff[q]: pixel offset of the lane in pixel tile q.
template <int EPI, int PTW>
__device__ __forceinline__ void epilogue_block(const ConvParams& p, const EpiCtx& e, const f32x16 (&acc_h)[2][PTW],
                                               const f32x16 (&acc_x)[2][PTW], int co_base, const unsigned (&poff)[PTW],
                                               int half, const float* bias_s, bool is_r, float& amax) {
  constexpr int NR = PTW * 4;
  // operand loads run TWO rounds ahead of the arithmetic + stores (three register sets): with one block per CU nothing else
  // hides their latency, and the epilogue is ~15 % of a 24-chunk block's life
  EpiRegs R[3];
  epi_load<EPI>(e, co_base, half, poff[0], 0, R[0]);
  if (NR > 1) epi_load<EPI>(e, co_base, half, poff[0], 1, R[1]);
  f32x16 v;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int q = i >> 2, c = (i >> 1) & 1, g = i & 1;
    if (i + 2 < NR) {
      const int i2 = i + 2;
      epi_load<EPI>(e, co_base + ((i2 >> 1) & 1) * 32, half, poff[i2 >> 2], i2 & 1, R[i2 % 3]);
    }
    if (g == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc_h[c][q][r] + acc_x[c][q][r] * (1.f / 2048.f);
    }
    epi_finish<EPI>(p, e, v, co_base + c * 32, half, g, bias_s, is_r, R[i % 3], amax, poff[q]);
  }
}

// KS: 1 or 3.  TW: tile width in pixels (tile = (64/TW) rows x TW cols; KS==1 uses TW=64 on the
// flattened H*W plane).
template <int KS, int TW, int EPI>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
  constexpr int KC = ConvCfg<KS>::KC;
  constexpr int TH = kBM / TW;
  constexpr int PAD = KS / 2;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
  constexpr int PATCH = PH * PW;
  constexpr int KROWS = KC * KS * KS;
  constexpr int PATCH_ELEMS = KC * PATCH;
  constexpr int W_ELEMS = KROWS * kBN;
  constexpr int STAGE = PATCH_ELEMS + W_ELEMS;            // floats per buffer (W slab first, 16-B aligned)
  constexpr int RED = 12 * 16 * 64;                       // 12 partial tiles of 16 regs x 64 lanes
  constexpr int LDS_FLOATS = (2 * STAGE > RED) ? 2 * STAGE : RED;
  constexpr int NP = (PATCH_ELEMS + 255) / 256;           // patch elements prefetched per thread
  constexpr int NW4 = (W_ELEMS / 4 + 255) / 256;          // weight float4s prefetched per thread
  constexpr int CPW = KC / 8;                             // channel pairs per wave per chunk
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;

  // block -> (co tile [slowest: one weight slab set stays L2-hot], batch, tile y, tile x)
  int id = blockIdx.x;
  const int tx = id % p.tiles_x;
  id /= p.tiles_x;
  const int ty = id % p.tiles_y;
  id /= p.tiles_y;
  const int b = id % p.B;
  const int nt = id / p.B;
  const int x0 = tx * TW, y0 = ty * TH;
  const int n0 = nt * kBN;
  const long long plane = (long long)p.H * p.W;

  // ---- per-thread staging descriptors (chunk independent) ----
  // Patch elements are fetched with raw buffer loads: the descriptor (chunk-uniform, in SGPRs) covers
  // the channels of the owning source tensor from this chunk's first channel on, so elements outside
  // the image (sentinel offset) and channels beyond Cin read as 0 by the hardware range check —
  // no branches, no selects, every load of a chunk in flight together.
  unsigned p_voff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int idx = tid + i * 256;
    const bool slot = idx < PATCH_ELEMS;
    if (!slot) idx = PATCH_ELEMS - 1;
    const int c = idx / PATCH;
    const int r = idx - c * PATCH;
    const int py = r / PW, px = r - py * PW;
    const int gy = y0 - PAD + py, gx = x0 - PAD + px;
    const bool in = slot && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    p_voff[i] = in ? (unsigned)(((long long)c * plane + (long long)gy * p.W + gx) * 4) : 0x7FFFFFF0u;
  }
  int w_idx[NW4];
#pragma unroll
  for (int i = 0; i < NW4; ++i) {
    int idx = tid + i * 256;
    if (idx >= W_ELEMS / 4) idx = W_ELEMS / 4 - 1;
    w_idx[i] = (idx / (kBN / 4)) * (p.Cout_pad >> 2) + (idx % (kBN / 4));
  }
  const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wpack + n0);
  const long long wchunk4 = (long long)KROWS * (p.Cout_pad >> 2);

  float pre_p[NP];
  f32x4 pre_w[NW4];

  // Source tensor of a chunk: chunk-uniform (the launcher guarantees that every source but the last
  // holds a multiple of KC channels), selected with scalar compares on the kernel arguments.
#define AS_CONV_FETCH(CHUNK)                                                                          \
  {                                                                                                   \
    const int cb = (CHUNK) * KC;                                                                      \
    const float* sp = p.src[0];                                                                       \
    int sc = p.src_c[0], sb = 0;                                                                      \
    if (p.n_src > 1 && cb >= p.src_end[0]) { sp = p.src[1]; sc = p.src_c[1]; sb = p.src_end[0]; }      \
    if (p.n_src > 2 && cb >= p.src_end[1]) { sp = p.src[2]; sc = p.src_c[2]; sb = p.src_end[1]; }      \
    if (p.n_src > 3 && cb >= p.src_end[2]) { sp = p.src[3]; sc = p.src_c[3]; sb = p.src_end[2]; }      \
    const float* spb = sp + ((long long)b * sc + (cb - sb)) * plane;                                   \
    const int recs = (int)((long long)(sc - (cb - sb)) * plane * 4);                                   \
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)spb, 0, recs, 0x00020000); \
    _Pragma("unroll") for (int i = 0; i < NP; ++i)                                                     \
      pre_p[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)p_voff[i], 0, 0)); \
    const f32x4* wc = wsrc + (long long)(CHUNK) * wchunk4;                                             \
    _Pragma("unroll") for (int i = 0; i < NW4; ++i) pre_w[i] = wc[w_idx[i]];                           \
  }
#define AS_CONV_COMMIT(BUF)                                                                           \
  {                                                                                                   \
    float* wdst = lds + (BUF) * STAGE;                                                                \
    float* pdst = wdst + W_ELEMS;                                                                     \
    _Pragma("unroll") for (int i = 0; i < NW4; ++i) {                                                  \
      const int idx = tid + i * 256;                                                                  \
      if (idx < W_ELEMS / 4) reinterpret_cast<f32x4*>(wdst)[idx] = pre_w[i];                           \
    }                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                                   \
      const int idx = tid + i * 256;                                                                  \
      if (idx < PATCH_ELEMS) pdst[idx] = pre_p[i];                                                    \
    }                                                                                                 \
  }

  // lane bases: wave w owns channel pairs {w, w+4, ...} of each chunk
  int poff[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int m = q * 32 + l31;
    poff[q] = W_ELEMS + (2 * wave + half) * PATCH + (m / TW) * PW + (m % TW);
  }
  const int woff = (2 * wave + half) * kBN + l31;

  f32x16 acc[2][2];  // [co tile][pixel tile]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][q][r] = 0.f;

  AS_CONV_FETCH(0)
  AS_CONV_COMMIT(0)
  __syncthreads();

  for (int chunk = 0; chunk < p.chunks; ++chunk) {
    const int cur = chunk & 1;
    const bool more = chunk + 1 < p.chunks;
    if (more) AS_CONV_FETCH(chunk + 1)
    const float* buf = lds + cur * STAGE;
    // software pipeline over the KS*KS*CPW k-steps of this wave: operands of step s+1 are read from LDS
    // before the 4 MFMAs of step s issue (sched_barrier pins that order), so at 1-2 waves per SIMD the
    // ~100-cycle ds_read latency hides under 256 cycles of MFMA instead of preceding them.
    constexpr int NS = KS * KS * CPW;
    float a_cur[2], b_cur[2], a_nxt[2], b_nxt[2];
#define AS_CONV_LDOPS(S, A, Bv)                                                            \
  {                                                                                        \
    constexpr int tap_ = (S) / CPW, j_ = (S) % CPW;                                         \
    constexpr int ky_ = tap_ / KS, kx_ = tap_ % KS;                                         \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) A[c] = buf[woff + (tap_ * KC + 8 * j_) * kBN + c * 32]; \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) Bv[q] = buf[poff[q] + (8 * j_) * PATCH + ky_ * PW + kx_]; \
  }
    AS_CONV_LDOPS(0, a_cur, b_cur)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 1 < NS) {
        // constexpr dispatch on s+1 (the loop is fully unrolled; S must be a constant expression)
        switch (s + 1) {
#define AS_CASE(N) case N: if constexpr (N < NS) AS_CONV_LDOPS(N, a_nxt, b_nxt) break;
          AS_CASE(1) AS_CASE(2) AS_CASE(3) AS_CASE(4) AS_CASE(5) AS_CASE(6) AS_CASE(7) AS_CASE(8)
          AS_CASE(9) AS_CASE(10) AS_CASE(11) AS_CASE(12) AS_CASE(13) AS_CASE(14) AS_CASE(15)
#undef AS_CASE
          default: break;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[c], b_cur[q], acc[c][q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 2; ++c) { a_cur[c] = a_nxt[c]; b_cur[c] = b_nxt[c]; }
    }
#undef AS_CONV_LDOPS
    if (more) AS_CONV_COMMIT(cur ^ 1)
    __syncthreads();
  }
#undef AS_CONV_FETCH
#undef AS_CONV_COMMIT

  // ---- reduce the four K-partials: wave w ends up owning tile t = w (co tile w>>1, pixel tile w&1) ----
  // slot(t, w) = t*3 + (w < t ? w : w-1), w != t.  Layout per slot: [16 regs][64 lanes] (conflict free).
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t == wave) continue;
    const int slot = t * 3 + (wave < t ? wave : wave - 1);
    float* dst = lds + slot * 1024 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r * 64] = acc[t >> 1][t & 1][r];
  }
  __syncthreads();
  f32x16 sum;
  {
    // select my own tile without dynamic register indexing
    const f32x16 own = (wave == 0) ? acc[0][0] : (wave == 1) ? acc[0][1] : (wave == 2) ? acc[1][0] : acc[1][1];
    sum = own;
    const float* srcp = lds + wave * 3 * 1024 + lane;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum[r] += srcp[s * 1024 + r * 64];
  }

  // ---- epilogue for tile (co tile = wave>>1, pixel tile = wave&1) ----
  __syncthreads();  // the reduction slots are consumed: reuse LDS for the block's bias
  if (tid < kBN) lds[tid] = (p.bias && n0 + tid < p.Cout) ? p.bias[n0 + tid] : 0.f;
  __syncthreads();
  {
    const EpiCtx e = make_epi_ctx<EPI>(p, b, n0, kBN);
    const int m = (wave & 1) * 32 + l31;
    const int gy = y0 + m / TW, gx = x0 + m % TW;
    const unsigned poff = (gy < p.H && gx < p.W) ? (unsigned)(((long long)gy * p.W + gx) * 4) : 0x7FFFFFF0u;
    const bool is_r = (EPI == AS_EPI_GRU_ZR) && n0 >= (p.Cout >> 1);
    epilogue_tile<EPI>(p, e, sum, (wave >> 1) * 32, half, poff, lds, is_r);
  }
}

// weight [Cout,Cin,KS,KS] -> wpack [chunks][tap][KC][Cout_pad]  (zero padded)
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout,
                                    int Cout_pad, int KS, int KC, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int co = (int)(idx % Cout_pad);
  long long t = idx / Cout_pad;
  const int cl = (int)(t % KC);
  t /= KC;
  const int tap = (int)(t % (KS * KS));
  const int chunk = (int)(t / (KS * KS));
  const int ci = chunk * KC + cl;
  float v = 0.f;
  if (co < Cout && ci < Cin) v = w[((long long)co * Cin + ci) * KS * KS + tap];
  wp[idx] = v;
}

// ---- direct kernels --------------------------------------------------------------------------

// 7x7, 1 -> Cout, + bias, ReLU  (convd1, update.py:81,87).  16x16 pixel tile, the 49-tap window of
// a pixel lives in registers, weights are wave-uniform (scalar loads).
__global__ __launch_bounds__(256) void conv7x7_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int H, int W, int Cout, int out_ctot, int out_coff) {
  __shared__ float patch[22 * 22];
  const int b = blockIdx.z;
  const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 16;
  const long long plane = (long long)H * W;
  const float* xp = x + (long long)b * plane;
  for (int idx = threadIdx.x; idx < 22 * 22; idx += 256) {
    const int py = idx / 22, px = idx - py * 22;
    const int gy = y0 - 3 + py, gx = x0 - 3 + px;
    patch[idx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? xp[(long long)gy * W + gx] : 0.f;
  }
  __syncthreads();
  const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
  float win[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) win[t] = patch[(ly + t / 7) * 22 + lx + t % 7];
  const int gy = y0 + ly, gx = x0 + lx;
  const bool ok = gy < H && gx < W;
  float* o = out + ((long long)b * out_ctot + out_coff) * plane + (long long)gy * W + gx;
  for (int co = 0; co < Cout; ++co) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 49; ++t) acc += win[t] * w[co * 49 + t];
    acc += bias ? bias[co] : 0.f;
    if (ok) o[(long long)co * plane] = fmaxf(acc, 0.f);
  }
}

// Blocked split-fp16 output (ConvParams' BS8 comment): one thread per (8-channel block, output pixel) writes the 16-B unit of
// the hi plane set and the one of the lo plane set.  Same per-element arithmetic as the fp32 kernels (shared helpers).
__device__ __forceinline__ void bs8_store(_Float16* __restrict__ out_bs, long long b, int c8tot, int blk, long long plane,
                                          long long pix, const float (&v)[8], float& amax) {
  half8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; j += 2) amax = fmaxf(amax, fmaxf(fabsf(v[j]), fabsf(v[j + 1])));
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 hj = (_Float16)v[j];
    hi[j] = hj;
    lo[j] = (_Float16)((v[j] - (float)hj) * 2048.f);
  }
  _Float16* rec = out_bs + ((b * 2 * c8tot + blk) * plane + pix) * 8;
  *reinterpret_cast<half8*>(rec) = hi;
  *reinterpret_cast<half8*>(rec + (long long)c8tot * plane * 8) = lo;
}

// Tap-major variant (the shipped path): weights as wt[49][Cout_pad].  The kernel above walks the output channels in its
// outer loop and pulls 49 scalar weights per channel (3136 dependent s_load dwords per wave) on 135 blocks — 30 us at
// 136x240, on the critical stream of the GRU loop.  Here a block owns a 16x16 pixel tile and CO = 8 output channels
// (8x the blocks), the kernel rows are the outer loop and one row's 7 x 8 weights arrive as seven s_load_dwordx8.
template <int CO>
__global__ __launch_bounds__(256) void conv7x7_c1_tm_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int H, int W, int Cout, int CP, int out_ctot, int out_coff,
                                                            float* __restrict__ copy_out, int copy_ctot, int copy_coff, int copy_bs,
                                                            int out_bs) {
  __shared__ float patch[22 * 22];
  as::fp16_saturate_mode();
  float ovf_amax = 0.f;
  const int groups = CP / CO;
  const int b = blockIdx.z / groups;
  const int c0 = (blockIdx.z - b * groups) * CO;
  const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 16;
  const long long plane = (long long)H * W;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long long)b * plane), 0, (int)(plane * 4), 0x00020000);
  for (int idx = threadIdx.x; idx < 22 * 22; idx += 256) {
    const int py = idx / 22, px = idx - py * 22;
    const int gy = y0 - 3 + py, gx = x0 - 3 + px;
    patch[idx] = as_bload(rs, (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (unsigned)((gy * W + gx) * 4) : 0x7FFFFFF0u);
  }
  __syncthreads();
  const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
  float acc[CO];
#pragma unroll
  for (int j = 0; j < CO; ++j) acc[j] = 0.f;
#pragma unroll 1  // one kernel row (7 x CO weights = 56 SGPRs) live at a time
  for (int ky = 0; ky < 7; ++ky) {
    const float* pr = patch + (ly + ky) * 22 + lx;
    const float* wr = wt + (ky * 7) * CP + c0;  // wave-uniform, unconditional (wt is zero padded to CP columns)
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const float v = pr[kx];
#pragma unroll
      for (int j = 0; j < CO; ++j) acc[j] = fmaf(v, wr[kx * CP + j], acc[j]);
    }
  }
  const int gy = y0 + ly, gx = x0 + lx;
  if (gy < H && gx < W) {
    if (out_bs) {  // `out` is a blocked split-fp16 tensor of out_ctot channels: the thread's CO = 8 channels are one block's pixel unit
      static_assert(CO == 8, "blocked output: one 8-channel block per thread");
      if (c0 < (Cout + 7) / 8 * 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < CO; ++j) v[j] = (c0 + j < Cout) ? fmaxf(acc[j] + (bias ? bias[c0 + j] : 0.f), 0.f) : 0.f;
        bs8_store(reinterpret_cast<_Float16*>(out), b, (out_ctot + 7) >> 3, (out_coff + c0) >> 3, plane, (long long)gy * W + gx, v, ovf_amax);
      }
    } else {
      float* o = out + ((long long)b * out_ctot + out_coff + c0) * plane + (long long)gy * W + gx;
#pragma unroll
      for (int j = 0; j < CO; ++j)
        if (c0 + j < Cout) o[(long long)j * plane] = fmaxf(acc[j] + (bias ? bias[c0 + j] : 0.f), 0.f);
    }
    // optional pass-through of the input plane (the `cat([out, disp])` of the motion encoder, update.py:91)
    if (copy_out && c0 == 0) {
      const float v = patch[(ly + 3) * 22 + lx + 3];
      if (copy_bs) {  // copy_out is a blocked split-fp16 tensor (ConvParams' BS8 comment): slot copy_coff % 8 of its block
        const int c8 = (copy_ctot + 7) >> 3;
        _Float16* rec = reinterpret_cast<_Float16*>(copy_out) + (((long long)b * 2 * c8 + (copy_coff >> 3)) * plane + (long long)gy * W + gx) * 8 + (copy_coff & 7);
        const _Float16 hk = (_Float16)v;
        rec[0] = hk;
        rec[(long long)c8 * plane * 8] = (_Float16)((v - (float)hk) * 2048.f);
        ovf_amax = fmaxf(ovf_amax, fabsf(v));
      } else {
        copy_out[((long long)b * copy_ctot + copy_coff) * plane + (long long)gy * W + gx] = v;
      }
    }
  }
  as::note_split_overflow(ovf_amax, &g_split_overflow_conv);
}

// 3x3, Cin -> 1, + bias (DispHead.conv2, update.py:19,24).  64 pixels of a row x 4 channel slices.
__global__ __launch_bounds__(256) void conv3x3_to1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int Cin, int H, int W) {
  __shared__ float red[4][64];
  const int lx = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int gx = blockIdx.x * 64 + lx, gy = blockIdx.y, b = blockIdx.z;
  const long long plane = (long long)H * W;
  const int c0 = slice * ((Cin + 3) / 4), c1 = min(Cin, c0 + (Cin + 3) / 4);
  float acc = 0.f;
  if (gx < W) {
    for (int c = c0; c < c1; ++c) {
      const float* xp = x + ((long long)b * Cin + c) * plane;
      const float* wc = w + c * 9;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = gy + ky - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = gx + kx - 1;
          const float v = (xx >= 0 && xx < W) ? xp[(long long)yy * W + xx] : 0.f;
          acc += v * wc[ky * 3 + kx];
        }
      }
    }
  }
  red[slice][lx] = acc;
  __syncthreads();
  if (slice == 0 && gx < W)
    out[(long long)b * plane + (long long)gy * W + gx] = ((red[0][lx] + red[1][lx]) + (red[2][lx] + red[3][lx])) + (bias ? bias[0] : 0.f);
}

// out[b,0,y,x] = bias + sum_t S[b,t,y+ky-1,x+kx-1]  (t = ky*3+kx, zero outside): second half of a 3x3,
// Cin -> 1 convolution whose per-tap channel reductions S were produced by a 1x1 MFMA conv (Cin -> 9).
__global__ __launch_bounds__(256) void tap_shift_sum_kernel(const float* __restrict__ S, const float* __restrict__ bias,
                                                            const float* __restrict__ addend, float* __restrict__ out, int H,
                                                            int W, long long P, int groups) {
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= P) return;
  const long long plane = (long long)H * W;
  const long long b = pix / plane;
  const int rem = (int)(pix - b * plane);
  const int y = rem / W, x = rem - y * W;
  const float* sp = S + b * groups * 9 * plane;
  float acc = 0.f;
  // channel-tile partials of the fused form (AS_EPI_RELU_TAPS) summed in a fixed order; four groups' 36 loads in flight together
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sp, 0, (int)((long long)groups * 9 * plane * 4), 0x00020000);
  unsigned toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    toff[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (unsigned)(((long long)t * plane + (long long)yy * W + xx) * 4) : 0x7FFFFFF0u;
  }
  const unsigned gstep = (unsigned)(9 * plane * 4);
  for (int g0 = 0; g0 < groups; g0 += 4) {
    float v[4][9];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int t = 0; t < 9; ++t)
        v[g][t] = as_bload(rs, (toff[t] == 0x7FFFFFF0u || g0 + g >= groups) ? 0x7FFFFFF0u : toff[t] + (unsigned)(g0 + g) * gstep);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc += v[g][t];  // out-of-image taps and missing groups read 0
  }
  const float delta = acc + (bias ? bias[0] : 0.f);
  out[pix] = addend ? addend[pix] + delta : delta;  // disp + delta_disp of the GRU loop fused (continuous_IGEVstereo.py:296)
}

// pool2x: 3x3 mean, stride 2, zero pad 1, divisor 9 (update.py:94-95)
__device__ __forceinline__ float pool2x_at(const float* __restrict__ xp, int yo, int xo, int H, int W) {
  float s = 0.f;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = 2 * yo - 1 + dy;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int xx = 2 * xo - 1 + dx;
      if (xx >= 0 && xx < W) s += xp[(long long)yy * W + xx];
    }
  }
  return s / 9.f;
}

__global__ __launch_bounds__(256) void pool2x_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W,
                                                     int Ho, int Wo, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int xo = (int)(idx % Wo);
  long long t = idx / Wo;
  const int yo = (int)(t % Ho);
  const long long bc = t / Ho;
  out[idx] = pool2x_at(x + bc * H * W, yo, xo, H, W);
}

__global__ __launch_bounds__(256) void pool2x_bs_kernel(const float* __restrict__ x, _Float16* __restrict__ out_bs, int C, int H, int W,
                                                        int Ho, int Wo, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  as::fp16_saturate_mode();
  float ovf_amax = 0.f;
  if (idx >= total) return;
  const int c8 = (C + 7) >> 3;
  const int xo = (int)(idx % Wo);
  long long t = idx / Wo;
  const int yo = (int)(t % Ho);
  t /= Ho;
  const int blk = (int)(t % c8);
  const long long b = t / c8;
  // all 72 taps of the thread's 8 channels as unconditional buffer loads (out-of-image taps and channels >= C read 0 through
  // the range check), summed per channel in pool2x_at's order
  const long long plane = (long long)H * W;
  const int nch = min(8, C - blk * 8);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (b * C + blk * 8) * plane), 0, (int)(nch * plane * 4), 0x00020000);
  unsigned off[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = 2 * yo - 1 + k / 3, xx = 2 * xo - 1 + k % 3;
    off[k] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (unsigned)(((long long)yy * W + xx) * 4) : 0x7FFFFFF0u;
  }
  const unsigned pl4 = (unsigned)(plane * 4);
  float raw[8][9];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k < 9; ++k) raw[j][k] = as_bload(rs, off[k] == 0x7FFFFFF0u ? off[k] : off[k] + (unsigned)j * pl4);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float sacc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) sacc += raw[j][k];
    v[j] = sacc / 9.f;
  }
  bs8_store(out_bs, b, c8, blk, (long long)Ho * Wo, (long long)yo * Wo + xo, v, ovf_amax);
  as::note_split_overflow(ovf_amax, &g_split_overflow_conv);
}

// pool2x_bs for an even input width: the three columns of a window row are ONE 8-byte load (columns 2xo, 2xo+1; rows of an even
// width keep it 8-B aligned) plus the left neighbour's second element through a lane shift — 24 wide loads per thread instead of
// 72 dword loads, every wave-load a fully used contiguous 512 B.  Same summation order as pool2x_at (row-major, zeros for the
// padding): bit-identical results.
__global__ __launch_bounds__(256) void pool2x_bs_even_kernel(const float* __restrict__ x, _Float16* __restrict__ out_bs, int C, int H, int W,
                                                             int Ho, int Wo, long long total) {
  const long long idx0 = (long long)blockIdx.x * 256 + threadIdx.x;
  as::fp16_saturate_mode();
  float ovf_amax = 0.f;
  const bool live = idx0 < total;  // no early return: every lane takes part in the shifts
  const long long idx = live ? idx0 : total - 1;
  const int c8 = (C + 7) >> 3;
  const int xo = (int)(idx % Wo);
  long long t = idx / Wo;
  const int yo = (int)(t % Ho);
  t /= Ho;
  const int blk = (int)(t % c8);
  const long long b = t / c8;
  const long long plane = (long long)H * W;
  const int nch = min(8, C - blk * 8);
  const int lane = threadIdx.x & 63;
  // the left column comes from the previous lane when that lane holds the previous output column of the same row
  const bool own_left = lane == 0 || xo == 0;
  // plain 8-byte global loads (this compiler lowers raw_buffer_load_b64 with a select-valued offset to ONE dword load): rows outside
  // the image and channels >= C are predicated off and read as zeros
  const float* xb = x + (b * C + blk * 8) * plane;
  f32x2 raw[8][3];
  float left[8][3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = 2 * yo - 1 + dy;
    const bool rin = yy >= 0 && yy < H;
    const long long e2 = (long long)(rin ? yy : 0) * W + 2 * xo;  // columns 2xo, 2xo+1 (< W: W even)
    const bool lin = rin && own_left && xo > 0;                    // column 2xo-1 for lanes without a donor
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool cin = j < nch;
      f32x2 z2;
      z2.x = 0.f; z2.y = 0.f;
      raw[j][dy] = (rin && cin) ? *reinterpret_cast<const f32x2*>(xb + (long long)j * plane + e2) : z2;
      left[j][dy] = (lin && cin) ? xb[(long long)j * plane + e2 - 1] : 0.f;
    }
  }
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float sacc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const float c0 = raw[j][dy].x, c1 = raw[j][dy].y;
      const float donor = __shfl_up(c1, 1);
      const float l = own_left ? left[j][dy] : donor;
      sacc += l;
      sacc += c0;
      sacc += c1;
    }
    v[j] = sacc / 9.f;
  }
  if (live) bs8_store(out_bs, b, c8, blk, (long long)Ho * Wo, (long long)yo * Wo + xo, v, ovf_amax);
  as::note_split_overflow(ovf_amax, &g_split_overflow_conv);
}

// interp: bilinear, align_corners=True (update.py:100-102)
__device__ __forceinline__ float interp_at(const float* __restrict__ xp, int yo, int xo, int H, int W, float sy, float sx) {
  const float fy = sy * (float)yo, fx = sx * (float)xo;
  const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float ty = fy - (float)y0, tx = fx - (float)x0;
  const float top = (1.f - tx) * xp[(long long)y0 * W + x0] + tx * xp[(long long)y0 * W + x1];
  const float bot = (1.f - tx) * xp[(long long)y1 * W + x0] + tx * xp[(long long)y1 * W + x1];
  return (1.f - ty) * top + ty * bot;
}

__global__ __launch_bounds__(256) void interp_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W,
                                                     int Ho, int Wo, float sy, float sx, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int xo = (int)(idx % Wo);
  long long t = idx / Wo;
  const int yo = (int)(t % Ho);
  const long long bc = t / Ho;
  out[idx] = interp_at(x + bc * H * W, yo, xo, H, W, sy, sx);
}

__global__ __launch_bounds__(256) void interp_bs_kernel(const float* __restrict__ x, _Float16* __restrict__ out_bs, int C, int H, int W,
                                                        int Ho, int Wo, float sy, float sx, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  as::fp16_saturate_mode();
  float ovf_amax = 0.f;
  if (idx >= total) return;
  const int c8 = (C + 7) >> 3;
  const int xo = (int)(idx % Wo);
  long long t = idx / Wo;
  const int yo = (int)(t % Ho);
  t /= Ho;
  const int blk = (int)(t % c8);
  const long long b = t / c8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = blk * 8 + j;
    v[j] = c < C ? interp_at(x + (b * C + c) * H * W, yo, xo, H, W, sy, sx) : 0.f;
  }
  bs8_store(out_bs, b, c8, blk, (long long)Ho * Wo, (long long)yo * Wo + xo, v, ovf_amax);
  as::note_split_overflow(ovf_amax, &g_split_overflow_conv);
}

// ================================================================================================
// Split-precision implicit GEMM: 3 x v_mfma_f32_32x32x16_f16 per 16-channel k-step.
//   operand x = hi + lo/2048, hi = fp16(x), lo = fp16((x - hi)*2048)   (22 significand bits)
//   w*x ~= w_hi*x_hi + (w_hi*x_lo + w_lo*x_hi)/2048,  two fp32 accumulators per output tile
// 16x the fp32-MFMA rate for 3 instructions => ~5x the matrix throughput at ~2^-22 relative error per
// product (fp32's own rounding is 2^-24).  Weights are split once at pack time; activations are split
// while they are staged into LDS.  Requires |x| < 65504.
//
// Block = 4 waves, tile 128 pixels x BN (128 | 64) output channels; a wave owns 64 co x 64 px (BN=128)
// or 64 co x 32 px (BN=64) for ALL of K (no split-K needed at this MFMA rate).
// LDS images are k-contiguous so a lane's MFMA fragment (8 consecutive channels) is ONE ds_read_b128:
//   weights  [tap][comp][h][co][8]   (global pack has the same order: staging is a linear 16-B copy)
//   patch    [comp][h][patch pixel][8]
// Pipeline unit = one kernel row (KS taps) of a 16-channel chunk: weights double-buffered per unit,
// the halo patch double-buffered per chunk; next unit/chunk is fetched into registers under the MFMAs.
// ================================================================================================
using half2v = __attribute__((ext_vector_type(2))) _Float16;

constexpr int kSplitKC = 16;
// cache policy bits of the staged epilogue's result stores (buffer-store aux: 1 = sc0, 2 = nt, 16 = sc1).  0 = write-back: the
// results stay dirty in the XCD's L2 and are written out at the kernel boundary.  A/B builds only (tools/conv_variant.sh).
#ifndef AS_EPI_STORE_AUX
#define AS_EPI_STORE_AUX 0
#endif

typedef __attribute__((address_space(3))) void as_lds_void;
typedef __attribute__((address_space(1))) const void as_gbl_void;

// Diagnostic builds only (tools/conv_variant.sh): never defined in the product build.
//   -DAS_CONV_STAMPS  s_memtime brackets around the segments of the chunk loop, summed in scalar registers by the first
//                     consumer and the first loader wave of each block and stored to a buffer of their own after the loop
//   -DAS_ABL_NO_W / -DAS_ABL_NO_P   timing-only: the loaders skip the global loads of the weight image / the halo patch
//                     after the first unit (stale registers are stored instead: results are wrong, the instruction
//                     stream, LDS traffic and barriers stay)
//   -DAS_ABL_NO_DMA   timing-only, all-DMA path: no unit is staged after the first one (the consumers alone)
//   -DAS_ABL_MFMA_TAPS=N  timing-only: the consumers run the operand reads + MFMAs of the first N taps of every unit only
//                     (N = 0: staging alone; N = 4 | 5 of 9: the resource profile of a Winograd F(2x2,3x3) | F(4,3) kernel of
//                     this tiling — the same bytes staged per block, 1/2.25 | 1/2 of the matrix instructions)
#ifdef AS_CONV_STAMPS
constexpr int kStampBlocks = 1024, kStampSlots = 16;
__device__ unsigned long long as_conv_stamp_buf[kStampBlocks * kStampSlots];
#define AS_STAMP(T)                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                     \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T) :: "memory");            \
  __builtin_amdgcn_sched_barrier(0);
// block lifetime: absolute s_memtime of thread 0 at 0 = kernel entry, 1 = first unit staged (consumers start), 2 = chunk loop
// done, 3 = tile parked + barrier (staged epilogue), 4 = kernel end  -> as_debug_conv_life
__device__ unsigned long long as_conv_life_buf[kStampBlocks * 8];
// slots 5 / 6: s_memrealtime (constant 100 MHz) at entry / end -> in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
#define AS_LIFE(I)                                                                       \
  if (threadIdx.x == 0 && blockIdx.x < kStampBlocks) {                                   \
    unsigned long long t_;                                                               \
    AS_STAMP(t_)                                                                         \
    as_conv_life_buf[blockIdx.x * 8 + (I)] = t_;                                         \
    if ((I) == 0 || (I) == 4) as_conv_life_buf[blockIdx.x * 8 + ((I) == 0 ? 5 : 6)] = __builtin_amdgcn_s_memrealtime(); \
  }
#define AS_STAMP_DECL unsigned long long st_a = 0, st_b = 0, st_sum[6] = {0, 0, 0, 0, 0, 0};
#define AS_STAMP_BEGIN AS_STAMP(st_a)
#define AS_STAMP_SEG(I) { AS_STAMP(st_b) st_sum[I] += st_b - st_a; st_a = st_b; }
#define AS_STAMP_FLUSH(ROLE)                                                             \
  if (blockIdx.x < kStampBlocks && lane == 0 && (wave & 3) == 0) {                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) as_conv_stamp_buf[blockIdx.x * kStampSlots + (ROLE) * 8 + i_] = st_sum[i_]; \
  }
#else
#define AS_LIFE(I)
#define AS_STAMP_DECL
#define AS_STAMP_BEGIN
#define AS_STAMP_SEG(I)
#define AS_STAMP_FLUSH(ROLE)
#endif

// Wave specialisation (8 waves, 2 per SIMD): waves 0-3 are CONSUMERS — their instruction stream is only
// ds_read_b128 + MFMA; waves 4-7 are LOADERS — weight-image DMA, halo-patch fetch, fp32 -> hi/lo split and
// the LDS commit.  Measured on the single-role version: an LDS-DMA instruction costs the issuing wave
// ~180 cycles, so 18 of them per chunk in front of 108 MFMAs could not overlap with them (in-order issue);
// on a sibling wave of the same SIMD they do.
template <int KS, int TW, int BN, int EPI, int NSUB = 1, int S = 1, bool FAST = false, bool LEAN = false>
__global__ __launch_bounds__(LEAN ? 256 : 512, 2) void conv_split_kernel(ConvParams p) {
  // LEAN: four waves that issue their own LDS-DMA and consume it (no loader waves, one weight image + one patch image:
  // ~68 KB), so TWO blocks share a CU: while one waits for its unit to land or runs its prologue / epilogue, the other's MFMAs
  // keep the matrix pipes busy.  All-DMA operand staging only (every source blocked split-fp16), 3x3, stride 1, 64-channel tiles.
  constexpr int NT = LEAN ? 256 : 512;
  static_assert(!LEAN || (KS == 3 && S == 1 && BN == 64), "lean blocks: 3x3, stride 1, 64-channel tiles");
  as::fp16_saturate_mode();      // |x| >= 65504 saturates in the operand split instead of producing inf / NaN (common.h)
  float ovf_amax = 0.f;      // max |x| this thread split (loader path) or emitted as a blocked split-fp16 result
  // FAST: fp16 operands (the hi parts only), ONE MFMA per product (as_set_fast16).  A compile-time variant: a run-time
  // branch around the two cross-term MFMAs costs the three-MFMA stream 25 % end to end (measured on one box: 25.8 vs 21.6 ms)
  constexpr bool fast16 = FAST;
  // Block = NSUB sub-tiles of 128 pixels (TH x TW each, consecutive tile ids of the image) x BN output channels.
  // NSUB = 2 with BN = 64 halves the weight bytes a CU pulls per MFMA (the per-CU L1 fill rate, ~45 GB/s, is what
  // the loader waves run into) while 255 tiles of a 136x240 map still pair up into exactly 2 rounds of 256 blocks.
  constexpr int BM = 128 * NSUB;
  constexpr int TH = 128 / TW;
  constexpr int PAD = KS / 2;
  // S: stride (1 | 2).  Tiles live in OUTPUT coordinates; the halo patch of a TH x TW output tile spans
  // (TH-1) S + KS input rows, and a consumer lane's pixel sits at S times its output offset inside it.
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  constexpr int PATCHP = PH * PW;
  constexpr int NTAP = KS * KS;
  // Pipeline unit = NSC 16-channel chunks.  3x3: one chunk (9 taps = 108 MFMAs per consumer wave between barriers).
  // 1x1: FOUR chunks (64 channels) — with a single chunk the 12 MFMAs of a unit cannot cover the latency of the next
  // unit's loads (128 -> 64 at 518 400 pixels: 136 us, latency bound); the four chunks are laid out and consumed
  // exactly like four taps of one chunk (weight image [sub-chunk][comp][h][co][8], one patch sub-image per sub-chunk).
  constexpr int NSC = (KS == 1) ? 4 : 1;
  constexpr int NTAPE = NTAP * NSC;          // "taps" of a unit's weight image
#ifdef AS_ABL_MFMA_TAPS
  constexpr int NTAPC = (AS_ABL_MFMA_TAPS) < NTAPE ? (AS_ABL_MFMA_TAPS) : NTAPE;  // diagnostic builds: taps the consumers execute
#else
  constexpr int NTAPC = NTAPE;
#endif
  constexpr int WSEG = BN * 16;              // bytes of one (tap, comp, h) weight segment
  constexpr int WCHUNK = NTAPE * 4 * WSEG;   // bytes of one unit's weight image
  constexpr int NWD = WCHUNK / 16 / 256;     // 16-B LDS-DMA pieces per loader thread per chunk
  constexpr int PATCHT = NSUB * PATCHP;      // patch pixels of the block (sub-tile patches back to back)
  constexpr int NPI = (2 * PATCHT + 255) / 256;  // (k-half, patch pixel) items per loader thread and sub-chunk
  constexpr int PIMG = 4 * PATCHT * 16;      // bytes of one sub-chunk's patch image [comp][h][pixel][8]
  constexpr int WPX = 256 / BN;              // consumer waves along the pixel dimension (4 consumers = (BN/64) x WPX)
  constexpr int PTW = BM / (WPX * 32);       // pixel MFMA tiles per consumer wave
  // All-DMA operand staging: when every source is a blocked split-fp16 tensor, both LDS images of a unit are plain copies of
  // global memory (weights: the pack's order; patch: 16-B pixel units), so the loaders issue `buffer_load_dwordx4 ... lds` for unit
  // c+1 into a SECOND patch image while the consumers work on unit c, wait for them to land, and meet the consumers at ONE barrier
  // per chunk — no VGPR round trip, no LDS write instructions, no commit phase.  Needs room for two patch images (3x3 only).
  constexpr bool PDB = KS == 3 && 2 * WCHUNK + 2 * NSC * PIMG <= 160 * 1024;
  const bool dma = PDB && p.all_bs;  // kernel-uniform
  static_assert(PTW == 1 || PTW == 2, "consumer tile is 64 co x 32|64 px");
  // Operand reads of tap t+1: ONE behind each of tap t's first MFMAs (default) or, with -DAS_CONV_LD_BURST (the round-4 form, kept
  // for A/B builds: tools/conv_variant.sh), a burst of eight between tap t's two MFMA groups.
#ifdef AS_CONV_LD_BURST
  constexpr bool LD_IL = false;
#else
  constexpr bool LD_IL = true;
#endif
  static_assert(WCHUNK % (16 * 256) == 0, "weight chunk must split evenly over the loader threads");
  // [W image 0][W image 1][patch image]
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = !LEAN && wave >= 4;
  const int l31 = lane & 31, half = lane >> 5;
  AS_LIFE(0)
  if (p.stagger) {
    // Blocks of one XCD (ids congruent mod 8) keep running in lock-step — they share halo patches and weight lines in that XCD's
    // L2 — but the eight XCDs start `stagger` x 64 clocks apart, so their staging bursts and store tails do not hit the fabric
    // at the same instant.
    const int steps = (int)(blockIdx.x & 7u) * p.stagger;
    for (int i = 0; i < steps; i += 8) __builtin_amdgcn_s_sleep(8);
  }
  if (LEAN && p.lean_offset && blockIdx.x >= (gridDim.x >> 1)) {
    // Two lean blocks share a CU.  Started together they run in lock-step — both stage, both compute, both store at the same
    // time, so the matrix pipes idle through a prologue and a DOUBLE store tail.  The blocks of the second half of the grid
    // (the ones that take the CUs' second slots) start `lean_offset` x 64 clocks later: their prologue and the first blocks'
    // finish then fall into the partner's MFMA loop.
    for (int i = 0; i < p.lean_offset; i += 8) __builtin_amdgcn_s_sleep(8);
  }

  const int ntile = p.tiles_x * p.tiles_y;
  const int ngroup = (ntile + NSUB - 1) / NSUB;
  // Block ids are dealt round-robin to the 8 XCDs (each with its own L2).  With xcd_map the blocks that share a
  // pixel tile but differ in output-channel tile get ids 8 apart — same XCD, same wave of dispatch — so the
  // halo patches they all read are fetched into that L2 once instead of once per channel tile
  // (PMC: 392 MB fetched per gru04 z|r launch against 137 MB algorithmic before).
  int group, b, nt, ks;
  {
    const int T = ngroup * p.B;  // pixel tiles incl. batch
    int id = blockIdx.x;
    ks = id / (T * p.n_tiles);   // K slice (0 unless split-K)
    id -= ks * T * p.n_tiles;
    int pt;
    if (p.xcd_map == 2) {
      // BANDED: XCD x (ids congruent to x mod 8) owns the contiguous run of pixel tiles [x T/8, (x+1) T/8) — about two tile rows of a
      // 136x240 map — in raster order, the channel tiles of a pixel tile still 8 ids apart.  Neighbouring pixel tiles then share
      // their halo rows and the 128-B lines that straddle a tile edge in ONE L2 instead of fetching them once per XCD
      // (mode 1 puts pixel tile t on XCD t mod 8: horizontal neighbours never meet).  The T mod 8 last tiles are dealt as in mode 1.
      const int a = T >> 3, full = a * 8 * p.n_tiles;
      if (id < full) {
        const int x = id & 7, j = id >> 3;
        const int pl = j / p.n_tiles;
        nt = j - pl * p.n_tiles;
        pt = x * a + pl;
      } else {
        const int rem = id - full, bb = T - 8 * a;
        nt = rem / bb;
        pt = 8 * a + (rem - nt * bb);
      }
    } else if (p.xcd_map) {
      const int per = 8 * p.n_tiles;
      const int chunk = id / per, r = id - chunk * per;
      const int m = min(8, T - chunk * 8);  // pixel tiles in this chunk (the last one may be short)
      nt = r / m;
      pt = chunk * 8 + (r - nt * m);
    } else {
      pt = id % T;
      nt = id / T;
    }
    group = pt % ngroup;
    b = pt / ngroup;
  }
  const int chunk_lo = (int)((long long)p.chunks * ks / p.ksplit);
  const int chunk_hi = (int)((long long)p.chunks * (ks + 1) / p.ksplit);
  int sx0[NSUB], sy0[NSUB];  // origin of each sub-tile; a missing one (odd tile count) sits below the image
#pragma unroll
  for (int u = 0; u < NSUB; ++u) {
    const int t = group * NSUB + u;
    sx0[u] = (t % p.tiles_x) * TW;
    sy0[u] = t < ntile ? (t / p.tiles_x) * TH : p.H + PAD + 1;
  }
  const bool second = p.dual && nt >= (p.n_tiles >> 1);  // block-uniform: this block belongs to the dual launch's second conv
  if (second) nt -= p.n_tiles >> 1;
  const float* const src0 = second ? p.src2 : p.src[0];
  const int src0_bs = second ? p.src2_bs : p.src_bs[0];
  const float* const wpack_sel = second ? p.wpack2 : p.wpack;
  const float* const bias_sel = second ? p.bias2 : p.bias;
  const int n0 = nt * BN;
  const long long plane = (long long)p.Hi * p.Wi;  // INPUT plane: source addressing of the loader waves

  f32x16 acc_h[2][PTW], acc_x[2][PTW];
  const int cw = wave & 3;
  const int co_base = (BN == 128) ? (cw >> 1) * 64 : 0;
  const int px_base = (BN == 128) ? (cw & 1) * (PTW * 32) : cw * (PTW * 32);

  // ---- staged epilogue: one thread per (pixel, 8-channel block); its global operands (context term, h, z) ----
  // EPG thread groups along the channel dimension, ENB8 blocks of 8 channels per thread.  On the all-DMA path the LOADER waves
  // fetch the operands of their blocks during the LAST chunks of the K loop (their registers are idle there): the epilogue's read
  // burst — every CU at once, HBM-bound — shrinks by the loaders' half, which moves under the MFMAs.
  constexpr bool STAGED = EPI != kEpiTaps && EPI != kEpiPartial;
  constexpr int EPG = NT / BM;
  constexpr int ENB8 = STAGED ? BN / 8 / EPG : 1;
  constexpr int EOPS = (EPI == AS_EPI_GRU_Q) ? 3 : 2;                  // buffer loads per channel (add; h; z)
  float pav[ENB8][8], phv[ENB8][8], pzv[ENB8][8];
  bool pre = false;  // wave-uniform: this wave's epilogue operands are already in pav / phv / pzv
#define AS_EPI_SETUP                                                                                   \
  const EpiCtx e = make_epi_ctx<EPI>(p, b, n0, BN, second);                                            \
  [[maybe_unused]] const bool is_r = (EPI == AS_EPI_GRU_ZR) && n0 >= (p.Cout >> 1);                    \
  const int mt = tid % BM, cg = tid / BM;                                                              \
  const int su = mt >> 7, m = mt & 127;                                                                \
  const int gy = ((NSUB > 1 && su) ? sy0[NSUB - 1] : sy0[0]) + m / TW, gx = ((NSUB > 1 && su) ? sx0[NSUB - 1] : sx0[0]) + m % TW; \
  const unsigned poff = (gy < p.H && gx < p.W) ? (unsigned)(((long long)gy * p.W + gx) * 4) : 0x7FFFFFF0u;
// INVARIANT (loader prefetch, all-DMA path): one AS_EPI_LOADK(K) issues EXACTLY 8 * EOPS buffer loads (8 channels x {add, h[, z]}),
// and the loader's counted wait `s_waitcnt vmcnt(8 * EOPS)` behind it relies on that count: it must cover the unit's LDS-DMA and every
// older load and leave only this block's loads in flight, because the barrier that follows is a bare s_barrier.  An edit (or a
// compiler) that merges, drops or predicates one of these loads turns that wait into a silent race on the LDS image; use
// vmcnt(0) there when in doubt (one exposed round trip per unit).  Round 5's advisor checked the build's disassembly by hand
// (16 / 24 loads per branch, no spills); `llvm-objdump --offloading` + `-d` on build/conv.o reproduces it.
#define AS_EPI_LOADK(K)                                                                               \
  _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                      \
    const int col = (cg * ENB8 + (K)) * 8 + j;                                                        \
    const unsigned off = poff == 0x7FFFFFF0u ? poff : (unsigned)col * e.plane4 + poff;                 \
    pav[K][j] = as_bload(e.r_add, off);                                                               \
    if (EPI == AS_EPI_GRU_ZR || EPI == AS_EPI_LINEAR) phv[K][j] = as_bload(e.r_h, off);                \
    if (EPI == AS_EPI_GRU_Q) { phv[K][j] = as_bload(e.r_h, off); pzv[K][j] = as_bload(e.r_z, off); }   \
  }

#define AS_SPLIT_LDOPS(TAP, S)                                                                          \
  {                                                                                                     \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                                      \
      a_hi[S][c] = *reinterpret_cast<const half8*>(wb + (((TAP) * 2 + 0) * 2) * WSEG + c * 512);          \
      if (!fast16) a_lo[S][c] = *reinterpret_cast<const half8*>(wb + (((TAP) * 2 + 1) * 2) * WSEG + c * 512); \
    }                                                                                                   \
    constexpr int tapoff_ = (KS == 1) ? (TAP) * PIMG : (((TAP) / KS) * PW + ((TAP) % KS)) * 16;          \
    _Pragma("unroll") for (int q = 0; q < PTW; ++q) {                                                    \
      b_hi[S][q] = *reinterpret_cast<const half8*>(pb + plane_off[q] + tapoff_);                         \
      if (!fast16) b_lo[S][q] = *reinterpret_cast<const half8*>(pb + 2 * PATCHT * 16 + plane_off[q] + tapoff_); \
    }                                                                                                   \
  }
#define AS_SPLIT_MFMA_C(S, c)                                                                           \
  _Pragma("unroll") for (int q = 0; q < PTW; ++q) {                                                      \
    acc_h[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[S][c], b_hi[S][q], acc_h[c][q], 0, 0, 0);   \
    if (!fast16) acc_x[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[S][c], b_lo[S][q], acc_x[c][q], 0, 0, 0);   \
  }                                                                                                     \
  if (!fast16) {                                                                                        \
  _Pragma("unroll") for (int q = 0; q < PTW; ++q)  /* second cross term: not back to back with the first on the same accumulator */ \
    acc_x[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[S][c], b_hi[S][q], acc_x[c][q], 0, 0, 0); \
  }
  // LD_IL: ONE operand read of tap t+1 behind each of tap t's first MFMAs instead of a burst of eight between its two MFMA groups:
  // an MFMA holds the SIMD's vector issue for 8 of its 32 cycles and a ds_read_b128 issued in the remaining gap is free, while a
  // burst delays the MFMA behind it by what does not fit one gap.  Same MFMAs on the same accumulators in the same order.
#define AS_LD_A(TAPN, SN, c, comp)                                                                      \
  if constexpr ((TAPN) < NTAPC) {                                                                       \
    if ((comp) == 0) a_hi[SN][c] = *reinterpret_cast<const half8*>(wb + (((TAPN) * 2 + 0) * 2) * WSEG + (c) * 512);          \
    else if (!fast16) a_lo[SN][c] = *reinterpret_cast<const half8*>(wb + (((TAPN) * 2 + 1) * 2) * WSEG + (c) * 512);         \
  }
#define AS_LD_B(TAPN, SN, q, comp)                                                                      \
  if constexpr ((TAPN) < NTAPC) {                                                                       \
    constexpr int tapoffn_ = (KS == 1) ? (TAPN) * PIMG : (((TAPN) / KS) * PW + ((TAPN) % KS)) * 16;      \
    if ((comp) == 0) b_hi[SN][q] = *reinterpret_cast<const half8*>(pb + plane_off[q] + tapoffn_);       \
    else if (!fast16) b_lo[SN][q] = *reinterpret_cast<const half8*>(pb + 2 * PATCHT * 16 + plane_off[q] + tapoffn_); \
  }
#define AS_MM_HH(S, c, q) acc_h[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[S][c], b_hi[S][q], acc_h[c][q], 0, 0, 0);
#define AS_MM_HX(S, c, q) if (!fast16) acc_x[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[S][c], b_lo[S][q], acc_x[c][q], 0, 0, 0);
#define AS_MM_LH(S, c, q) if (!fast16) acc_x[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[S][c], b_hi[S][q], acc_x[c][q], 0, 0, 0);
#define AS_SB __builtin_amdgcn_sched_barrier(0);
#define AS_SPLIT_STEP_IL(TAP)                                                                           \
  if constexpr ((TAP) < NTAPC) {                                                                        \
    constexpr int S_ = (TAP) & 1, N_ = ((TAP) + 1) & 1, T1_ = (TAP) + 1;                                \
    if constexpr (PTW == 2) {                                                                           \
      AS_MM_HH(S_, 0, 0) AS_LD_A(T1_, N_, 0, 0) AS_SB                                                    \
      AS_MM_HX(S_, 0, 0) AS_LD_A(T1_, N_, 0, 1) AS_SB                                                    \
      AS_MM_HH(S_, 0, 1) AS_LD_B(T1_, N_, 0, 0) AS_SB                                                    \
      AS_MM_HX(S_, 0, 1) AS_LD_B(T1_, N_, 0, 1) AS_SB                                                    \
      AS_MM_LH(S_, 0, 0) AS_LD_A(T1_, N_, 1, 0) AS_SB                                                    \
      AS_MM_LH(S_, 0, 1) AS_LD_A(T1_, N_, 1, 1) AS_SB                                                    \
      AS_MM_HH(S_, 1, 0) AS_LD_B(T1_, N_, 1, 0) AS_SB                                                    \
      AS_MM_HX(S_, 1, 0) AS_LD_B(T1_, N_, 1, 1) AS_SB                                                    \
      AS_MM_HH(S_, 1, 1) AS_SB                                                                           \
      AS_MM_HX(S_, 1, 1) AS_SB                                                                           \
      AS_MM_LH(S_, 1, 0) AS_SB                                                                           \
      AS_MM_LH(S_, 1, 1) AS_SB                                                                           \
    } else {                                                                                            \
      AS_MM_HX(S_, 0, 0) AS_LD_A(T1_, N_, 0, 0) AS_SB                                                    \
      AS_MM_HH(S_, 0, 0) AS_LD_B(T1_, N_, 0, 0) AS_SB                                                    \
      AS_MM_LH(S_, 0, 0) AS_LD_B(T1_, N_, 0, 1) AS_SB                                                    \
      AS_MM_HX(S_, 1, 0) AS_LD_A(T1_, N_, 0, 1) AS_SB                                                    \
      AS_MM_HH(S_, 1, 0) AS_LD_A(T1_, N_, 1, 0) AS_SB                                                    \
      AS_MM_LH(S_, 1, 0) AS_LD_A(T1_, N_, 1, 1) AS_SB                                                    \
    }                                                                                                   \
  }
#define AS_SPLIT_STEP(TAP)                                                                              \
  if constexpr (LD_IL) { AS_SPLIT_STEP_IL(TAP) } else                                                   \
  if constexpr ((TAP) < NTAPC) {                                                                        \
    AS_SPLIT_MFMA_C((TAP) & 1, 0)                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if constexpr ((TAP) + 1 < NTAPC) AS_SPLIT_LDOPS((TAP) + 1, ((TAP) + 1) & 1)                          \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    AS_SPLIT_MFMA_C((TAP) & 1, 1)                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }
  if constexpr (LEAN) {
    // =========================== LEAN BLOCK: every wave stages and consumes ===========================
    constexpr int SEG_PER_STEP = 256 / BN;
    const int ltid = tid, lwave = wave;
    unsigned p_boff[NPI];
    bool p_slot[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      int idx = ltid + i * 256;
      const bool slot = idx < 2 * PATCHT;
      if (!slot) idx = 2 * PATCHT - 1;
      const int hh = idx / PATCHT, ppt = idx - hh * PATCHT;
      const int su = ppt / PATCHP, pp = ppt - su * PATCHP;
      const int py = pp / PW, px = pp - py * PW;
      const int gy = (NSUB > 1 && su ? sy0[NSUB - 1] : sy0[0]) - PAD + py, gx = (NSUB > 1 && su ? sx0[NSUB - 1] : sx0[0]) - PAD + px;
      const bool in = slot && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
      p_boff[i] = in ? (unsigned)(((long long)hh * plane + (long long)gy * p.Wi + gx) * 16) : 0x7FFFFFF0u;
      p_slot[i] = slot;
    }
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wpack_sel, 0, 0x7FFFFFF0, 0x00020000);
    const unsigned wvoff = (unsigned)((n0 + (long long)(ltid / BN) * p.Cout_pad + (ltid % BN)) * 16);
    const long long wstep16 = (long long)SEG_PER_STEP * p.Cout_pad;
    const long long wchunk16 = (long long)NTAPE * 4 * p.Cout_pad;
    const int wlane = half * WSEG + (co_base + l31) * 16;
    int plane_off[PTW];
#pragma unroll
    for (int q = 0; q < PTW; ++q) {
      const int mt = px_base + q * 32 + l31;
      const int m = mt & 127;
      plane_off[q] = half * (PATCHT * 16) + ((mt >> 7) * PATCHP + (m / TW) * S * PW + (m % TW) * S) * 16;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < PTW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_h[c][q][r] = 0.f; acc_x[c][q][r] = 0.f; }
    for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
      {  // this wave's quarter of the unit: weight image -> lds[0, WCHUNK), patch image -> lds[WCHUNK, WCHUNK + PIMG)
#pragma unroll
        for (int g = 0; g < NWD; ++g)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (as_lds_void*)(lds + g * 4096 + lwave * 1024), 16, wvoff,
                                                   (unsigned)(((long long)chunk * wchunk16 + (long long)g * wstep16) * 16), 0, 0);
        const int cb = chunk * kSplitKC;
        const float* sp = src0;
        int sc = p.src_c[0], sb = 0;
        if (p.n_src > 1 && cb >= p.src_end[0]) { sp = p.src[1]; sc = p.src_c[1]; sb = p.src_end[0]; }
        if (p.n_src > 2 && cb >= p.src_end[1]) { sp = p.src[2]; sc = p.src_c[2]; sb = p.src_end[1]; }
        if (p.n_src > 3 && cb >= p.src_end[2]) { sp = p.src[3]; sc = p.src_c[3]; sb = p.src_end[2]; }
        const int left = sc - (cb - sb);
        const int c8 = (sc + 7) >> 3, blk = (cb - sb) >> 3;
        const _Float16* spb = reinterpret_cast<const _Float16*>(sp) + ((long long)b * 2 * c8 + (left > 0 ? blk : 0)) * plane * 8;
        const int recs = left > 0 ? (int)((long long)(c8 - blk) * plane * 16) : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)spb, 0, recs, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(spb + (long long)c8 * plane * 8), 0, recs, 0x00020000);
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
          const int vo_ = (int)p_boff[i];
          if (p_slot[i]) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (as_lds_void*)(lds + WCHUNK + (i * 256 + lwave * 64) * 16), 16, vo_, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsl, (as_lds_void*)(lds + WCHUNK + 2 * PATCHT * 16 + (i * 256 + lwave * 64) * 16), 16, vo_, 0, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // the unit has landed
      const unsigned char* wb = lds + wlane;
      const unsigned char* pb = lds + WCHUNK;
      half8 a_hi[2][2], a_lo[2][2], b_hi[2][PTW], b_lo[2][PTW];
      AS_SPLIT_LDOPS(0, 0)
      AS_SPLIT_STEP(0) AS_SPLIT_STEP(1) AS_SPLIT_STEP(2) AS_SPLIT_STEP(3) AS_SPLIT_STEP(4)
      AS_SPLIT_STEP(5) AS_SPLIT_STEP(6) AS_SPLIT_STEP(7) AS_SPLIT_STEP(8)
      __syncthreads();  // every wave is done with the images before the next unit overwrites them
    }
  } else if (loader) {
    // =========================== LOADER WAVES ===========================
    const int ltid = tid - 256;
    const int lwave = wave - 4;
    // A patch item is (k-half hh, patch pixel): the lane fetches the 8 channels 8 hh .. 8 hh + 7 of its pixel and
    // commits them as ONE ds_write_b128 per component — the fragment layout [comp][h][pixel][8] makes a pixel's 8
    // channels 16 contiguous bytes, and a wave's 64 pixels 1 KB: full-rate, conflict-free LDS writes.  (Items of
    // (channel pair, pixel) with 4-B writes at a 16-B lane stride were 4-way bank conflicted: 4 x the instructions at a
    // quarter of the rate, on the LDS port the consumers' operand reads already keep 2/3 busy.)
    unsigned p_voff[NPI];  // byte offset of channel 8*hh of the chunk at this patch pixel, or OOB sentinel
    unsigned p_boff[NPI];  // the same position inside a plane set of a blocked split-fp16 source (16-B pixel units)
    int p_lds[NPI];        // byte offset inside one comp image of the patch buffer
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      int idx = ltid + i * 256;
      const bool slot = idx < 2 * PATCHT;
      if (!slot) idx = 2 * PATCHT - 1;
      const int hh = idx / PATCHT, ppt = idx - hh * PATCHT;
      const int su = ppt / PATCHP, pp = ppt - su * PATCHP;
      const int py = pp / PW, px = pp - py * PW;
      const int gy = (NSUB > 1 && su ? sy0[NSUB - 1] : sy0[0]) * S - PAD + py, gx = (NSUB > 1 && su ? sx0[NSUB - 1] : sx0[0]) * S - PAD + px;
      const bool in = slot && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
      p_voff[i] = in ? (unsigned)(((long long)(8 * hh) * plane + (long long)gy * p.Wi + gx) * 4) : 0x7FFFFFF0u;
      p_boff[i] = in ? (unsigned)(((long long)hh * plane + (long long)gy * p.Wi + gx) * 16) : 0x7FFFFFF0u;
      p_lds[i] = slot ? hh * (PATCHT * 16) + ppt * 16 : -1;
    }
    // piece i of this thread is 16-B unit (ltid + 256 i) of the chunk image = segment (ltid + 256 i) / BN,
    // column (ltid + 256 i) % BN; the global pack has the same order, so the DMA destination is linear
    constexpr int SEG_PER_STEP = 256 / BN;
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(wpack_sel) + n0 + (long long)(ltid / BN) * p.Cout_pad + (ltid % BN);
    const long long wstep16 = (long long)SEG_PER_STEP * p.Cout_pad;
    const long long wchunk16 = (long long)NTAPE * 4 * p.Cout_pad;  // 16-B units of one pipeline unit in the pack
    half8 c_hi[NSC][NPI], c_lo[NSC][NPI];

    // weights: plain 16-B global loads into registers, then ds_write_b128 (an LDS-DMA instruction costs the
    // issuing wave ~150-180 cycles per 1-KB piece on a busy CU — 18 of them per chunk made the loaders
    // issue-bound; a global_load + ds_write pair costs ~20, and the loader waves have the VGPRs to spare)
    f32x4 wreg[NWD];
#define AS_SPLIT_LOAD_W(CHUNK)                                                                        \
  {                                                                                                   \
    const f32x4* wc = wsrc + (long long)(CHUNK) * wchunk16;                                            \
    _Pragma("unroll") for (int i = 0; i < NWD; ++i) wreg[i] = wc[i * wstep16];                         \
  }
#define AS_SPLIT_STORE_W(BUF)                                                                         \
  {                                                                                                   \
    f32x4* wd = reinterpret_cast<f32x4*>(lds + (BUF) * WCHUNK) + ltid;                                 \
    _Pragma("unroll") for (int i = 0; i < NWD; ++i) wd[i * 256] = wreg[i];                             \
  }
#define AS_SPLIT_FETCH_SPLIT_P(CHUNK)                                                                 \
  _Pragma("unroll") for (int sc_ = 0; sc_ < NSC; ++sc_) {                                              \
    const int cb = ((CHUNK) * NSC + sc_) * kSplitKC;                                                  \
    const float* sp = src0;                                                                           \
    int sc = p.src_c[0], sb = 0, sbs = src0_bs;                                                       \
    if (p.n_src > 1 && cb >= p.src_end[0]) { sp = p.src[1]; sc = p.src_c[1]; sb = p.src_end[0]; sbs = p.src_bs[1]; } \
    if (p.n_src > 2 && cb >= p.src_end[1]) { sp = p.src[2]; sc = p.src_c[2]; sb = p.src_end[1]; sbs = p.src_bs[2]; } \
    if (p.n_src > 3 && cb >= p.src_end[2]) { sp = p.src[3]; sc = p.src_c[3]; sb = p.src_end[2]; sbs = p.src_bs[3]; } \
    const int left = sc - (cb - sb);  /* channels of the source from this chunk on; <= 0 in the zero padding of the last unit */ \
    if (sbs) {  /* blocked split-fp16 source: a k-half of a pixel is one 16-B unit of the hi and one of the lo plane set */ \
      const int c8 = (sc + 7) >> 3, blk = (cb - sb) >> 3;                                              \
      const _Float16* spb = reinterpret_cast<const _Float16*>(sp) + ((long long)b * 2 * c8 + (left > 0 ? blk : 0)) * plane * 8; \
      const int recs = left > 0 ? (int)((long long)(c8 - blk) * plane * 16) : 0;                       \
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)spb, 0, recs, 0x00020000); \
      const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(spb + (long long)c8 * plane * 8), 0, recs, 0x00020000); \
      u32x4 qh[NPI], ql[NPI];                                                                         \
      _Pragma("unroll") for (int i = 0; i < NPI; ++i) {                                                \
        qh[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)p_boff[i], 0, 0);                       \
        ql[i] = __builtin_amdgcn_raw_buffer_load_b128(rsl, (int)p_boff[i], 0, 0);                      \
      }                                                                                               \
      _Pragma("unroll") for (int i = 0; i < NPI; ++i) {                                                \
        c_hi[sc_][i] = __builtin_bit_cast(half8, qh[i]);                                              \
        c_lo[sc_][i] = __builtin_bit_cast(half8, ql[i]);                                              \
      }                                                                                               \
    } else {                                                                                          \
      const float* spb = sp + ((long long)b * sc + (left > 0 ? cb - sb : 0)) * plane;                  \
      const int recs = left > 0 ? (int)((long long)left * plane * 4) : 0;                              \
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)spb, 0, recs, 0x00020000); \
      const unsigned pl4 = (unsigned)(plane * 4);                                                     \
      float v[NPI][8];                                                                                \
      _Pragma("unroll") for (int i = 0; i < NPI; ++i) {                                                \
        const unsigned o0 = p_voff[i];                                                                \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                  \
          v[i][j] = as_bload(rs, o0 == 0x7FFFFFF0u ? o0 : o0 + (unsigned)j * pl4);                     \
      }                                                                                               \
      _Pragma("unroll") for (int i = 0; i < NPI; ++i) {                                                \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                \
          const _Float16 hj = (_Float16)v[i][j];                                                      \
          c_hi[sc_][i][j] = hj;                                                                       \
          c_lo[sc_][i][j] = (_Float16)((v[i][j] - (float)hj) * 2048.f);                                \
        }                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 8; j += 2)                                               \
          ovf_amax = fmaxf(ovf_amax, fmaxf(fabsf(v[i][j]), fabsf(v[i][j + 1])));                       \
      }                                                                                               \
    }                                                                                                 \
  }
#define AS_SPLIT_COMMIT_P()                                                                           \
  _Pragma("unroll") for (int sc_ = 0; sc_ < NSC; ++sc_) {                                              \
    unsigned char* pd = lds + 2 * WCHUNK + sc_ * PIMG;                                                 \
    _Pragma("unroll") for (int i = 0; i < NPI; ++i) {                                                  \
      if (p_lds[i] >= 0) {                                                                            \
        *reinterpret_cast<half8*>(pd + p_lds[i]) = c_hi[sc_][i];                                       \
        *reinterpret_cast<half8*>(pd + 2 * PATCHT * 16 + p_lds[i]) = c_lo[sc_][i];                     \
      }                                                                                               \
    }                                                                                                 \
  }
    if (dma) {
      if constexpr (PDB) {
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wpack_sel, 0, 0x7FFFFFF0, 0x00020000);
        const unsigned wvoff = (unsigned)((n0 + (long long)(ltid / BN) * p.Cout_pad + (ltid % BN)) * 16);
        bool p_slot[NPI];
#pragma unroll
        for (int i = 0; i < NPI; ++i) p_slot[i] = p_lds[i] >= 0;
#define AS_DMA_UNIT(CHUNK, BUF)                                                                       \
  {                                                                                                   \
    _Pragma("unroll") for (int g = 0; g < NWD; ++g)                                                    \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (as_lds_void*)(lds + (BUF) * WCHUNK + g * 4096 + lwave * 1024), 16, wvoff, \
                                               (unsigned)(((long long)(CHUNK) * wchunk16 + (long long)g * wstep16) * 16), 0, 0); \
    const int cb = (CHUNK) * kSplitKC;                                                                \
    const float* sp = src0;                                                                           \
    int sc = p.src_c[0], sb = 0;                                                                      \
    if (p.n_src > 1 && cb >= p.src_end[0]) { sp = p.src[1]; sc = p.src_c[1]; sb = p.src_end[0]; }      \
    if (p.n_src > 2 && cb >= p.src_end[1]) { sp = p.src[2]; sc = p.src_c[2]; sb = p.src_end[1]; }      \
    if (p.n_src > 3 && cb >= p.src_end[2]) { sp = p.src[3]; sc = p.src_c[3]; sb = p.src_end[2]; }      \
    const int left = sc - (cb - sb);                                                                  \
    const int c8 = (sc + 7) >> 3, blk = (cb - sb) >> 3;                                                \
    const _Float16* spb = reinterpret_cast<const _Float16*>(sp) + ((long long)b * 2 * c8 + (left > 0 ? blk : 0)) * plane * 8; \
    const int recs = left > 0 ? (int)((long long)(c8 - blk) * plane * 16) : 0;                         \
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)spb, 0, recs, 0x00020000); \
    const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)(spb + (long long)c8 * plane * 8), 0, recs, 0x00020000); \
    const int pimg = 2 * WCHUNK + (BUF) * PIMG;  /* byte offset of the unit's patch image */          \
    _Pragma("unroll") for (int i = 0; i < NPI; ++i) {                                                  \
      const int vo_ = (int)p_boff[i];  /* (a subscript expression as the builtin's voffset silently drops the kernel's host stub) */ \
      if (p_slot[i]) {  /* lanes past the image's last unit stay out (EXEC) */                        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (as_lds_void*)(lds + pimg + (i * 256 + lwave * 64) * 16), 16, vo_, 0, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsl, (as_lds_void*)(lds + pimg + 2 * PATCHT * 16 + (i * 256 + lwave * 64) * 16), 16, vo_, 0, 0, 0); \
      }                                                                                               \
    }                                                                                                 \
  }
        AS_DMA_UNIT(chunk_lo, 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // epilogue operands of this loader thread: ONE of its ENB8 blocks per unit, the last one three units before the end (a
        // thin stream beside the unit's DMA instead of a burst: chip-wide a block is 6 MB) — issued behind the unit's DMA (pinned
        // by sched_barriers), so that the counted wait covers the DMA and every older load and leaves only this block's loads
        // in flight.  From the first block on the loader's barriers are bare s_barriers: __syncthreads' fence would wait for
        // every outstanding load, and a loader issues no LDS instruction.
        int pre0 = chunk_hi;  // first unit of the prefetch window (none)
        if constexpr (STAGED && !FAST) {
#ifndef AS_CONV_NO_EPI_PREFETCH
          if (chunk_hi - chunk_lo >= ENB8 + 3 && (p.add != nullptr || EPI != AS_EPI_LINEAR || p.h != nullptr)) pre0 = chunk_hi - (ENB8 + 2);
#endif
        }
        for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
#ifndef AS_ABL_NO_DMA
          if (chunk + 1 < chunk_hi) AS_DMA_UNIT(chunk + 1, ((chunk - chunk_lo) & 1) ^ 1)
#endif
          if constexpr (STAGED && !FAST) {
            if (chunk >= pre0) {
              const int bi = chunk - pre0;  // wave-uniform
              if (bi < ENB8) {
                __builtin_amdgcn_sched_barrier(0);
                AS_EPI_SETUP
#pragma unroll
                for (int k = 0; k < ENB8; ++k)
                  if (bi == k) AS_EPI_LOADK(k)
                pre = true;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * EOPS) : "memory");
              } else if (chunk + 1 < chunk_hi) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last unit, issued behind the last block's loads
              }
              __builtin_amdgcn_s_barrier();
              continue;
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();  // unit c+1 landed, consumers finished unit c
        }
#undef AS_DMA_UNIT
      }
    } else {
    AS_SPLIT_LOAD_W(chunk_lo)
    AS_SPLIT_FETCH_SPLIT_P(chunk_lo)
    AS_SPLIT_STORE_W(0)
    AS_SPLIT_COMMIT_P()
    __syncthreads();
    AS_STAMP_DECL
    AS_STAMP_BEGIN
    for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
      const bool more = chunk + 1 < chunk_hi;
      if (more) {
#ifndef AS_ABL_NO_W
        AS_SPLIT_LOAD_W(chunk + 1)
#endif
        AS_STAMP_SEG(5)  // weight loads issued
#ifndef AS_ABL_NO_P
        AS_SPLIT_FETCH_SPLIT_P(chunk + 1)
#endif
        AS_STAMP_SEG(0)  // patch loads issued, returned, split
        AS_SPLIT_STORE_W(((chunk - chunk_lo) & 1) ^ 1)  // the other W image is free while the consumers work on this one
        AS_STAMP_SEG(1)  // weight image stored
      }
      __syncthreads();  // consumers finished chunk: patch and W image `cur` are free
      AS_STAMP_SEG(2)    // waited for the consumers
      if (more) {
        AS_SPLIT_COMMIT_P()
        AS_STAMP_SEG(3)  // patch committed
        __syncthreads();
        AS_STAMP_SEG(4)
      }
    }
    AS_STAMP_FLUSH(1)
    }
#undef AS_SPLIT_LOAD_W
#undef AS_SPLIT_STORE_W
#undef AS_SPLIT_FETCH_SPLIT_P
#undef AS_SPLIT_COMMIT_P
  } else {
    // =========================== CONSUMER WAVES ===========================
    const int wlane = half * WSEG + (co_base + l31) * 16;
    int plane_off[PTW];
#pragma unroll
    for (int q = 0; q < PTW; ++q) {
      const int mt = px_base + q * 32 + l31;  // pixel of the block; sub-tile mt / 128, pixel m inside it
      const int m = mt & 127;
      plane_off[q] = half * (PATCHT * 16) + ((mt >> 7) * PATCHP + (m / TW) * S * PW + (m % TW) * S) * 16;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < PTW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_h[c][q][r] = 0.f; acc_x[c][q][r] = 0.f; }
    __syncthreads();
    AS_LIFE(1)
    AS_STAMP_DECL
    AS_STAMP_BEGIN
    for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
      const bool more = chunk + 1 < chunk_hi;
      const unsigned char* wb = lds + ((chunk - chunk_lo) & 1) * WCHUNK + wlane;
      const unsigned char* pb = lds + 2 * WCHUNK + (dma ? ((chunk - chunk_lo) & 1) * PIMG : 0);
      // operand software pipeline: the ds_read_b128 of tap t+1 sit between the two halves of tap t's MFMAs
      half8 a_hi[2][2], a_lo[2][2], b_hi[2][PTW], b_lo[2][PTW];
      AS_SPLIT_LDOPS(0, 0)
      AS_SPLIT_STEP(0) AS_SPLIT_STEP(1) AS_SPLIT_STEP(2) AS_SPLIT_STEP(3) AS_SPLIT_STEP(4)
      AS_SPLIT_STEP(5) AS_SPLIT_STEP(6) AS_SPLIT_STEP(7) AS_SPLIT_STEP(8)
      AS_STAMP_SEG(0)  // operand reads + MFMAs of the chunk
      __syncthreads();
      AS_STAMP_SEG(1)  // waited for the loaders (next weight image stored, next patch fetched)
      if (more && !dma) {
        __syncthreads();
        AS_STAMP_SEG(2)  // waited for the patch commit
      }
    }
    AS_STAMP_FLUSH(0)
  }

  // ---- epilogue (consumer waves hold the accumulators) ----
  AS_LIFE(2)
  float* bias_s = reinterpret_cast<float*>(lds);  // the weight images are dead after the last barrier
  if (tid < BN) bias_s[tid] = (bias_sel && n0 + tid < p.Cout) ? bias_sel[n0 + tid] : 0.f;
  __syncthreads();
  if constexpr (EPI == kEpiTaps) {
    // act(conv + bias) is consumed on the spot by the NEXT layer's 3x3, Cout -> 1 convolution (DispHead.conv2, update.py:19,24):
    // per tap t the reduction sum_c w2[c][t] act(.)[c] over this block's 64 channels -> plane (n0/64)*9 + t of out; the
    // shifted 9-tap sum over all channel tiles is as_tap_shift_sum's.  The 256-channel hidden layer never exists in memory.
    static_assert(EPI != kEpiTaps || BN == 64, "tap reduction: one consumer wave must hold all channels of the tile");
    float* w2s = bias_s + 64;  // [64][12]
    for (int i = tid; i < 64 * 9; i += NT) {
      const int c = i / 9, t = i - c * 9;
      w2s[c * 12 + t] = (n0 + c < p.Cout) ? p.tap_w[(long long)(n0 + c) * 9 + t] : 0.f;
    }
    __syncthreads();
    // stage relu(acc + bias) as fp32 [64 channels][BM pixels] in LDS (the operand images are dead), then one thread per pixel
    // walks the 64 channels: no accumulator registers are live next to the nine tap sums, the summation order is fixed
    float* stage = bias_s + 1024;
    if (!loader) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < PTW; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int col = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            stage[col * BM + px_base + q * 32 + l31] = fmaxf(acc_h[c][q][r] + acc_x[c][q][r] * (1.f / 2048.f) + bias_s[col], 0.f);
          }
    }
    __syncthreads();
    if (tid < BM) {
      float ta[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) ta[t] = 0.f;
#pragma unroll 8
      for (int c = 0; c < 64; ++c) {
        const float o = stage[c * BM + tid];
        const f32x4 wa = *reinterpret_cast<const f32x4*>(w2s + c * 12), wb = *reinterpret_cast<const f32x4*>(w2s + c * 12 + 4);
        const float wc = w2s[c * 12 + 8];
        ta[0] += wa[0] * o; ta[1] += wa[1] * o; ta[2] += wa[2] * o; ta[3] += wa[3] * o;
        ta[4] += wb[0] * o; ta[5] += wb[1] * o; ta[6] += wb[2] * o; ta[7] += wb[3] * o;
        ta[8] += wc * o;
      }
      const int su = tid >> 7, m = tid & 127;
      const int gy = ((NSUB > 1 && su) ? sy0[NSUB - 1] : sy0[0]) + m / TW, gx = ((NSUB > 1 && su) ? sx0[NSUB - 1] : sx0[0]) + m % TW;
      if (gy < p.H && gx < p.W) {
        const long long oplane = (long long)p.H * p.W;
        float* outp = p.out + ((long long)b * p.n_tiles + nt) * 9 * oplane + (long long)gy * p.W + gx;
#pragma unroll
        for (int t = 0; t < 9; ++t) outp[(long long)t * oplane] = ta[t];
      }
    }
  } else if constexpr (EPI != kEpiPartial) {
    // Staged epilogue: the consumers park the tile (hh + cross/2048) as fp32 [BN channels][BM pixels] in LDS — the operand
    // images are dead — and ALL eight waves finish it, one thread per (pixel, 8-channel block) step: with one block per CU
    // the epilogue's global operand loads (context term, h, z) are pure latency, and the four loader waves double the loads in
    // flight; a thread owns whole 8-channel blocks of its pixel, so the blocked copy is written as full 16-B units.
    // Per element the arithmetic is the one of epi_finish (same operations, same order): results are unchanged.
    float* stage = bias_s + 256;
    // the epilogue's global operands (context term, h, z) do not depend on the tile: the first block's loads are issued
    // BEFORE the consumers park it and before the barrier (issued after it they cost one exposed round trip per block:
    // gru04 z|r 149.7 -> 148.0 us, +0.6 % pairs/s)
    AS_EPI_SETUP
    constexpr int NB8 = ENB8;              // 8-channel blocks per thread
    if (!pre) AS_EPI_LOADK(0)
    if (!loader) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < PTW; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int col = co_base + c * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            stage[col * BM + px_base + q * 32 + l31] = acc_h[c][q][r] + acc_x[c][q][r] * (1.f / 2048.f);
          }
    }
    __syncthreads();
    AS_LIFE(3)
    // the accumulators are dead: the loads of ALL remaining blocks go out together (a wave that prefetched has them already)
    if (!pre) {
#pragma unroll
      for (int k = 1; k < NB8; ++k) AS_EPI_LOADK(k)
    }
#pragma unroll
    for (int k = 0; k < NB8; ++k) {
      float ov[8];
      const int col0 = (cg * NB8 + k) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = col0 + j;
        const float x = stage[col * BM + mt] + bias_s[col] + pav[k][j];
        float o;
        if (EPI == AS_EPI_LINEAR) {
          o = act_apply(x, e.act);
          if (p.h) o = fmaxf(o + phv[k][j], 0.f);
        } else if (EPI == AS_EPI_GRU_ZR) {
          const float gte = 1.f / (1.f + expf(-x));
          o = is_r ? gte * phv[k][j] : gte;
        } else {
          o = (1.f - pzv[k][j]) * phv[k][j] + pzv[k][j] * tanhf(x);
        }
        ov[j] = o;
        if (!e.skip_out) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), e.r_out, (int)(poff == 0x7FFFFFF0u ? poff : (unsigned)col * e.plane4 + poff), 0, AS_EPI_STORE_AUX);
      }
      if (e.has_bs) {
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const _Float16 hj = (_Float16)ov[j];
          hi[j] = hj;
          lo[j] = (_Float16)((ov[j] - (float)hj) * 2048.f);
        }
#pragma unroll
        for (int j = 0; j < 8; j += 2) ovf_amax = fmaxf(ovf_amax, fmaxf(fabsf(ov[j]), fabsf(ov[j + 1])));
        const unsigned off = poff == 0x7FFFFFF0u ? poff : (unsigned)(col0 >> 3) * (e.plane4 * 4u) + poff * 4u;
        if (col0 + 8 <= e.cvalid) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), e.r_bs, (int)off, 0, AS_EPI_STORE_AUX);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), e.r_bsl, (int)off, 0, AS_EPI_STORE_AUX);
        } else {  // the result's last channels end inside or before this block: own slots + zeroed padding, 2 B each
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool own = col0 + j < e.cvalid, pad = col0 + j >= e.cend;
            if (own || pad) {
              const unsigned ok = off == 0x7FFFFFF0u ? off : off + 2u * j;
              const _Float16 zero = (_Float16)0.f;
              __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, own ? hi[j] : zero), e.r_bs, (int)ok, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, own ? lo[j] : zero), e.r_bsl, (int)ok, 0, 0);
            }
          }
        }
      }
    }
  } else if (!loader) {
    const EpiCtx e = make_epi_ctx<EPI>(p, b, n0, BN, second);
    const bool is_r = (EPI == AS_EPI_GRU_ZR) && n0 >= (p.Cout >> 1);
    unsigned poff[PTW];
#pragma unroll
    for (int q = 0; q < PTW; ++q) {
      const int mt = px_base + q * 32 + l31, m = mt & 127;
      const int su = (NSUB > 1) ? __builtin_amdgcn_readfirstlane(mt >> 7) : 0;  // a wave's 64 pixels lie in one sub-tile
      const int gy = (su ? sy0[NSUB - 1] : sy0[0]) + m / TW, gx = (su ? sx0[NSUB - 1] : sx0[0]) + m % TW;
      poff[q] = (gy < p.H && gx < p.W) ? (unsigned)(((long long)gy * p.W + gx) * 4) : 0x7FFFFFF0u;
    }
    epilogue_block<EPI, PTW>(p, e, acc_h, acc_x, co_base, poff, half, bias_s, is_r, ovf_amax);
  }
  AS_LIFE(4)
  as::note_split_overflow(ovf_amax, &g_split_overflow_conv);
}

#undef AS_EPI_SETUP
#undef AS_EPI_LOADK
#undef AS_SPLIT_STEP
#undef AS_SPLIT_STEP_IL
#undef AS_LD_A
#undef AS_LD_B
#undef AS_MM_HH
#undef AS_MM_HX
#undef AS_MM_LH
#undef AS_SB
#undef AS_SPLIT_MFMA_C
#undef AS_SPLIT_LDOPS

// weight [Cout,Cin,KS,KS] fp32 -> split pack [chunk16][tap][comp][h][Cout_pad][8] fp16 (zero padded)
// transposed != 0: pack of the data-gradient convolution's weight W'[ci][co][ky][kx] = W[co][ci][K-1-ky][K-1-kx] straight from W
// ([Cin, Cout, K, K] as the caller passes Cin / Cout of W': W is then [Cout][Cin][K][K]) — no flipped / transposed copy in memory
__global__ void pack_weights_split_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int Cin, int Cout,
                                          int Cout_pad, int ntap, long long total, int transposed = 0) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int j = (int)(idx & 7);
  long long t = idx >> 3;
  const int co = (int)(t % Cout_pad);
  t /= Cout_pad;
  const int h = (int)(t & 1);
  t >>= 1;
  const int comp = (int)(t & 1);
  t >>= 1;
  const int tap = (int)(t % ntap);
  const int chunk = (int)(t / ntap);
  const int ci = chunk * kSplitKC + 8 * h + j;
  float v = 0.f;
  if (co < Cout && ci < Cin) v = transposed ? w[((long long)ci * Cout + co) * ntap + (ntap - 1 - tap)] : w[((long long)co * Cin + ci) * ntap + tap];
  as::fp16_saturate_mode();
  const _Float16 hi = (_Float16)v;
  wp[idx] = comp == 0 ? hi : (_Float16)((v - (float)hi) * 2048.f);
}

// ---- split-K finish: sum the K-slice slabs and apply the fused epilogue (one lane per output element) ----
template <int EPI>
__global__ __launch_bounds__(256) void conv_finish_kernel(ConvParams p) {
  const long long plane = (long long)p.H * p.W;
  // with a blocked split-fp16 copy the channel range is walked to the end of its last 8-block (the pad slots are zeroed)
  const int cwalk = p.out_bs ? ((EPI == AS_EPI_GRU_ZR) ? p.Cout : (p.Cout + 7) / 8 * 8) : p.Cout;
  const long long total = (long long)p.B * cwalk * plane;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long long pix = idx % plane;
  const int co = (int)((idx / plane) % cwalk);
  const int b = (int)(idx / (plane * cwalk));
  float o = 0.f;
  bool to_bs = p.out_bs != nullptr;
  if (co >= p.Cout && p.out_bs_coff8 * 8 + co < p.out_bs_ctot) to_bs = false;  // another producer's slot, not padding
  int cb_ = co;  // channel inside the mirrored tensor
  if (co < p.Cout) {
    // all (<= 8) K-slice partials as unconditional buffer loads in flight together (missing slices read 0 through the range
    // check), summed in slice order: a run-time trip count made this a chain of dependent load -> add round trips
    const long long slab = (long long)p.B * p.Cout_pad * plane;  // elements of one K slice
    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc((void*)p.ws, 0, (int)((long long)p.ksplit * slab * 4), 0x00020000);
    const unsigned o0 = (unsigned)((((long long)b * p.Cout_pad + co) * plane + pix) * 4), sl4 = (unsigned)(slab * 4);
    float part[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) part[ks] = as_bload(rws, ks < p.ksplit ? o0 + (unsigned)ks * sl4 : 0x7FFFFFF0u);
    float x = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) x += part[ks];
    if (p.bias) x += p.bias[co];
    if (p.add) x += p.add[((long long)b * p.add_ctot + p.add_coff + co) * plane + pix];
    if (EPI == AS_EPI_LINEAR) {
      o = act_apply(x, p.act);
      if (p.h) o = fmaxf(o + p.h[((long long)b * p.Cout + co) * plane + pix], 0.f);
      if (!p.bs_only) p.out[((long long)b * p.out_ctot + p.out_coff + co) * plane + pix] = o;
    } else if (EPI == AS_EPI_GRU_ZR) {
      const int ch = p.Cout >> 1;
      const float g = 1.f / (1.f + expf(-x));
      if (co < ch) { p.out[((long long)b * ch + co) * plane + pix] = g; to_bs = false; }
      else {
        const long long oo = ((long long)b * ch + (co - ch)) * plane + pix;
        o = g * p.h[oo];
        if (!p.bs_only) p.out2[oo] = o;
        cb_ = co - ch;
      }
    } else {
      const long long oo = ((long long)b * p.Cout + co) * plane + pix;
      const float zz = p.z[oo];
      o = (1.f - zz) * p.h[oo] + zz * tanhf(x);
      p.out[oo] = o;
    }
  }
  if (to_bs) {
    _Float16* rec = p.out_bs + (((long long)b * 2 * p.out_bs_c8tot + p.out_bs_coff8 + (cb_ >> 3)) * plane + pix) * 8 + (cb_ & 7);
    as::fp16_saturate_mode();
    const _Float16 hk = (_Float16)o;
    rec[0] = hk;
    rec[(long long)p.out_bs_c8tot * plane * 8] = (_Float16)((o - (float)hk) * 2048.f);
    as::note_split_overflow(fabsf(o), &g_split_overflow_conv);
  }
}

template <int KS, int TW, int BN, int EPI, int NSUB = 1, int S = 1, bool FAST = false, bool LEAN = false>
int launch_conv_split_epi(const ConvParams& p, hipStream_t s) {
  if constexpr (!FAST && KS == 3 && S == 1) {  // the one-MFMA variant exists for the stride-1 3x3 convolutions (the GRU loop, the context net)
    if (p.fast16) return launch_conv_split_epi<KS, TW, BN, EPI, NSUB, S, true>(p, s);
  }
  if constexpr (!LEAN && !FAST && KS == 3 && S == 1 && BN == 64) {
    // two 4-wave blocks per CU instead of one 8-wave block when every source is blocked (AS_CONV_LEAN: 0 off, 1 = the
    // 256-pixel blocks of the big maps, 2 = every 64-channel 3x3 launch)
    static const int lean_mode = getenv("AS_CONV_LEAN") ? atoi(getenv("AS_CONV_LEAN")) : 1;
    // measured (cfg 2): with >= 2 blocks for every CU the pair overlaps one block's staging / prologue / epilogue with the
    // other's MFMAs (gru04 z|r 149.3 -> 146.6 us, head conv1 58.9 -> 54.4); with fewer blocks a CU holds ONE single-buffered
    // block and loses (gru04 q 85.5 -> 102.8, gru08 z|r 62 -> 87), as do 128-pixel lean blocks at three per CU (156.6 vs 144.4)
    const long long nblk = (long long)p.B * as::cdiv64((long long)p.tiles_x * p.tiles_y, NSUB) * p.n_tiles * p.ksplit;
    // 3 (A/B knob): mode 1 + the 128-pixel blocks of the small maps (49 KB: up to three per CU, co-resident with another
    // launch's blocks) — never the big maps' 256-block launches, which run alone and need their own double buffering
    if (p.all_bs && (lean_mode == 2 || ((lean_mode == 1 || lean_mode == 3) && NSUB == 2 && nblk >= 2 * kNumCU) || (lean_mode == 3 && NSUB == 1)))
      return launch_conv_split_epi<KS, TW, BN, EPI, NSUB, S, false, true>(p, s);
  }
  constexpr int TH = 128 / TW, PATCHP = ((TH - 1) * S + KS) * ((TW - 1) * S + KS);
  constexpr int NSC = (KS == 1) ? 4 : 1;
  constexpr size_t wimg = (size_t)(KS * KS * NSC * 4 * BN * 16), pimg = (size_t)NSC * (4 * NSUB * PATCHP * 16);
  // LEAN: one weight + one patch image, and room for the epilogue's staging (bias / tap weights 4 KB + the fp32 tile)
  constexpr size_t lds_lean = (wimg + pimg > 4096 + (size_t)BN * 128 * NSUB * 4) ? wimg + pimg : 4096 + (size_t)BN * 128 * NSUB * 4;
  constexpr size_t lds = LEAN ? lds_lean
                              : 2 * wimg + ((KS == 3 && 2 * wimg + 2 * pimg <= 160 * 1024) ? 2 : 1) * pimg;  // the kernel's PDB rule
  static_assert(lds <= 160 * 1024, "conv_split: LDS budget");
  static_assert(!LEAN || lds <= 80 * 1024, "conv_split (lean): two blocks per CU");
  static bool configured = false;  // per instantiation; the attribute is idempotent
  if (!configured && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_split_kernel<KS, TW, BN, EPI, NSUB, S, FAST, LEAN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    configured = true;
  }
  const long long groups = as::cdiv64((long long)p.tiles_x * p.tiles_y, NSUB);
  const dim3 grid((unsigned)((long long)p.B * groups * p.n_tiles * p.ksplit));
  static const int xcd_mode = getenv("AS_CONV_XCD") ? atoi(getenv("AS_CONV_XCD")) : 2;  // 0 off | 1 channel tiles together | 2 + banded pixel tiles
  ConvParams q = p;
  q.xcd_map = xcd_mode == 2 ? 2 : ((xcd_mode && p.n_tiles > 1) ? 1 : 0);  // 2: banded pixel tiles per XCD (any channel-tile count)
  static const int stagger = getenv("AS_CONV_XCD_STAGGER") ? atoi(getenv("AS_CONV_XCD_STAGGER")) : 0;
  q.stagger = stagger;
  static const int lean_offset = getenv("AS_CONV_LEAN_OFFSET") ? atoi(getenv("AS_CONV_LEAN_OFFSET")) : 0;
  q.lean_offset = lean_offset;
  hipLaunchKernelGGL((conv_split_kernel<KS, TW, BN, EPI, NSUB, S, FAST, LEAN>), grid, dim3(LEAN ? 256 : 512), lds, s, q);
  return as::check_launch("conv2d(split)");
}

template <int KS, int TW, int BN, int NSUB = 1>
int launch_conv_split(const ConvParams& p, int epi, hipStream_t s) {
  if (p.ksplit > 1) {
    int rc = launch_conv_split_epi<KS, TW, BN, kEpiPartial>(p, s);
    if (rc != AS_OK) return rc;
    const int cwalk = (p.out_bs && epi != AS_EPI_GRU_ZR) ? (p.Cout + 7) / 8 * 8 : p.Cout;  // conv_finish_kernel's channel walk
    const long long total = (long long)p.B * cwalk * p.H * p.W;
    const dim3 g((unsigned)as::cdiv64(total, 256));
    if (epi == AS_EPI_LINEAR) hipLaunchKernelGGL(conv_finish_kernel<AS_EPI_LINEAR>, g, dim3(256), 0, s, p);
    else if (epi == AS_EPI_GRU_ZR) hipLaunchKernelGGL(conv_finish_kernel<AS_EPI_GRU_ZR>, g, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_finish_kernel<AS_EPI_GRU_Q>, g, dim3(256), 0, s, p);
    return as::check_launch("conv2d(split-K finish)");
  }
  if (epi == AS_EPI_LINEAR) return launch_conv_split_epi<KS, TW, BN, AS_EPI_LINEAR, NSUB>(p, s);
  if (epi == AS_EPI_GRU_ZR) return launch_conv_split_epi<KS, TW, BN, AS_EPI_GRU_ZR, NSUB>(p, s);
  if (epi == AS_EPI_RELU_TAPS) {
    if constexpr (BN == 64 && KS == 3) return launch_conv_split_epi<KS, TW, BN, kEpiTaps, NSUB>(p, s);
    else return as::fail(AS_ERR_BAD_ARG, "conv2d(RELU_TAPS): 3x3, 64-channel tiles only");
  }
  return launch_conv_split_epi<KS, TW, BN, AS_EPI_GRU_Q, NSUB>(p, s);
}

template <int KS, int TW>
int launch_conv(const ConvParams& p, int epi, hipStream_t s) {
  const dim3 grid((unsigned)((long long)p.B * p.tiles_x * p.tiles_y * p.n_tiles));
  if (epi == AS_EPI_LINEAR) hipLaunchKernelGGL((conv_igemm_kernel<KS, TW, AS_EPI_LINEAR>), grid, dim3(256), 0, s, p);
  else if (epi == AS_EPI_GRU_ZR) hipLaunchKernelGGL((conv_igemm_kernel<KS, TW, AS_EPI_GRU_ZR>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((conv_igemm_kernel<KS, TW, AS_EPI_GRU_Q>), grid, dim3(256), 0, s, p);
  return as::check_launch("conv2d");
}

int conv_kc(int KS) { return KS == 3 ? ConvCfg<3>::KC : ConvCfg<1>::KC; }

// Small feature maps give too few blocks to pull the weight stream (each CU fills its LDS at ~35 GB/s, so
// a 1/16-res GRU conv on 36 CUs is bound by 36 x that): split K over more blocks when a workspace is given.
void conv_pick_ksplit(ConvParams& p, const as_conv_desc* d) {
  p.ksplit = 1;
  p.ws = nullptr;
  if (!d->ws || d->ws_elems <= 0) return;
  const long long blocks = (long long)p.B * p.tiles_x * p.tiles_y * p.n_tiles;
  const long long slab = (long long)p.B * p.Cout_pad * p.H * p.W;
  // one block per CU (the tile's LDS footprint): more blocks than CUs would just queue a second round
  int ks = (int)(kNumCU / blocks);
  // AS_CONV_KSPLIT_MAX (A/B knob): 1 = never split.  Alone a split launch is shorter; beside the other chain's kernels the
  // unsplit one costs fewer CU-microseconds (no partial slabs, no finish launch) on half the CUs.
  static const int ks_max = getenv("AS_CONV_KSPLIT_MAX") ? atoi(getenv("AS_CONV_KSPLIT_MAX")) : 8;
  if (ks > ks_max) ks = ks_max;
  if (ks > 8) ks = 8;
  while (ks > 1 && (p.chunks / ks < 2 || slab * ks > d->ws_elems)) --ks;
  if (ks > 1) { p.ksplit = ks; p.ws = d->ws; }
}
int conv_cout_pad(int Cout) { return ((Cout + kBN - 1) / kBN) * kBN; }

// second convolution of a dual launch: twice the channel tiles, the upper half addressed to the second problem
void conv_apply_dual(ConvParams& p, const as_conv_desc* d) {
  p.dual = 1;
  p.n_tiles *= 2;
  p.src2 = d->src2;
  p.src2_bs = d->src2_bs ? 1 : 0;
  p.wpack2 = d->wpack2;
  p.bias2 = d->bias2;
  p.out_coff2 = d->out_coff2;
  p.out_bs_coff8_2 = d->out_bs_coff2 / 8;
  p.h2 = d->h2;
  p.act2 = d->dual_act2 ? d->act2 : d->act;
  p.dual_sep = (d->out_b || d->out_bs_b) ? 1 : 0;
  p.out_b = d->out_b;
  p.out_bs_b = reinterpret_cast<_Float16*>(d->out_bs_b);
}

}  // namespace

extern "C" {

#ifdef AS_CONV_STAMPS
// diagnostic build only: copy the stamp sums [block][16] (slots 0-5 consumer, 8-13 loader) to the host and clear them
int as_debug_conv_stamps(unsigned long long* out, int n) {
  if (!out || n <= 0 || n > kStampBlocks * kStampSlots) return AS_ERR_BAD_ARG;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(as_conv_stamp_buf), sizeof(unsigned long long) * n) != hipSuccess) return AS_ERR_LAUNCH;
  static unsigned long long zeros[kStampBlocks * kStampSlots];
  if (hipMemcpyToSymbol(HIP_SYMBOL(as_conv_stamp_buf), zeros, sizeof(zeros)) != hipSuccess) return AS_ERR_LAUNCH;
  return AS_OK;
}
// diagnostic build only: the block lifetime stamps [block][8] of the last conv_split_kernel launch
int as_debug_conv_life(unsigned long long* out, int n) {
  if (!out || n <= 0 || n > kStampBlocks * 8) return AS_ERR_BAD_ARG;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(as_conv_life_buf), sizeof(unsigned long long) * n) != hipSuccess) return AS_ERR_LAUNCH;
  return AS_OK;
}
#endif

int64_t as_conv_pack_size(int Cin, int Cout, int KS) {
  if (Cin <= 0 || Cout <= 0 || (KS != 1 && KS != 3)) return -1;
  const int KC = conv_kc(KS);
  const int64_t chunks = (Cin + KC - 1) / KC;
  return chunks * KS * KS * KC * conv_cout_pad(Cout);
}

int as_conv_pack_weights(const float* weight, float* wpack, int Cin, int Cout, int KS, void* stream) {
  AS_REQUIRE(weight && wpack, AS_ERR_BAD_ARG, "conv_pack: null pointer");
  const int64_t total = as_conv_pack_size(Cin, Cout, KS);
  AS_REQUIRE(total > 0, AS_ERR_BAD_ARG, "conv_pack: unsupported Cin=%d Cout=%d KS=%d", Cin, Cout, KS);
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream),
                     weight, wpack, Cin, Cout, conv_cout_pad(Cout), KS, conv_kc(KS), (long long)total);
  return as::check_launch("conv_pack_weights");
}

int64_t as_conv_pack_size_split(int Cin, int Cout, int KS) {
  if (Cin <= 0 || Cout <= 0 || (KS != 1 && KS != 3)) return -1;
  int64_t chunks = (Cin + kSplitKC - 1) / kSplitKC;
  if (KS == 1) chunks = (chunks + 3) / 4 * 4;  // 1x1: the kernel's pipeline unit is four chunks (zero padded)
  return chunks * KS * KS * 4 * conv_cout_pad(Cout) * 8;  // fp16 elements
}

int as_conv_pack_weights_split_t(const float* weight, void* wpack, int Cin, int Cout, int KS, void* stream) {
  AS_REQUIRE(weight && wpack, AS_ERR_BAD_ARG, "conv_pack_split_t: null pointer");
  const int64_t total = as_conv_pack_size_split(Cin, Cout, KS);
  AS_REQUIRE(total > 0, AS_ERR_BAD_ARG, "conv_pack_split_t: unsupported Cin=%d Cout=%d KS=%d", Cin, Cout, KS);
  hipLaunchKernelGGL(pack_weights_split_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream),
                     weight, (_Float16*)wpack, Cin, Cout, conv_cout_pad(Cout), KS * KS, (long long)total, 1);
  return as::check_launch("conv_pack_weights_split_t");
}

int as_conv_pack_weights_split(const float* weight, void* wpack, int Cin, int Cout, int KS, void* stream) {
  AS_REQUIRE(weight && wpack, AS_ERR_BAD_ARG, "conv_pack_split: null pointer");
  const int64_t total = as_conv_pack_size_split(Cin, Cout, KS);
  AS_REQUIRE(total > 0, AS_ERR_BAD_ARG, "conv_pack_split: unsupported Cin=%d Cout=%d KS=%d", Cin, Cout, KS);
  hipLaunchKernelGGL(pack_weights_split_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream),
                     weight, (_Float16*)wpack, Cin, Cout, conv_cout_pad(Cout), KS * KS, (long long)total, 0);
  return as::check_launch("conv_pack_weights_split");
}

int64_t as_conv_ws_elems(int B, int Cout, int H, int W) {
  if (B <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  const int cpad = conv_cout_pad(Cout);
  const int bn = (cpad % 128 == 0) ? 128 : 64;
  const long long tiles = (long long)B * as::cdiv64((long long)H * W, 128) * (cpad / bn);
  int ks = (int)(kNumCU / tiles);
  if (ks > 8) ks = 8;
  return ks > 1 ? (int64_t)ks * B * cpad * H * W : 0;
}

int as_conv2d(const as_conv_desc* d, void* stream);
static int conv2d_dual_sequential(const as_conv_desc* d, void* stream) {
  as_conv_desc a = *d;
  a.dual = 0;
  const int rc = as_conv2d(&a, stream);
  if (rc != AS_OK) return rc;
  a.src[0] = d->src2; a.src_bs[0] = d->src2_bs; a.wpack = d->wpack2; a.bias = d->bias2;
  a.out_coff = d->out_coff2; a.out_bs_coff = d->out_bs_coff2;
  a.h = d->h2;
  if (d->dual_act2) a.act = d->act2;
  if (d->out_b || d->out_bs_b) {  // dense outputs of its own
    a.out = d->out_b; a.out_ctot = d->Cout; a.out_coff = 0;
    a.out_bs = d->out_bs_b; a.out_bs_ctot = d->Cout; a.out_bs_coff = 0;
    a.bs_only = (d->out_b == nullptr) ? 1 : 0;
  }
  a.h2 = nullptr; a.out_b = nullptr; a.out_bs_b = nullptr; a.dual_act2 = 0;
  return as_conv2d(&a, stream);
}

int as_conv2d(const as_conv_desc* d, void* stream) {
  AS_REQUIRE(d, AS_ERR_BAD_ARG, "conv2d: null descriptor");
  if (d->dual) {
    AS_REQUIRE(d->n_src == 1 && d->epilogue == AS_EPI_LINEAR && !d->add && (d->stride == 0 || d->stride == 1), AS_ERR_BAD_ARG,
               "conv2d(dual): one source, LINEAR epilogue, no add, stride 1");
    AS_REQUIRE((d->h == nullptr) == (d->h2 == nullptr), AS_ERR_BAD_ARG, "conv2d(dual): a residual for both convolutions or for neither");
    AS_REQUIRE(!d->dual_act2 || (d->act2 >= AS_ACT_NONE && d->act2 <= AS_ACT_LEAKY), AS_ERR_BAD_ARG, "conv2d(dual): act2=%d", d->act2);
    AS_REQUIRE(!d->out_bs_b || (reinterpret_cast<uintptr_t>(d->out_bs_b) & 15) == 0, AS_ERR_BAD_ARG, "conv2d(dual): out_bs_b not 16-B aligned");
    AS_REQUIRE(!(d->out_b || d->out_bs_b) || ((d->out_b != nullptr) == (d->out != nullptr && !d->bs_only) && (d->out_bs_b != nullptr) == (d->out_bs != nullptr)),
               AS_ERR_BAD_ARG, "conv2d(dual): separate second outputs mirror the first convolution's (fp32 and / or blocked)");
    AS_REQUIRE(d->src2 && d->wpack2 && (reinterpret_cast<uintptr_t>(d->wpack2) & 15) == 0, AS_ERR_BAD_ARG, "conv2d(dual): null / misaligned src2 / wpack2");
    AS_REQUIRE(!d->src2_bs || (reinterpret_cast<uintptr_t>(d->src2) & 15) == 0, AS_ERR_BAD_ARG, "conv2d(dual): blocked src2 not 16-B aligned");
    AS_REQUIRE(d->out_coff2 >= 0 && d->out_bs_coff2 >= 0 && d->out_bs_coff2 % 8 == 0, AS_ERR_BAD_SHAPE, "conv2d(dual): bad second output window");
    if (d->precision != 1) return conv2d_dual_sequential(d, stream);
  }
  AS_REQUIRE(d->KS == 1 || d->KS == 3, AS_ERR_BAD_ARG, "conv2d: KS=%d (supported: 1, 3)", d->KS);
  AS_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, AS_ERR_BAD_ARG, "conv2d: non-positive size");
  AS_REQUIRE(d->n_src >= 1 && d->n_src <= AS_MAX_SRCS, AS_ERR_BAD_ARG, "conv2d: n_src=%d", d->n_src);
  AS_REQUIRE(d->wpack, AS_ERR_BAD_ARG, "conv2d: null wpack");
  const bool bs_only = d->out_bs && d->bs_only;
  // the fp32 result may be omitted only where the blocked copy replaces it (LINEAR: out; GRU_ZR: out2)
  AS_REQUIRE(d->out || (bs_only && d->epilogue == AS_EPI_LINEAR), AS_ERR_BAD_ARG, "conv2d: null out");
  AS_REQUIRE((reinterpret_cast<uintptr_t>(d->wpack) & 15) == 0, AS_ERR_BAD_ARG, "conv2d: wpack not 16-B aligned");
  ConvParams p{};
  int csum = 0;
  for (int i = 0; i < d->n_src; ++i) {
    AS_REQUIRE(d->src[i] && d->src_c[i] > 0, AS_ERR_BAD_ARG, "conv2d: source %d null or empty", i);
    p.src[i] = d->src[i];
    p.src_c[i] = d->src_c[i];
    csum += d->src_c[i];
    p.src_end[i] = csum;
    p.src_bs[i] = d->src_bs[i] ? 1 : 0;
    AS_REQUIRE(!d->src_bs[i] || (d->precision == 1 && (d->stride == 0 || d->stride == 1)), AS_ERR_BAD_ARG,
               "conv2d: blocked split-fp16 sources need the split-precision kernel, stride 1");
    AS_REQUIRE(!d->src_bs[i] || (reinterpret_cast<uintptr_t>(d->src[i]) & 15) == 0, AS_ERR_BAD_ARG, "conv2d: blocked source %d not 16-B aligned", i);
  }
  if (d->out_bs) {
    AS_REQUIRE(d->precision == 1 && (d->stride == 0 || d->stride == 1), AS_ERR_BAD_ARG, "conv2d: out_bs needs the split-precision kernel, stride 1");
    AS_REQUIRE((reinterpret_cast<uintptr_t>(d->out_bs) & 15) == 0, AS_ERR_BAD_ARG, "conv2d: out_bs not 16-B aligned");
    const int cres = d->epilogue == AS_EPI_GRU_ZR ? d->Cout / 2 : d->Cout;
    const int ctot = d->out_bs_ctot > 0 ? d->out_bs_ctot : cres;
    AS_REQUIRE(d->out_bs_coff >= 0 && d->out_bs_coff % 8 == 0 && d->out_bs_coff + cres <= (ctot + 7) / 8 * 8, AS_ERR_BAD_SHAPE,
               "conv2d: out_bs channel window [%d,%d) must start at a multiple of 8 inside %d channels", d->out_bs_coff, d->out_bs_coff + cres, ctot);
    AS_REQUIRE(d->epilogue != AS_EPI_GRU_Q || !d->bs_only, AS_ERR_BAD_ARG, "conv2d(GRU_Q): the fp32 hidden state is always written");
    p.out_bs = reinterpret_cast<_Float16*>(d->out_bs);
    p.out_bs_c8tot = (ctot + 7) / 8;
    p.out_bs_ctot = ctot;
    p.out_bs_coff8 = d->out_bs_coff / 8;
    p.bs_only = d->bs_only ? 1 : 0;
  }
  AS_REQUIRE(csum == d->Cin, AS_ERR_BAD_SHAPE, "conv2d: sources hold %d channels, Cin=%d", csum, d->Cin);
  AS_REQUIRE(d->precision == 0 || d->precision == 1, AS_ERR_BAD_ARG, "conv2d: precision=%d", d->precision);
  const bool split = d->precision == 1;
  const int kc_req = split ? kSplitKC : conv_kc(d->KS);
  for (int i = 0; i + 1 < d->n_src; ++i)
    AS_REQUIRE(d->src_c[i] % kc_req == 0, AS_ERR_BAD_SHAPE,
               "conv2d: source %d has %d channels; every source but the last must hold a multiple of %d (concatenate first)",
               i, d->src_c[i], kc_req);
  p.n_src = d->n_src;
  {
    bool all = d->precision == 1;
    for (int i = 0; i < d->n_src; ++i) all = all && d->src_bs[i];
    if (d->dual) all = all && d->src2_bs;
    static const int dma_mode = getenv("AS_CONV_DMA") ? atoi(getenv("AS_CONV_DMA")) : 1;
    p.all_bs = (all && dma_mode) ? 1 : 0;
  }
  p.wpack = d->wpack; p.bias = d->bias; p.add = d->add;
  p.add_ctot = d->add_ctot; p.add_coff = d->add_coff;
  AS_REQUIRE(!d->add || (d->add_coff >= 0 && d->add_coff + d->Cout <= d->add_ctot), AS_ERR_BAD_SHAPE, "conv2d: add channel window [%d,%d) outside %d", d->add_coff, d->add_coff + d->Cout, d->add_ctot);
  p.h = d->h; p.z = d->z; p.out = d->out; p.out2 = d->out2;
  p.B = d->B; p.Cin = d->Cin; p.Cout = d->Cout; p.act = d->act;
  p.fast16 = (d->precision == 1 && as::fast16_mode()) ? 1 : 0;
  p.Cout_pad = conv_cout_pad(d->Cout);
  const int epi = d->epilogue;
  if (epi == AS_EPI_LINEAR) {
    p.out_ctot = d->out_ctot > 0 ? d->out_ctot : d->Cout;
    p.out_coff = d->out_coff;
    AS_REQUIRE(bs_only || (p.out_coff >= 0 && p.out_coff + d->Cout <= p.out_ctot), AS_ERR_BAD_SHAPE, "conv2d: out channel window outside out_ctot");
    AS_REQUIRE(!d->dual || bs_only || d->out_coff2 + d->Cout <= p.out_ctot, AS_ERR_BAD_SHAPE, "conv2d(dual): second out channel window outside out_ctot");
    AS_REQUIRE(d->act >= AS_ACT_NONE && d->act <= AS_ACT_LEAKY, AS_ERR_BAD_ARG, "conv2d: act=%d", d->act);
  } else if (epi == AS_EPI_GRU_ZR) {
    AS_REQUIRE(d->h && (d->out2 || bs_only) && (d->Cout % (2 * kBN)) == 0, AS_ERR_BAD_ARG, "conv2d(GRU_ZR): needs h, out2 and Cout %% 128 == 0");
  } else if (epi == AS_EPI_GRU_Q) {
    AS_REQUIRE(d->h && d->z, AS_ERR_BAD_ARG, "conv2d(GRU_Q): needs h and z");
  } else if (epi == AS_EPI_RELU_TAPS) {
    AS_REQUIRE(d->tap_w && d->out && !d->add && !d->h && !d->out_bs && !d->dual && d->precision == 1 && d->KS == 3 && (d->stride == 0 || d->stride == 1),
               AS_ERR_BAD_ARG, "conv2d(RELU_TAPS): needs tap_w and out; 3x3, split precision, stride 1, no add / residual / blocked copy / dual");
    AS_REQUIRE(d->act == AS_ACT_RELU, AS_ERR_BAD_ARG, "conv2d(RELU_TAPS): act must be AS_ACT_RELU");
    p.tap_w = d->tap_w;
  } else {
    return as::fail(AS_ERR_BAD_ARG, "conv2d: epilogue=%d", epi);
  }
  const int KC = split ? kSplitKC : conv_kc(d->KS);
  for (int i = 0; i < d->n_src; ++i)  // 32-bit buffer offsets inside one (batch, source) tensor
    AS_REQUIRE((long long)d->src_c[i] * d->H * d->W * 4 < 0x7FFFFFF0ll, AS_ERR_BAD_SHAPE,
               "conv2d: source %d exceeds 2 GiB per batch element", i);
  p.chunks = (d->Cin + KC - 1) / KC;
  p.n_tiles = p.Cout_pad / kBN;
  hipStream_t s = as::as_stream(stream);
  p.ksplit = 1;
  p.ws = nullptr;
  const int stride = d->stride ? d->stride : 1;  // 0 (zero-initialised descriptor) = 1
  AS_REQUIRE(stride == 1 || (stride == 2 && split && d->KS == 3 && epi == AS_EPI_LINEAR), AS_ERR_BAD_ARG,
             "conv2d: stride=%d (stride 2: 3x3, split precision, LINEAR epilogue only)", stride);
  if (split) {
    const int bn = (p.Cout_pad % 128 == 0) ? 128 : 64;
    p.n_tiles = p.Cout_pad / bn;
    AS_REQUIRE((long long)d->H * d->W < 2147483647ll, AS_ERR_BAD_SHAPE, "conv2d: plane too large");
    p.Hi = d->H;
    p.Wi = d->W;
    if (stride == 2) {
      // output plane (H-1)/2+1: 8x16 output tiles x 64 channels (the 17x33 halo patch leaves LDS room for one 64-wide weight image pair)
      p.H = (d->H - 1) / 2 + 1;
      p.W = (d->W - 1) / 2 + 1;
      p.n_tiles = p.Cout_pad / 64;
      p.tiles_x = as::cdiv(p.W, 16);
      p.tiles_y = as::cdiv(p.H, 8);
      AS_REQUIRE((long long)p.B * p.tiles_x * p.tiles_y * p.n_tiles < 2147483647ll, AS_ERR_BAD_SHAPE, "conv2d: grid too large");
      return launch_conv_split_epi<3, 16, 64, AS_EPI_LINEAR, 1, 2>(p, s);
    }
    if (d->KS == 1) {
      p.chunks = (p.chunks + 3) / 4;  // pipeline units of four 16-channel chunks
      p.Hi = 1;
      p.Wi = d->H * d->W;
      p.H = 1;
      p.W = d->H * d->W;
      p.tiles_x = as::cdiv(p.W, 128);
      p.tiles_y = 1;
      AS_REQUIRE((long long)p.B * p.tiles_x * p.n_tiles < 2147483647ll, AS_ERR_BAD_SHAPE, "conv2d: grid too large");
      conv_pick_ksplit(p, d);
      if (d->dual) {
        if (p.ksplit > 1) return conv2d_dual_sequential(d, stream);
        conv_apply_dual(p, d);
      }
      return bn == 128 ? launch_conv_split<1, 128, 128>(p, epi, s) : launch_conv_split<1, 128, 64>(p, epi, s);
    }
    p.H = d->H;
    p.W = d->W;
    // 128-pixel tile as 8x16 or 4x32, whichever pads the image less (ties: 8x16, smaller halo)
    const long long a16 = (long long)as::cdiv(p.W, 16) * 16 * as::cdiv(p.H, 8) * 8;
    const long long a32 = (long long)as::cdiv(p.W, 32) * 32 * as::cdiv(p.H, 4) * 4;
    const int tw = a32 < a16 ? 32 : 16;
    p.tiles_x = as::cdiv(p.W, tw);
    p.tiles_y = as::cdiv(p.H, 128 / tw);
    AS_REQUIRE((long long)p.B * p.tiles_x * p.tiles_y * p.n_tiles < 2147483647ll, AS_ERR_BAD_SHAPE, "conv2d: grid too large");
    conv_pick_ksplit(p, d);
    if (d->dual && p.ksplit > 1) return conv2d_dual_sequential(d, stream);
    if (epi == AS_EPI_RELU_TAPS) {  // 64-channel tiles, whole K per block (the reduction is over a tile's channels)
      p.ksplit = 1;
      p.ws = nullptr;
      p.n_tiles = p.Cout_pad / 64;
      const long long groups2 = (long long)p.B * as::cdiv64((long long)p.tiles_x * p.tiles_y, 2) * p.n_tiles;
      if (groups2 >= 2 * kNumCU) return tw == 16 ? launch_conv_split<3, 16, 64, 2>(p, epi, s) : launch_conv_split<3, 32, 64, 2>(p, epi, s);
      return tw == 16 ? launch_conv_split<3, 16, 64>(p, epi, s) : launch_conv_split<3, 32, 64>(p, epi, s);
    }
    // big maps: 256-pixel x 64-channel blocks (two sub-tiles) pull 30 % fewer bytes per MFMA through the CU's L1
    static const int wide_mode = getenv("AS_CONV_WIDE") ? atoi(getenv("AS_CONV_WIDE")) : 1;
    const long long wide_blocks = (long long)p.B * as::cdiv64((long long)p.tiles_x * p.tiles_y, 2) * (p.Cout_pad / 64);
    const long long now_blocks = (long long)p.B * p.tiles_x * p.tiles_y * p.n_tiles;
    bool wide_ok = bn == 128 ? as::cdiv64(wide_blocks, kNumCU) <= as::cdiv64(now_blocks, kNumCU)
                             : wide_blocks >= 2 * kNumCU;  // bn == 64: a wide block is twice the work of a current one
    // 64-channel layers below that bar (the encoder's convc2 || convd2 dual launch: 2 x 255 narrow blocks = two rounds, or 2 x 128
    // wide blocks = exactly one): rounds x (fixed cost + chunks x chunk time) from the block-lifetime stamps of both forms
    // (narrow: 6.8 us prologue + park + finish, 1.2 us per chunk; wide: 10.8 us, 2.3 us per chunk; tools/conv_stamps.py)
    static const int wide64_mode = getenv("AS_CONV_WIDE64") ? atoi(getenv("AS_CONV_WIDE64")) : 1;
    if (!wide_ok && bn == 64 && wide64_mode) {
      const int mult = d->dual ? 2 : 1;
      const double t_now = (double)as::cdiv64(now_blocks * mult, kNumCU) * (6.8 + 1.2 * p.chunks);
      const double t_wide = (double)as::cdiv64(wide_blocks * mult, kNumCU) * (10.8 + 2.3 * p.chunks);
      wide_ok = t_wide < t_now;
    }
    if (wide_mode && p.ksplit == 1 && wide_ok) {
      p.n_tiles = p.Cout_pad / 64;
      if (d->dual) conv_apply_dual(p, d);
      return tw == 16 ? launch_conv_split<3, 16, 64, 2>(p, epi, s) : launch_conv_split<3, 32, 64, 2>(p, epi, s);
    }
    // small maps with every source blocked: 64-channel tiles (two patch images fit next to their weight images -> all-DMA
    // staging) with K split over the blocks that leaves, instead of 128-channel tiles on the register-staged path
    static const int small_dma = getenv("AS_CONV_SMALL_DMA") ? atoi(getenv("AS_CONV_SMALL_DMA")) : 1;
    bool use64 = bn == 64;
    if (small_dma && p.all_bs && bn == 128 && !d->dual) {
      p.n_tiles = p.Cout_pad / 64;
      conv_pick_ksplit(p, d);
      use64 = true;
    }
    // A/B knob AS_CONV_PREFER64=1 (off by default: measured slower on the cfg-4 step, 53.6 vs 52.8 ms): fp32 sources, 64-channel
    // tiles when they need NO K split where the 128-channel tiles do — one fused launch instead of partial sums + a finish launch
    static const int prefer64 = getenv("AS_CONV_PREFER64") ? atoi(getenv("AS_CONV_PREFER64")) : 0;
    if (prefer64 && !use64 && bn == 128 && !d->dual && p.ksplit > 1) {
      const long long blocks64 = (long long)p.B * p.tiles_x * p.tiles_y * (p.Cout_pad / 64);
      if (blocks64 <= kNumCU) {
        p.n_tiles = p.Cout_pad / 64;
        p.ksplit = 1;
        p.ws = nullptr;
        use64 = true;
      }
    }
    if (d->dual) conv_apply_dual(p, d);
    if (tw == 16) return use64 ? launch_conv_split<3, 16, 64>(p, epi, s) : launch_conv_split<3, 16, 128>(p, epi, s);
    return use64 ? launch_conv_split<3, 32, 64>(p, epi, s) : launch_conv_split<3, 32, 128>(p, epi, s);
  }
  if (d->KS == 1) {
    // no halo: run on the flattened H*W plane
    AS_REQUIRE((long long)d->H * d->W < 2147483647ll, AS_ERR_BAD_SHAPE, "conv2d: plane too large");
    p.H = 1;
    p.W = d->H * d->W;
    p.tiles_x = as::cdiv(p.W, kBM);
    p.tiles_y = 1;
    AS_REQUIRE((long long)p.B * p.tiles_x * p.n_tiles < 2147483647ll, AS_ERR_BAD_SHAPE, "conv2d: grid too large");
    return launch_conv<1, 64>(p, epi, s);
  }
  p.H = d->H;
  p.W = d->W;
  AS_REQUIRE((long long)d->H * d->W < 2147483647ll, AS_ERR_BAD_SHAPE, "conv2d: plane too large");
  // pick the tile shape (TH x TW = 64 pixels) that wastes the fewest pixels on this image;
  // ties prefer 4x16 (64-B store runs, least halo) over 2x32 over 8x8
  const int cand[3] = {16, 32, 8};
  int best_tw = 16;
  long long best = -1;
  for (int i = 0; i < 3; ++i) {
    const int tw = cand[i], th = kBM / tw;
    const long long area = (long long)as::cdiv(p.W, tw) * tw * as::cdiv(p.H, th) * th;
    if (best < 0 || area < best) { best = area; best_tw = tw; }
  }
  p.tiles_x = as::cdiv(p.W, best_tw);
  p.tiles_y = as::cdiv(p.H, kBM / best_tw);
  AS_REQUIRE((long long)p.B * p.tiles_x * p.tiles_y * p.n_tiles < 2147483647ll, AS_ERR_BAD_SHAPE, "conv2d: grid too large");
  if (best_tw == 32) return launch_conv<3, 32>(p, epi, s);
  if (best_tw == 16) return launch_conv<3, 16>(p, epi, s);
  return launch_conv<3, 8>(p, epi, s);
}

int as_conv7x7_c1_relu(const float* x, const float* weight, const float* bias, float* out, int B, int H, int W,
                       int Cout, int out_ctot, int out_coff, int tap_major, float* copy_out, int copy_ctot, int copy_coff,
                       int copy_bs, int out_bs, void* stream) {
  AS_REQUIRE(x && weight && out, AS_ERR_BAD_ARG, "conv7x7_c1: null pointer");
  AS_REQUIRE((long long)H * W * 4 < 0x7FFFFFF0ll, AS_ERR_BAD_SHAPE, "conv7x7_c1: plane too large");
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0, AS_ERR_BAD_ARG, "conv7x7_c1: non-positive size");
  AS_REQUIRE(out_coff >= 0 && out_coff + Cout <= (out_bs ? (out_ctot + 7) / 8 * 8 : out_ctot), AS_ERR_BAD_SHAPE, "conv7x7_c1: out channel window outside out_ctot");
  AS_REQUIRE(!out_bs || (out_coff % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0), AS_ERR_BAD_ARG, "conv7x7_c1: blocked output window must start at a multiple of 8, 16-B aligned");
  AS_REQUIRE(B <= 65535 && as::cdiv(H, 16) <= 65535, AS_ERR_BAD_SHAPE, "conv7x7_c1: grid too large");
  dim3 grid((unsigned)as::cdiv(W, 16), (unsigned)as::cdiv(H, 16), (unsigned)B);
  if (tap_major) {
    const int CP = (Cout + 63) / 64 * 64;
    AS_REQUIRE((long long)B * (CP / 8) <= 65535, AS_ERR_BAD_SHAPE, "conv7x7_c1: grid too large");
    const dim3 g3(grid.x, grid.y, (unsigned)(B * (CP / 8)));
    AS_REQUIRE(!copy_out || (copy_coff >= 0 && copy_coff < copy_ctot), AS_ERR_BAD_SHAPE, "conv7x7_c1: copy channel outside copy_ctot");
    hipLaunchKernelGGL(conv7x7_c1_tm_kernel<8>, g3, dim3(256), 0, as::as_stream(stream), x, weight, bias, out, H, W, Cout, CP, out_ctot, out_coff,
                       copy_out, copy_ctot, copy_coff, copy_bs, out_bs);
  } else {
    AS_REQUIRE(!out_bs, AS_ERR_BAD_ARG, "conv7x7_c1: a blocked output needs tap_major weights");
    AS_REQUIRE(!copy_out, AS_ERR_BAD_ARG, "conv7x7_c1: the input pass-through needs tap_major weights");
    hipLaunchKernelGGL(conv7x7_c1_kernel, grid, dim3(256), 0, as::as_stream(stream), x, weight, bias, out, H, W, Cout, out_ctot, out_coff);
  }
  return as::check_launch("conv7x7_c1_relu");
}

int as_conv3x3_to1(const float* x, const float* weight, const float* bias, float* out, int B, int Cin, int H, int W, void* stream) {
  AS_REQUIRE(x && weight && out, AS_ERR_BAD_ARG, "conv3x3_to1: null pointer");
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0, AS_ERR_BAD_ARG, "conv3x3_to1: non-positive size");
  AS_REQUIRE(B <= 65535 && H <= 65535, AS_ERR_BAD_SHAPE, "conv3x3_to1: grid too large");
  dim3 grid((unsigned)as::cdiv(W, 64), (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL(conv3x3_to1_kernel, grid, dim3(256), 0, as::as_stream(stream), x, weight, bias, out, Cin, H, W);
  return as::check_launch("conv3x3_to1");
}

int as_tap_shift_sum(const float* S, const float* bias, const float* addend, float* out, int B, int H, int W, int groups, void* stream) {
  AS_REQUIRE(S && out, AS_ERR_BAD_ARG, "tap_shift_sum: null pointer");
  AS_REQUIRE(B > 0 && H > 0 && W > 0 && groups > 0, AS_ERR_BAD_ARG, "tap_shift_sum: non-positive size");
  const long long P = (long long)B * H * W;
  hipLaunchKernelGGL(tap_shift_sum_kernel, dim3((unsigned)as::cdiv64(P, 256)), dim3(256), 0, as::as_stream(stream), S, bias, addend, out, H, W, P, groups);
  return as::check_launch("tap_shift_sum");
}

int as_pool2x(const float* x, float* out, int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(x && out, AS_ERR_BAD_ARG, "pool2x: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "pool2x: non-positive size");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)B * C * Ho * Wo;
  hipLaunchKernelGGL(pool2x_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream), x, out, H, W, Ho, Wo, total);
  return as::check_launch("pool2x");
}

int as_interp_bilinear_ac(const float* x, float* out, int B, int C, int H, int W, int Ho, int Wo, void* stream) {
  AS_REQUIRE(x && out, AS_ERR_BAD_ARG, "interp: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, AS_ERR_BAD_ARG, "interp: non-positive size");
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const long long total = (long long)B * C * Ho * Wo;
  hipLaunchKernelGGL(interp_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream), x, out, H, W, Ho, Wo, sy, sx, total);
  return as::check_launch("interp_bilinear_ac");
}

/* pool2x / interp with a blocked split-fp16 result (as_conv_desc.src_bs): the resampled map only feeds convolutions */
int as_pool2x_bs(const float* x, void* out_bs, int B, int C, int H, int W, void* stream) {
  AS_REQUIRE(x && out_bs, AS_ERR_BAD_ARG, "pool2x_bs: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "pool2x_bs: non-positive size");
  AS_REQUIRE((reinterpret_cast<uintptr_t>(out_bs) & 15) == 0, AS_ERR_BAD_ARG, "pool2x_bs: out_bs not 16-B aligned");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)B * ((C + 7) / 8) * Ho * Wo;
  static const int even_mode = getenv("AS_POOL2X_EVEN") ? atoi(getenv("AS_POOL2X_EVEN")) : 1;
  if (even_mode && (W & 1) == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0 && (long long)H * W * 4 * 8 < 0x7FFFFFF0ll)
    hipLaunchKernelGGL(pool2x_bs_even_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream), x,
                       reinterpret_cast<_Float16*>(out_bs), C, H, W, Ho, Wo, total);
  else
    hipLaunchKernelGGL(pool2x_bs_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream), x,
                       reinterpret_cast<_Float16*>(out_bs), C, H, W, Ho, Wo, total);
  return as::check_launch("pool2x_bs");
}

int as_interp_bilinear_ac_bs(const float* x, void* out_bs, int B, int C, int H, int W, int Ho, int Wo, void* stream) {
  AS_REQUIRE(x && out_bs, AS_ERR_BAD_ARG, "interp_bs: null pointer");
  AS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, AS_ERR_BAD_ARG, "interp_bs: non-positive size");
  AS_REQUIRE((reinterpret_cast<uintptr_t>(out_bs) & 15) == 0, AS_ERR_BAD_ARG, "interp_bs: out_bs not 16-B aligned");
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const long long total = (long long)B * ((C + 7) / 8) * Ho * Wo;
  hipLaunchKernelGGL(interp_bs_kernel, dim3((unsigned)as::cdiv64(total, 256)), dim3(256), 0, as::as_stream(stream), x,
                     reinterpret_cast<_Float16*>(out_bs), C, H, W, Ho, Wo, sy, sx, total);
  return as::check_launch("interp_bilinear_ac_bs");
}

unsigned as_conv_split_overflow(int reset) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_split_overflow_conv), sizeof(v)) != hipSuccess) return 0xFFFFFFFFu;
  if (reset && v) {
    const unsigned z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_split_overflow_conv), &z, sizeof(z));
  }
  return v;
}

}  // extern "C"
