// One launch per MobileNetV2 inverted-residual block of the feature trunk (extractor.py:327-342 = timm's mobilenetv2_100
// `InvertedResidual`: 1x1 expand -> BN -> ReLU6 -> depthwise 3x3 (stride 1|2) -> BN -> ReLU6 -> 1x1 project -> BN [+ x]), inference:
// the 6x-expanded tensor never reaches HBM.  SURVEY.md §8 f4 (backbone one-shot operators); round-6 review item 1a.
//
// Block = 256 threads = 4 waves, one output tile of TH x TW pixels (8x8 at stride 1, 4x8 at stride 2) of one batch element:
//   1. the input tile with the depthwise conv's halo ((TH-1)S+3 x (TW-1)S+3 pixels, all Cin channels) is read once, split into
//      fp16 hi / lo parts (x = hi + lo/2048, the library's split precision) and parked PIXEL-major in LDS (a lane's MFMA B
//      fragment = 8 consecutive channels of its pixel = one 16-B read);
//   2. per chunk of 32 expanded channels:  expand GEMM [32 x Cin] x [Cin x tile+halo] on the matrix cores (3 fp16 MFMAs per
//      product: hi.hi, hi.lo, lo.hi; fp32 accumulate), + bias, ReLU6, ZERO outside the image (the depthwise conv pads the EXPANDED
//      tensor) -> fp32 LDS tile;  depthwise 3x3 + bias + ReLU6 on the vector ALU -> split again, pixel-major;  project GEMM
//      [Cout x 32] x [32 x tile] accumulated over the chunks in registers;
//   3. + bias (+ the fp32 residual x) -> out.
// Weights arrive as MFMA A fragments (as_ir_block_pack_*: built once per weight version by the host wrapper, ops.IrBlockPack)
// and are read from global memory (a chunk's fragments are shared by every block: L2-resident).
#include "common.h"

namespace {

using f32x16 = float __attribute__((ext_vector_type(16)));
using half8 = _Float16 __attribute__((ext_vector_type(8)));

struct IrParams {
  const float* x;       // [B, Cin, H, W]
  float* out;           // [B, Cout, Ho, Wo]
  const half8* w1;      // expand fragments   [nch][nks1][hi|lo][64 lanes]  (8 halfs each)
  const float* b1;      // [mid_pad]
  const float* wd;      // depthwise taps     [mid_pad][9]
  const float* b2;      // [mid_pad]
  const half8* w3;      // project fragments  [nch][nrt][2 k-steps][hi|lo][64 lanes]
  const float* b3;      // [Cout]
  int B, Cin, Cin_pad, mid_pad, Cout, H, W, Ho, Wo, residual;
  int nch, nks1, nrt, tiles_x, tiles_y;
};

__device__ __forceinline__ int acc_row(int i, int half) { return (i & 3) + 8 * (i >> 2) + 4 * half; }
__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

template <int S>
struct IrGeom {
  static constexpr int TH = S == 1 ? 8 : 4, TW = 8;
  static constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;  // input tile with halo
  static constexpr int PIN = PH * PW;                                   // 100 | 153
  static constexpr int NP = (PIN + 31) / 32 * 32;                       // 128 | 160: MFMA column tiles of the expand GEMM
  static constexpr int NCT = NP / 32;                                   // 4 | 5
  static constexpr int OP = TH * TW;                                    // 64 | 32 output pixels
  static constexpr int OCT = OP / 32;                                   // 2 | 1
  static constexpr int MP = NP + 1;                                     // pitch (floats) of the fp32 expanded tile: odd -> no bank conflicts over channels
  static constexpr int DP = 40;                                         // pitch (halfs) of the depthwise result rows: 32 channels + 8
};

// dynamic LDS: xs_hi | xs_lo [NP][XP] halfs, XP = Cin_pad + 8;  mid [32][MP] floats (MP*32*4 is a multiple of 16);  dwo_hi | dwo_lo
// [OP][DP] halfs;  w1s: a chunk's expand fragments [nks1][hi|lo][64] x 16 B;  b1 | b2 [mid_pad] floats
template <int S>
__host__ __device__ constexpr size_t ir_lds_bytes(int cin_pad, int mid_pad) {
  using G = IrGeom<S>;
  return (size_t)2 * G::NP * (cin_pad + 8) * 2 + (size_t)32 * G::MP * 4 + (size_t)2 * G::OP * G::DP * 2 + (size_t)(cin_pad / 16) * 2 * 64 * 16 +
         (size_t)2 * mid_pad * 4;
}

template <int S>
__global__ __launch_bounds__(256) void ir_block_kernel(IrParams p) {
  using G = IrGeom<S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  as::fp16_saturate_mode();
  const int XP = p.Cin_pad + 8;
  _Float16* xs_hi = reinterpret_cast<_Float16*>(smem);
  _Float16* xs_lo = xs_hi + (size_t)G::NP * XP;
  float* mid = reinterpret_cast<float*>(xs_lo + (size_t)G::NP * XP);
  _Float16* dwo_hi = reinterpret_cast<_Float16*>(mid + 32 * G::MP);
  _Float16* dwo_lo = dwo_hi + G::OP * G::DP;
  half8* w1s = reinterpret_cast<half8*>(dwo_lo + G::OP * G::DP);
  float* b1s = reinterpret_cast<float*>(w1s + p.nks1 * 2 * 64);
  float* b2s = b1s + p.mid_pad;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x;
  bid /= p.tiles_x;
  const int ty = bid % p.tiles_y, b = bid / p.tiles_y;
  const int oy0 = ty * G::TH, ox0 = tx * G::TW;        // output tile origin
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;      // input tile origin (with the halo)
  const long long plane = (long long)p.H * p.W;
  const float* __restrict__ xb = p.x + (long long)b * p.Cin * plane;

  // ---- 1. input tile -> split fp16, pixel-major (q fastest over the threads: runs of PW pixels of one channel row).  Eight
  //         loads per thread are in flight before the first is consumed (one exposed round trip per 2048 elements, not per 256)
  {
    const int total = p.Cin_pad * G::NP;
    for (int base = tid; base < total; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = base + 256 * k;
        const int c = idx / G::NP, q = idx - c * G::NP;
        const int qy = q / G::PW, qx = q - qy * G::PW;
        const int gy = iy0 + qy, gx = ix0 + qx;
        const bool ok = idx < total && q < G::PIN && c < p.Cin && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        v[k] = ok ? xb[(long long)c * plane + (long long)gy * p.W + gx] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = base + 256 * k;
        if (idx < total) {
          const int c = idx / G::NP, q = idx - c * G::NP;
          const _Float16 h = (_Float16)v[k];
          xs_hi[q * XP + c] = h;
          xs_lo[q * XP + c] = (_Float16)((v[k] - (float)h) * 2048.f);
        }
      }
    }
  }
  for (int i = tid; i < p.mid_pad; i += 256) { b1s[i] = p.b1[i]; b2s[i] = p.b2[i]; }

  // this lane's expand columns (pixels of the input tile): inside the image?  (the expanded tensor is zero-padded, not expand(0))
  bool col_in[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = (wave + 4 * u) * 32 + l31;
    const int qy = q / G::PW, qx = q - qy * G::PW;
    const int gy = iy0 + qy, gx = ix0 + qx;
    col_in[u] = q < G::PIN && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
  }

  // project accumulators: (row tile, output column tile) pairs wave, wave + 4, wave + 8 of nrt * OCT (<= 10)
  const int npair = p.nrt * G::OCT;
  f32x16 ph[3], px[3];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int i = 0; i < 16; ++i) { ph[u][i] = 0.f; px[u][i] = 0.f; }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // a chunk's expand fragments are shared by the four waves: staged through LDS, the NEXT chunk's in flight (registers) while
  // this chunk computes.  nw1 16-B units per chunk (<= 16 x 2 x 64 = 2048): at most 8 per thread
  const int nw1 = p.nks1 * 2 * 64;
  half8 w1n[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (tid + 256 * k < nw1) w1s[tid + 256 * k] = p.w1[tid + 256 * k];
  __syncthreads();

  for (int ch = 0; ch < p.nch; ++ch) {
    if (ch + 1 < p.nch) {
      const half8* __restrict__ nx = p.w1 + (long long)(ch + 1) * nw1;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (tid + 256 * k < nw1) w1n[k] = nx[tid + 256 * k];
    }
    // ---- 2a. expand: mid[32][NP] = relu6(W1[ch] . x + b1), zero outside the image ----
    {
      f32x16 eh[2] = {zero16, zero16}, ex[2] = {zero16, zero16};
      for (int ks = 0; ks < p.nks1; ++ks) {
        const half8 ah = w1s[(ks * 2 + 0) * 64 + lane], al = w1s[(ks * 2 + 1) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int ct = wave + 4 * u;  // wave-uniform
          if (ct < G::NCT) {
            const int off = (ct * 32 + l31) * XP + ks * 16 + 8 * half;
            const half8 bh = *reinterpret_cast<const half8*>(xs_hi + off), bl = *reinterpret_cast<const half8*>(xs_lo + off);
            eh[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, eh[u], 0, 0, 0);
            ex[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ex[u], 0, 0, 0);
            ex[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, ex[u], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ct = wave + 4 * u;
        if (ct < G::NCT) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int r = acc_row(i, half);
            const float v = relu6f(fmaf(ex[u][i], 1.f / 2048.f, eh[u][i]) + b1s[ch * 32 + r]);
            mid[r * G::MP + ct * 32 + l31] = col_in[u] ? v : 0.f;
          }
        }
      }
    }
    __syncthreads();
    // this wave's project fragments of the chunk: requested now, consumed behind the depthwise stage
    half8 w3h[3][2], w3l[3][2];
    {
      const half8* __restrict__ w3c = p.w3 + (long long)ch * p.nrt * 4 * 64;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int pr = wave + 4 * u;
        if (pr < npair) {
          const int rt = pr / G::OCT;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            w3h[u][ks] = w3c[((rt * 2 + ks) * 2 + 0) * 64 + lane];
            w3l[u][ks] = w3c[((rt * 2 + ks) * 2 + 1) * 64 + lane];
          }
        }
      }
    }
    // ---- 2b. depthwise 3x3 + bias + ReLU6 -> split fp16, pixel-major [OP][32] ----
    {
      const int c = tid & 31, g = tid >> 5;  // channel of the chunk, pixel group (8 groups)
      const float* __restrict__ wdc = p.wd + (long long)(ch * 32 + c) * 9;
      float wt[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) wt[t] = wdc[t];
      const float bias = b2s[ch * 32 + c];
      const float* mc = mid + c * G::MP;
#pragma unroll
      for (int k = 0; k < G::OP / 8; ++k) {
        const int o = g + 8 * k;
        const int py = o / G::TW, pxx = o - py * G::TW;
        const float* m0 = mc + (py * S) * G::PW + pxx * S;
        float a = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) a = fmaf(wt[ky * 3 + kx], m0[ky * G::PW + kx], a);
        const float v = relu6f(a + bias);
        const _Float16 h = (_Float16)v;
        dwo_hi[o * G::DP + c] = h;
        dwo_lo[o * G::DP + c] = (_Float16)((v - (float)h) * 2048.f);
      }
    }
    // the next chunk's expand fragments -> LDS (every wave finished this chunk's expand before the barrier above)
    if (ch + 1 < p.nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (tid + 256 * k < nw1) w1s[tid + 256 * k] = w1n[k];
    }
    __syncthreads();
    // ---- 2c. project: acc[Cout x OP] += W3[:, chunk] . dw ----
    {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int pr = wave + 4 * u;  // wave-uniform
        if (pr < npair) {
          const int oc = pr % G::OCT;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const half8 ah = w3h[u][ks], al = w3l[u][ks];
            const int off = (oc * 32 + l31) * G::DP + ks * 16 + 8 * half;
            const half8 bh = *reinterpret_cast<const half8*>(dwo_hi + off), bl = *reinterpret_cast<const half8*>(dwo_lo + off);
            ph[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, ph[u], 0, 0, 0);
            px[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, px[u], 0, 0, 0);
            px[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, px[u], 0, 0, 0);
          }
        }
      }
    }
    // the next chunk's expand writes `mid` (its readers passed the barrier above); its depthwise stage writes `dwo` behind the
    // barrier that follows the expand, which every wave reaches only after these reads
  }

  // ---- 3. + bias (+ residual) -> out ----
  const long long oplane = (long long)p.Ho * p.Wo;
  float* __restrict__ ob = p.out + (long long)b * p.Cout * oplane;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int pr = wave + 4 * u;
    if (pr < npair) {
      const int rt = pr / G::OCT, oc = pr - rt * G::OCT;
      const int o = oc * 32 + l31;
      const int oy = oy0 + o / G::TW, ox = ox0 + o % G::TW;
      if (oy < p.Ho && ox < p.Wo) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = rt * 32 + acc_row(i, half);
          if (co < p.Cout) {
            float v = fmaf(px[u][i], 1.f / 2048.f, ph[u][i]) + p.b3[co];
            if (p.residual) v += xb[(long long)co * plane + (long long)oy * p.W + ox];
            ob[(long long)co * oplane + (long long)oy * p.Wo + ox] = v;
          }
        }
      }
    }
  }
}

// weight [rows][cols] fp32 (row stride ld) -> A fragments of 32-row x 16-column MFMA steps, hi and lo parts:
// out[((rt * nks + ks) * 2 + hl) * 64 + lane][j] = part(w[rt * 32 + lane % 32][ks * 16 + 8 * (lane / 32) + j]), zero outside
__global__ __launch_bounds__(256) void ir_pack_frag_kernel(const float* __restrict__ w, int rows, int cols, int ld, int nrt, int nks, int chunk_major,
                                                           int k_per_chunk, _Float16* __restrict__ out, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63), hl = (int)((idx >> 9) & 1);
  long long blk = idx >> 10;  // (rt, ks) block index in the OUTPUT order
  int rt, ks;
  if (chunk_major) {  // project: [chunk][rt][ks_in_chunk]: ks = chunk * k_per_chunk + ks_in_chunk
    const int ksc = (int)(blk % k_per_chunk);
    blk /= k_per_chunk;
    rt = (int)(blk % nrt);
    ks = (int)(blk / nrt) * k_per_chunk + ksc;
  } else {            // expand: [rt][ks]
    ks = (int)(blk % nks);
    rt = (int)(blk / nks);
  }
  const int r = rt * 32 + (lane & 31), c = ks * 16 + 8 * (lane >> 5) + j;
  float v = (r < rows && c < cols) ? w[(long long)r * ld + c] : 0.f;
  v = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  const _Float16 h = (_Float16)v;
  out[idx] = hl == 0 ? h : (_Float16)((v - (float)h) * 2048.f);
}

}  // namespace

extern "C" {

int64_t as_ir_block_pack_bytes(int Cin, int mid, int Cout) {
  if (Cin <= 0 || mid <= 0 || Cout <= 0) return -1;
  const int64_t cin_pad = (Cin + 15) / 16 * 16, mid_pad = (mid + 31) / 32 * 32, nrt = (Cout + 31) / 32;
  const int64_t w1 = (mid_pad / 32) * (cin_pad / 16) * 2 * 64 * 8 * 2;    // bytes
  const int64_t w3 = (mid_pad / 32) * nrt * 2 * 2 * 64 * 8 * 2;
  return w1 + w3;
}

/* w1 [mid][Cin], w3 [Cout][mid] (BatchNorm folded by the caller) -> MFMA fragments */
int as_ir_block_pack(const float* w1, const float* w3, int Cin, int mid, int Cout, void* pack, void* stream) {
  AS_REQUIRE(w1 && w3 && pack, AS_ERR_BAD_ARG, "ir_block_pack: null pointer");
  AS_REQUIRE(Cin > 0 && mid > 0 && Cout > 0 && Cout <= 160 && Cin <= 256, AS_ERR_BAD_SHAPE, "ir_block_pack: Cin=%d mid=%d Cout=%d", Cin, mid, Cout);
  AS_REQUIRE((reinterpret_cast<uintptr_t>(pack) & 15) == 0, AS_ERR_BAD_ARG, "ir_block_pack: pack not 16-B aligned");
  const int cin_pad = (Cin + 15) / 16 * 16, mid_pad = (mid + 31) / 32 * 32, nrt = (Cout + 31) / 32, nch = mid_pad / 32, nks1 = cin_pad / 16;
  hipStream_t s = as::as_stream(stream);
  char* base = static_cast<char*>(pack);
  const long long n1 = (long long)nch * nks1 * 2 * 64 * 8, n3 = (long long)nch * nrt * 2 * 2 * 64 * 8;
  _Float16* f1 = reinterpret_cast<_Float16*>(base);
  _Float16* f3 = f1 + n1;
  // expand: rows = mid channels (row tile = chunk), cols = Cin
  hipLaunchKernelGGL(ir_pack_frag_kernel, dim3((unsigned)as::cdiv64(n1, 256)), dim3(256), 0, s, w1, mid, Cin, Cin, nch, nks1, 0, 1, f1, n1);
  // project: rows = Cout, cols = mid; output order [chunk][rt][2 k-steps]
  hipLaunchKernelGGL(ir_pack_frag_kernel, dim3((unsigned)as::cdiv64(n3, 256)), dim3(256), 0, s, w3, Cout, mid, mid, nrt, 2 * nch, 1, 2, f3, n3);
  return as::check_launch("ir_block_pack");
}

/* fparams [11 * mid_pad + Cout] fp32: b1 [mid_pad] | b2 [mid_pad] | wd [mid_pad][9] | b3 [Cout], mid_pad = ceil(mid / 32) * 32, padding zero */
int as_ir_block(const float* x, const void* pack, const float* fparams, float* out, int B, int Cin, int mid, int Cout, int H, int W, int stride,
                int residual, void* stream) {
  AS_REQUIRE(x && pack && fparams && out, AS_ERR_BAD_ARG, "ir_block: null pointer");
  AS_REQUIRE((reinterpret_cast<uintptr_t>(pack) & 15) == 0, AS_ERR_BAD_ARG, "ir_block: pack not 16-B aligned");
  AS_REQUIRE(B > 0 && Cin > 0 && mid > 0 && Cout > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "ir_block: non-positive size");
  AS_REQUIRE(stride == 1 || stride == 2, AS_ERR_BAD_ARG, "ir_block: stride=%d", stride);
  AS_REQUIRE(Cout <= 160 && Cin <= 256, AS_ERR_BAD_SHAPE, "ir_block: Cin=%d Cout=%d (<= 256 / <= 160)", Cin, Cout);
  AS_REQUIRE(!residual || (stride == 1 && Cin == Cout), AS_ERR_BAD_ARG, "ir_block: a residual needs stride 1 and Cin == Cout");
  AS_REQUIRE((long long)H * W < 2147483647ll, AS_ERR_BAD_SHAPE, "ir_block: plane too large");
  IrParams p{};
  p.x = x; p.out = out; p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.residual = residual ? 1 : 0;
  p.Cin_pad = (Cin + 15) / 16 * 16; p.mid_pad = (mid + 31) / 32 * 32; p.nrt = (Cout + 31) / 32; p.nch = p.mid_pad / 32; p.nks1 = p.Cin_pad / 16;
  p.Ho = (H - 1) / stride + 1; p.Wo = (W - 1) / stride + 1;
  const char* base = static_cast<const char*>(pack);
  const long long n1 = (long long)p.nch * p.nks1 * 2 * 64 * 8, n3 = (long long)p.nch * p.nrt * 2 * 2 * 64 * 8;
  p.w1 = reinterpret_cast<const half8*>(base);
  p.w3 = reinterpret_cast<const half8*>(base + n1 * 2);
  (void)n3;
  const float* fb = fparams;
  p.b1 = fb; p.b2 = fb + p.mid_pad; p.wd = fb + 2 * p.mid_pad; p.b3 = fb + 11 * p.mid_pad;
  hipStream_t s = as::as_stream(stream);
  if (stride == 1) {
    using G = IrGeom<1>;
    p.tiles_x = as::cdiv(p.Wo, G::TW); p.tiles_y = as::cdiv(p.Ho, G::TH);
    const size_t lds = ir_lds_bytes<1>(p.Cin_pad, p.mid_pad);
    AS_REQUIRE(lds <= 160 * 1024, AS_ERR_BAD_SHAPE, "ir_block: LDS plan of %zu bytes", lds);
    AS_REQUIRE((long long)B * p.tiles_x * p.tiles_y < 2147483647ll, AS_ERR_BAD_SHAPE, "ir_block: grid too large");
    as::lds_opt_in(reinterpret_cast<const void*>(ir_block_kernel<1>));
    hipLaunchKernelGGL(ir_block_kernel<1>, dim3((unsigned)((long long)B * p.tiles_x * p.tiles_y)), dim3(256), lds, s, p);
  } else {
    using G = IrGeom<2>;
    p.tiles_x = as::cdiv(p.Wo, G::TW); p.tiles_y = as::cdiv(p.Ho, G::TH);
    const size_t lds = ir_lds_bytes<2>(p.Cin_pad, p.mid_pad);
    AS_REQUIRE(lds <= 160 * 1024, AS_ERR_BAD_SHAPE, "ir_block: LDS plan of %zu bytes", lds);
    AS_REQUIRE((long long)B * p.tiles_x * p.tiles_y < 2147483647ll, AS_ERR_BAD_SHAPE, "ir_block: grid too large");
    as::lds_opt_in(reinterpret_cast<const void*>(ir_block_kernel<2>));
    hipLaunchKernelGGL(ir_block_kernel<2>, dim3((unsigned)((long long)B * p.tiles_x * p.tiles_y)), dim3(256), lds, s, p);
  }
  return as::check_launch("ir_block");
}

}  // extern "C"
