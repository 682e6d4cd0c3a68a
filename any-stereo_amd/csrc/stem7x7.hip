// §8 f4: the 7x7, 3 -> 64 channel stem of the context / feature encoders (extractor.py:127 `conv1`, stride 1 at
// n_downsample = 2) as a split-precision MFMA implicit GEMM — the layer that stayed on MIOpen in round 1 (igemm + two layout
// transposes + separate bias and ReLU passes: ~290 us at 544x960; here one launch).
//
// K = 3 channels x 7 kernel rows = 21 "rows" of 8 taps (7 + one zero weight), two rows per 16-wide MFMA k-step -> 11 k-steps
// (the 22nd row is zero).  The B operand of a k-step is, per pixel, 8 CONSECUTIVE input values of one image row — so the block
// expands its halo patch once into an "im2row" LDS image [comp][patch row][pixel][8 halves] (hi / lo fp16 of x = hi + lo/2048,
// like conv.hip) and every operand fetch is one aligned, conflict-free ds_read_b128.  The whole weight set (45 KB of fragments)
// stays in LDS; blocks are persistent over 8 x 16 pixel tiles, two blocks per CU.
#include "common.h"

namespace {

using half8 = _Float16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

constexpr int kTH = 8, kTW = 16;                 // pixel tile
constexpr int kPH = kTH + 6, kPWraw = kTW + 6;   // halo patch rows / raw columns
constexpr int kRows = 3 * kPH;                   // patch rows over the three channels
constexpr int kSteps = 11;                       // k-steps of 16 (22 rows of 8)
constexpr int kWBytes = kSteps * 2 * 2 * 64 * 16;   // [step][comp][k-half][co 64][8 halves]
constexpr int kIBytes = 2 * kRows * kTW * 16;       // [comp][patch row][px][8 halves]
constexpr int kRawFloats = kRows * kPWraw;

struct StemParams {
  const float* x;
  const unsigned char* wpack;
  const float* bias;
  float* out;
  int B, H, W, tiles_x, tiles_y, act;
};

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case AS_ACT_RELU: return fmaxf(v, 0.f);
    case AS_ACT_LEAKY: return v >= 0.f ? v : 0.01f * v;
    default: return v;
  }
}

// fragments of W [64][3][7][7] (row-major fp32): k = row*8 + kx, row = ci*7 + ky; A operand of lane (co, k-half) = 8 k values
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ pack) {
  const int t = blockIdx.x * 256 + threadIdx.x;  // (step, half, co, j)
  if (t >= kSteps * 2 * 64 * 8) return;
  const int j = t & 7, co = (t >> 3) & 63, half = (t >> 9) & 1, step = t >> 10;
  const int row = 2 * step + half;
  float v = 0.f;
  if (row < 21 && j < 7) v = w[(co * 3 + row / 7) * 49 + (row % 7) * 7 + j];
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)((v - (float)hi) * 2048.f);
  pack[(((step * 2 + 0) * 2 + half) * 64 + co) * 8 + j] = hi;
  pack[(((step * 2 + 1) * 2 + half) * 64 + co) * 8 + j] = lo;
}

__global__ __launch_bounds__(256, 2) void stem7x7_kernel(StemParams p) {
  as::fp16_saturate_mode();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* wl = lds;                       // weight fragments
  unsigned char* il = lds + kWBytes;             // im2row image
  float* raw = reinterpret_cast<float*>(lds + kWBytes + kIBytes);  // raw halo patch [kRows][kPWraw]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  for (int i = tid; i < kWBytes / 16; i += 256)
    reinterpret_cast<uint4*>(wl)[i] = reinterpret_cast<const uint4*>(p.wpack)[i];
  // this lane's pixel of the tile: wave w owns pixels 32 w .. 32 w + 31 = tile rows 2 w, 2 w + 1
  const int py = 2 * wave + (l31 >> 4), px = l31 & 15;
  const long long plane = (long long)p.H * p.W;
  const int ntile = p.B * p.tiles_y * p.tiles_x;
  // the lane's 32 output channels' bias, once per (persistent) block
  float bias_r[2][16];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[c][r] = p.bias ? p.bias[c * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.f;
  // raw halo patch: kRawFloats = 924 values, <= 4 per thread, fetched ONE TILE AHEAD into registers (the global latency then
  // hides behind the previous tile's MFMAs and stores instead of standing between two barriers)
  constexpr int NRAW = (kRawFloats + 255) / 256;
  int r_pr[NRAW], r_c[NRAW];
#pragma unroll
  for (int k = 0; k < NRAW; ++k) {
    const int i = tid + k * 256;
    r_pr[k] = i < kRawFloats ? i / kPWraw : -1;
    r_c[k] = i - (i / kPWraw) * kPWraw;
  }
  float rv[NRAW];
  auto fetch = [&](int tile_) {
    const int tx_ = tile_ % p.tiles_x, ty_ = (tile_ / p.tiles_x) % p.tiles_y, b_ = tile_ / (p.tiles_x * p.tiles_y);
    const float* xb_ = p.x + (long long)b_ * 3 * plane;
#pragma unroll
    for (int k = 0; k < NRAW; ++k) {
      const int ci = r_pr[k] / kPH, r = r_pr[k] - ci * kPH;
      const int gy = ty_ * kTH + r - 3, gx = tx_ * kTW + r_c[k] - 3;
      rv[k] = (r_pr[k] >= 0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? xb_[ci * plane + (long long)gy * p.W + gx] : 0.f;
    }
  };
  if ((int)blockIdx.x < ntile) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const int y0 = ty * kTH, x0 = tx * kTW;
    __syncthreads();  // the previous tile's operand reads are done (and, first trip, nothing yet)
#pragma unroll
    for (int k = 0; k < NRAW; ++k)
      if (r_pr[k] >= 0) raw[tid + k * 256] = rv[k];
    __syncthreads();
    if (tile + (int)gridDim.x < ntile) fetch(tile + gridDim.x);
    for (int e = tid; e < kRows * kTW; e += 256) {  // entry (patch row, px): raw[pr][px .. px+7] split into hi / lo
      const int pr = e / kTW, ex = e - pr * kTW;
      const float* rp = raw + pr * kPWraw + ex;
      half8 h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = j < 7 ? rp[j] : 0.f;  // the 8th tap meets a zero weight; keep it finite
        const _Float16 hj = (_Float16)v;
        h[j] = hj;
        l[j] = (_Float16)((v - (float)hj) * 2048.f);
      }
      *reinterpret_cast<half8*>(il + e * 16) = h;
      *reinterpret_cast<half8*>(il + (kRows * kTW + e) * 16) = l;
    }
    __syncthreads();
    f32x16 acc_h[2], acc_x[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_h[c][r] = 0.f; acc_x[c][r] = 0.f; }
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      // rows 2s (k-half 0) and 2s+1 (k-half 1): patch row = ci*kPH + py + ky; the zero 22nd row reads row 0
      constexpr int dummy = 0;
      const int rA = 2 * s, rB = 2 * s + 1;
      const int prA = (rA / 7) * kPH + (rA % 7);
      const int prB = rB < 21 ? (rB / 7) * kPH + (rB % 7) : dummy;
      const int pr = py + (half ? prB : prA);
      const half8 b_hi = *reinterpret_cast<const half8*>(il + (pr * kTW + px) * 16);
      const half8 b_lo = *reinterpret_cast<const half8*>(il + ((kRows + pr) * kTW + px) * 16);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const half8 a_hi = *reinterpret_cast<const half8*>(wl + ((((s * 2 + 0) * 2 + half) * 64) + c * 32 + l31) * 16);
        const half8 a_lo = *reinterpret_cast<const half8*>(wl + ((((s * 2 + 1) * 2 + half) * 64) + c * 32 + l31) * 16);
        acc_h[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc_h[c], 0, 0, 0);
        acc_x[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc_x[c], 0, 0, 0);
        acc_x[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc_x[c], 0, 0, 0);
      }
    }
    // D layout of the 32x32 tile: register r of lane (l31, half) = row (co) (r & 3) + 8 (r >> 2) + 4 half, column (pixel) l31
    const int gy = y0 + py, gx = x0 + px;
    if (gy < p.H && gx < p.W) {
      float* op = p.out + (long long)b * 64 * plane + (long long)gy * p.W + gx;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const float v = acc_h[c][r] + acc_x[c][r] * (1.f / 2048.f) + bias_r[c][r];
          op[co * plane] = act_apply(v, p.act);
        }
    }
  }
}

}  // namespace

extern "C" {

long long as_conv7x7_c3_pack_bytes(void) { return kWBytes; }

int as_conv7x7_c3_pack(const float* weight, void* wpack, void* stream) {
  AS_REQUIRE(weight && wpack, AS_ERR_BAD_ARG, "conv7x7_c3_pack: null pointer");
  hipLaunchKernelGGL(stem_pack_kernel, dim3(as::cdiv(kSteps * 2 * 64 * 8, 256)), dim3(256), 0, as::as_stream(stream), weight,
                     reinterpret_cast<_Float16*>(wpack));
  return as::check_launch("conv7x7_c3_pack");
}

int as_conv7x7_c3(const float* x, const void* wpack, const float* bias, float* out, int B, int H, int W, int act, void* stream) {
  AS_REQUIRE(x && wpack && out, AS_ERR_BAD_ARG, "conv7x7_c3: null pointer");
  AS_REQUIRE(B > 0 && H > 0 && W > 0, AS_ERR_BAD_ARG, "conv7x7_c3: non-positive size");
  AS_REQUIRE(act == AS_ACT_NONE || act == AS_ACT_RELU || act == AS_ACT_LEAKY, AS_ERR_BAD_ARG, "conv7x7_c3: act=%d", act);
  AS_REQUIRE(as::use_split_precision(), AS_ERR_BAD_ARG, "conv7x7_c3: split-precision mode only (the caller keeps the library path in fp32 mode)");
  StemParams p{};
  p.x = x; p.wpack = static_cast<const unsigned char*>(wpack); p.bias = bias; p.out = out;
  p.B = B; p.H = H; p.W = W; p.act = act;
  p.tiles_x = as::cdiv(W, kTW);
  p.tiles_y = as::cdiv(H, kTH);
  const long long ntile = (long long)B * p.tiles_x * p.tiles_y;
  AS_REQUIRE(ntile < 2147483647ll, AS_ERR_BAD_SHAPE, "conv7x7_c3: too many tiles");
  constexpr size_t lds = kWBytes + kIBytes + kRawFloats * 4;
  static_assert(lds <= 80 * 1024, "stem7x7: two blocks per CU");
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stem7x7_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return as::fail(AS_ERR_LAUNCH, "conv7x7_c3: LDS attribute: %s", hipGetErrorString(e));
    configured = true;
  }
  const int grid = (int)(ntile < 512 ? ntile : 512);  // persistent: 2 blocks per CU, each walks tiles grid apart
  hipLaunchKernelGGL(stem7x7_kernel, dim3(grid), dim3(256), lds, as::as_stream(stream), p);
  return as::check_launch("conv7x7_c3");
}

}  // extern "C"
