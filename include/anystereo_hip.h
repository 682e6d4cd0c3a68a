/*
 * anystereo_hip.h — C ABI of libanystereo_hip.so, the MI355X (gfx950) implementation of
 * Any-Stereo's data-parallel hot path.
 *
 * Conventions
 *   - plain C: device pointers + sizes + a stream; no torch / C++ types cross this boundary.
 *   - every pointer is a DEVICE pointer to fp32 data unless stated otherwise; tensors are dense,
 *     row-major in the index order written in the comment.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Launchers never
 *     allocate, synchronise or touch global state, so they are re-entrant and safe under
 *     hipStreamBeginCapture (hipGraph).
 *   - return value: AS_OK (0) or a negative AS_ERR_* code; as_last_error_string() describes the
 *     last failure on the calling thread.  Shapes are validated on the host BEFORE any launch
 *     (the reference performs no shape checks: sampler/sampler.cpp:20-22 checks device+contiguity only).
 *   - outputs are caller-allocated and fully overwritten (no zero-init requirement; the reference
 *     allocates zeros and accumulates, sampler/sampler_kernel.cu:122-124,148).
 *
 * "replaces" = the reference interface (file:line under /root/reference) each entry point stands in for.
 */
#ifndef ANYSTEREO_HIP_H
#define ANYSTEREO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AS_OK 0
#define AS_ERR_BAD_ARG (-1)    /* null pointer / non-positive size / unsupported combination */
#define AS_ERR_BAD_SHAPE (-2)  /* sizes inconsistent with each other or above the supported maximum */
#define AS_ERR_LAUNCH (-3)     /* hipGetLastError() != hipSuccess after the launch */
#define AS_ERR_NO_DEVICE (-4)

#define AS_MAX_LEVELS 4
#define AS_MAX_SRCS 4

/* dtype codes for as_corr_sampler_* (the reference dispatches f64/f32/f16: sampler_kernel.cu:126) */
#define AS_F32 0
#define AS_F16 1
#define AS_F64 2

/* activation / epilogue codes for as_conv2d */
#define AS_ACT_NONE 0
#define AS_ACT_RELU 1
#define AS_ACT_SIGMOID 2
#define AS_ACT_TANH 3
#define AS_ACT_RELU6 4 /* min(max(x,0),6): MobileNetV2 blocks of the feature net (extractor.py:331-342) */
#define AS_ACT_LEAKY 5 /* LeakyReLU(0.01): BasicConv / BasicConv_IN (submodule.py:6-33)                 */
#define AS_ACT_GELU 6  /* exact (erf) GELU: HighRes_Aggregation_LN_GeLU head (submodule.py:233-252); norm kernels only */
#define AS_EPI_LINEAR 0 /* out = act(acc + bias + add); with h != NULL: out = relu(h + act(...))    (extractor.py:56-62) */
#define AS_EPI_GRU_ZR 1 /* co <  Cout/2: z  = sigmoid(acc+bias+add)        -> out  [B,Cout/2,H,W]
                           co >= Cout/2: rh = sigmoid(acc+bias+add) * h    -> out2 [B,Cout/2,H,W]  */
#define AS_EPI_RELU_TAPS 4 /* act(acc+bias) is not stored: per 64-channel tile g and tap t, out[b][g*9+t] = sum_{c in tile} tap_w[c][t] * act(.)[c]
                              — the channel reductions of a FOLLOWING 3x3, Cout -> 1 convolution (DispHead: conv1 -> relu -> conv2,
                              update.py:23-24), finished by as_tap_shift_sum(groups = ceil(Cout/64)).  3x3, precision 1, stride 1. */
#define AS_EPI_GRU_Q 2  /* out = (1-z)*h + z*tanh(acc+bias+add)            (update.py:39-40)      */

/* Matrix-core arithmetic of the GEMM-shaped kernels (as_corr_build_pyramid, as_conv2d):
 *   0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32),
 *   1 = split precision, 3 x fp16 MFMA per product (operand = hi + lo/2048, fp32 accumulation; relative
 *       error ~2^-22 per product, requires |x| < 65504).  Process-wide; initial value from the environment
 *       variable ANYSTEREO_PRECISION ("fp32" -> 0, otherwise 1). */
/* Reduced-precision variant of mode 1 for the convolution kernels: plain fp16 operands (the hi parts of the split), fp32
 * accumulate, ONE MFMA per product — the counterpart of the reference's autocast path (`autocast(enabled=args.mixed_precision)`,
 * continuous_IGEVstereo.py:287; evaluation.py:558,644).  Process-wide; own, looser tolerance. */
int as_set_fast16(int on);
int as_get_fast16(void);
int as_set_precision(int mode);
int as_get_precision(void);

const char* as_last_error_string(void);
int as_abi_version(void);        /* bumped on any signature change */

/* Post-capture surgery on a captured hipGraph (the training step as one graph, harness/train.py): every memset node — ATen's
 * reduction semaphores, MIOpen's split-K zero-initialisation — is replaced by a fill KERNEL node with the same edges (memset nodes
 * inside a ~5 000-node chain are not reliably ordered with their neighbours on this ROCm stack).  graph: the
 * hipGraph_t before (re-)instantiation; *replaced / *left (optional) count the converted nodes and the ones left alone. */
int as_graph_replace_memsets(void* graph, int* replaced, int* left);

/* Timeline marker (measurement, no reference counterpart: the reference's only timing is the host perf_counter pair of
 * evaluation.py:248-250): buf[slot] = device wall clock (constant 100 MHz, 10 ns ticks) when `stream` reaches this launch.
 * buf: device memory, >= slot + 1 unsigned 64-bit words.  Capturable: a marker is an ordinary kernel node of the forward's graph. */
int as_stamp(unsigned long long* buf, int slot, void* stream);
/* 16 hex digits: sha256 over the .hip / .h files of csrc and this header at build time (any-stereo_amd/build.py); the Python binding
 * recomputes it from the tree and refuses a library built from other sources */
const char* as_source_hash(void);
int as_device_count(void);       /* hipGetDeviceCount; 0 on a CPU-only host */

/* ---------------------------------------------------------------------------------------------
 * a18  corr_sampler — replaces sampler/sampler.cpp:24-45 (`corr_sampler.forward/backward`),
 *      kernels sampler/sampler_kernel.cu:19-60 and :63-104.
 *   volume [N,H1,W1,W2] (dtype), coords [N,2,H1,W1] fp32 (channel 0 = x; channel 1 is never read —
 *   the reference reads it only for an unused `dy`, :40-43), out/corr_grad [N,2r+1,H1,W1] (dtype),
 *   volume_grad [N,H1,W1,W2] (dtype).  coords_channels is 1 or 2 (batch stride of coords).
 * ------------------------------------------------------------------------------------------- */
int as_corr_sampler_fwd(const void* volume, const float* coords, void* out,
                        int N, int H1, int W1, int W2, int radius, int coords_channels, int dtype, void* stream);
int as_corr_sampler_bwd(const float* coords, const void* corr_grad, void* volume_grad,
                        int N, int H1, int W1, int W2, int radius, int coords_channels, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a1+a2  all-pairs correlation + pooled pyramid in one pass — replaces
 *   Combined_Geo_Encoding_Volume.corr + the init_corr pyramid (coreContinuous_IGEV/geometry.py:63-72,:27-29)
 *   and CorrBlock1D (corePrune_RAFT/geometry.py:46-55,:17-19).
 *   f1 [B,C,H,W1], f2 [B,C,H,W2]; levels[i] [B,H,W1,W2>>i] for i < L (1..AS_MAX_LEVELS).
 * ------------------------------------------------------------------------------------------- */
int as_corr_build_pyramid(const float* f1, const float* f2, float* const* levels,
                          int B, int C, int H, int W1, int W2, int L, void* stream);

/* a2 (geo half): gev [B,G,D,H,W] -> levels[i] [B,H,W,D>>i,G] (disparity-major, channel-minor so a
 * (2r+2)-tap window of all G channels is one contiguous run) — replaces geometry.py:17-25.       */
int as_geo_pyramid(const float* gev, float* const* levels, int B, int G, int D, int H, int W, int L, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a3  fused multi-level lookup — replaces Combined_Geo_Encoding_Volume.__call__ (geometry.py:34-60),
 *   CorrBlock1D.__call__ (corePrune_RAFT/geometry.py:24-43) and bilinear_sampler (utils.py:59-73).
 *   geo[i] as produced by as_geo_pyramid (G may be 0 and geo NULL: RAFT), corr[i] as produced by
 *   as_corr_build_pyramid, disp [B,1,H,W]; out [B, L*(2r+1)*(G+1), H, W].  The x coordinate grid
 *   (`coords`, continuous_IGEVstereo.py:280) is the pixel column and is generated in-kernel.
 * ------------------------------------------------------------------------------------------- */
int as_geo_corr_lookup_fwd(const float* const* geo, const float* const* corr, const float* disp, float* out,
                           int B, int H, int W, int W2, int D, int G, int L, int radius, void* stream);
/* backward of the above w.r.t. the volumes (disp is detached by the caller, continuous_IGEVstereo.py:285):
 * d_geo[i], d_corr[i] must be ZERO-FILLED by the caller; rows are pixel-private so no atomics are used. */
int as_geo_corr_lookup_bwd(const float* disp, const float* d_out, float* const* d_geo, float* const* d_corr,
                           int B, int H, int W, int W2, int D, int G, int L, int radius, void* stream);
/* the same, ADDING each pixel's windows to d_geo / d_corr: the lookup runs once per GRU iteration on the same pyramid
 * (continuous_IGEVstereo.py:286), so one zero-filled gradient per level collects all iterations of a training step instead of
 * `iters` zero-filled volumes that autograd then sums.  Calls on one stream are ordered; within a call every window is private. */
int as_geo_corr_lookup_bwd_accum(const float* disp, const float* d_out, float* const* d_geo, float* const* d_corr,
                                 int B, int H, int W, int W2, int D, int G, int L, int radius, void* stream);

/* a3 + the first conv of a6 fused (csrc/lookup.hip): out = act(convc1(lookup(disp))) — replaces
 *   `corr = geo_fn(disp, coords)` (continuous_IGEVstereo.py:286 -> geometry.py:34-60) followed by
 *   `F.relu(self.convc1(corr))` (update.py:84-85); the [B, L*9*(G+1), H, W] lookup result stays in LDS.
 *   Built for radius 4 with (G, L) = (8, 2) [IGEV] or (0, 4) [RAFT]; convc1 = 1x1, L*9*(G+1) -> 64 channels.
 *   wimage = as_lookup_convc1_pack(weight [64][cin] fp32) (as_lookup_convc1_pack_bytes(cin) bytes); bias [64]|NULL.
 *   Results (either or both): out_bs = blocked split-fp16 link tensor [B][2][ceil(ctot/8)][H][W][8] fp16 (as_conv_desc.out_bs),
 *   channels [coff, coff+64); out_f32 [B,64,H,W].  Split-precision arithmetic.  as_lookup_split_overflow: see
 *   as_liif_split_overflow. */
int64_t as_lookup_convc1_pack_bytes(int cin);
int as_lookup_convc1_pack(const float* w, int cin, void* image, void* stream);
int as_lookup_convc1_fwd(const float* const* geo, const float* const* corr, const float* disp, const void* wimage, const float* bias,
                         void* out_bs, int out_bs_ctot, int out_bs_coff, float* out_f32, int relu,
                         int B, int H, int W, int W2, int D, int G, int L, int radius, void* stream);
/* The front of a GRU iteration in ONE launch (inference): what followed the disparity head's first conv as three dependent
 *   launches — as_tap_shift_sum (`disp = disp + delta_disp`, continuous_IGEVstereo.py:295 with update.py:24 folded),
 *   as_lookup_convc1_fwd (geometry.py:34-60 + update.py:84-85) and as_conv7x7_c1_relu (update.py:87, with the disparity
 *   pass-through of update.py:91) — runs side by side; every block derives the new disparity of the pixels it needs from
 *   the tap planes with as_tap_shift_sum's own arithmetic (bit-identical).
 *   taps [B][groups*9][H][W] (as_conv2d AS_EPI_RELU_TAPS), head_bias [1]|NULL, disp_old -> disp_new [B,1,H,W];
 *   wimage / bias_c1: as_lookup_convc1_pack; cor_bs: blocked split-fp16 [B][2][8][H][W][8] = relu(convc1(lookup(disp_new)));
 *   w7: 7x7 weights tap-major [49][cp7 >= 64] (zero padded), b7 [64]|NULL; d1_bs: blocked relu(conv7x7(disp_new) + b7);
 *   copy_bs (optional): blocked tensor of copy_ctot channels that receives disp_new in channel copy_coff.
 *   w7 == NULL: the head's finish + lookup + convc1 only (d1_bs / copy_bs untouched); the caller launches as_conv7x7_c1_relu on
 *   disp_new itself — two launches instead of three on the loop's critical chain. */
int as_loop_front_fwd(const float* const* geo, const float* const* corr, const float* taps, int groups, const float* head_bias,
                      const float* disp_old, float* disp_new, const void* wimage, const float* bias_c1, void* cor_bs,
                      const float* w7, int cp7, const float* b7, void* d1_bs, void* copy_bs, int copy_ctot, int copy_coff,
                      int B, int H, int W, int W2, int D, int G, int L, int radius, void* stream);
unsigned as_lookup_split_overflow(int reset);
/* the same counter for the convolution kernels (conv.hip) and the all-pairs correlation build (volumes.hip) */
unsigned as_conv_split_overflow(int reset);
unsigned as_volumes_split_overflow(int reset);

/* ---------------------------------------------------------------------------------------------
 * a4  group-wise correlation volume — replaces build_gwc_volume / groupwise_correlation
 *   (coreContinuous_IGEV/submodule.py:253-271).  fl, fr [B,C,H,W]; out [B,G,D,H,W]; C % G == 0.
 * a5  (softmax over D +) disparity regression — replaces F.softmax(...)+disparity_regression
 *   (continuous_IGEVstereo.py:267-268, submodule.py:321-325).  cost [B,D,H,W] -> out [B,1,H,W];
 *   apply_softmax != 0: cost holds logits (fused path); 0: cost is already a probability volume.
 * ------------------------------------------------------------------------------------------- */
int as_gwc_volume_fwd(const float* fl, const float* fr, float* out, int B, int C, int H, int W, int D, int G, void* stream);
int as_disparity_regression(const float* cost, float* out, int B, int D, int H, int W, int apply_softmax, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a6/a7/a9/a15  implicit-GEMM convolution on fp32 MFMA with fused epilogues — replaces the nn.Conv2d
 *   calls of BasicMotionEncoder (update.py:84-92), ConvGRU (update.py:33-41), DispHead (update.py:23-24)
 *   and the nn.Linear stack of MLP (liif.py:22-25; a Linear over [B*Q,C] is a 1x1 conv over [B,C,1,Q]).
 *   stride 1, "same" zero padding (pad = K/2), square odd kernel KS in {1,3}.
 *   The input is the channel concatenation of n_src tensors src[s] [B,src_c[s],H,W] (torch.cat-free).
 *   wpack: weights re-laid out by as_conv_pack_weights.  bias [Cout] or NULL.  add [B,Cout,H,W] or NULL
 *   (the GRU context terms cz/cr/cq).  See AS_EPI_* for out/out2/h/z.  out_ctot/out_coff write the
 *   result into channels [out_coff, out_coff+Cout) of a [B,out_ctot,H,W] tensor (cat-free producers).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const float* src[AS_MAX_SRCS];
  int src_c[AS_MAX_SRCS];
  int n_src;
  const float* wpack;
  const float* bias;
  const float* add;   /* [B, add_ctot, H, W], channels [add_coff, add_coff+Cout) are used */
  int add_ctot, add_coff;
  const float* h;     /* AS_EPI_GRU_ZR / AS_EPI_GRU_Q: hidden state [B,Cout(/2),H,W]; AS_EPI_LINEAR: optional residual [B,Cout,H,W] */
  const float* z;     /* AS_EPI_GRU_Q */
  float* out;
  float* out2;        /* AS_EPI_GRU_ZR */
  int out_ctot, out_coff;
  int B, H, W, Cin, Cout, KS;
  int act, epilogue;
  int precision;      /* which pack `wpack` is: 0 = fp32 (as_conv_pack_weights), 1 = split fp16 (as_conv_pack_weights_split) */
  float* ws;          /* optional scratch for split-K on small feature maps (precision 1): partial-sum slabs */
  int64_t ws_elems;   /* capacity of ws in floats; as_conv_ws_elems() gives the useful maximum; 0/NULL = never split K */
  int stride;         /* 0 or 1: 'same' conv, output H x W.  2 (KS 3, precision 1, AS_EPI_LINEAR): padding 1, H and W are the INPUT
                         plane, out / add / h are [.., (H-1)/2+1, (W-1)/2+1] — the stride-2 convs of the encoders (extractor.py:14,83) */
  /* Blocked split-fp16 tensors (precision 1, stride 1): [B][2][ceil(C/8)][H][W][8] fp16 = a plane set of hi parts and one of lo
     parts of the kernel's operand split (x = hi + lo/2048), the 8 channels of a block contiguous per pixel — the 16-B units
     of its LDS patch image, so the loader fetches a k-half of a pixel with two coalesced 16-B loads and no arithmetic.  Written by one as_conv2d (out_bs), read by the next (src_bs):
     an internal link format between convolutions, bit-identical in effect to passing the fp32 tensor. */
  int src_bs[AS_MAX_SRCS];   /* != 0: src[s] points to such a tensor holding src_c[s] logical channels */
  void* out_bs;       /* optional blocked copy of the result: AS_EPI_LINEAR / AS_EPI_GRU_Q: of out; AS_EPI_GRU_ZR: of out2 (r*h) */
  int out_bs_ctot, out_bs_coff;  /* the copy goes to channels [out_bs_coff, +Cout) (multiple of 8) of a blocked tensor of out_bs_ctot channels (0: Cout) */
  int bs_only;        /* != 0: do not write the fp32 form of that result (LINEAR: out may be NULL; GRU_ZR: out2 may be NULL) */
  /* Dual launch (AS_EPI_LINEAR, n_src == 1, no add / residual, stride 1): a second convolution of the SAME shape (Cin, Cout, KS,
     act, plane) runs in the same grid with its own source, weights, bias and output channel window inside out / out_bs — the two
     64 -> 64 branch convolutions of BasicMotionEncoder (update.py:86,88) as one launch.  Equivalent to two calls (and executed as
     two where the fused form does not apply: fp32 precision, split-K). */
  const float* tap_w; /* AS_EPI_RELU_TAPS: [Cout][9] fp32 weights of the following Cout -> 1 convolution (tap = ky*3+kx); out is [B, ceil(Cout/64)*9, H, W] */
  int dual;
  const float* src2;
  int src2_bs;
  const float* wpack2;
  const float* bias2;
  int out_coff2, out_bs_coff2;
  /* Dual launch, round 6: the second convolution may carry its own residual (h2: both convolutions or neither), its own
     activation (dual_act2 != 0: act2 instead of act) and DENSE outputs of its own — out_b [B,Cout,H,W] and / or out_bs_b (blocked,
     Cout channels), mirroring which of out / out_bs the first one writes — instead of a channel window of out / out_bs: the two
     heads of a context-network scale (extractor.py:254-273: ResidualBlock + conv per head on the same input) layer by layer. */
  const float* h2;
  int dual_act2, act2;
  float* out_b;
  void* out_bs_b;
} as_conv_desc;
int as_conv2d(const as_conv_desc* d, void* stream);
/* floats of split-K scratch worth passing in as_conv_desc.ws for this problem (0: the problem is large enough) */
int64_t as_conv_ws_elems(int B, int Cout, int H, int W);
/* weight [Cout,Cin,KS,KS] (nn.Conv2d layout) -> wpack; returns the element count needed when wpack==NULL */
int64_t as_conv_pack_size(int Cin, int Cout, int KS);
int as_conv_pack_weights(const float* weight, float* wpack, int Cin, int Cout, int KS, void* stream);
/* the split-precision pack (fp16 hi/lo, k-contiguous): element count is in fp16 units */
int64_t as_conv_pack_size_split(int Cin, int Cout, int KS);
int as_conv_pack_weights_split(const float* weight, void* wpack, int Cin, int Cout, int KS, void* stream);
/* the pack of the DATA-GRADIENT convolution straight from the forward weight: weight is [Cout_w, Cin_w, KS, KS] with Cout_w = Cin and
 * Cin_w = Cout of this call, packed as W'[ci][co][ky][kx] = W[co][ci][KS-1-ky][KS-1-kx] (transposed, taps flipped; autograd of
 * update.py:16-92 and of every stride-1 convolution the library runs: no flipped / transposed copy of the weight in memory) */
int as_conv_pack_weights_split_t(const float* weight, void* wpack, int Cin, int Cout, int KS, void* stream);

/* direct (VALU) convolutions for the two shapes where an MFMA tile would be mostly padding:
 *   convd1: 7x7, 1 -> Cout, +bias, ReLU   (update.py:81,87);   conv2 of DispHead: 3x3, Cin -> 1, +bias (update.py:19,24)
 *   tap_major = 0: weight is the module's [Cout,1,7,7]; tap_major = 1: weight is its transpose [49][Cout_pad], Cout_pad =
 *   Cout rounded up to 64, zero padded (one tap's weights contiguous -> vector scalar loads; the fast path).
 *   copy_out != NULL (tap_major only): x is also copied into channel copy_coff of copy_out [B,copy_ctot,H,W] — the
 *   `torch.cat([out, disp])` tail of BasicMotionEncoder (update.py:91) without a copy kernel. */
int as_conv7x7_c1_relu(const float* x, const float* weight, const float* bias, float* out,
                       int B, int H, int W, int Cout, int out_ctot, int out_coff, int tap_major,
                       float* copy_out, int copy_ctot, int copy_coff, int copy_bs /* != 0: copy_out is a blocked split-fp16 tensor of copy_ctot channels (as_conv_desc.src_bs) */,
                       int out_bs /* != 0 (tap_major only): out is a blocked split-fp16 tensor of out_ctot channels, out_coff a multiple of 8 */,
                       void* stream);
int as_conv3x3_to1(const float* x, const float* weight, const float* bias, float* out,
                   int B, int Cin, int H, int W, void* stream);

/* second stage of the MFMA form of a 3x3, Cin -> 1 convolution: S [B,9,H,W] holds, per tap t = ky*3+kx,
 * the channel reduction sum_c w[c,ky,kx]*x[c] (a 1x1 as_conv2d with the 9 taps as output channels);
 * out[b,0,y,x] = bias + sum_t S[b,t,y+ky-1,x+kx-1] with zero padding — DispHead.conv2 (update.py:19,24). */
int as_tap_shift_sum(const float* S /* [B, groups*9, H, W] */, const float* bias, const float* addend /* [B,1,H,W] or NULL: out = addend + (...) */,
                     float* out, int B, int H, int W, int groups /* channel-tile partials summed in order (1: the plain form) */, void* stream);

/* a8  pool2x = avg_pool2d(3,stride 2,pad 1) (update.py:94-95); interp = bilinear align_corners=True
 *     resize (update.py:100-102).  x [B,C,H,W] -> out [B,C,Ho,Wo].                                */
int as_pool2x(const float* x, float* out, int B, int C, int H, int W, void* stream);
int as_interp_bilinear_ac(const float* x, float* out, int B, int C, int H, int W, int Ho, int Wo, void* stream);
/* the same resamplers with a blocked split-fp16 result [B][2][ceil(C/8)][Ho][Wo][8] (as_conv_desc.src_bs) for maps that only
 * feed convolutions (pool2x(net) and interp(net) inside BasicMultiUpdateBlock.forward, update.py:124-131) */
int as_pool2x_bs(const float* x, void* out_bs, int B, int C, int H, int W, void* stream);
int as_interp_bilinear_ac_bs(const float* x, void* out_bs, int B, int C, int H, int W, int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * f4  backbone-side one-shot operators, direct convolutions with fused bias + activation;
 *     eval-mode BatchNorm is folded into weight/bias by the caller.
 *   as_dwconv3x3: depthwise 3x3, padding 1, stride 1|2 — conv_dw (+bn, +ReLU6) of the MobileNetV2 blocks of
 *     Feature (extractor.py:327-342).  x [B,C,H,W], weight [C,1,3,3], bias [C]|NULL, residual [B,C,Ho,Wo]|NULL
 *     (added after the activation) -> out [B,C,Ho,Wo], Ho = (H-1)/stride+1.
 *   as_conv3d_k3: Conv3d 3x3x3, padding 1, stride 1|2 (all dims), no groups — corr_stem / classifier / hourglass
 *     convs (continuous_IGEVstereo.py:22-89,:139,:158).  x [B,Cin,D,H,W]; wpack [Cin,27,Cout] (= weight
 *     [Cout,Cin,3,3,3] permuted so one tap's Cout weights are contiguous); bias [Cout]|NULL -> out [B,Cout,Do,Ho,Wo].
 * ------------------------------------------------------------------------------------------- */
int as_dwconv3x3(const float* x, const float* weight, const float* bias, const float* residual, float* out,
                 int B, int C, int H, int W, int stride, int act, void* stream);

/* f4 (round 6)  One launch per MobileNetV2 inverted-residual block of Feature (extractor.py:327-342 = timm mobilenetv2_100
 *   InvertedResidual, eval): relu6(bn1(conv_pw x)) -> relu6(bn2(conv_dw .)) (3x3, padding 1, stride 1|2) -> bn3(conv_pwl .) [+ x when
 *   stride 1 and Cin == Cout].  The 6x-expanded tensor stays in LDS.  BatchNorm is folded by the caller:
 *   as_ir_block_pack: w1 [mid][Cin], w3 [Cout][mid] -> MFMA fragments (as_ir_block_pack_bytes bytes, 16-B aligned);
 *   fparams [11 * mid_pad + Cout] fp32 = b1 [mid_pad] | b2 [mid_pad] | wd [mid_pad][9] | b3 [Cout], mid_pad = ceil(mid/32)*32, padding 0.
 *   x [B,Cin,H,W] -> out [B,Cout,(H-1)/stride+1,(W-1)/stride+1].  Cin <= 256, Cout <= 160.  Split-precision arithmetic. */
int64_t as_ir_block_pack_bytes(int Cin, int mid, int Cout);
int as_ir_block_pack(const float* w1, const float* w3, int Cin, int mid, int Cout, void* pack, void* stream);
int as_ir_block(const float* x, const void* pack, const float* fparams, float* out, int B, int Cin, int mid, int Cout, int H, int W,
                int stride, int residual, void* stream);
int as_conv3d_k3(const float* x, const float* wpack, const float* bias, float* out,
                 int B, int Cin, int Cout, int D, int H, int W, int stride, int act, void* stream);
/* as_conv3d_k3 with FeatureAtt's channel gate in the epilogue (round 6): out = act(conv + bias) * gate[b, co, y, x] for every depth
 * slice — `torch.sigmoid(feat_att(feat).unsqueeze(2)) * cv` (submodule.py:328-341) without a pass over the volume.
 * gate [B,Cout,Ho,Wo] or NULL (= as_conv3d_k3). */
int as_conv3d_k3_gated(const float* x, const float* wpack, const float* bias, const float* gate, float* out,
                       int B, int Cin, int Cout, int D, int H, int W, int stride, int act, void* stream);
/*   as_deconv3d_k4s2: ConvTranspose3d kernel 4, stride 2, padding 1 (all dims) — the hourglass up-convolutions
 *     (continuous_IGEVstereo.py:43-51).  x [B,Cin,D,H,W]; wpack [Cin,4,4,4,Cout] (= weight [Cin,Cout,4,4,4] with
 *     Cout moved last); bias [Cout]|NULL -> out [B,Cout,2D,2H,2W]. */
int as_deconv3d_k4s2(const float* x, const float* wpack, const float* bias, float* out,
                     int B, int Cin, int Cout, int D, int H, int W, int act, void* stream);
/*   as_instance_norm_act: act(InstanceNorm2d/3d(x)) with affine = False and biased variance — the IN + LeakyReLU / ReLU tail
 *     of BasicConv_IN and the stems (submodule.py:76-103, continuous_IGEVstereo.py:105-118).  x, out [planes = B*C][HW];
 *     ws: caller scratch of as_instance_norm_ws_bytes(planes) bytes (fp64 partial sums), 8-B aligned.
 *   as_layernorm2d_act: act(w * (x - mean_c) / sqrt(var_c + eps) + b) per pixel over the C <= 64 channels of NCHW x —
 *     LayerNorm2d (submodule.py:148-187) + the ReLU / GELU that follows it in the HighRes_Aggregation heads. */
int64_t as_instance_norm_ws_bytes(int planes);
int as_instance_norm_act(const float* x, const float* residual /* same shape as x, or NULL: out = relu(residual + act(IN(x))) */,
                         float* out, void* ws, int planes, int64_t HW, float eps, int act, void* stream);
int as_layernorm2d_act(const float* x, const float* weight, const float* bias, float* out, int B, int C, int H, int W, float eps,
                       int act, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a12/a13  cosine affinity to the 8 neighbours, written straight into channels [C, C+8) of the
 *   concatenated structure feature — replaces AffinityFeature.forward (liif.py:432-446) and the
 *   'with_v2ISU' branch of StructureFeature.forward (liif.py:496-499).
 *   x [B,C,H,W] -> out [B,C+8,H,W] (channels [0,C) = x copied, [C,C+8) = affinity); ws = caller-provided
 *   [B,H,W] scratch that receives the clamped per-pixel L2 norm.
 * a14  nearest gather + relative coordinate — replaces liif_feat_multiscale_train (liif.py:108-137).
 *   feat [B,C,H,W], coord [B,Q,2] (row,col) -> latent[B, lat_ctot, Q] channels [lat_coff, lat_coff+C+2)
 *   = [q_feat (C), rel_row, rel_col]   (channel-major so the MLP runs as a 1x1 conv over Q).
 * a16/a17  (softmax over the 9 mask logits +) convex combination of the 3x3 neighbourhood of the nearest
 *   low-res pixel — replaces F.softmax + context_upsample_multiscale_train
 *   (continuous_IGEVstereo.py:204,212-214; submodule.py:357-372).
 *   disp [B,1,H,W]; mask [B,9,Q]; coord [B,Q,2] -> out [B,1,Q].
 *   mask_is_logits != 0: softmax is applied to mask first.  scale != NULL ([B]): the disparity is the
 *   UNscaled 1/4-res field and is multiplied by 4*scale_b on the fly (continuous_IGEVstereo.py:204);
 *   scale == NULL: disp is used as given (the reference function's own contract).
 * ------------------------------------------------------------------------------------------- */
int as_structure_feature(const float* x, float* out, float* ws /* [B,H,W] scratch */, int B, int C, int H, int W, void* stream);
int as_liif_gather(const float* feat, const float* coord, float* latent,
                   int B, int C, int H, int W, int Q, int lat_ctot, int lat_coff, void* stream);
/* a14 + first layer of a15 fused: out[b,c,q] = relu(u0[b,c,n0(q)] + u1[b,c,n1(q)] + wrel[c,:]·rel(q) + bias[c]) where
 *   u_i = W1_i·feat_i are the first Linear layer's blocks applied at LOW resolution (1x1 as_conv2d), n_i the nearest
 *   source pixel and rel(q) = (rel_row0, rel_col0, rel_row1, rel_col1) as in as_liif_gather (liif.py:108-137 followed
 *   by the first Linear + ReLU of the MLP, liif.py:9-25).  u0 [B,C,H0,W0]; u1 [B,C,H1,W1] or NULL; wrel [C,2*n_src];
 *   bias [C]|NULL; coord [B,Q,2] -> out [B,C,Q]. */
int as_liif_gather_mlp1(const float* u0, const float* u1, const float* coord, const float* wrel, const float* bias, float* out,
                        int B, int C, int H0, int W0, int H1, int W1, int Q, void* stream);
int as_convex_upsample(const float* disp, const float* scale, const float* mask, const float* coord,
                       float* out, int B, int H, int W, int Q, int mask_is_logits, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a12-a17 fused for inference (csrc/liif_fused.hip): the whole `upsample_disp` tail
 *   (continuous_IGEVstereo.py:192-214 / prune_raft_stereo.py:200-226: StructureFeature -> liif_feat_multiscale_train ->
 *   MLP -> softmax -> context_upsample_multiscale_train; liif.py:432-446,:496-499,:108-137,:9-25,:644-678;
 *   submodule.py:357-372) with no per-query intermediate in HBM.  Same function as the staged entry points above.
 *
 * as_liif_affinity   aff [B,8,H,W] = AffinityFeature(cat(srcs)) for up to 3 NCHW sources (never concatenated; every
 *                    source but the last needs a multiple of 8 channels); ws = scratch of
 *                    as_liif_affinity_ws_bytes(B, H, W, ceil(sum of channels / 8)) bytes (per-channel-group partial sums).
 * as_liif_lowres_pack  W1[:, koff : koff+K] ([128][ldw] fp32, K <= 192) as split-fp16 MFMA fragments
 *                    (as_liif_lowres_pack_bytes(K) bytes).
 * as_liif_lowres_cl  out[b][y*W+x][0..128) = W1[:, koff : koff+K] . cat(srcs)[b,:,y,x]  — the first Linear layer's feature
 *                    block applied per LOW-resolution pixel (it commutes with the nearest gather), result channels-last so
 *                    that a query's 128-vector is one 512-B run.  wimage = as_liif_lowres_pack of that column block;
 *                    K = sum of the source channels; every source but the last needs a multiple of 16 channels.
 * as_liif_tail_pack  the 128->64->64->9 layers + the relative-coordinate columns / bias of the first layer as the tail
 *                    kernel's LDS weight image (as_liif_tail_image_bytes() bytes, split fp16 fragments in MFMA order).
 *                    wrel [128][2*n_src] (rel_row, rel_col per source), b1 [128]|NULL, w2 [64][128], w3 [64][64], w4 [9][64].
 * as_liif_tail       per query: relu(u0[n0(q)] + u1[n1(q)] + wrel.rel(q) + b1) -> MLP -> softmax -> sum_k p_k * d[3x3 nbr k of
 *                    the nearest pixel of disp], d = disp * 4 * scale_b (scale == NULL: disp as given) -> out [B,1,Q].
 *                    u0 [B][H0*W0][128], u1 [B][H1*W1][128]|NULL (as_liif_lowres_cl); disp [B,1,Hd,Wd]; coord [B,Q,2]
 *                    (row, col); clamp_inplace != 0 writes the clamped coordinates back (the reference's in-place
 *                    `hr_coord.clamp_`, submodule.py:366); logits [B,9,Q]|NULL also receives the mask logits (`liif_up`'s
 *                    return value).  Split-precision arithmetic (3 x fp16 MFMA per product, fp32 accumulate).
 * as_liif_rows_cl    channels-last copy of a small input's structure feature: out [B][H*W][as_liif_rows_pitch() = 48] =
 *                    cat(srcs)[b, :, y, x], zero padded (<= 48 channels: stem_2x | its affinity, liif.py:496-499).
 * as_liif_tail_direct  as_liif_tail with the SECOND input handed over as those raw rows instead of its first-layer rows:
 *                    image1 = as_liif_lowres_pack of W1[:, that input's columns]; the tail takes the product per query
 *                    (three k-steps) — the 128-channel table at the second input's resolution (4 x the first one's) is
 *                    neither written nor gathered.  Same result up to fp32 summation order.
 * as_liif_split_overflow  number of waves (since the last reset) in which an operand of these kernels left the fp16
 *                    range and was saturated to +-65504 (synchronises; diagnostics only).
 * ------------------------------------------------------------------------------------------- */
int64_t as_liif_affinity_ws_bytes(int B, int H, int W, int n_chunks);
int as_liif_affinity(const float* const* srcs, const int* channels, int n_src, float* aff, float* ws, int B, int H, int W, void* stream);
int64_t as_liif_lowres_pack_bytes(int K);
int as_liif_lowres_pack(const float* w, int ldw, int koff, int K, void* image, void* stream);
int as_liif_lowres_cl(const float* const* srcs, const int* channels, int n_src, const void* wimage, float* out,
                      int B, int H, int W, void* stream);
int64_t as_liif_tail_image_bytes(void);
int as_liif_tail_pack(const float* wrel, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                      const float* w4, const float* b4, int n_src, void* image, void* stream);
int as_liif_tail(const float* u0, const float* u1, float* coord, const void* image, const float* disp, const float* scale,
                 float* out, float* logits, int B, int Q, int H0, int W0, int H1, int W1, int Hd, int Wd, int clamp_inplace,
                 const int* row_len, void* stream);
/* Query ORDER hint for as_liif_tail / as_liif_tail_direct (round 6): *row_len (device memory) = the row length of a raster-ordered
 * query grid (make_coord, liif.py:27-45, flattened row-major as evaluation.py:67-89 builds hr_coord), found on the device without a
 * host synchronisation — the first index where the row coordinate changes, accepted when the grid structure repeats — or 0.  The
 * tail then walks 4 (8) query rows x 32 columns per block step, so the low-resolution rows a patch shares are fetched once
 * (HBM-side traffic of the tail at 960x540: 200 -> ~100 MB).  row_len may be NULL.  Any value yields the same results. */
int as_liif_query_rows(const float* coord, int B, int Q, int* row_len, void* stream);
int as_liif_rows_pitch(void);
int as_liif_rows_cl(const float* const* srcs, const int* channels, int n_src, float* out, int B, int H, int W, void* stream);
int as_liif_tail_direct(const float* u0, const float* rows1, float* coord, const void* image, const void* image1, const float* disp,
                        const float* scale, float* out, float* logits, int B, int Q, int H0, int W0, int H1, int W1, int Hd, int Wd,
                        int clamp_inplace, const int* row_len, void* stream);

/* Training of the per-query MLP (liif.py:9-25, :644-678 under autograd; train_continuous_IGEV.py:214-239) without per-query
 * activations saved by the forward.
 * as_liif_mlp_fwd     mask logits [B,9,Q] of as_liif_tail's MLP (same kernel, no softmax / convex combination output, the
 *                    caller's coordinates are NOT clamped).  B1 > 0: u1 holds B1 batch elements and query batch b reads element
 *                    b % B1 (B % B1 == 0) — the loop-invariant second input of a call batched over (iteration, sample).
 * as_liif_mlp_bwd_pack  W2 [64][128], W3 [64][64], W4 [9][64] as the fragments of the TRANSPOSED layers
 *                    (as_liif_mlp_bwd_image_bytes() bytes).
 * as_liif_mlp_bwd     recomputes h1, h2, h3 per 32-query tile exactly as the forward did, runs the data-gradient chain
 *                    d3 = [h3>0] W4^T dlogits, d2 = [h2>0] W3^T d3, d1 = [h1>0] W2^T d2 on the matrix cores and writes h1 [B,128,Q],
 *                    h2, h3, d3, d2 [B,64,Q], d1 [B,128,Q] (post-ReLU activations / gradients w.r.t. the pre-activations): the
 *                    operands of the layers' weight gradients (as_conv2d_wgrad) and of the first layer's scatter-add
 *                    (as_liif_scatter_add).  image = as_liif_tail_pack, imageT = as_liif_mlp_bwd_pack of the same weights.
 *                    Either d1, or (d1 = NULL) the first layer's consumers fused: du0 [B,128,H0,W0] and du1 [B1,128,H1,W1] receive
 *                    the scatter-add of d1 (zero-filled here; runs of queries in one source pixel are pre-summed inside the wave,
 *                    evaluations of a shared second input add into element b % B1), dwrel [128][4] the gradient of the
 *                    relative-coordinate columns — d1 then never reaches memory. */
int64_t as_liif_mlp_bwd_image_bytes(void);
int as_liif_mlp_bwd_pack(const float* w2, const float* w3, const float* w4, void* imageT, void* stream);
int as_liif_mlp_fwd(const float* u0, const float* u1, const float* coord, const void* image, float* logits, int B, int B1, int Q,
                    int H0, int W0, int H1, int W1, void* stream);
int as_liif_mlp_bwd(const float* u0, const float* u1, const float* coord, const void* image, const void* imageT, const float* dlogits,
                    float* h1, float* h2, float* h3, float* d3, float* d2, float* d1, float* du0, float* du1, float* dwrel, int B, int B1,
                    int Q, int H0, int W0, int H1, int W1, void* stream);
unsigned as_liif_split_overflow(int reset);

/* ---------------------------------------------------------------------------------------------
 * Backward (training, cfg 4: train_continuous_IGEV.py:214-239) of the HBM-bound operators above — what autograd
 * derives for the reference's Python call sites.  All fp32, every destination element is written (no zero-fill
 * contract) — the two scatters (liif_gather_bwd's d_feat, convex_upsample_bwd's d_disp) clear their destination on
 * `stream` and accumulate with float atomics, like ATen's grid_sample backward.
 *   a2^T  as_corr_pyramid_bwd: d_levels[i] [rows, W2>>i] (rows = B*H*W1) -> d_corr0 [rows, W2] = sum_i 2^-i d_levels[i][r, x>>i]
 *         (avg_pool2d([1,2]) chain, geometry.py:27-29 / corePrune_RAFT/geometry.py:16-19).  d(f1), d(f2) then are two plain
 *         batched GEMMs of d_corr0 with f2 / f1 (einsum 'aijk,aijh->ajkh', geometry.py:70) — left to rocBLAS by the binding.
 *   a2^T  as_geo_pyramid_bwd: d_levels[i] [B,H,W,D>>i,G] -> d_gev [B,G,D,H,W] (geometry.py:17-25).
 *   a4^T  as_gwc_volume_bwd: d_vol [B,G,D,H,W] -> d_fl, d_fr [B,C,H,W] (submodule.py:253-271).
 *   a5^T  as_disparity_regression_bwd: d_out [B,1,H,W] -> d_cost [B,D,H,W]; apply_softmax as in the forward.
 *   a14^T as_liif_gather_bwd: d_latent [B,lat_ctot,Q] channels [lat_coff, lat_coff+C) -> d_feat [B,C,H,W]
 *         (grid_sample nearest backward, liif.py:122-125; the relative coordinates carry no gradient to feat).
 *   a16/a17^T as_convex_upsample_bwd: d_out [B,1,Q] -> d_mask [B,9,Q] (w.r.t. the logits when mask_is_logits) and
 *         d_disp [B,1,H,W] (may be NULL) w.r.t. the UNscaled disparity when scale != NULL.
 * ------------------------------------------------------------------------------------------- */
/*   a6/a7/a9 weight gradient  as_conv2d_wgrad: x [B,Cin,H,W], dy [B,Cout,H,W] -> dw [Cout,Cin,KS,KS] (+ db [Cout] when db != NULL) of a
 *         stride-1 "same" convolution, KS = 1 | 3 — what autograd derives for the nn.Conv2d layers of update.py:16-92
 *         (train_continuous_IGEV.py:214-239 applies each `iters` times per step; the binding stacks all iterations along B and
 *         reduces them in one launch).  bf16 hi/lo operand split (hi*hi + hi*lo + lo*hi, one fp32 accumulator: ~2^-16 relative
 *         per product, fp32 exponent range), split-K over a workspace of as_conv2d_wgrad_ws_bytes() bytes, summed in a fixed
 *         order (deterministic).  dw / db are overwritten. */
int64_t as_conv2d_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W, int KS);
int as_conv2d_wgrad(const float* x, const float* dy, float* dw, float* db, int B, int Cin, int Cout, int H, int W, int KS,
                    void* ws, int64_t ws_bytes, void* stream);
/* the same over n <= 32 (x, dy) tensor pairs of `per` images each — the GRU iterations of one training step, reduced in ONE launch
 * without stacking them into one tensor first (the stacking copies were 3-4 ms of a step); workspace as for B = n * per */
int as_conv2d_wgrad_multi(const float* const* xs, const float* const* dys, int n, int per, float* dw, float* db, int Cin, int Cout,
                          int H, int W, int KS, void* ws, int64_t ws_bytes, void* stream);

/* Weight [Cout][1][7][7] and bias gradient of the motion encoder's 7x7 convolution of the one-channel disparity map (update.py:81,87,
 * convd1) for up to 32 (x [per,1,H,W], dy [per,Cout,H,W]) pairs — the GRU iterations of a step — in one launch; Cout <= 64.
 * ws: as_conv7x7_c1_wgrad_ws_bytes(n*per, Cout, H, W) bytes of scratch.  Fixed summation order (deterministic). */
int64_t as_conv7x7_c1_wgrad_ws_bytes(int B, int Cout, int H, int W);
int as_conv7x7_c1_wgrad_multi(const float* const* xs, const float* const* dys, int n, int per, float* dw, float* db, int Cout, int H, int W,
                              void* ws, int64_t ws_bytes, void* stream);
/*   a8^T  as_pool2x_bwd / as_interp_bilinear_ac_bwd: d_out [B,C,Ho,Wo] -> d_x [B,C,H,W], the transposes of as_pool2x /
 *         as_interp_bilinear_ac (what autograd derives for F.avg_pool2d / F.interpolate at update.py:94-102); gather form, one
 *         thread per input element, fixed summation order. */
int as_pool2x_bwd(const float* d_out, float* d_x, int B, int C, int H, int W, void* stream);
/*   f4^T  depthwise 3x3 (padding 1) under autograd — conv_dw of the MobileNetV2 blocks (extractor.py:331-342; MIOpen has only
 *         its naive reference solvers for fp32 convolutions with groups == channels).  Forward: as_dwconv3x3.  Data gradient:
 *         stride 1 = as_dwconv3x3 on the flipped taps; stride 2 = as_dwconv3x3_s2_bwd_data (d_out [B,C,Ho,Wo] -> d_x [B,C,H,W],
 *         gather form).  Weight gradient: as_dwconv3x3_wgrad writes partial[C][slices][9] (slices = as_dwconv3x3_wgrad_slices);
 *         the caller sums the slices in order (deterministic). */
int as_dwconv3x3_s2_bwd_data(const float* d_out, const float* weight, float* d_x, int B, int C, int H, int W, void* stream);
int as_dwconv3x3_wgrad_slices(int B, int C, int H, int W, int stride);
int as_dwconv3x3_wgrad(const float* x, const float* d_out, float* partial, int slices, int B, int C, int H, int W, int stride, void* stream);
int as_interp_bilinear_ac_bwd(const float* d_out, float* d_x, int B, int C, int H, int W, int Ho, int Wo, void* stream);
int as_corr_pyramid_bwd(const float* const* d_levels, float* d_corr0, long long rows, int W2, int L, void* stream);
int as_geo_pyramid_bwd(const float* const* d_levels, float* d_gev, int B, int G, int D, int H, int W, int L, void* stream);
int as_gwc_volume_bwd(const float* fl, const float* fr, const float* d_vol, float* d_fl, float* d_fr,
                      int B, int C, int H, int W, int D, int G, void* stream);
int as_disparity_regression_bwd(const float* cost, const float* d_out, float* d_cost, int B, int D, int H, int W,
                                int apply_softmax, void* stream);
int as_liif_gather_bwd(const float* d_latent, const float* coord, float* d_feat,
                       int B, int C, int H, int W, int Q, int lat_ctot, int lat_coff, void* stream);
/* deterministic form (no atomics): `order` [B,Q] = the queries of each batch element stably sorted by source pixel, `starts`
 * [B,npix+1] = first sorted position of every pixel (and Q at npix); out [B,C,npix] is written completely, in a fixed summation order */
int as_liif_gather_bwd_det(const float* d_rows, const int* order, const int* starts, float* out, int B, int C, int npix, int Q,
                           int ctot, int coff, void* stream);
/* a7 in training: the ConvGRU gate math (update.py:33-41) as two fused pointwise stages and their transposes (the convs
 *   run through as_conv2d with the LINEAR epilogue; inference uses the fused GRU epilogues instead).
 *   ZR: lin [B,2C,H,W] = convz||convr output, ctx [B,ctx_ctot,H,W] with [cz||cr] at channel ctx_coff, h [B,C,H,W]
 *       -> z, r, rh = r*h.   bwd: d_z, d_rh (either may be NULL = zero) -> d_lin [B,2C,H,W] (= gradient of the context window
 *       as well) and this stage's contribution d_h.
 *   Q : lin [B,C,H,W] = convq output, cq at channel ctx_coff -> t = tanh(lin + cq), out = (1-z) h + z t.
 *       bwd: d_out -> d_lin (= d cq), d_z, d_h (this stage's contribution). */
int as_gru_gates_zr(const float* lin, const float* ctx, int ctx_ctot, int ctx_coff, const float* h, float* z, float* r, float* rh,
                    int B, int C, int H, int W, void* stream);
int as_gru_gates_zr_bwd(const float* d_z, const float* d_rh, const float* z, const float* r, const float* h, float* d_lin,
                        float* d_h, int B, int C, int H, int W, void* stream);
int as_gru_gates_q(const float* lin, const float* ctx, int ctx_ctot, int ctx_coff, const float* z, const float* h, float* out,
                   float* t, int B, int C, int H, int W, void* stream);
int as_gru_gates_q_bwd(const float* d_out, const float* z, const float* t, const float* h, float* d_lin, float* d_z, float* d_h,
                       int B, int C, int H, int W, void* stream);
/* the same transposes, also ADDING the context-window gradient (= d_lin) into d_ctx [B,ctx_ctot,H,W] at channel ctx_coff: the
 *   context tensor is the same in every GRU iteration (continuous_IGEVstereo.py:273), so one zero-filled accumulator collects all
 *   iterations of a training step instead of 3 x iters sliced gradients that autograd sums one by one */
int as_gru_gates_zr_bwd_ctx(const float* d_z, const float* d_rh, const float* z, const float* r, const float* h, float* d_lin,
                            float* d_h, float* d_ctx, int ctx_ctot, int ctx_coff, int B, int C, int H, int W, void* stream);
int as_gru_gates_q_bwd_ctx(const float* d_out, const float* z, const float* t, const float* h, float* d_lin, float* d_z, float* d_h,
                           float* d_ctx, int ctx_ctot, int ctx_coff, int B, int C, int H, int W, void* stream);
/* relative coordinates of a14 alone and the query sort key of the training path: rel [B, 2*n_src, Q] (rows
 *   rel_row_s, rel_col_s as in as_liif_gather; may be NULL), key [B,Q] int32 = (nearest pixel of source 0) * 4 + parity of
 *   the nearest pixel of source 1 (may be NULL).  Sorting the queries by key makes the scatter of as_liif_gather_bwd
 *   run-coherent (it pre-sums runs of equal source pixels inside a wave before the atomics). */
int as_liif_rel_key(const float* coord, float* rel, int* key, int B, int Q, int n_src, int H0, int W0, int H1, int W1, void* stream);
int as_convex_upsample_bwd(const float* disp, const float* scale, const float* mask, const float* coord, const float* d_out,
                           float* d_mask, float* d_disp, int B, int H, int W, int Q, int mask_is_logits, void* stream);

/* ---------------------------------------------------------------------------------------------
 * §8 f4: the implicit upsampler's off-by-default options (csrc/liif_variants.hip)
 *
 * as_liif_latent   one source's block of the MLP input, CHANNEL-major, in ONE launch (liif_out_multi_scale_Training.forward,
 *                  liif.py:652-676): latent[b, lat_coff + ..., q] = [ gathered features | rel_row, rel_col | sin/cos encoding |
 *                  cell ].  Features: C channels of the nearest pixel (liif_feat_multiscale_train, liif.py:108-137); with
 *                  unfold9 its zero-padded 3x3 neighbourhood, channel c*9 + ky*3 + kx (F.unfold, liif.py:655); with n_samp = 4
 *                  the four half-cell shifted nearest samples, blocks in (vx, vy) = (-1,-1), (-1,1), (1,-1), (1,1) order, and
 *                  rel taken to the mean of the first and last sample's cell centres (liif_feat_multiscale_train_quater,
 *                  liif.py:140-176).  emb [n_enc,2] (NULL when n_enc = 0) are the frequency rows of SpatialEncoding: the block
 *                  [rel, sin(rel·emb^T), cos(rel·emb^T)] replaces rel (liif.py:339-370, cat_input).  cell [B,Q,2] or NULL is
 *                  appended as is (decode_cell, liif.py:111-114,673).  Block width = (unfold9 ? 9C : C)*n_samp + 2 + 2*n_enc +
 *                  (cell ? 2 : 0), checked against [lat_coff, lat_ctot).
 * as_liif_latent_bwd  d_feat [B,C,H,W] = scatter-add of the block's feature channels (zero-filled here).
 * as_convex_upsample_quater(_bwd)  out[b,q] = sum_k w_k(q) * disp[b, nearest(coord + shift_k)] * (4*scale_b), four shifted
 *                  samples instead of the 3x3 window, mask [B,4,Q] (context_upsample_multiscale_train_quaterp,
 *                  submodule.py:375-399; scale / logits handling as in as_convex_upsample); the coordinates are NOT clamped
 *                  in place by this variant.
 * as_affinity_bwd  gradient of cat(x, AffinityFeature(x)) / AffinityFeature(x) w.r.t. x when the affinity sees the live map
 *                  ('with_ISU', 'with_1_4ISU', 'only_ISU': liif.py:493-495,501-503,534-535 over :432-446): aff / g_aff point
 *                  at the 8 affinity channels of the forward output and of its gradient (batch strides in floats), g_x [C
 *                  channels] or NULL is the pass-through gradient of the concat; ws = B*H*W floats of scratch. */
int as_liif_latent(const float* feat, const float* coord, const float* emb, const float* cell, float* latent, int B, int C, int H,
                   int W, int Q, int lat_ctot, int lat_coff, int unfold9, int n_samp, int n_enc, void* stream);
int as_liif_latent_bwd(const float* d_latent, const float* coord, float* d_feat, int B, int C, int H, int W, int Q, int lat_ctot,
                       int lat_coff, int unfold9, int n_samp, void* stream);
int as_convex_upsample_quater(const float* disp, const float* scale, const float* mask, const float* coord, float* out, int B, int H,
                              int W, int Q, int mask_is_logits, void* stream);
int as_convex_upsample_quater_bwd(const float* disp, const float* scale, const float* mask, const float* coord, const float* d_out,
                                  float* d_mask, float* d_disp, int B, int H, int W, int Q, int mask_is_logits, void* stream);
int as_affinity_bwd(const float* x, const float* aff, long long aff_batch_stride, const float* g_aff, long long g_aff_batch_stride,
                    const float* g_x, long long g_x_batch_stride, float* dx, float* ws, int B, int C, int H, int W, void* stream);

/* f4: the encoders' 7x7, 3 -> 64 channel stem (`conv1`, extractor.py:127 in BasicEncoder / MultiBasicEncoder; stride 1, padding 3)
 *   as one split-precision MFMA launch (csrc/stem7x7.hip): out [B,64,H,W] = act(conv7x7(x [B,3,H,W]) + bias).
 *   wpack = as_conv7x7_c3_pack_bytes() bytes written by as_conv7x7_c3_pack from the fp32 weight [64,3,7,7] (BatchNorm folded by the
 *   caller where the norm is a frozen BatchNorm); act in {AS_ACT_NONE, AS_ACT_RELU, AS_ACT_LEAKY}.  Split-precision mode only. */
long long as_conv7x7_c3_pack_bytes(void);
int as_conv7x7_c3_pack(const float* weight, void* wpack, void* stream);
int as_conv7x7_c3(const float* x, const void* wpack, const float* bias, float* out, int B, int H, int W, int act, void* stream);

/* f4: 3x3, padding 1, stride 1|2 convolution of an image-like input (Cin <= 8: the 3-channel stems, e.g. MobileNetV2's `conv_stem`,
 *   extractor.py:331-336) to Cout = multiple of 8 channels, + bias (folded BatchNorm) + activation.  wpack [Cin][9][Cout] fp32
 *   (= weight.permute(1,2,3,0)).  Plain fp32 FMA arithmetic (27 MAC per output: HBM-bound). */
int as_conv3x3_few(const float* x, const float* wpack, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                   int stride, int act, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ANYSTEREO_HIP_H */
