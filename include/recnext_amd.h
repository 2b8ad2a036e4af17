/*
 * recnext_amd -- C ABI of the MI355X (gfx950) RecConv2d / RecAttn2d token-mixer kernels.
 *
 * The reference (suous/RecNeXt) has no FFI layer: its boundary for this path is the Python class
 *     RecConv2d(in_channels, kernel_size=5, bias=False, level=2, mode='bilinear')   model/recnext.py:9
 *     RecAttn2d(dim, num_heads, kernel_size=5, stage=1, mode="nearest")             model/recattn.py:55
 * whose bodies are chains of ATen calls.  This header is what a binding for that class would
 * call instead (see INTEGRATION.md for the ctypes stub); each entry point cites the reference
 * lines it replaces.
 *
 * Conventions
 *   - Activations: device pointers to NHWC-contiguous memory (the storage of a torch.channels_last
 *     tensor of logical shape N x C x H x W).  dtype: RCX_DTYPE_F32, RCX_DTYPE_BF16 or RCX_DTYPE_F16 (arithmetic is float32 for all three).
 *   - Weights: float32 device memory, *packed* tap-major (k,k,C) by rcx_pack_dw_weight from the
 *     reference's (C,1,k,k) parameter layout; biases float32 (C).  A RecConv2d parameter pack is
 *     (level+2) such blocks back to back: [down, convs[0], ..., convs[level]]  (convs[0] pairs with
 *     the coarsest level, model/recnext.py:32-34).
 *   - The library never allocates, frees, synchronises or retains pointers; all work is enqueued
 *     on the hipStream_t passed as `stream` (NULL = the default stream).  Calls are re-entrant.
 *   - Return value: 0 ok; <0 argument / unsupported-configuration error; >0 a hipError_t.
 *     rcx_last_error() returns a thread-local message for the last non-zero return.
 */
#ifndef RECNEXT_AMD_H
#define RECNEXT_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCX_ABI_VERSION 7

enum { RCX_DTYPE_F32 = 0, RCX_DTYPE_BF16 = 1, RCX_DTYPE_F16 = 2 };   /* F16: the reference's autocast dtype (engine.py:48) */
enum { RCX_MODE_BILINEAR = 0, RCX_MODE_NEAREST = 1 };   /* F.interpolate(mode=...), model/recnext.py:33 */
enum { RCX_MAX_LEVEL = 8 };

enum {
    RCX_ERR_BAD_ARG = -1,       /* null pointer, non-positive extent, even k, level out of range */
    RCX_ERR_UNSUPPORTED = -2,   /* configuration outside what the kernels implement */
    RCX_ERR_WORKSPACE = -3      /* workspace smaller than rcx_recconv2d_fwd_workspace_bytes() */
};

int rcx_abi_version(void);
const char* rcx_last_error(void);
/* The RCX_* environment switches (A/B measurement knobs) are read once, at the first call that consults one; this
 * re-reads them.  For tests and A/B tools that flip a switch inside one process; not to be called while another thread is in the library. */
void rcx_reload_options(void);

/* Start-up self-test of one hardware behaviour the 16-bit load paths rely on and the ISA does not promise: a D16 "hi" load
 * (global_load_short_d16_hi, buffer_load_short_d16_hi, ds_read_u16_d16_hi) ZEROES the other half of its destination register on gfx950, so a
 * bf16 element lands in float32 position with no conversion (rcx_cpt_kernel.h row loads, rcx_cpl14_pieces.h, rcx_lanes.h).  One 64-lane launch:
 * reads the 64 16-bit values at `src`, ORs a failure mask into *flag (caller-zeroed uint32: bit 0 global, 1 buffer, 2 LDS form); asynchronous on
 * `stream` -- the caller reads *flag after synchronising.  The Python binding runs it once per device before its first 16-bit launch and raises
 * if the flag is not 0.  Replaces nothing in the reference. */
int rcx_selftest_d16(const void* src, void* flag, void* stream);

/* Description of the kernel schedule rcx_recconv2d_fwd would use for this problem, e.g. "generic" or
 * "plane(cb=16,band8,nt=512,lds=157760)"; thread-local storage, valid until the next call on this thread. */
const char* rcx_recconv2d_fwd_plan(int N, int C, int H, int W, int level, int k, int mode, int dtype);

/* Repack one depthwise weight (C,1,k,k) [dtype f32|bf16|f16] -> float32 (k,k,C).
 * Replaces nothing in the reference (layout plumbing for nn.Conv2d(groups=C).weight, model/recnext.py:21-22). */
int rcx_pack_dw_weight(const void* w_ckk, float* dst_kkc, int C, int k, int dtype, void* stream);
/* All parameters of one RecConv2d block in one launch: w[j] (C,1,k,k) and b[j] (C) [dtype f32|bf16|f16], j = 0 .. count-1 in pack
 * order [down, convs[0], ..., convs[level]] (HOST arrays of DEVICE pointers; b may be NULL, b[j] may be NULL) -> wpack (count,k,k,C),
 * wpack_flipped (the same with every k x k rotated by 180 degrees: the taps rcx_recconv2d_bwd wants; may be NULL) and bpack (count,C)
 * (may be NULL), all float32.  A training step repacks after every optimizer step (engine.py:38-71): this replaces 2*(level+2)
 * rcx_pack_dw_weight / rcx_pack_bias launches and the flip. */
int rcx_pack_recconv_params(const void* const* w, const void* const* b, float* wpack, float* wpack_flipped, float* bpack,
                            int count, int C, int k, int dtype, void* stream);
/* The inverse layout change for the gradients: gwpack (count,k,k,C) float32 -> gw[j] (C,1,k,k) float32 contiguous, one launch. */
int rcx_unpack_recconv_grads(const float* gwpack, void* const* gw, int count, int C, int k, void* stream);
/* Convert one bias vector (C) [dtype f32|bf16|f16] -> float32 (C). */
int rcx_pack_bias(const void* b, float* dst, int C, int dtype, void* stream);

/* Scratch bytes rcx_recconv2d_fwd needs for this problem (may be 0). */
size_t rcx_recconv2d_fwd_workspace_bytes(int N, int C, int H, int W, int level, int k, int dtype);

/*
 * RecConv2d.forward -- model/recnext.py:24-34 (down ladder :27-29, up recursion :31-33, final conv :34).
 *   x, y    : N x H x W x C activations of `dtype` (y may not alias x)
 *   wpack   : (level+2, k, k, C) float32, [down, convs[0..level]]
 *   bpack   : (level+2, C) float32 or NULL (bias=False)
 *   level>=0, k odd, mode RCX_MODE_*.
 */
int rcx_recconv2d_fwd(const void* x, void* y, const float* wpack, const float* bpack,
                      void* workspace, size_t workspace_bytes,
                      int N, int C, int H, int W, int level, int k, int mode, int dtype, void* stream);

/*
 * Measurement hook (nothing in the reference; bench.py's roofline object): the next rcx_recconv2d_fwd of this thread that runs as ONE fused kernel
 * (plan "cpt(...)" or "cpl(...)": the 56x56 / 28x28 / 14x14 / 7x7 blocks) has the two HIP events recorded by the command processor at the kernel's
 * start and end (hipExtLaunchKernelGGL) -- its duration as rocprofv3 sees it, without the dispatch gaps an event pair recorded around the call
 * brackets, and without delaying the next kernel.  Either may be NULL.  Cleared when that call returns, consumed or not
 * (rcx_launch_events_pending() == 1 after arming, 0 once consumed or cleared).
 */
int rcx_time_next_launch(void* start_event, void* stop_event);
int rcx_launch_events_pending(void);

/*
 * Training (engine.py:48-64): a forward that keeps the fp32 pyramid F_1..F_L, C_1..C_L in `saved`, and the backward pass.
 *   rcx_recconv2d_fwd_train   same contract as rcx_recconv2d_fwd (the blocks of RecNeXt at 224x224 run their inference launch, which then also
 *                             writes the pyramid; other shapes the per-level schedule); `saved` must hold
 *                             rcx_recconv2d_train_saved_bytes() and stay untouched until rcx_recconv2d_bwd has run.
 *   rcx_recconv2d_bwd         gy: N x H x W x C of `gy_dtype` (dL/dy): float32 always works; the block's own 16-bit `dtype` -- dL/dy exactly as
 *                             autograd hands it over under autocast, no float32 copy -- where rcx_recconv2d_bwd_gy_dtype() returns it (the
 *                             tiled 56x56 / level 4 and 28x28 / level 3 backward);  gx: N x H x W x C of `dtype` (dL/dx);
 *                             wpack_flipped: wpack with every k x k tap block rotated by 180 degrees (transpose convs);
 *                             gwpack: (level+2, k, k, C) float32 = dL/d[down, convs[0..level]] in the packed layout, the
 *                             shared down weight accumulated over all levels (model/recnext.py:21,28);
 *                             gbpack: (level+2, C) float32 or NULL.
 *                             gw_out / gb_out (HOST arrays of level+2 DEVICE pointers, or NULL): if gw_out is given the gradients leave in the
 *                             PARAMETERS' own layout and type instead -- gw_out[i] (C,1,k,k) contiguous, gb_out[i] (C) (gb_out may be NULL when
 *                             the block has no bias), elements of `grad_dtype` -- written by the final reduction itself (no unpack launch, no
 *                             dtype copy); gwpack / gbpack are then not written and may be NULL.
 *                             Needs C % 4 == 0.  Deterministic (no atomics).
 *   rcx_recconv2d_bwd_gy_dtype  the element type rcx_recconv2d_bwd wants gy in for this problem: `dtype` where a 16-bit gy is read as it is,
 *                             else RCX_DTYPE_F32.
 */
size_t rcx_recconv2d_train_saved_bytes(int N, int C, int H, int W, int level, int k);
size_t rcx_recconv2d_bwd_workspace_bytes(int N, int C, int H, int W, int level, int k);
int rcx_recconv2d_fwd_train(const void* x, void* y, const float* wpack, const float* bpack, void* saved, size_t saved_bytes,
                            int N, int C, int H, int W, int level, int k, int mode, int dtype, void* stream);
int rcx_recconv2d_bwd_gy_dtype(int N, int C, int H, int W, int level, int k, int dtype);
int rcx_recconv2d_bwd(const void* x, const void* gy, int gy_dtype, const float* wpack, const float* wpack_flipped, const void* saved,
                      void* gx, float* gwpack, float* gbpack, void* const* gw_out, void* const* gb_out, int grad_dtype,
                      void* workspace, size_t workspace_bytes,
                      int N, int C, int H, int W, int level, int k, int mode, int dtype, void* stream);

/*
 * Depthwise conv, zero padding k/2, stride 1 or 2 -- nn.Conv2d(groups=C) as used at
 * model/recnext.py:21-22 and by RecAttn2d's ConvNorm(dw k5 s2) after BN folding (model/recattn.py:61, :89-111).
 *   x: N x H x W x C (in_dtype);  y: N x Ho x Wo x C (out_dtype), Ho = (H + 2*(k/2) - k)/stride + 1.
 */
int rcx_dwconv2d_fwd(const void* x, void* y, const float* w_kkc, const float* bias,
                     int N, int C, int H, int W, int k, int stride,
                     int in_dtype, int out_dtype, void* stream);

/*
 * Depthwise conv with channel multiplier 2: nn.Conv2d(Cin, 2*Cin, k, stride, padding=k/2, groups=Cin), i.e. output
 * channel o reads input channel o/2 -- Downsample.token_mixer, model/recnext.py:165 / model/recattn.py:178
 * (with the eval-mode BatchNorm that follows it, :166/:170, folded into w/bias by the caller).
 *   x: N x H x W x Cin ; y: N x Ho x Wo x 2*Cin ; w: (k,k,2*Cin) float32 packed ; x and y share `dtype`.
 */
int rcx_dwconv2d_mult2_fwd(const void* x, void* y, const float* w_kkc, const float* bias,
                           int N, int Cin, int H, int W, int k, int stride, int dtype, void* stream);

/*
 * y = dwconv_k(x + resize(coarse -> (H,W), mode)) -- the body of one up-recursion step
 * (model/recnext.py:33-34) and the tail of RecAttn2d.forward (model/recattn.py:67).
 *   x: N x H x W x C (x_dtype); coarse: N x Hc x Wc x C (coarse_dtype) or NULL (then y = dwconv_k(x));
 *   y: N x H x W x C (out_dtype).
 */
/* Which kernel rcx_upadd_dwconv_fwd would run ("upadd_cpt(k_upadd_cpt<mode, pitch>,...)", "upadd_cpl14(...)", "upadd_lanes(...)",
 * "conv5_lanes(...)", "generic"); has_coarse = 0 for the plain conv (coarse == NULL).  Thread-local storage, valid until the next call. */
const char* rcx_upadd_dwconv_fwd_plan(int N, int C, int H, int W, int Hc, int Wc, int k, int mode,
                                      int x_dtype, int coarse_dtype, int out_dtype, int has_coarse);
int rcx_upadd_dwconv_fwd(const void* x, const void* coarse, void* y, const float* w_kkc, const float* bias,
                         int N, int C, int H, int W, int Hc, int Wc, int k, int mode,
                         int x_dtype, int coarse_dtype, int out_dtype, void* stream);

/*
 * Backward of rcx_dwconv2d_fwd (a depthwise ConvNorm.conv of RecAttn2d in a training step, engine.py:48-64): what autograd
 * derives for nn.Conv2d(groups=C, padding=k/2, stride 1|2), model/recattn.py:61,64 and model/recnext.py:21-22.
 *   x: N x H x W x C (x_dtype); gy: N x Ho x Wo x C float32; w_kkc / w_flipped_kkc: (k,k,C) float32 packs, the second with
 *   both tap axes reversed; gx: like x, may be NULL; gw: (k,k,C) float32, gb: (C) float32 or NULL -- both overwritten.
 *   workspace: rcx_dwconv2d_bwd_workspace_bytes(C, k) bytes.  Deterministic.  C must be a multiple of 4.
 */
size_t rcx_dwconv2d_bwd_workspace_bytes(int C, int k);
int rcx_dwconv2d_bwd(const void* x, const float* gy, const float* w_kkc, const float* w_flipped_kkc,
                     void* gx, float* gw, float* gb, void* workspace, size_t workspace_bytes,
                     int N, int C, int H, int W, int k, int stride, int x_dtype, void* stream);

/*
 * Backward of rcx_upadd_dwconv_fwd with a coarse plane: what autograd derives for the last line of RecAttn2d.forward,
 * `conv(x + F.interpolate(a, size=x.shape[2:], mode))` (model/recattn.py:67; also model/recnext.py:33-34), in a training step (engine.py:48-64).
 *   x: N x H x W x C (`dtype`); coarse: N x Hc x Wc x C float32 (the forward's `a`); gy: N x H x W x C of `gy_dtype` -- float32 always works, the
 *   16-bit `dtype` itself where rcx_upadd_dwconv_bwd_gy_dtype() returns it (56x56 / 28x28 planes with an exact 2x coarse plane: the tiled adjoint
 *   kernels of rcx_recconv2d_bwd); w_kkc / w_flipped_kkc as for rcx_dwconv2d_bwd.
 *   gx: like x (= K^T gy) or NULL; gcoarse: N x Hc x Wc x C float32 (= R^T K^T gy) or NULL; gw: (k,k,C) float32 (= <x + R(coarse), gy>), gb: (C) float32
 *   or NULL -- all overwritten.  workspace: rcx_upadd_dwconv_bwd_workspace_bytes() bytes.  Deterministic.  C must be a multiple of 4.
 */
size_t rcx_upadd_dwconv_bwd_workspace_bytes(int N, int C, int H, int W, int Hc, int Wc, int k);
int rcx_upadd_dwconv_bwd_gy_dtype(int N, int C, int H, int W, int Hc, int Wc, int k, int dtype);
int rcx_upadd_dwconv_bwd(const void* x, const float* coarse, const void* gy, int gy_dtype, const float* w_kkc, const float* w_flipped_kkc,
                         void* gx, float* gcoarse, float* gw, float* gb, void* workspace, size_t workspace_bytes,
                         int N, int C, int H, int W, int Hc, int Wc, int k, int mode, int dtype, void* stream);

/*
 * Backward of rcx_dwconv2d_mult2_fwd with stride 2 (Downsample.token_mixer, model/recnext.py:165, in a training step).
 *   x: N x H x W x Cin (dtype); gy: N x Ho x Wo x 2*Cin float32; w_kkc: (k,k,2*Cin) float32; gx: like x or NULL;
 *   gw: (k,k,2*Cin) float32, gb: (2*Cin) float32 or NULL; workspace: rcx_dwconv2d_bwd_workspace_bytes(2*Cin, k) bytes.
 *   k in {3,5,7}, Cin even.  Deterministic.
 */
int rcx_dwconv2d_mult2_bwd(const void* x, const float* gy, const float* w_kkc, void* gx, float* gw, float* gb,
                           void* workspace, size_t workspace_bytes, int N, int Cin, int H, int W, int k, int dtype, void* stream);

/*
 * Linear-attention core of RecAttn2d's coarse level: everything after the grouped 1x1 `qk` conv of
 * LinearAttention1.forward (model/recattn.py:21-28) / LinearAttention2.forward (:44-51; the same function):
 *     q = elu(qpre)+1, k = elu(kpre)+1; out = q^T (k v^T) / (n * (q^T mean_n(k) + 1e-6)) + pe
 *   qpre, kpre: B x n x C pre-activations of the q / k halves of the `qk` conv (BatchNorm folded, bias added);
 *   v: B x n x C, the attention input itself (:22); pe: B x n x C, output of the `pe` depthwise 3x3 ConvNorm (:27);
 *   out: B x n x C.  n = h*w tokens, token-major (= NHWC); head h owns channels [h*C/heads, (h+1)*C/heads).
 *   All five tensors share `dtype`; C/heads at most 64 (at most 32 unless it is a multiple of 4).
 */
int rcx_linear_attention_fwd(const void* qpre, const void* kpre, const void* v, const void* pe, void* out,
                             int B, int n, int C, int heads, int dtype, void* stream);
/*
 * The same with `pe` computed where it is used (round 3): pe = depthwise 3x3 conv (pad 1) of v + bias -- LinearAttention's `pe`
 * ConvNorm (model/recattn.py:14, :27 / :50) with its BatchNorm folded -- from the (3, 3, C) float32 pack of rcx_pack_dw_weight and the
 * (C) bias pack (or NULL); v is the H x W plane in NHWC.  One launch and one tensor less than rcx_dwconv2d_fwd + rcx_linear_attention_fwd.
 * RCX_ERR_UNSUPPORTED unless C/heads is a multiple of 4 (every head of the A-series is 32 wide) and the call runs on the vector-pipe kernel
 * (fewer than 512 tokens for 32-wide heads and 16-bit I/O: on the matrix-core kernel the fused form measured slower and is not offered).
 */
int rcx_linear_attention_pe_fwd(const void* qpre, const void* kpre, const void* v, const float* w_pe_kkc, const float* b_pe, void* out,
                                int B, int H, int W, int C, int heads, int dtype, void* stream);

/*
 * RecAttn2d's whole coarse level on the matrix cores, for 16-bit-activation runs (round 4): the grouped 1x1 `qk` conv (model/recattn.py:13 / :36
 * with its BatchNorm folded, :21 / :44), the activation, k v^T, the normaliser, q kv and + pe (:22-27 / :45-50) -- what rcx_linear_attention_pe_fwd
 * does plus the projection that used to be two library GEMMs; q and k never exist in memory.
 *   d      : B x (H W) x C float32, the output of the stride-2 ConvNorm (:61), NHWC; also v and the input of `pe`;
 *   wqk    : (2C) x (C/2) bf16, rows [0, C) = the q half of the conv's weight, rows [C, 2C) = the k half (BatchNorm folded in float32, then rounded);
 *   bqk    : (2C) float32 biases; w_pe_kkc / b_pe: as rcx_linear_attention_pe_fwd (b_pe may be NULL);  out: B x (H W) x C float32;
 *   workspace: rcx_recattn_qkcore_workspace_bytes(...) bytes (0 bytes / may be NULL when one launch does it).  Every pointer 16-byte aligned.
 * 1, 2, 4, 8 or 16 heads of dimension C / heads = 32, or of 4, 8 .. 28 when heads is even (a head is padded to 32 inside the kernels; memory stays
 * compact: RecNeXt-A0 / A1 / A2's 20 / 24 / 28).  Planes of at most 64 tokens whose image fits the CU's LDS: ONE launch, a workgroup
 * per image, a wave per head (16 heads: at most 32 tokens; RecNeXt-A's stages 2 and 3 at 224 x 224).  Other planes: TWO launches (the k^T v partial
 * sums of every image go through the workspace, summed in a fixed order: deterministic), at most 8 heads.  rcx_recattn_qkcore_launches() tells which
 * (0: no kernel for the shape -> the call returns RCX_ERR_UNSUPPORTED).  The products run on the matrix cores with bf16 operands and float32
 * accumulation, everything else is float32: meant for 16-bit activations (results within north_star's 1e-2 of the float32 forward); float32 callers
 * keep rcx_linear_attention_pe_fwd.
 */
int rcx_recattn_qkcore_launches(int B, int H, int W, int C, int heads);
size_t rcx_recattn_qkcore_workspace_bytes(int B, int H, int W, int C, int heads);
int rcx_recattn_qkcore_fwd(const float* d, const void* wqk_bf16, const float* bqk, const float* w_pe_kkc, const float* b_pe, float* out,
                           void* workspace, size_t workspace_bytes, int B, int H, int W, int C, int heads, void* stream);

/*
 * The same preceded by RecAttn2d's stride-2 depthwise 5x5 ConvNorm (model/recattn.py:12 / :61, BatchNorm folded) in the same launch: x -> d stays in
 * LDS, d never exists in memory.  x: B x H x W x C bf16 / f16 NHWC; w_down_kkc (5,5,C) / b_down (C, may be NULL) float32 as rcx_dwconv2d_fwd takes
 * them; the rest as rcx_recattn_qkcore_fwd; out: B x ceil(H/2) x ceil(W/2) x C float32.  Planes 14 x 14 (1 .. 8 heads) and 7 x 7 (1 .. 16 heads) only --
 * RecNeXt-A's stages 2 and 3 at 224 x 224 -- rcx_recattn_down_qkcore_supported() says (1 / 0); else RCX_ERR_UNSUPPORTED.
 */
int rcx_recattn_down_qkcore_supported(int B, int H, int W, int C, int heads, int x_dtype);
int rcx_recattn_down_qkcore_fwd(const void* x, const float* w_down_kkc, const float* b_down, const void* wqk_bf16, const float* bqk,
                                const float* w_pe_kkc, const float* b_pe, float* out, int B, int H, int W, int C, int heads, int x_dtype, void* stream);

/*
 * RecAttn2d.forward whole (model/recattn.py:54-67 in eval mode, BatchNorms folded): y = ConvNorm_k5(x + interpolate(LinearAttention(ConvNorm_k5s2(x)),
 * size = x's, mode = nearest)) in ONE launch -- a workgroup per image, x's plane read twice (the stride-2 conv, the final conv), d and the attention
 * output only ever in LDS.  x, y: B x H x W x C bf16 / f16 NHWC; w_down_kkc / b_down, w_conv_kkc / b_conv: (5,5,C) / (C) float32 packs (biases may be
 * NULL); wqk / bqk / w_pe_kkc / b_pe as rcx_recattn_qkcore_fwd.  Planes 14 x 14 and 7 x 7, 1 .. 8 heads of 32 channels, RCX_MODE_NEAREST;
 * rcx_recattn2d_fwd_supported() says (1 / 0); else RCX_ERR_UNSUPPORTED -- the caller then chains the entry points above and rcx_upadd_dwconv_fwd.
 */
int rcx_recattn2d_fwd_supported(int B, int H, int W, int C, int heads, int mode, int dtype);
int rcx_recattn2d_fwd(const void* x, void* y, const float* w_down_kkc, const float* b_down, const void* wqk_bf16, const float* bqk,
                      const float* w_pe_kkc, const float* b_pe, const float* w_conv_kkc, const float* b_conv,
                      int B, int H, int W, int C, int heads, int mode, int dtype, void* stream);

/*
 * The channel mixer of a MetaNeXtBlock / Downsample with the residual add around it, inference, ONE launch (model/recnext.py:125-132 `mlp`,
 * :157-158 `x + drop_path(channel_mixer(norm(token_mixer(x))))`, :169-171; model/recattn.py:171, :184), both 1x1 ConvNorms BN-folded (:75-97):
 *     y[m][:] = x[m][:] + W2 gelu(W1 z[m][:] + b1) + b2        m = 0 .. M-1 tokens (M = N H W of an NHWC tensor), gelu = the exact (erf) form
 *   z, x, y : M x C bf16 (z = the token mixer's output, x = the block's input; z == x for Downsample); y may alias neither;
 *   wfrag   : rcx_channel_mlp_pack_bytes(C, H) bytes -- W1 (H x C) and W2 (C x H) rounded to bf16 and laid out as the matrix-core fragments the kernel
 *             reads (recnext_amd/ops.py::pack_channel_mlp builds it; layout in recnext_amd/csrc/rcx_mlp.hip); 16-byte aligned;
 *   bias    : 32 (H/32 + ceil(C/32)) floats: b1 (H), then b2 padded with zeros to a multiple of 32.
 * H % 32 == 0 (pad the hidden layer with zero units: gelu(0) = 0 meets zero weights), C % 8 == 0, M C 2 < 2^31, the weights must fit the LDS.
 * rcx_channel_mlp_supported() says whether there is a kernel for (C, H): today C = 40 / 48 with H = 96, 56 / 64 with 128, 80 / 160, 96 / 192, 128 / 256 (weights resident in LDS) and
 * C = 128 / 256, 160 / 320, 192 / 384, 256 / 512, 320 / 640 (weights streamed through LDS; W2 / b2 padded to an even number of 32-row output tiles when C > 128);
 * else RCX_ERR_UNSUPPORTED and the caller keeps the GEMM library.  The products run on the matrix cores with bf16 operands (the hidden activations are
 * rounded to bf16 once, after the GELU) and float32 accumulation.
 */
int rcx_channel_mlp_supported(int M, int C, int H, int dtype);
size_t rcx_channel_mlp_pack_bytes(int C, int H);
int rcx_channel_mlp_fwd(const void* z, const void* x, void* y, const void* wfrag, const float* bias, int M, int C, int H, int dtype, void* stream);

/*
 * RecNextStem, inference, ONE launch (model/recnext.py:134-146; both 3x3 stride-2 ConvNorms BN-folded, :75-97; nn.GELU between them):
 *     y = conv3x3_s2_p1(gelu(conv3x3_s2_p1(x) + b1)) + b2
 *   x  : N x H x W x 3 bf16 (NHWC);  y : N x ceil(ceil(H/2)/2) x ceil(ceil(W/2)/2) x CO bf16 (NHWC);
 *   w1frag : 2 ceil(CM/32) KB, the first conv's (CM, 3, 3, 3) weight rounded to bf16 as matrix-core fragments (K = 27 taps padded to 32);  b1 : 32 ceil(CM/32) float32;
 *   w2frag : rcx_stem_pack_bytes(CM, CO) bytes, the second conv's (CO, CM, 3, 3) weight likewise (recnext_amd/ops.py::pack_stem builds all four);
 *   b2 : 32 ceil(CO/32) float32 (zero padded).  All four 16-byte aligned.  CM in {20, 24, 28, 32, 40} (the registered models' embed_dim[0] / 2), CO % 4 == 0, CO <= 96.
 * The 112 x 112 x CM intermediate exists only as 17 x 17 tiles in LDS, rounded to bf16 once (as the library path does).  rcx_stem_supported() says whether the
 * shape has a kernel; else RCX_ERR_UNSUPPORTED and the caller keeps the conv library.
 */
int rcx_stem_supported(int N, int H, int W, int CM, int CO, int dtype);
size_t rcx_stem_pack_bytes(int CM, int CO);
int rcx_stem_fwd(const void* x, void* y, const void* w1frag, const float* b1, const void* w2frag, const float* b2, int N, int H, int W, int CM, int CO, int dtype, void* stream);

/*
 * Backward of rcx_linear_attention_fwd (the gradients engine.py:48-64 needs through RecAttn2d, model/recattn.py:16-28 / :39-51):
 *   given gout = dL/dout (B x n x C), writes gq = dL/dqpre, gk = dL/dkpre, gv = dL/dv (all B x n x C, `dtype`); dL/dpe = gout is the
 *   caller's.  float32 arithmetic, deterministic (fixed summation order).  C/heads at most 64.
 */
int rcx_linear_attention_bwd(const void* qpre, const void* kpre, const void* v, const void* gout, void* gq, void* gk, void* gv,
                             int B, int n, int C, int heads, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RECNEXT_AMD_H */
