/*
 * rpnet_abi.h — C ABI of librpnet_hip.so, the MI355X (gfx950) kernels behind the
 * RP-Net hot path (uci-cbcl/RP-Net: net/rp_net.py, net/unet.py, net/modules.py).
 *
 * The reference has no native code and no FFI: the path is ordinary nn.Module code on
 * stock torch operators.  Each entry point below therefore names the reference
 * operator sequence (file:line under /root/reference) that it replaces; the
 * reference-side binding a maintainer would add is the ctypes stub shown in
 * INTEGRATION.md (rpnet_amd/hip.py is exactly that stub).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller;
 *     the library never allocates, frees or synchronises; work is enqueued on `stream`.
 *   - activations are fp32 NHWC ("pixel-major": [N][H][W][C]); the module boundary
 *     tensors of the reference (images [B,1,H,W], masks [B,H,W], logits [B,K,H,W]) are
 *     read/written in their own NCHW layout by the kernels that touch them.
 *   - "groups": G consecutive equal slices of the N images that keep separate
 *     BatchNorm batch statistics (the reference calls the encoder once for the support
 *     images and once for the query images, net/rp_net.py:248,257 — one launch here,
 *     two statistic groups, two running-stat updates in call order).
 *   - the 3x3 convolutions, their weight gradients and the local correlation also exist on "split-bf16"
 *     operands (an fp32 value as 3 exact bf16 planes, 6 partial products on the bf16 matrix pipe, fp32
 *     accumulation: fp32 accuracy at 16/6 of the fp32 matrix rate) — the default arithmetic of the host side;
 *     see rpnet_split_bf16 and the split_planes field of rpnet_conv_desc.
 *   - return value: 0 on success, a negative rpnet_status, or a positive hipError_t;
 *     rpnet_last_error_string() gives thread-local text for the last failure.
 *   - re-entrant: no global mutable state.
 */
#ifndef RPNET_ABI_H
#define RPNET_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rpnet_stream_t; /* hipStream_t */

enum rpnet_status {
    RPNET_OK = 0,
    RPNET_ERR_SHAPE = -1,   /* a dimension violates the kernel's tiling contract */
    RPNET_ERR_ARG = -2,     /* null / inconsistent argument */
    RPNET_ERR_WORKSPACE = -3
};

/* Version of this header.  History: 100 rounds 1 - 3; 104 round 4 (rpnet_bn_relu / rpnet_bn_bwd / rpnet_conv1_wgrad_bn gained
 * arguments in front of `stream`, rpnet_conv_desc grew skip_*, tile_skip, y_enc — a caller built against 100 would pass its stream
 * where a pointer is expected: hence the number); 105 round 5 (rpnet_refine_glue_*, rpnet_conv_up4*, rpnet_conv_wgrad_up4*,
 * rpnet_upconv_collapse_weights added; nothing existing changed).  A caller MUST zero-initialise rpnet_conv_desc (fields added
 * later are optional features that are off at zero) and SHOULD compare rpnet_version() with the RPNET_ABI_VERSION it was built
 * against. */
#define RPNET_ABI_VERSION 108
int rpnet_version(void);
const char* rpnet_last_error_string(void);

/* ------------------------------------------------------------------ weight packing
 * nn.Conv2d weight [Cout][Cin][kh][kw] (state_dict layout, net/modules.py:47) ->
 *   wp  [taps][Cin_pad/4][Cout][4]  forward  implicit-GEMM B operand (k-quads interleaved)
 *   wd  [taps][Cout/4][Cin_pad][4]  dgrad B operand: taps flipped, Cin/Cout swapped
 * `cin_off`/`cin_pad`: input channel c of the weight lands on packed row cin_off + c of
 * cin_pad rows; the other rows are zero (the 121 correlation channels are padded to 128,
 * net/rp_net.py:65-69,81).  wd may be NULL. taps = 9 (3x3) or 1 (1x1). */
int rpnet_pack_conv_weight(const float* w, float* wp, float* wd, int cout, int cin, int taps,
                           int cin_off0, int cin_split, int cin_off1, int cin_pad, rpnet_stream_t stream);

/* ------------------------------------------------------- split-bf16 operands (fp32-accurate)
 * The bf16 matrix pipe of gfx950 is 16x the fp32 one.  An fp32 value x is carried as NP bf16 planes
 * x = h + m (+ l), h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (exact: 3 x 8 significand bits),
 * and a product x*y as the partial products of the planes, accumulated in fp32 by
 * v_mfma_f32_32x32x16_bf16:
 *   NP = 3: hh + hm + mh + hl + lh + mm   (dropped terms <= 2^-23 |x*y|: fp32 round-off level), 6 MFMAs
 * Two planes are FP16 planes (v_mfma_f32_32x32x16_f16, same rate): x / s = h + l, h = fp16(x / s), l = fp16(x / s - h)
 * with a power-of-two scale s (tensor scale of an activation / gradient, row scale of a weight) that maps a rigorous
 * bound of the operand to <= 2^15 — 22 significand bits for everything within 2^-18 of the bound, an absolute floor of
 * 2^-40 of the bound below — and a product is hh + hl + lh (dropped term <= 2^-22 |x*y|), 3 MFMAs; the accumulator is
 * multiplied by the scales in the epilogue (exact).  Used where such a bound exists (every 3x3 convolution of the
 * training step: BatchNorm outputs, BatchNorm gradients, weights); three bf16 planes everywhere else.
 * rpnet_split_bf16: x [rows][C] fp32 (optionally times a per-row factor s or 1-s: x*mask of
 * net/rp_net.py:275,283) -> out [3][rows][C] bf16.  C % 8 == 0.
 * rpnet_split_f16: the same into two fp16 planes of x f(mask) / s, s = max(*s_a, *s_b) (device scalars: the tensor
 * scale(s) the producer(s) published — rpnet_bn_relu; s_b NULL or the second source of a concatenation), *s_out = s.
 * rpnet_pack_conv_weight_split with planes == 2 also writes the row scales row_scale_wp [Cout] and row_scale_wd
 * [cin_pad] (preset the padding rows to 1) = rpnet_conv_desc.acc_scale_col of the forward / dgrad launch.
 * rpnet_pack_conv_weight_split: as rpnet_pack_conv_weight, into
 *   wp [planes][taps][Cin_pad/32][Cout][32]   (k = input channel, contiguous per output channel)
 *   wd [planes][taps][Cout/32][Cin_pad][32]   (dgrad: k = output channel, taps flipped); wd may be NULL.
 * cin_pad, cout multiples of 32; channel ranges that are not multiples of 8 (the 377 = 121 + 256 channels of the 1x1
 * convolution, packed as 128 + 256) take a gathering form of the kernel. */
int rpnet_split_bf16(const float* x, const float* scale, int scale_mode, void* out, size_t rows, int C, int planes,
                     rpnet_stream_t stream);
/* *s_out = the power-of-two tensor scale for a tensor bounded by *bound (maps the bound to <= 2^15); see out_absmax */
/* Predicted fp16 tensor scales for eval-mode layers (running statistics give no a-priori bound, net/modules.py:48): slot i of
 * `measured` holds max |output| of layer launch i of the call that just ended (rpnet_conv_desc.out_absmax).  For i < n:
 * if check != 0 and measured[i] > bound[i] (the bound the call RAN with), *violations is incremented — the caller must redo
 * the call on measured scales; then bound[i] = pow2ceil(measured[i] * safety), scale[i] = bound[i] * 2^-15 for the next call. */
int rpnet_predict_scales(const float* measured, float* bound, float* scale, int n, float safety, int check, int* violations,
                         rpnet_stream_t stream);
int rpnet_pow2_scale(const float* bound, float* s_out, rpnet_stream_t stream);
int rpnet_split_f16(const float* x, const float* mask, int mask_mode, const float* s_a, const float* s_b, float* s_out,
                    void* out, size_t rows, int C, int planes /* 2, or 1: plain fp16 (RPNET_CONV_MATH=f16) */,
                    int a_is_bound /* 1: *s_a is a measured bound of |x| (rpnet_conv_desc.out_absmax), s = its power-of-two
                                      scale; s_b must be NULL */,
                    rpnet_stream_t stream);
int rpnet_pack_conv_weight_split(const float* w, void* wp, void* wd, int cout, int cin, int taps, int cin_off0,
                                 int cin_split, int cin_off1, int cin_pad, int planes, float* row_scale_wp,
                                 float* row_scale_wd, rpnet_stream_t stream);
/* the same for up to 24 layers in ONE launch per kernel (a training step repacks every layer's weights: 2 launches
 * instead of 2 per layer).  `items` is a HOST array of n descriptors (device pointers inside), copied by value. */
typedef struct rpnet_pack_item {
    const float* w; void* wp; void* wd;        /* as rpnet_pack_conv_weight_split; wd may be NULL */
    float* row_scale_wp; float* row_scale_wd;  /* planes 1 / 2: [cout] / [cin_pad] outputs; NULL for planes == 3 */
    int cout, cin, taps, cin_off0, cin_split, cin_off1, cin_pad;   /* taps: 9, 1, or 4 (the collapsed up_conv weights) */
} rpnet_pack_item;
int rpnet_pack_conv_weights_split(const rpnet_pack_item* items, int n, int planes, rpnet_stream_t stream);

/* ------------------------------------------------------------- conv (implicit GEMM)
 * Replaces nn.Conv2d(k=3,s=1,p=1,bias=True) / nn.Conv2d(k=1) forward and its
 * autograd input-gradient (net/modules.py:47,50,67; net/rp_net.py:51,56,66), fp32 MFMA
 * (v_mfma_f32_32x32x2_f32).  One descriptor drives forward and dgrad:
 *   A operand  = input pixels, gathered from up to two NHWC sources that are
 *                concatenated along C (torch.cat skip connections, net/unet.py:460,464;
 *                torch.cat([corr, fm1]), net/rp_net.py:81), optionally through a
 *                nearest x2 up-sampling (nn.Upsample, net/modules.py:66) and optionally
 *                scaled per input pixel by s or 1-s (x*mask, x*(1-mask), net/rp_net.py:275,283);
 *   B operand  = packed weights;
 *   epilogue   = +bias, optional per-channel affine+ReLU (eval-mode BatchNorm+ReLU,
 *                net/modules.py:48-49), optional per-output-pixel scale and accumulate
 *                (dgrad of the two masked CRE branches into one tensor), split of the
 *                output channels over two destinations (dgrad of a concat). */
typedef struct rpnet_conv_desc {
    const float* x0; const float* x1;  /* NHWC sources, C0 + C1 = Cin (x1 NULL if C1 == 0) */
    int C0, C1;
    const float* w;                    /* packed [taps][Cin/4][Cout][4] */
    const float* bias;                 /* [Cout] or NULL */
    const float* in_scale;             /* [N*Hin*Win] or NULL */
    int in_scale_mode;                 /* 0 none, 1: *s, 2: *(1-s) */
    float* y0; float* y1;              /* NHWC destinations, Co0 + Co1 = Cout */
    int Co0, Co1;
    const float* ep_scale; const float* ep_shift; /* [G][Cout] eval affine or NULL */
    int ep_relu;
    const float* out_scale;            /* [N*H*W] or NULL */
    int out_scale_mode;                /* 0 none, 1: *s, 2: *(1-s) */
    int accumulate;                    /* y += result */
    int N, H, W;                       /* output (= conv input after up-sampling) size */
    int taps;                          /* 9 or 1 */
    int upsample;                      /* sources are [N][H/2][W/2][C] */
    int groups;                        /* for ep_scale/ep_shift rows and the statistics below */
    int dilation;                      /* 3x3 tap spacing: 0/1 = dense, 2 = the dilated last VGG block (net/vgg.py:31) */
    double* stats_partial;             /* optional: per (M tile, channel) sum / sum-of-squares of the
                                          output, [groups * rpnet_conv_stats_blocks()][Cout][2] — the
                                          train-mode BatchNorm batch statistics fused into the epilogue */
    int split_planes;                  /* 0: x0/x1/w are fp32 (v_mfma_f32_32x32x2_f32).  3: x0/x1/w point at
                                          split-bf16 operands (rpnet_split_bf16 / rpnet_pack_conv_weight_split);
                                          2 / 1: at two / one fp16 plane(s) of operand / scale (rpnet_split_f16; 2 is
                                          fp32-equivalent, 1 is plain fp16 operands with fp32 accumulation);
                                          plane p of a source at +p*N*Hin*Win*C elements, of w at
                                          +p*taps*Cin*Cout; in_scale must already be folded into the split */
    void* y_split;                     /* optional (single destination, Co1 == 0): the final output also as split-bf16
                                          planes [split_out_planes][N*H*W][Cout] — what the next convolution reads */
    int split_out_planes;              /* 2 or 3 when y_split is set */
    /* power-of-two scales of fp16 split operands (split_planes == 2 or 1; all NULL otherwise): the accumulator is
       multiplied by acc_scale_col[column] * *acc_scale_x before the bias (column = output channel: the per-row scale of
       the packed weights, rpnet_pack_conv_weight_split; *acc_scale_x = the tensor scale of the activation operand).
       rpnet_conv_wgrad multiplies dW by *acc_scale_x * *acc_scale_dy (the scales of its two operands). */
    const float* acc_scale_col; const float* acc_scale_x; const float* acc_scale_dy;
    /* RESERVED, must be NULL / 0.  (Rounds 1 - 4 could run the reduction pass of the SOURCE layer's BatchNorm backward in this
       launch's epilogue; measured slower than the separate pass every round — the strided reads of y and the fp64 sums cost the
       matrix-bound kernel more than the pass saves — and removed in round 5; the fields stay so that the layout does not move.) */
    const float* bnb_y; const float* bnb_stats; double* bnb_partial; float* bnb_pmax; int bnb_groups;
    const float* acc_scale_x1;         /* optional (fp16 planes, two sources, rpnet_conv_fwd with taps == 1 and rpnet_conv_wgrad with
                                          taps == 1 only): the tensor scale of source x1 when it differs from source x0's
                                          (*acc_scale_x) — cat([corr, fm1]) of net/rp_net.py:81: the correlation's scale is
                                          measured per call, fm1's comes from its BatchNorm bound.  NULL: one joint scale */
    float* out_absmax;                 /* optional: *out_absmax = max(*out_absmax, max |final output value|) (device scalar the
                                          caller zeroed; an order-independent atomic max, so the result is deterministic):
                                          the data-dependent bound from which an eval-mode BatchNorm output gets its fp16
                                          tensor scale (rpnet_pow2_scale) — running statistics give no a-priori bound */
    int tune;                          /* 0: the library picks the tile variant.  Tuning / tests, low byte: v + 1 forces variant v of
                                          the split forward kernels (where the shape allows it); in rpnet_conv_wgrad: 4 / 8 = the register-staged
                                          split weight gradient (4-wave / 12-wave layout), 16 = round 5's row-major LDS-DMA kernel, 17 = the
                                          ring kernel with the shallow DMA pipeline, 18 = the ring kernel with its DMAs through the compiler's
                                          builtin (diagnostic, slow); bits 8-10 there: ablation forms of the LDS-DMA weight-gradient kernels.
                                          Bits 8-9 in rpnet_conv_fwd: ablation switches of the LDS-DMA kernel (tools/bench_conv_split.py: 1 = first channel chunk only, 2 = no
                                          epilogue; results are then meaningless).  Bit 16: the default policy without the
                                          LDS-DMA kernel (A/B).  Carried here, not in the environment: the library keeps no
                                          global state */
    const float* y_split_scale;        /* optional with y_split and split_out_planes 2 / 1: the planes are fp16 planes of output / *y_split_scale
                                          (a power-of-two tensor scale the CALLER predicted, e.g. from the layer's maximum in the
                                          previous call: rpnet_predict_scales; out_absmax of the same launch is the check).
                                          NULL: the planes of the unscaled output (three bf16 planes, or fp16 planes of values the
                                          caller knows to be <= 2^15) */
    void* splitk_ws;                   /* optional workspace of rpnet_conv_splitk_workspace_bytes(d) bytes: lets rpnet_conv_fwd cut the */
    size_t splitk_ws_bytes;            /* K range (input channels) of a launch whose grid would leave more than half of the CUs idle
                                          (eval-mode calls at batch 2: M = 8192 ... 1024) into 2 ... 8 parts computed by separate blocks
                                          (fp32 partial outputs in the workspace) and summed, in a fixed order, by a second launch that
                                          does the epilogue (bias, ep_*, out_absmax, y_split).  NULL / too small: one block per tile */
    /* Optional: whole output tiles whose result is known to be zero before the bias skip their K loop (round 4).  The CRE's
       w_k(x * mask) / w_q(x * (1 - mask)) (net/rp_net.py:275,283) read an input that is EXACTLY zero wherever the pooled mask is 0
       (resp. 1): skip_mask [N][H][W] (the conv's resolution; no up-sampling), skip_mode 1: factor = mask, 2: factor = 1 - mask;
       skip_halo 1: the factor multiplies the INPUT (forward: a tile is skipped when the factor is zero on the tile and its
       one-pixel halo), 0: it multiplies the OUTPUT (input gradient with out_scale: zero on the tile itself).  skip_ws: caller's
       scratch of >= N*H*W / 128 bytes for the per-tile flags (rpnet_conv_fwd fills it with one small launch in front).  Only the
       LDS-DMA patch kernels honour it (others compute every tile); skipped and computed tiles are numerically equal (0 * w adds 0;
       the SIGN of a zero may differ: a negative accumulator times a zero factor is -0 where the skipped tile writes +0),
       unless a weight or gradient is Inf / NaN. */
    const float* skip_mask; int skip_mode; int skip_halo; unsigned char* skip_ws;
    const unsigned char* tile_skip;    /* internal (set by the launcher): flags [M tiles], 0 = skip */
    /* RESERVED, must be NULL / 0.  (Round 4 could write the pre-BatchNorm output as 2-byte codes here — measured neutral in time and
       worse in error on BASELINE configs[4], removed in round 5; the fields stay so that the layout does not move.) */
    const float* y_enc; int y_enc_stride;
} rpnet_conv_desc;

int rpnet_conv_fwd(const rpnet_conv_desc* d, rpnet_stream_t stream);
/* partial-statistic rows per group rpnet_conv_fwd writes for this descriptor (0: this shape cannot
 * fuse them — a group does not split into whole tiles — use rpnet_bn_stats on the output instead) */
int rpnet_conv_stats_blocks(const rpnet_conv_desc* d);
/* which tile variant of the split kernels rpnet_conv_fwd launches for this descriptor (incl. its `tune` override; -1 for
 * fp32 operands): 0-3 plain split implicit GEMM, 7 the 8-wave patch kernel, 8-9 the 4-wave patch kernels, 11 the LDS-DMA
 * patch kernel (conv_split_dma.hip).  Tests use it to assert that a forced variant actually ran. */
int rpnet_conv_tile_variant(const rpnet_conv_desc* d);
/* bytes of rpnet_conv_desc.splitk_ws with which rpnet_conv_fwd would split the K range of this launch (0: it would not — the
 * grid fills the machine, or the descriptor asks for an epilogue feature the reduce launch does not have: batch statistics,
 * BatchNorm-backward sums, accumulate, per-row output scales, a second output tensor) */
size_t rpnet_conv_splitk_workspace_bytes(const rpnet_conv_desc* d);

/* ------------------------------------------------------------- up_conv without its redundant products (round 5)
 * nn.Upsample(scale_factor=2) -> nn.Conv2d 3x3 (net/modules.py:61-75: Up5, Up4 of net/unet.py:457-465).  A 3x3 convolution over a
 * nearest-x2 up-sampled image reads only a 2 x 2 block of SOURCE pixels per output pixel; with the weights of coinciding taps
 * added up front the layer is 4 / 9 of its multiply-adds, forward, input gradient and weight gradient (csrc/conv_up4_dma.hip,
 * csrc/conv_wgrad_up4.hip).  Result: the reference's to the rounding of the weight sums (2^-24 relative).
 *   rpnet_upconv_collapse_weights: w [Cout][Cin][3][3] -> wc [4 Cout][Cin][2][2], row (py * 2 + px) * Cout + co = the weights of
 *     output phase (py, px) = (Y & 1, X & 1); pack wc with rpnet_pack_conv_weights_split (taps = 4, cout = 4 Cout).
 *   rpnet_conv_up4(d, mode): fp16 planes (split_planes 2).  d.N / d.H / d.W: the HIGH-resolution tensor.
 *     mode 1, forward: x0 = planes of the low-resolution input [N][H/2][W/2][C0], w = the `wp` pack of wc, y0 [N][H][W][Co0];
 *       bias, ep_scale / ep_shift / ep_relu, stats_partial (rpnet_conv_up4_stats_blocks rows per group), accumulate,
 *       out_absmax, acc_scale_col = the pack's row scales [4 Co0], acc_scale_x as in rpnet_conv_fwd.
 *     mode 2, input gradient: x0 = planes of dy [N][H][W][C0], w = the `wd` pack of wc, y0 = dx [N][H/2][W/2][Co0] (low resolution:
 *       the 2 x 2 sum of rpnet_upsample2_bwd is part of the launch); acc_scale_col = the pack's wd row scales [Co0].
 *   rpnet_conv_up4_supported: 1 when the shapes fit (whole 256-pixel low-resolution patches, one source / destination). */
int rpnet_upconv_collapse_weights(const float* w, float* wc, int Cout, int Cin, rpnet_stream_t stream);
int rpnet_conv_up4_supported(const rpnet_conv_desc* d, int mode);
int rpnet_conv_up4_stats_blocks(const rpnet_conv_desc* d);
int rpnet_conv_up4(const rpnet_conv_desc* d, int mode, rpnet_stream_t stream);
/* weight gradient of the same layer on the collapsed form (csrc/conv_wgrad_up4.hip): x0 = planes of the low-resolution input
 * [N][H/2][W/2][C0], dy = planes of the high-resolution output gradient [N][H][W][Co0], dw [Co0][C0][3][3] (state_dict layout);
 * acc_scale_x / acc_scale_dy: the operands' tensor scales; `accumulate` adds into dw.  Two-phase form as rpnet_conv_wgrad
 * (dw == NULL: the split-K GEMM only; dy == NULL: the reduce only).  Power-of-two low-resolution images >= 8 pixels wide, C0 and
 * Co0 multiples of 64 (rpnet_conv_wgrad_up4_supported); otherwise rpnet_conv_wgrad with d->upsample. */
int rpnet_conv_wgrad_up4_supported(const rpnet_conv_desc* d);
size_t rpnet_conv_wgrad_up4_workspace_bytes(int N, int H, int W, int cin, int cout);
int rpnet_conv_wgrad_up4(const rpnet_conv_desc* d, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                         rpnet_stream_t stream);

/* weight gradient of the same convolution (autograd of nn.Conv2d wrt weight):
 * dW[cout][cin][kh][kw] = sum_pixels A[pixel+tap][cin] * dy[pixel][cout], A gathered
 * exactly as in rpnet_conv_fwd.  Split-K over pixels into `workspace`
 * (rpnet_conv_wgrad_workspace_bytes), then reduced and transposed into the state_dict
 * layout.  Channels [cin_off0, cin_off0+cin_split) and [cin_off1, ...) of the gathered A
 * map to dW input channels 0.. (skips the zero padding rows).
 * d->split_planes != 0 (dense 3x3 only): x0/x1 AND dy are split-bf16 planes ([planes][pixels][C]); then the call
 * may also be made in two phases on two streams — dw == NULL: the split-K GEMM only (partial sums stay in
 * `workspace`), dy == NULL: the reduce + transpose only — so that the HBM-bound reduce of one layer runs under the
 * GEMM of the next. */
size_t rpnet_conv_wgrad_workspace_bytes(int N, int H, int W, int cin_gathered, int cout, int taps);
int rpnet_conv_wgrad(const rpnet_conv_desc* d, const float* dy, float* dw, int cin_w,
                     int cin_off0, int cin_split, int cin_off1,
                     void* workspace, size_t workspace_bytes, rpnet_stream_t stream);

/* first layer, Cin = 1 (net/unet.py:407 Conv1.conv.0): direct convolution */
int rpnet_conv1_fwd(const float* x, const float* w /*[Cout][1][3][3]*/, const float* bias, float* y,
                    const float* ep_scale, const float* ep_shift, int N, int H, int W, int cout,
                    float* out_absmax /* may be NULL; as rpnet_conv_desc.out_absmax */,
                    double* stats_partial /* may be NULL; the train-mode BatchNorm statistics of y fused as in
                                             rpnet_conv_desc.stats_partial: [groups * rpnet_conv1_stats_blocks()][cout][2]
                                             for rpnet_bn_stats_from_partial (ep_scale must be NULL) */,
                    int groups, rpnet_stream_t stream);
int rpnet_conv1_stats_blocks(int N, int H, int W, int cout, int groups);
size_t rpnet_conv1_wgrad_workspace_bytes(int N, int H, int W, int cout);
int rpnet_conv1_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int cout,
                      void* workspace, size_t workspace_bytes, rpnet_stream_t stream);
/* the same weight gradient straight from the gradient dz of the layer's BatchNorm + ReLU OUTPUT: dy is formed on the spot
 * from dz, the pre-BatchNorm tensor y, stats [4][groups][cout] (scale, shift, mean, invstd of rpnet_bn_stats, contiguous) and
 * coef [groups][cout][2] (rpnet_bn_bwd with dy == dy_split == NULL leaves it at workspace + rpnet_bn_bwd_coef_offset) —
 * Conv1.conv.0 has no input gradient, so its dy has no other reader and the BatchNorm backward's apply pass is not run */
int rpnet_conv1_wgrad_bn(const float* x, const float* dz, const float* y /* NULL: made again from x, w, bias */,
                         const float* stats, const float* coef, float* dw,
                         int N, int H, int W, int cout, int groups, void* workspace, size_t workspace_bytes,
                         const float* w /* may be NULL when y is given */, const float* bias, rpnet_stream_t stream);
/* Train mode without the pre-BatchNorm tensor y of this layer in memory (nine multiply-adds make a value, eight bytes write
 * and re-read it): rpnet_conv1_fwd(y = NULL, stats_partial) only sums y for rpnet_bn_stats_from_partial;
 * rpnet_conv1_bn_relu = nn.BatchNorm2d + nn.ReLU (net/modules.py:48-49) of conv(x) made on the spot -> z (fp32, may be NULL) and
 * the operand planes z_split / split_scale exactly as rpnet_bn_relu writes them; rpnet_conv1_bn_bwd_partial = the reduction
 * pass of rpnet_bn_bwd with y made on the spot: partial [groups * rpnet_conv1_bn_bwd_rows()][cout][2] for
 * rpnet_bn_bwd(y = NULL, dy = dy_split = NULL, given_partial, given_rows); then rpnet_conv1_wgrad_bn(y = NULL, w, bias).
 * Every value of y is made by one definition (explicit fused multiply-adds in a fixed order): the same bits each time. */
int rpnet_conv1_bn_relu(const float* x, const float* w, const float* bias, const float* scale /*[groups][cout]*/,
                        const float* shift, float* z, void* z_split, int planes, const float* gamma, const float* beta,
                        float* split_scale, int N, int H, int W, int cout, int groups, rpnet_stream_t stream);
int rpnet_conv1_bn_bwd_rows(int N, int H, int W, int cout, int groups);
int rpnet_conv1_bn_bwd_partial(const float* x, const float* w, const float* bias, const float* dz,
                               const float* stats /*[4][groups][cout]: scale, shift, mean, invstd*/, double* partial,
                               int N, int H, int W, int cout, int groups, rpnet_stream_t stream);

/* ---------------------------------------------------------------------- BatchNorm
 * Train-mode nn.BatchNorm2d + nn.ReLU(inplace) (net/modules.py:48-49,51-52,68-69),
 * split around the grid-wide reduction:
 *   rpnet_bn_stats   per (group, channel) batch mean / biased variance (fp64
 *                    accumulation) -> scale = gamma*invstd, shift = beta - mean*scale,
 *                    save mean & invstd; running_mean/var momentum update with the
 *                    unbiased variance, one update per group in group order;
 *                    num_batches_tracked (int64, may be NULL) += groups.
 *   rpnet_bn_relu    z = relu(y*scale + shift); z_split (may be NULL): also the split-bf16 planes of z
 *                    ([planes][N*HW][C], the operand format of the next convolution, see rpnet_split_bf16);
 *                    with z_split, z may be NULL: the fp32 form is not written (its only consumer reads the planes)
 *   rpnet_bn_bwd     given dz: dgamma, dbeta (summed over groups; accumulate != 0: added to what the
 *                    pointers hold, i.e. straight into the parameters' gradient buffers) and
 *                    dy = scale*(dz*[z>0] - mean(dz*[z>0]) - xhat*mean(dz*[z>0]*xhat)), in fp32 (dy, may be
 *                    NULL when dy_split is given) and / or as split-bf16 planes (dy_split, may be NULL); both NULL: the
 *                    reduction pass only (dgamma, dbeta, coefficients: rpnet_bn_bwd_coef_offset). */
size_t rpnet_bn_workspace_bytes(int C, int groups);
/* byte offset, in rpnet_bn_bwd's workspace, of the coefficients coef [groups][C][2] = (mean(dz m), mean(dz m xhat)) */
size_t rpnet_bn_bwd_coef_offset(int C, int groups);
int rpnet_bn_stats(const float* y, int N, int HW, int C, int groups, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, long long* num_batches_tracked,
                   float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                   void* workspace, size_t workspace_bytes, rpnet_stream_t stream);
/* finalize half of rpnet_bn_stats on partial sums produced by rpnet_conv_fwd (stats_partial) */
int rpnet_bn_stats_from_partial(const double* partial, int nblk, int N, int HW, int C, int groups,
                                const float* gamma, const float* beta, float* running_mean, float* running_var,
                                long long* num_batches_tracked, float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                                rpnet_stream_t stream);
int rpnet_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, float* scale, float* shift, int C,
                         rpnet_stream_t stream);
/* Split outputs: planes == 3 -> bf16 planes of the value itself.  planes == 2 -> fp16 planes of value / s with a
 * power-of-two TENSOR scale s written to *split_scale (device scalar), chosen from a rigorous bound of the output so
 * that fp16 cannot overflow whatever the data: rpnet_bn_relu  |z| <= max_c(|gamma_c| sqrt(n) + |beta_c|) (needs gamma,
 * beta; |xhat| <= sqrt(n) for any batch), rpnet_bn_bwd  |dy_c| <= |scale_c| (max|dz m| + |s1|/n + |s2|/sqrt(n)) with the
 * maxima gathered by the reduction pass; s = pow2ceil(bound) 2^-15.  The consumer multiplies its accumulator by s
 * (rpnet_conv_desc.acc_scale_x / acc_scale_dy). */
/* pool_w > 0 (the image width W; H = HW / W, both even): BatchNorm + ReLU + nn.MaxPool2d(2, 2) (net/unet.py:397,442-448) in
 * one pass for an output that feeds nothing but its pool — z (may be NULL) and z_split (required) are then
 * [N, H/2, W/2, C], the full-resolution z is never written; rpnet_bn_bwd with the same pool_w takes dz of that pooled
 * shape, finds each window's first maximum (row, column scan order, as rpnet_maxpool2_bwd) again from y and writes dy at
 * full resolution (dy_split required, no given_partial). */
/* y_dec / y_dec_stride: RESERVED, must be NULL / 0 (the decode side of rpnet_conv_desc.y_enc, removed in round 5) */
int rpnet_bn_relu(const float* y, const float* scale, const float* shift, float* z, void* z_split, int planes,
                  const float* gamma, const float* beta, float* split_scale, int N, int HW, int C, int groups,
                  int pool_w, const float* y_dec, int y_dec_stride, rpnet_stream_t stream);
/* the fp16 tensor scale of a BatchNorm + ReLU output alone (same value rpnet_bn_relu writes with planes == 2) */
int rpnet_bn_act_scale(const float* gamma, const float* beta, float* split_scale, int N, int HW, int C, int groups,
                       rpnet_stream_t stream);
int rpnet_bn_bwd(const float* dz, const float* y, const float* gamma, const float* scale, const float* shift,
                 const float* mean, const float* invstd, float* dy, void* dy_split, int planes, float* split_scale,
                 float* dgamma, float* dbeta, int N, int HW, int C, int groups, int accumulate,
                 const double* given_partial, const float* given_pmax, int given_rows /* NULL, NULL, 0: the reduction pass
                 runs here; else the caller made the sums itself (the first layer: rpnet_conv1_bn_bwd_partial) */,
                 int pool_w, void* workspace, size_t workspace_bytes, const float* y_dec /* reserved */, int y_dec_stride,
                 rpnet_stream_t stream);

/* conv + bias + ReLU without BatchNorm (vgg.Encoder, net/vgg.py:39-58) — backward pieces:
 * dy = dz * [z > 0] (z may be NULL: no ReLU behind the conv) and db[c] = sum_pixels dy[p][c] */
size_t rpnet_bias_relu_bwd_workspace_bytes(int C);
int rpnet_bias_relu_bwd(const float* dz, const float* z, float* dy, float* db, size_t P, int C,
                        void* workspace, size_t workspace_bytes, rpnet_stream_t stream);
/* nn.MaxPool2d(kernel_size=3, stride=s, padding=1), s in {1,2} (net/vgg.py:23-29), NHWC; backward
 * routes each window's gradient to its first maximum (gather form, no atomics) */
int rpnet_maxpool3_fwd(const float* z, float* out, int N, int H, int W, int C, int stride, rpnet_stream_t stream);
int rpnet_maxpool3_bwd(const float* z, const float* dpool, float* dz, int N, int H, int W, int C, int stride,
                       rpnet_stream_t stream);

/* ------------------------------------------------------------ pooling / up-sampling
 * nn.MaxPool2d(2,2) (net/unet.py:397) forward; backward routes the gradient to the
 * first maximum of each window and adds `skip` (the gradient arriving through the
 * U-Net skip connection) if given.  nn.Upsample(scale_factor=2) backward = 2x2 sum. */
int rpnet_maxpool2_fwd(const float* z, float* out, int N, int H, int W, int C, rpnet_stream_t stream);
int rpnet_maxpool2_bwd(const float* z, const float* dpool, const float* skip, float* dz, int N, int H, int W,
                       int C, rpnet_stream_t stream);
int rpnet_upsample2_bwd(const float* dyu, float* dx, int N, int H, int W, int C, rpnet_stream_t stream);
/* out = srcs[0] + ... + srcs[n-1] (n <= 16 device pointers in a HOST array, 16-byte aligned, numel each): the gradient
 * fan-in of a feature map with many consumers (autograd's chain of pairwise adds for the query features that feed the
 * 2 T masked convolutions of the refinement loop, net/rp_net.py:275,281-312) in one pass. */
int rpnet_sum_n(const float* const* srcs, int n, float* out, size_t numel, rpnet_stream_t stream);

/* --------------------------------------------------------- context-correlation block
 * F.avg_pool2d(mask[:,None], scale) (net/rp_net.py:269-272,311): [B][H][W] -> [B][h][w] */
int rpnet_mask_avgpool(const float* mask, float* out, int B, int H, int W, int scale, rpnet_stream_t stream);

/* Correlation() (net/rp_net.py:153-181) in its exact local-window form:
 * corr[b,y,x, a*(2r+1)+c] = <f1[b,y,x,:], f2[b,y+c-r,x+a-r,:]>/sqrt(C), zero outside;
 * NHWC, output channel stride `cstride` >= (2r+1)^2, channels beyond the window are written 0. */
int rpnet_local_corr_fwd(const float* f1, const float* f2, float* corr, int B, int h, int w, int C, int r,
                         int cstride, rpnet_stream_t stream);
size_t rpnet_local_corr_bwd_workspace_bytes(int B, int h, int w, int cstride);
/* df1_add (may be NULL): a second gradient of f1 that is summed into df1 by the kernel's own store — f1 also feeds the
 * 1x1 convolution as the second source of cat([corr, fm1]) (net/rp_net.py:81); autograd would add the two in a
 * separate pass.  df1_add may alias df1. */
int rpnet_local_corr_bwd(const float* f1, const float* f2, const float* dcorr, float* df1, float* df2,
                         int B, int h, int w, int C, int r, int cstride, const float* df1_add,
                         void* workspace, size_t workspace_bytes, rpnet_stream_t stream);
/* the same on the bf16 matrix pipe with split-bf16 operands (rpnet_split_bf16 planes of f1 / f2, r = 5): the
 * 64 x 324 tile-by-halo score matrix as a GEMM over the channels, the window gathered out of it; backward as a
 * GEMM over the halo positions (C % 128 == 0).  dcorr and the outputs stay fp32.
 * planes == 3: bf16 planes.  planes == 2 (or 1: plain fp16): fp16 planes of f1 / scale1 and f2 / scale2 (device scalars: the
 * tensor scales rpnet_bn_relu wrote with the planes); the window gradients of the backward get a block-local
 * power-of-two scale from the maximum of the tile's own values. */
int rpnet_local_corr_split_fwd(const void* f1_split, const void* f2_split, float* corr, int B, int h, int w, int C, int r,
                               int cstride, int planes, const float* scale1, const float* scale2,
                               float* out_absmax /* may be NULL; as rpnet_conv_desc.out_absmax: the correlation's own fp16
                                                    planes (operand of the 1x1 convolution) are scaled by this bound */,
                               void* corr_planes /* optional (planes 1 / 2; round 6): the correlation also as fp16 planes
                                                    [planes][B][h][w][cstride] of corr / *corr_plane_scale, a power-of-two scale the CALLER
                                                    predicted (rpnet_predict_scales; out_absmax of the same launch is the check) — the
                                                    eval-mode call then needs no split pass over the tensor */,
                               const float* corr_plane_scale, rpnet_stream_t stream);
int rpnet_local_corr_split_bwd(const void* f1_split, const void* f2_split, const float* dcorr, float* df1, float* df2,
                               int B, int h, int w, int C, int r, int cstride, int planes, const float* scale1,
                               const float* scale2, const float* df1_add, void* workspace, size_t workspace_bytes,
                               rpnet_stream_t stream);

/* ------------------------------------------------------------------------- matcher
 * getFeatures + the mask sums of net/rp_net.py:366-376 in adjoint form:
 *   rpnet_mask_adjoint: am[b,k,:,:] = U^T mask_k[b] (U = bilinear up-sampler h,w -> H,W,
 *                       align_corners=False), msum[b,k] = sum(mask_k[b]); k indexes `nmask`
 *                       mask tensors given as an array of nmask device pointers' worth of
 *                       contiguous [B][H][W] blocks (masks + k*B*H*W).
 *   rpnet_masked_pool_fwd: proto[b,k,c] = sum_q f[b,q,c]*am[b,k,q] / (msum[b,k] + 1e-5)
 *   rpnet_masked_pool_bwd: df[b,q,c]   = sum_k dproto[b,k,c]*am[b,k,q]/(msum[b,k]+1e-5) */
int rpnet_mask_adjoint(const float* masks, float* am, float* msum, int B, int nmask, int H, int W, int h, int w,
                       rpnet_stream_t stream);
size_t rpnet_masked_pool_workspace_bytes(int B, int nmask, int hw, int C);
int rpnet_masked_pool_fwd(const float* f, const float* am, const float* msum, float* proto, int B, int nmask,
                          int hw, int C, void* workspace, size_t workspace_bytes, rpnet_stream_t stream);
int rpnet_masked_pool_bwd(const float* dproto, const float* am, const float* msum, float* df, int B, int nmask,
                          int hw, int C, int accumulate, rpnet_stream_t stream);

/* calDist (net/rp_net.py:353-363): pred[b,k,q] = scaler * cos(f[b,q,:], proto[b,k,:]),
 * torch cosine_similarity semantics (each norm clamped at 1e-8).  f NHWC [B][hw][C],
 * pred NCHW-lowres [B][K][hw]. */
int rpnet_cosine_match_fwd(const float* f, const float* proto, float* pred, int B, int K, int hw, int C,
                           float scaler, rpnet_stream_t stream);
size_t rpnet_cosine_match_bwd_workspace_bytes(int B, int K, int hw, int C);
int rpnet_cosine_match_bwd(const float* f, const float* proto, const float* dpred, float* df, float* dproto,
                           int B, int K, int hw, int C, float scaler, int accumulate_df,
                           void* workspace, size_t workspace_bytes, rpnet_stream_t stream);

/* F.interpolate(pred, size=(H,W), mode='bilinear') (net/rp_net.py:303,337) on [BK][h][w]
 * planes, and its adjoint. */
int rpnet_bilinear_up_fwd(const float* in, float* out, int planes, int h, int w, int H, int W, rpnet_stream_t stream);
int rpnet_bilinear_up_bwd(const float* dout, float* din, int planes, int h, int w, int H, int W, rpnet_stream_t stream);

/* softmax(dim=1)[:,1] -> (>0.5) unless soft -> avg_pool2d(scale) (net/rp_net.py:308-311):
 * logits [B][K][H][W] -> next mask [B][H/scale][W/scale] */
int rpnet_softmax_thresh_pool(const float* logits, float* mask, int B, int K, int H, int W, int scale, int soft,
                              rpnet_stream_t stream);

/* The glue between two refinement iterations as ONE launch (round 5; csrc/refine.hip) — replaces, per iteration of
 * net/rp_net.py:281-312, the chain  cre.q's BatchNorm + ReLU (:65-69)  ->  calDist x (1 + Wa) (:301,353-363)  ->  stack +
 * F.interpolate(bilinear) (:302-303)  ->  softmax(1)[:, 1] (:308)  ->  > 0.5 unless soft (:309-310)  ->  avg_pool2d(4) (:311)
 * and the next iteration's  x * mask, x * (1 - mask)  (:283) as the operand planes of its two 3x3 convolutions:
 *   y [B][h][w][F]       cre.q's 1x1-convolution output; with bn_scale / bn_shift [F] (the batch affine of
 *                        rpnet_bn_stats_from_partial) z = relu(y * scale + shift) is written to z [B][h][w][F]; bn_scale == NULL:
 *                        y IS the feature map (eval mode: the convolution's epilogue applied the folded BatchNorm), z is not touched
 *   proto [B][K][F]      -> pred [B][K][h][w] = scaler * cos(z, proto), logits [B][K][4h][4w]
 *   mask_next [B][h][w]  (NULL: the last iteration, nothing behind logits is computed)
 *   x [B][h][w][C] (NULL: no planes), x_scale (device scalar, fp16 planes), xk_planes / xq_planes [planes][B][h][w][C]:
 *                        the 16-bit operand planes (rpnet_split_f16 / rpnet_split_bf16 formats) of x * mask_next and
 *                        x * (1 - mask_next)
 * Same bits as the separate entry points (rpnet_bn_relu, rpnet_cosine_match_fwd, rpnet_bilinear_up_fwd,
 * rpnet_softmax_thresh_pool, rpnet_split_f16 / rpnet_split_bf16).  rpnet_refine_glue_supported: 1 when the shapes fit the kernel
 * (F == 64, 2 <= K <= 4, h and w multiples of 8; planes: 0 or C a multiple of 8 with 256 % (C / 8) == 0) — the caller otherwise
 * runs the separate entry points.
 * rpnet_refine_glue_bwd: autograd of pred / logits wrt the feature map and the prototypes (the adjoint of the up-sampling and
 * the cosine backward in one launch + the prototype-gradient reduce): df [B][h][w][F], dproto [B][K][F]. */
int rpnet_refine_glue_supported(int K, int h, int w, int F, int C, int planes);
int rpnet_refine_glue_fwd(const float* y, const float* bn_scale, const float* bn_shift, const float* proto, float scaler,
                          float* z, float* pred, float* logits, float* mask_next, int soft, const float* x,
                          const float* x_scale, void* xk_planes, void* xq_planes, int planes, int B, int K, int h, int w,
                          int F, int C, rpnet_stream_t stream);
size_t rpnet_refine_glue_bwd_workspace_bytes(int B, int K, int h, int w, int F);
int rpnet_refine_glue_bwd(const float* dlogits, const float* f, const float* proto, float scaler, float* df, float* dproto,
                          int B, int K, int h, int w, int F, void* workspace, size_t workspace_bytes, rpnet_stream_t stream);

/* soft_mask: True (yaml) — the fed-back mask stays differentiable (net/rp_net.py:309 skips the threshold):
 *   rpnet_rowdot_scale     autograd of x*s / x*(1-s) wrt BOTH factors given g = d/d(x*f(s)):
 *                          dx = g*f(s), dscale[p] = +-<g[p,:], x[p,:]> (mode 1: s, mode 2: 1-s)
 *   rpnet_softmax_pool_bwd autograd of avg_pool2d(softmax(logits,1)[:,1], scale) wrt logits */
int rpnet_rowdot_scale(const float* g, const float* x, const float* scale, float* dx, float* dscale, size_t P, int C,
                       int mode, int accumulate_dscale, rpnet_stream_t stream);
int rpnet_softmax_pool_bwd(const float* logits, const float* dmask, float* dlogits, int B, int K, int H, int W,
                           int scale, rpnet_stream_t stream);

/* -------------------------------------------------------------------------- losses
 * dice_ce (net/rp_net.py:87-127) for K-class logits [B][K][H][W], int64 labels [B][H][W]:
 *   loss = [with_dice] (1 - mean_k 2*I_k/(C_k + 1e-7)) + cross-entropy,
 *   cross-entropy = sum(-log_softmax[label]) / #valid            (per_sample = 0, nn.CrossEntropyLoss)
 *                 = (1/B) sum_b w_b * CE_b / #valid_b            (per_sample = 1: alignLoss calls
 *                   F.cross_entropy once per episode, net/rp_net.py:438, sums and divides by B, :349;
 *                   w_b = 0 skips an episode whose predicted foreground is empty, :414,421)
 * forward writes loss[0] and the sums backward needs into `stats`
 * ((B+1)*(2K+2) floats: per sample inter_k, card_k, ce_sum, #valid; then the totals).
 * backward writes (or accumulates) dlogits = gscale[0] * d loss / d logits. */
size_t rpnet_loss_workspace_bytes(int B, int K, int H, int W);
int rpnet_dice_ce_fwd(const float* logits, const int64_t* labels, float* loss, float* stats, int B, int K, int H,
                      int W, int with_dice, int ignore_index, int per_sample, const float* sample_weight,
                      void* workspace, size_t workspace_bytes, rpnet_stream_t stream);
int rpnet_dice_ce_bwd(const float* logits, const int64_t* labels, const float* stats, const float* gscale,
                      float* dlogits, int B, int K, int H, int W, int with_dice, int ignore_index, int per_sample,
                      const float* sample_weight, int accumulate, rpnet_stream_t stream);
/* The training objective sums dice_ce over the final output and every refinement iteration's output against the SAME
 * labels (the reference ships no training loop; train_rpnet.py / bench.py do what its paper describes): n <= 16 logit
 * tensors of one shape in two launches instead of 2 n, and one backward launch instead of n.  `logits` / `dlogits` are
 * HOST arrays of n device pointers.  loss: n + 1 floats — loss[i] = dice_ce(logits[i], labels) (bit-identical to
 * rpnet_dice_ce_fwd), loss[n] = their sum; stats: n * (B+1)*(2K+2) floats; workspace: n * rpnet_loss_workspace_bytes.
 * backward: dlogits[i] = gscale[0] * d loss[i] / d logits[i] (gscale = the gradient arriving at the sum). */
int rpnet_dice_ce_multi_fwd(const float* const* logits, int n, const int64_t* labels, float* loss, float* stats, int B,
                            int K, int H, int W, void* workspace, size_t workspace_bytes, rpnet_stream_t stream);
int rpnet_dice_ce_multi_bwd(const float* const* logits, float* const* dlogits, int n, const int64_t* labels,
                            const float* stats, const float* gscale, int B, int K, int H, int W, rpnet_stream_t stream);
/* The whole training objective in the same two launches (round 6): total = sum_i weights[i] * dice_ce(logits[i], labels) +
 * extra_scale * extra[0] — `weights` a HOST array of n multiplicities (the final output IS the last refinement iteration's output,
 * net/rp_net.py:314-337: it is summed once with weight 2 instead of twice), `extra` a device scalar or NULL (the align loss,
 * net/rp_net.py:394-440, with its yaml scaler `align_loss_scaler`).  loss: n + 1 floats as rpnet_dice_ce_multi_fwd (loss[n] = total;
 * the products and the sum are formed as the tensor expression `sum + scaler * align` forms them: bit-identical).
 * backward: dlogits[i] = gscale[0] * weights[i] * d dice_ce_i / d logits[i]; dextra[0] (optional) = gscale[0] * extra_scale. */
int rpnet_objective_fwd(const float* const* logits, const float* weights, int n, const int64_t* labels, const float* extra,
                        float extra_scale, float* loss, float* stats, int B, int K, int H, int W, void* workspace,
                        size_t workspace_bytes, rpnet_stream_t stream);
int rpnet_objective_bwd(const float* const* logits, float* const* dlogits, const float* weights, int n, const int64_t* labels,
                        const float* stats, const float* gscale, float* dextra, float extra_scale, int B, int K, int H, int W,
                        rpnet_stream_t stream);

/* alignLoss pieces: arg-max class masks of the low-res prediction with their pixel
 * counts (net/rp_net.py:412-417) — masks [B][K][hw] (0/1), counts [B][K], keep [K][B] (optional; 1 where the count is positive:
 * the per-episode skip of a way whose predicted mask is empty, net/rp_net.py:414,421, as the sample weight of rpnet_dice_ce_fwd);
 * and the support label map 1 = fore, 0 = back, 255 = ignore (net/rp_net.py:433-436). */
int rpnet_argmax_masks(const float* pred, float* masks, float* counts, float* keep, int B, int K, int hw, rpnet_stream_t stream);
int rpnet_align_labels(const float* fore, const float* back, int64_t* labels, size_t n, rpnet_stream_t stream);

/* ------------------------------------------------- diagnostics (not on the hot path)
 * LDS canary (csrc/debug_probe.hip): `blocks` workgroups fill `lds_bytes` (1 KB .. 64 KB) of LDS with a pattern, re-read it for
 * `spin_ticks` ticks of the 100 MHz wall clock and count changed words.  Launched beside the LDS-DMA kernels with an allocation that
 * fits on their CUs it tests whether a neighbouring workgroup's `buffer_load ... lds` ever writes outside its own allocation
 * (tools/lds_canary.py; the pooled-pass fault, lds_dma.h).  mismatches[3] (caller-zeroed): changed words, blocks run, sweeps. */
int rpnet_debug_lds_canary(int blocks, int lds_bytes, long long spin_ticks, unsigned* mismatches, rpnet_stream_t stream);
/* Host-only self-test of the 32-bit index division the element-wise passes use (csrc/common.h FastDiv: Granlund-Montgomery multiply-shift
 * with a host-made multiplier) against the C operators: 49 divisors (1, powers of two and their neighbours, up to 2^32 - 1), edge values
 * and `random_per_divisor` random ones each.  Returns the number of mismatches: 0.  Needs no GPU. */
long long rpnet_debug_fastdiv_selftest(int random_per_divisor);
/* MFMA spinner (csrc/debug_probe.hip): `blocks` workgroups of 4 waves issue v_mfma_f32_32x32x16_f16 back to back for `spin_ticks` ticks of
 * the 100 MHz wall clock (<= 1 s), `lds_bytes` of LDS each (144 KB = one block per CU, as the weight-gradient kernel).  out[3]: shader
 * cycles, wall ticks, MFMAs per wave.  `blocks` = count + 65536 * pad: pad (0 .. 3) `s_nop 7` statements behind every MFMA.  The aggressor of
 * tools/corun_probe.py's AGG=spin:<blocks> legs. */
int rpnet_debug_mfma_spin(int blocks, int lds_bytes, long long spin_ticks, unsigned long long* out, rpnet_stream_t stream);

/* ------------------------------------------------- registration pre-step
 * dataset/few_shot_reader.py:109-198 get_registration_field with do_deformable=False (yamls/example.yml:101):
 * per slice, AffineRegistration (net/registration.py:316-357): theta [2][3] from identity by `iters` steps of
 * torch.optim.Adam(lr, betas, eps) on MSE(grid_sample(moving, affine_grid(theta)), fixed) (align_corners=False,
 * zero padding).  ONE launch: a block per slice runs the whole optimisation; slices in parallel.
 * moving / fixed [B][H][W] in [0, 1]; xs [W], ys [H] = the base grid of F.affine_grid (torch.linspace(-1, 1, n) *
 * (n - 1) / n, handed in so that the samples that sit exactly on pixel centres at theta = identity — a kink of the
 * interpolant — round as in the reference); theta out [B][2][3]; loss out [B] (may be NULL) = MSE at the last
 * evaluated theta.
 * rpnet_affine_warp          out = post(grid_sample(x, affine_grid(theta)))        (:337-345)
 * rpnet_identity_grid_warp   out = post(grid_sample(x, compute_grid()))            (:171-187,246-261: the demons stage
 *                            with zero iterations: an align_corners=True identity grid sampled with align_corners=False)
 * post(v) = scale * (threshold >= 0 ? (v > threshold) : v) + shift                  (few_shot_reader.py:168,172,190,196) */
int rpnet_affine_register(const float* moving, const float* fixed, const float* xs, const float* ys, float* theta,
                          float* loss, int B, int H, int W, int iters, double lr, double beta1, double beta2, double eps,
                          rpnet_stream_t stream);
int rpnet_affine_warp(const float* x, const float* theta, const float* xs, const float* ys, float* out, int B, int H, int W,
                      float threshold, float scale, float shift, rpnet_stream_t stream);
int rpnet_identity_grid_warp(const float* x, float* out, int B, int H, int W, float threshold, float scale, float shift,
                             rpnet_stream_t stream);

/* Deformable ("demons") stage, do_deformable: True (few_shot_reader.py:133-180; net/registration.py:195-212 Diffeomorphic
 * with scaling 10, :225-313 DemonsRegistration.train_registraion with the NCC loss :157-160, :106-135 GaussianRegulariser).
 * For every slice: flow [2][H][W] from zero by `iters` x { d = scaling-and-squaring(flow / 2^10); NCC(grid_sample(moving,
 * compute_grid() + d), fixed); Adam(lr, betas, eps) on the flow; flow <- conv2d(flow, kernel, zero padding, per channel) }.
 * moving = the AFFINE-warped source (net/registration.py:491: affine_reg(moving).detach()), fixed [S][H][W] in [0, 1];
 * kernel [ksize][ksize] fp32 = the reference's normalised Gaussian (sigma [2,2] -> 9x9), built by the host;
 * out: flow [S][2][H][W] (channel 0 = x), disp [S][2][H][W] = the displacement of the final flow, loss [S] (may be NULL)
 * = NCC at the last evaluated flow.  All slices advance together: 24 launches per step enqueued from C on `stream`, no
 * host synchronisation.  The bilinear scatter of the backward uses fp32 atomics (run-to-run order differs, as in the
 * reference's own CUDA grid_sampler backward).  The optimiser's scalars are doubles (python floats in the reference).
 * rpnet_displacement_warp   out = post(grid_sample(x, compute_grid() + disp))      (:246-261; post as above) */
size_t rpnet_demons_workspace_bytes(int S, int H, int W);
int rpnet_demons_register(const float* moving, const float* fixed, const float* kernel, int ksize, float* flow, float* disp,
                          float* loss, int S, int H, int W, int iters, double lr, double beta1, double beta2, double eps,
                          void* workspace, size_t workspace_bytes, rpnet_stream_t stream);
int rpnet_displacement_warp(const float* x, const float* disp, float* out, int S, int H, int W, float threshold, float scale,
                            float shift, rpnet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RPNET_ABI_H */
