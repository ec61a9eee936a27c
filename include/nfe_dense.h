/*
 * nfe_dense.h — C ABI of the dense (MFMA) half of the hot path in libnfe_render.so: StyleGAN2 mapping
 * network, modulated convolutions, ToRGB / skip-upsampling and the super-resolution pre-resize.
 * Same conventions as nfe_render.h (device fp32 pointers, caller allocates, async on `stream`,
 * 0 / negative error code + nfe_last_error()).
 *
 * Activations between layers are NHWC fp32 ([N,H,W,C]); the reference's NCHW tensors exist only at
 * the module boundary (nfe_nchw_to_nhwc / nfe_nhwc_to_nchw).  Convolutions run on
 * v_mfma_f32_32x32x16_bf16 with fp32 operands split into bf16 hi+lo (3 MFMAs per product, fp32
 * accumulate: NFE_CONV_BF16X3) or rounded to bf16 (1 MFMA: NFE_CONV_BF16, the throughput mode of
 * BASELINE config 3).
 *
 * The reference's native entry points on this path are the two plugins
 *   bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)    torch_utils/ops/bias_act.cpp:36
 *   upfirdn2d(x, f, upx, upy, downx, downy, padx0.., flip, gain)         torch_utils/ops/upfirdn2d.cpp:20
 * plus cuDNN conv2d / conv_transpose2d (torch_utils/ops/conv2d_gradfix.py:127-129).  Here their forward
 * semantics are fused into the convolution kernels' epilogues; each function names what it replaces.
 */
#ifndef NFE_DENSE_H
#define NFE_DENSE_H

#include "nfe_render.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NFE_CONV_BF16X3 0
#define NFE_CONV_BF16 1
#define NFE_CONV_F16 2        /* ABI v12: fp16 MFMA operands (v_mfma_f32_32x32x16_f16), fp32 accumulate and fp32 activations between layers -
                               * the reference's own GPU arithmetic for its fp16 layers (networks_stylegan2.py:421-423, superresolution.py:271-277
                               * with train.py:183 sr_num_fp16_res = 4).  Needs weights packed by nfe_conv_pack_f16; same sizes, scratch and
                               * kernel variants as NFE_CONV_BF16, 8x finer operand rounding (11 significand bits against 8). */

/* conv modes */
#define NFE_CONV_3X3 0        /* SynthesisLayer up=1: 3x3, pad 1 (networks_stylegan2.py:311-330) */
#define NFE_CONV_3X3_UP2 1    /* SynthesisLayer up=2: conv_transpose2d stride 2 + 4x4 FIR (conv2d_resample.py:114-128) */
#define NFE_CONV_1X1 2        /* ToRGBLayer: 1x1, no demodulation (networks_stylegan2.py:353-357) */

/* ---- layouts -------------------------------------------------------------------------------- */
int nfe_nchw_to_nhwc(const float* in, int n, int c, int h, int w, float* out, nfe_stream_t stream);
int nfe_nhwc_to_nchw(const float* in, int n, int c, int h, int w, float* out, nfe_stream_t stream);
/* NHWC image [N,H,W,96] -> tri-plane gather layout [N,3,H,W,32] (what nfe_render reads) */
int nfe_nhwc_to_planes(const float* in, int n, int h, int w, float* out, nfe_stream_t stream);
/* compute_mean_var (triplane.py:56-60) on an NHWC tensor: [N,H*W,C] -> mean,std [N,C].
 * scratch: n*c*16 bytes of device memory (fp64 partial sums). */
int nfe_plane_stats_nhwc(const float* x, int n, int hw, int c, float* mean, float* std, void* scratch,
                         nfe_stream_t stream);

/* ---- FullyConnectedLayer.forward (networks_stylegan2.py:114-127) ------------------------------
 * y[n,o] = act( (sum_i x[n,i] * w[o,i]) * weight_gain + b[o] * bias_gain ) with act = linear or
 * lrelu(0.2)*sqrt(2) (bias_act.py:23-33).  b may be NULL.  y_stride lets the caller write into a
 * wider row (the concat of MappingNetwork.forward, :244). */
int nfe_fully_connected(const float* x, const float* w, const float* b, int n, int in_features, int out_features,
                        float weight_gain, float bias_gain, int lrelu, float* y, int y_stride, nfe_stream_t stream);

/* Every style affine of a network in one launch (SynthesisLayer.affine / ToRGBLayer.affine, networks_stylegan2.py:316,354:
 * linear FullyConnectedLayers of the rows ws[:, i]): y_g[n,:] = (x_g[n,:] . w_g^T) * weight_gain + b_g * bias_gain.
 * `groups` is a host array of at most NFE_MAX_GROUPS entries (passed by value to the kernel). */
#define NFE_MAX_GROUPS 32
typedef struct nfe_fc_group {
    const float* x; int64_t x_stride;     /* [n, in_features] rows x_stride floats apart (a column block of ws) */
    const float* w; const float* b;       /* [out_features, in_features], [out_features] or NULL */
    float* y;                             /* [n, out_features] */
    int32_t in_features, out_features;
    float weight_gain, bias_gain;
} nfe_fc_group;
int nfe_fully_connected_grouped(const nfe_fc_group* groups, int n_groups, int n, nfe_stream_t stream);

/* nfe_conv_demod for several layers in one launch */
typedef struct nfe_demod_group { const float* styles; const float* wsq; float* dcoef; int32_t cin, cout;
                                 float* styles_norm;   /* ABI v15, optional [N,cin]: see nfe_conv_demod */ } nfe_demod_group;
int nfe_conv_demod_grouped(const nfe_demod_group* groups, int n_groups, int n, nfe_stream_t stream);

/* normalize_2nd_moment (networks_stylegan2.py:24-26): y = x * rsqrt(mean(x^2, dim=1) + 1e-8) */
int nfe_normalize_2nd_moment(const float* x, int n, int features, float* y, int y_stride, nfe_stream_t stream);

/* MappingNetwork tail (networks_stylegan2.py:257-267): broadcast w[n,:] to num_ws rows and lerp the first
 * `cutoff` rows toward w_avg: ws[n,k,:] = k < cutoff ? w_avg + psi*(w - w_avg) : w.  w_avg may be NULL
 * when psi == 1. */
int nfe_broadcast_truncate(const float* w, const float* w_avg, int n, int w_dim, int num_ws, float psi, int cutoff,
                           float* ws, nfe_stream_t stream);

/* ---- modulated convolution (modulated_conv2d, networks_stylegan2.py:34-91) --------------------- */
/* weight [Cout,Cin,k,k] fp32 -> MFMA fragment image (bf16 hi/lo) + per-(o,i) squared norms.
 * packed must hold nfe_conv_packed_words(cout,cin,k) 4-byte words; wsq is [Cout,Cin]. */
uint64_t nfe_conv_packed_words(int cout, int cin, int k);
int nfe_conv_pack(const float* weight, int cout, int cin, int k, float* packed, float* wsq, nfe_stream_t stream);
/* the same image with fp16 operand words (hi part; the lo part is zero): for math = NFE_CONV_F16 only.
 * prenormalize != 0 (ABI v15; every DEMODULATED layer: modulated_conv2d's `x.dtype == float16 and demodulate`, networks_stylegan2.py:53-55):
 * the weights of output channel o are multiplied by 1 / max |w[o]| before they are rounded (the reference's further 1 / sqrt(Cin k k)
 * protects its fp16 accumulator; this library accumulates in fp32 and leaves it out: it would push small weights into fp16's subnormal
 * range), wsq is formed from the scaled weights, and `packed` must hold nfe_conv_packed_words() + cout words (the scales are kept behind the image).  Use it together with
 * the pre-normalised styles of nfe_conv_demod(.., styles_norm): the coefficient cancels both scales.  ToRGB (no demodulation): 0. */
int nfe_conv_pack_f16(const float* weight, int cout, int cin, int k, int prenormalize, float* packed, float* wsq, nfe_stream_t stream);

/* demodulation coefficients: dcoef[n,o] = rsqrt(sum_i styles[n,i]^2 * wsq[o,i] + 1e-8)  (:64-65).
 * styles_norm (ABI v15, optional [N,cin]; the fp16 operand mode): the reference's pre-normalisation of the styles (:56): styles_norm[n,:] =
 * styles[n,:] / max_i |styles[n,i]| is written, and dcoef is formed from it (with the wsq of nfe_conv_pack_f16(prenormalize = 1)); the
 * convolution is then given styles_norm as its `styles` (and as the producing layer's `next_styles`). */
int nfe_conv_demod(const float* styles, const float* wsq, int n, int cin, int cout, float* dcoef, float* styles_norm, nfe_stream_t stream);

typedef struct nfe_conv_args {
    uint32_t struct_size;
    int32_t mode;                 /* NFE_CONV_3X3 / _3X3_UP2 / _1X1 */
    int32_t math;                 /* NFE_CONV_BF16X3 / NFE_CONV_BF16 / NFE_CONV_F16 */
    const float* x;               /* [N,H,W,Cin] */
    const float* styles;          /* [N,Cin] (ToRGB: already times weight_gain) */
    const float* packed;          /* from nfe_conv_pack */
    const float* dcoef;           /* [N,Cout] or NULL (no demodulation) */
    const float* noise;           /* [Ho,Wo] noise_const (or [N,Ho,Wo] per-sample noise) or NULL */
    int64_t noise_n_stride;       /* floats between samples of `noise`; 0 = shared noise_const */
    float noise_strength;
    const float* bias;            /* [Cout] */
    int32_t n, h, w, cin, cout;   /* input size; output is h x w (modes 0,2) or 2h x 2w (mode 1) */
    int32_t lrelu;                /* 1: lrelu(0.2); 0: linear */
    float act_gain;               /* def_gain*gain (sqrt(2) for lrelu layers) */
    float clamp;                  /* conv_clamp*gain, < 0 = none */
    const float* skip;            /* mode 2 only: previous-resolution image [N,H/2,W/2,Cout] to be
                                     upsample2d()'d (upfirdn2d.py:315-350) and added, or NULL */
    int32_t out_planes;           /* mode 2 only: 1 = write [N,3,H,W,32] instead of [N,H,W,Cout] */
    float* out;
    float* scratch;               /* mode 1 (required): at least N*(2H+1)*(2W+1)*Cout floats, the transposed-conv
                                     result before the FIR (unused where the fast path runs the FIR in the conv kernel's own
                                     epilogue on overlapping tiles: nfe_conv_describe() says which; results are the same bits
                                     either way).  With nfe_conv_scratch_floats() floats (modes 0 and 1) the
                                     layer additionally keeps a bf16 hi (+lo) image of the modulated input there and
                                     runs the LDS-DMA fast path; less (or NULL in mode 0) = generic path */
    uint64_t scratch_floats;      /* capacity of `scratch` in floats */
    /* Layer chaining without an fp32 round trip (conv0 -> conv1 of a SynthesisBlock, conv1 -> conv0 of the next block): */
    const float* next_styles;     /* [N,Cout] styles of the 3x3 layer that consumes this output, or NULL */
    float* next_split;            /* out: bf16 hi(+lo) image of out * next_styles, nfe_conv_split_floats() floats, opaque to the
                                     caller (ABI v10: stored as cout/16 planes of [H][W][16 channels] per sample, so that a
                                     16-channel K-group of the consuming GEMM reads contiguous patch rows; cout % 16 == 0).  Mode 1 writes
                                     it from the FIR epilogue, mode 0 from the conv epilogue where nfe_conv_splits_in_epilogue()
                                     says so (elsewhere by one more pass over `out`); in those two cases `out` may be NULL (the
                                     fp32 output is then not written) */
    const float* x_split;         /* in: such an image of this layer's modulated input (written by the producer with this
                                     layer's styles); `x` may then be NULL.  Needs the fast path (scratch as above). */
    /* Fused ToRGB: the image path of SynthesisBlock.forward (networks_stylegan2.py:450-457: y = torgb(x, w); img =
     * upsample2d(img) + y) evaluated in this layer's epilogue, in fp32, on the activation while it is still in registers
     * (the M-block groups of a pixel leave partial sums; a small kernel adds them in order, then bias, clamp, skip).  Only
     * where nfe_conv_fuses_rgb() says so; `out` may then be NULL (the fp32 activation is not written at all). */
    const float* rgb_weight;      /* [rgb_channels, Cout] ToRGBLayer.weight (1x1), or NULL = no fusion */
    const float* rgb_styles;      /* [N, Cout] ToRGB styles, already times weight_gain (ToRGBLayer.forward :353-354) */
    const float* rgb_bias;        /* [rgb_channels] */
    const float* rgb_skip;        /* previous-resolution image [N,H/2,W/2,rgb_channels] or NULL */
    float* rgb_out;               /* [N,H,W,rgb_channels] */
    int32_t rgb_channels;         /* 1..4 */
    float rgb_clamp;              /* conv_clamp of the ToRGB layer, < 0 = none */
} nfe_conv_args;
/* 1 if a NFE_CONV_3X3 call of these sizes evaluates rgb_* in its epilogue (LDS-DMA path, no split-K, rgb_channels <= 4) */
int nfe_conv_fuses_rgb(int mode, int math, int n, int h, int w, int cin, int cout, int rgb_channels);
/* 1 if a NFE_CONV_3X3 call of these sizes writes next_split from its own epilogue (LDS-DMA path, no split-K): `out` may then be
 * NULL when nobody else reads the fp32 activation.  Otherwise next_split is made by one more pass over `out`. */
int nfe_conv_splits_in_epilogue(int mode, int n, int h, int w, int cin, int cout);
/* ABI v10.  Human-readable description of the kernels nfe_modulated_conv() launches for a layer of these sizes ("conv3[32x16/8w]
 * bf16 ksplit=0 fuse_rgb=1 ..."): which tile shape, split-K factor and fused epilogues the batch size n selects.  Diagnostic only
 * (tests log it per layer); rgb_channels = 0 when no ToRGB fusion is requested.  Writes a NUL-terminated string into buf. */
int nfe_conv_describe(int mode, int math, int n, int h, int w, int cin, int cout, int rgb_channels, char* buf, int buf_len);
int nfe_modulated_conv(const nfe_conv_args* args, nfe_stream_t stream);
/* floats of a bf16 hi(+lo) activation image [n,h,w,c] (hi only for NFE_CONV_BF16) */
uint64_t nfe_conv_split_floats(int math, int n, int h, int w, int c);
/* 1 if a 3x3 layer of these sizes can take a pre-split input image (x_split), i.e. runs the LDS-DMA path */
int nfe_conv_accepts_split(int mode, int h, int w, int cin, int cout);
/* floats of scratch a call with these sizes can use (0 = none) */
uint64_t nfe_conv_scratch_floats(int mode, int math, int n, int h, int w, int cin, int cout);

/* ABI v10.  upfirdn2d (torch_utils/ops/upfirdn2d.py:120-205) with the path's filter setup_filter([1,3,3,1]) (separable, symmetric,
 * normalised), NHWC fp32: zero-insert by `up` (1 or 2), pad pad0 / pad1 on both axes, 4x4 FIR, keep every `down`-th (1 or 2) sample,
 * multiply by gain.  out is [n, oh, ow, c] with oh = (h*up + pad0 + pad1 - 4) / down + 1.  upsample2d(x) (upfirdn2d.py:315-350) is
 * up=2, pad=(2,1), gain=4.  The forward path fuses its FIRs into conv / ToRGB epilogues; this entry serves the transposed
 * filters of the SR-head input gradient and callers that want the reference's op by itself. */
int nfe_upfirdn2d(const float* in, int n, int h, int w, int c, int up, int down, int pad0, int pad1, float gain, float* out,
                  nfe_stream_t stream);

/* nfe_upfirdn2d(up = 1, down = 1) whose result [N,OH,OW,C] (OH = h + pad0 + pad1 - 3) leaves as its four polyphase images stacked along
 * the channels (ABI v13): out [N, ceil(OH / 2), ceil(OW / 2), 4 C], out[n][Y/2][X/2][((Y&1)*2 + (X&1)) * C + c], zeros where Y >= OH or
 * X >= OW.  The operand of the up-sampling layers' backward-data convolution (sr_grad.py). */
int nfe_upfirdn2d_polyphase(const float* in, int n, int h, int w, int c, int pad0, int pad1, float gain, float* out, nfe_stream_t stream);

/* ---- F.interpolate(mode='bilinear', align_corners=False, antialias=...) (superresolution.py:283-286)
 * NHWC [N,H,W,C] -> [N,OH,OW,C] */
int nfe_resize_bilinear(const float* in, int n, int h, int w, int c, int oh, int ow, int antialias, float* out,
                        nfe_stream_t stream);
/* its adjoint (the input gradient autograd derives for that F.interpolate call): grad_out [N,OH,OW,C] -> grad_in [N,H,W,C],
 * written (not accumulated).  Serves the SR-head input gradient when the feature image is not at the head's input resolution. */
int nfe_resize_bilinear_backward(const float* grad_out, int n, int h, int w, int c, int oh, int ow, int antialias, float* grad_in,
                                 nfe_stream_t stream);

/* ---- backward of bias_act('lrelu', gain, clamp) with the ToRGB branch folded in (ABI v13; the SR-head gradient, sr_grad.py) ----
 * One pass over a layer's saved output `out` [N,H,W,C] (bias_act.py:93-125: the derivative is taken from the OUTPUT):
 *   g_total = (grad ? grad : 0) + (grad_rgb ? sum_k grad_rgb[n,y,x,k] * rgb_w[k][c] * rgb_s[n][c] : 0)
 *   dst     = g_total * gain * (out < 0 ? 0.2 : 1) * (clamp > 0 ? (|out| < clamp) : 1) * (scale ? scale[n][c] : 1)
 * grad: gradient arriving from the next conv layer (or null); grad_rgb [N,H,W,rgb_k] with rgb_w [rgb_k][C], rgb_s [N][C]: the transposed
 * 1x1 ToRGB of the block (networks_stylegan2.py:455; rgb_k <= 4), or null; scale: a per-(view, channel) factor for the consumer (the
 * demodulation coefficients of an up-sampling layer's backward-data form), or null.  C % 4 == 0.  dst may alias grad. */
int nfe_bias_act_backward(const float* out, const float* grad, const float* grad_rgb, const float* rgb_w, const float* rgb_s, int rgb_k,
                          const float* scale, float gain, float clamp, int n, long long pixels, int c, float* dst, nfe_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NFE_DENSE_H */
