// Dense half of the hot path for MI355X (gfx950): StyleGAN2 mapping network, modulated convolutions as
// implicit GEMMs on v_mfma_f32_32x32x16_bf16 (fp32 operands split into bf16 hi+lo, or rounded to bf16),
// ToRGB with fused skip upsampling, the up-conv FIR epilogue and the SR pre-resize.  See include/nfe_dense.h
// and DESIGN.md §5.  Replaces, on this path, modulated_conv2d (training/networks_stylegan2.py:34-91),
// conv2d_resample (torch_utils/ops/conv2d_resample.py:48-143), upfirdn2d (upfirdn2d.py:120-350) and
// bias_act (bias_act.py:54-125).
#include <algorithm>
#include "nfe_common.h"
#include "nfe_dense.h"

namespace nfe {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
union Frag8 { bf16x8 v; uint4 q; unsigned u[4]; };
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// Operand formats of the implicit GEMMs, the TERMS template argument of every kernel below:
//   3  split-bf16 (hi + lo, three MFMAs per product, fp32-grade)      NFE_CONV_BF16X3
//   1  bf16 (one MFMA per product)                                    NFE_CONV_BF16
//   2  fp16 (one MFMA per product, v_mfma_f32_32x32x16_f16)           NFE_CONV_F16: the operand FORMAT of the reference's fp16 layers
//      (networks_stylegan2.py:421-423: fp16 operands, clamp +-256) - 11 significand bits against bf16's 8 at the same MFMA rate;
//      accumulation stays fp32 and the activations between layers stay fp32 (only the MFMA operands are rounded).  Round 6: the
//      reference's pre-normalisation of a demodulated fp16 layer (networks_stylegan2.py:53-56) - weights / max|w[o]| at pack time
//      (conv_wmax_kernel + conv_pack_kernel; the reference's further 1 / sqrt(I k k) guards its fp16 accumulation, see conv_wmax_kernel), styles / max|s| per sample in the demodulation pass (demod_wave), both cancelled
//      in the demodulation coefficient formed from the normalised values (:64-66) - so neither the modulated activation nor any weight
//      leaves the fp16 range whatever the style's magnitude; the saturating conversion (f16_pair, +-65504) stays as the last resort for
//      what no normalisation bounds (an unclamped activation itself beyond 65504; ToRGB, which the reference does not normalise either)
//      (tests/test_dense_gpu.py::test_fp16_huge_styles_are_prenormalised).
// Everything that is "one part or two" asks TERMS == 3; TERMS 1 and 2 differ only in the conversion and the MFMA opcode.
template <int TERMS>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c, int, int, int) {
    if constexpr (TERMS == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ unsigned f16_pair(float a, float b) {           // round to nearest even, saturating at +-65504 (was +-inf)
    a = __builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f); b = __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f);
    f16x2 p = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, p);
}

__device__ __forceinline__ unsigned bf16_rne(float v) {
    bf16x2 p = {(__bf16)v, (__bf16)0.0f};
    return *reinterpret_cast<unsigned*>(&p) & 0xffffu;
}
// (a,b) -> packed bf16 pair: hi word and lo word.  TERMS==3: hi = truncated top 16 bits, lo = bf16(x - hi).
// TERMS==1: hi = round-to-nearest-even bf16, lo unused.
template <int TERMS>
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    if (TERMS == 3) {
        // a and b usually arrive as products (activation x style).  Pin them in registers: with -ffp-contract=fast the compiler may
        // otherwise fold the multiply into the subtraction below (fma(x, s, -hi): the lo part of the EXACT product) in one producer
        // kernel and not in another, and the consumer's image would depend on which kernel wrote it (round 3: the fused up-sampling
        // epilogue and upfir_kernel differed in the last bit of lo in rare elements).  Defined: hi + lo split the ROUNDED product.
        asm volatile("; nfe_launder %0 %1" : "+v"(a), "+v"(b));
        const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
        hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
        bf16x2 p = {(__bf16)(a - __uint_as_float(ua & 0xffff0000u)), (__bf16)(b - __uint_as_float(ub & 0xffff0000u))};
        lo = *reinterpret_cast<unsigned*>(&p);
    } else if (TERMS == 2) {
        hi = f16_pair(a, b);
        lo = 0;
    } else {
        bf16x2 p = {(__bf16)a, (__bf16)b};
        hi = *reinterpret_cast<unsigned*>(&p);
        lo = 0;
    }
}
// single-part split whose format is known at run time only (the HBM-bound passes that are not templated on the operand format)
__device__ __forceinline__ void split2_single(int f16, float a, float b, unsigned& hi) {
    unsigned lo;
    if (f16) split2<2>(a, b, hi, lo); else split2<1>(a, b, hi, lo);
}

// ------------------------------------------------------------------------------------------------
// layouts
// ------------------------------------------------------------------------------------------------
// [N,C,HW] <-> [N,HW,C] through a 32x32 LDS tile
template <bool TO_NHWC>
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int c, int hw, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + (long long)n * c * hw;
    float* dst = out + (long long)n * c * hw;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        if (TO_NHWC) { const int cc = c0 + r, pp = p0 + tx; tile[r][tx] = (cc < c && pp < hw) ? src[(long long)cc * hw + pp] : 0.0f; }
        else { const int pp = p0 + r, cc = c0 + tx; tile[r][tx] = (cc < c && pp < hw) ? src[(long long)pp * c + cc] : 0.0f; }
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        if (TO_NHWC) { const int pp = p0 + r, cc = c0 + tx; if (cc < c && pp < hw) dst[(long long)pp * c + cc] = tile[tx][r]; }
        else { const int cc = c0 + r, pp = p0 + tx; if (cc < c && pp < hw) dst[(long long)cc * hw + pp] = tile[tx][r]; }
    }
}

__global__ void nhwc_to_planes_kernel(const float4* __restrict__ in, long long n_pix, int hw, float4* __restrict__ out) {
    // in [N,HW,96] -> out [N,3,HW,32]; one thread per float4 (24 per pixel)
    const long long total = n_pix * 24;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / 24; const int q = (int)(i % 24);
        const long long n = pix / hw, p = pix % hw;
        out[((n * 3 + q / 8) * hw + p) * 8 + (q % 8)] = in[i];
    }
}

// per-(n,c) mean / unbiased std over HW of an NHWC tensor (compute_mean_var, triplane.py:56-60).
// grid (C/64, N, SPLIT): fp64 partial sums via atomics into scratch, finalised by stats_finish_kernel.
__global__ __launch_bounds__(256) void stats_nhwc_kernel(const float* __restrict__ x, int hw, int c, double* __restrict__ sums) {
    const int n = blockIdx.y, ch = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const int rows_per = (hw + gridDim.z - 1) / gridDim.z;
    const int r0 = blockIdx.z * rows_per, r1 = min(hw, r0 + rows_per);
    double s = 0.0, ss = 0.0;
    if (ch < c) {
        const float* col = x + (long long)n * hw * c + ch;
        int r = r0 + sub;
        for (; r + 28 < r1; r += 32) {          // eight independent loads in flight per lane
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(long long)(r + 4 * u) * c];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const double d = v[u]; s += d; ss += d * d; }
        }
        for (; r < r1; r += 4) { const double d = col[(long long)r * c]; s += d; ss += d * d; }
    }
    __shared__ double sh[2][4][64];
    sh[0][sub][threadIdx.x & 63] = s; sh[1][sub][threadIdx.x & 63] = ss;
    __syncthreads();
    if (sub == 0 && ch < c) {
        const int l = threadIdx.x;
        atomicAdd(&sums[((long long)n * c + ch) * 2 + 0], sh[0][0][l] + sh[0][1][l] + sh[0][2][l] + sh[0][3][l]);
        atomicAdd(&sums[((long long)n * c + ch) * 2 + 1], sh[1][0][l] + sh[1][1][l] + sh[1][2][l] + sh[1][3][l]);
    }
}
__global__ void stats_finish_kernel(const double* __restrict__ sums, int total, int hw, float* mean, float* stdv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double s = sums[2 * i], ss = sums[2 * i + 1], mu = s / hw;
    double var = (ss - s * mu) / (double)(hw - 1);
    mean[i] = (float)mu; stdv[i] = (float)sqrt(var > 0.0 ? var : 0.0);
}

// ------------------------------------------------------------------------------------------------
// small dense ops of the mapping network / style affines
// ------------------------------------------------------------------------------------------------
// one wave per output element; FullyConnectedLayer.forward (networks_stylegan2.py:114-127)
__global__ __launch_bounds__(256) void fc_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                 int n, int fin, int fout, float wg, float bg, int lrelu, float* __restrict__ y, int ys) {
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wid >= (long long)n * fout) return;
    const int row = (int)(wid / fout), o = (int)(wid % fout);
    const float* xr = x + (long long)row * fin; const float* wr = w + (long long)o * fin;
    float acc = 0.0f;
    for (int i = lane; i < fin; i += 64) acc = fmaf(xr[i], wr[i], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        float v = acc * wg + (b ? b[o] * bg : 0.0f);
        if (lrelu) v = (v < 0.0f ? v * 0.2f : v) * 1.4142135623730951f;     // bias_act 'lrelu': alpha 0.2, gain sqrt(2)
        y[(long long)row * ys + o] = v;
    }
}

__global__ __launch_bounds__(64) void norm2_kernel(const float* __restrict__ x, int f, float* __restrict__ y, int ys) {
    const int row = blockIdx.x, lane = threadIdx.x;
    float s = 0.0f;
    for (int i = lane; i < f; i += 64) { const float v = x[(long long)row * f + i]; s = fmaf(v, v, s); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float r = rsqrtf(s / (float)f + 1e-8f);
    for (int i = lane; i < f; i += 64) y[(long long)row * ys + i] = x[(long long)row * f + i] * r;
}

__global__ void broadcast_truncate_kernel(const float* __restrict__ w, const float* __restrict__ w_avg, int n, int d, int num_ws,
                                          float psi, int cutoff, float* __restrict__ ws) {
    const long long total = (long long)n * num_ws * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i % d); const int k = (int)((i / d) % num_ws); const long long row = i / ((long long)d * num_ws);
        float v = w[row * d + j];
        if (k < cutoff && psi != 1.0f) { const float a = w_avg[j]; v = a + psi * (v - a); }   // torch.lerp(w_avg, x, psi)
        ws[i] = v;
    }
}

// One wave per (sample, output channel): dcoef = rsqrt(sum_i s_i^2 wsq[o][i] + 1e-8) (networks_stylegan2.py:64-65).  With `snorm` (the fp16
// operand mode) the styles are pre-normalised as the reference does before an fp16 modulated conv (:53-56: styles / max_i |styles|, so that
// no modulated activation leaves the fp16 range; the weights' half of it is in conv_pack_kernel): every wave of a sample finds the
// sample's maximum again (cin <= 512: eight loads per lane), the wave of channel 0 writes the normalised row for the convolution's
// modulation, and the coefficient is formed from the normalised values - the normalisation cancels in styles x dcoef exactly as in the
// reference (up to its 1e-8, which now sits beside normalised magnitudes, as there).
__device__ __forceinline__ void demod_wave(const float* __restrict__ styles, const float* __restrict__ wsq, int cin, int cout, long long wid, int lane,
                                           float* __restrict__ dcoef, float* __restrict__ snorm) {
    const int row = (int)(wid / cout), o = (int)(wid % cout);
    const float* sr = styles + (long long)row * cin;
    float inv = 1.0f;
    if (snorm) {
        float m = 0.0f;
        for (int i = lane; i < cin; i += 64) m = fmaxf(m, fabsf(sr[i]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        inv = m > 0.0f ? m : 1.0f;                          // an all-zero style row stays zero (the reference would divide by zero)
    }
    float acc = 0.0f;
    for (int i = lane; i < cin; i += 64) {
        const float s = snorm ? sr[i] / inv : sr[i];
        if (snorm && o == 0) snorm[(long long)row * cin + i] = s;
        acc = fmaf(s * s, wsq[(long long)o * cin + i], acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) dcoef[wid] = rsqrtf(acc + 1e-8f);
}
__global__ __launch_bounds__(256) void demod_kernel(const float* __restrict__ styles, const float* __restrict__ wsq, int n, int cin, int cout,
                                                    float* __restrict__ dcoef, float* __restrict__ snorm) {
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long long)n * cout) return;
    demod_wave(styles, wsq, cin, cout, wid, threadIdx.x & 63, dcoef, snorm);
}

// All style affines / demodulation coefficients of a network in ONE launch each (they depend on ws only): a forward pass
// has ~30 of these 5-microsecond kernels otherwise.  blockIdx.y = group.
struct FcGroups { nfe_fc_group g[NFE_MAX_GROUPS]; };
struct DemodGroups { nfe_demod_group g[NFE_MAX_GROUPS]; };

__global__ __launch_bounds__(256) void fc_grouped_kernel(FcGroups G, int n) {
    const nfe_fc_group g = G.g[blockIdx.y];
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wid >= (long long)n * g.out_features) return;
    const int row = (int)(wid / g.out_features), o = (int)(wid % g.out_features);
    const float* xr = g.x + (long long)row * g.x_stride; const float* wr = g.w + (long long)o * g.in_features;
    float acc = 0.0f;
    for (int i = lane; i < g.in_features; i += 64) acc = fmaf(xr[i], wr[i], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) g.y[(long long)row * g.out_features + o] = acc * g.weight_gain + (g.b ? g.b[o] * g.bias_gain : 0.0f);
}

__global__ __launch_bounds__(256) void demod_grouped_kernel(DemodGroups G, int n) {
    const nfe_demod_group g = G.g[blockIdx.y];
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long long)n * g.cout) return;
    demod_wave(g.styles, g.wsq, g.cin, g.cout, wid, threadIdx.x & 63, g.dcoef, g.styles_norm);
}

// ------------------------------------------------------------------------------------------------
// weights -> MFMA A-fragment image: [Cout/32][Cin/16][taps][part][lane 64][4 words]; word w of lane l
// holds elements e = 2w, 2w+1 of the 8-vector: out channel 32*mb + (l&31), in channel 16g + 8(l>>5) + e.
// ------------------------------------------------------------------------------------------------
// fp16 pre-normalisation of the weights (networks_stylegan2.py:55): alpha[o] = 1 / max |w[o]|, one wave per output channel.
// The reference's factor is 1 / (max |w[o]| sqrt(Cin k k)); its sqrt(Cin k k) keeps the SUM of Cin k k products inside fp16, because its
// convolution accumulates and returns fp16.  Here the accumulators and the activations are fp32 and only the operands must fit, so that
// half of the factor is left out - deliberately: with it a weight below 0.4 % of its channel's maximum (512 x 9 products: 1/68) lands in
// fp16's subnormal range, which the MFMA's operand path does not keep, and config 3's per-channel mean error against the fp32 capture
// grew four-fold (image_raw 1.7e-5 -> 7.6e-5, depth 4.7e-6 -> 2.6e-5; commit 74f5a14 has the switch that rebuilds that form).
__global__ __launch_bounds__(256) void conv_wmax_kernel(const float* __restrict__ weight, int cout, long long per_o, float* __restrict__ alpha) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= cout) return;
    float m = 0.0f;
    for (long long i = lane; i < per_o; i += 64) m = fmaxf(m, fabsf(weight[(long long)o * per_o + i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) alpha[o] = m > 0.0f ? 1.0f / m : 1.0f;
}
// alpha (fp16 operand mode of a demodulated layer only, else null): every weight is multiplied by its output channel's alpha before it is
// rounded to fp16, and wsq is formed from the scaled weights, so that the demodulation coefficient cancels the scale (:53-66).
__global__ void conv_pack_kernel(const float* __restrict__ weight, int cout, int cin, int taps, float* __restrict__ packed, float* __restrict__ wsq, int f16,
                                 const float* __restrict__ alpha) {
    const int G = (cin + 15) / 16, MB = (cout + 31) / 32;
    const long long total = (long long)MB * G * taps * 2 * 256;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int word = (int)(i & 3), lane = (int)((i >> 2) & 63), part = (int)((i >> 8) & 1);
        long long r = i >> 9;
        const int t = (int)(r % taps); r /= taps;
        const int g = (int)(r % G); const int mb = (int)(r / G);
        const int o = 32 * mb + (lane & 31), h = lane >> 5;
        const float al = (alpha && o < cout) ? alpha[o] : 1.0f;
        unsigned bits[2];
        for (int k = 0; k < 2; ++k) {
            const int ch = 16 * g + 8 * h + 2 * word + k;
            const float v = (o < cout && ch < cin) ? weight[((long long)o * cin + ch) * taps + t] * al : 0.0f;
            const unsigned hi = bf16_rne(v);
            bits[k] = f16 ? (part == 0 ? (f16_pair(v, 0.0f) & 0xffffu) : 0u) : (part == 0 ? hi : bf16_rne(v - __uint_as_float(hi << 16)));
        }
        packed[i] = __uint_as_float(bits[0] | (bits[1] << 16));
    }
    const long long nw = (long long)cout * cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (long long)gridDim.x * blockDim.x) {
        const float al = alpha ? alpha[i / cin] : 1.0f;
        float s_ = 0.0f;
        for (int t = 0; t < taps; ++t) { const float v = weight[i * taps + t] * al; s_ = fmaf(v, v, s_); }
        wsq[i] = s_;
    }
}

// ------------------------------------------------------------------------------------------------
// modulated convolution as implicit GEMM.  Workgroup = 4 waves = one 32-channel M-block x a 16x16 pixel
// tile of one sample; wave w owns tile rows 4w..4w+3 as two 32-pixel N-blocks.  Per 16-channel K-group
// the block stages (a) the weight fragments of its M-block (taps x hi/lo x 1 KiB, straight copy) and
// (b) the input patch with halo, multiplied by the styles and split to bf16 hi/lo, laid out
// [part][row][channel-half][col][8 bf16] so a lane's 16-byte B fragment read is conflict-free.
// ------------------------------------------------------------------------------------------------
struct ConvK {
    const float* x; const float* styles; const uint4* packed; const float* dcoef; const float* noise; long long noise_n_stride; float noise_strength;
    const float* bias; int N, H, W, Cin, Cout; int lrelu; float act_gain, clamp; const float* skip; int out_planes;
    float* out; float* scratch;
    const float* next_styles; uint2* split_hi; uint2* split_lo;      // up-conv: modulated bf16 image for the consuming layer
    float* partial; int ksplit;                                       // split-K: raw partial sums [ksplit][...], reduced by splitk_reduce_kernel
    int f16;                                                          // single-part operand images are fp16 (NFE_CONV_F16), not bf16
};

constexpr int PATCH = 18;                                         // 16 + halo
constexpr int PATCH_PART_BYTES = PATCH * 2 * PATCH * 16;          // one of hi / lo

// bias_act.py:93-125: lrelu(0.2) -> gain -> clamp.  Branch-free and four instructions (round 3; the select / two-compare form was a
// fifth of the epilogues' arithmetic): max(v, 0.2 v) IS the leaky ReLU (0.2 v > v exactly where v < 0, the same product either
// way), slope 1 switches it off, the median of (v, -c, c) is the clamp and c = +inf switches that off.  Same bits as the branchy
// form for every input incl. -0 and NaN (v_med3 returns the minimum when an operand is NaN, as fminf(fmaxf(NaN, -c), c) = -c).
__device__ __forceinline__ float epilogue_act(float v, int lrelu, float gain, float clamp) {
    const float slope = lrelu ? 0.2f : 1.0f, c = clamp >= 0.0f ? clamp : INFINITY;      // uniform: scalar selects
    v = __builtin_amdgcn_fmed3f(v, v * slope, INFINITY);                                // = max(v, slope * v)
    return __builtin_amdgcn_fmed3f(v * gain, -c, c);
}

// upsample2d (upfirdn2d.py:315-350: zero-insert x2, pad (2,1), [1,3,3,1]/8*2 per axis) evaluated at (y,x)
__device__ __forceinline__ float4 skip_up2(const float* __restrict__ skip, int n, int hs, int ws, int c, int y, int x, int o) {
    const int ya = (y & 1) ? (y >> 1) : (y >> 1) - 1, yb = ya + 1;
    const int xa = (x & 1) ? (x >> 1) : (x >> 1) - 1, xb = xa + 1;
    const float wya = (y & 1) ? 0.75f : 0.25f, wyb = 1.0f - wya, wxa = (x & 1) ? 0.75f : 0.25f, wxb = 1.0f - wxa;
    float4 r = make_float4(0, 0, 0, 0);
    const int ys[2] = {ya, yb}, xs[2] = {xa, xb};
    const float wy[2] = {wya, wyb}, wx[2] = {wxa, wxb};
    if ((c & 3) == 0) {
        // Branch-free (round 3): the four taps are loaded unconditionally from clamped coordinates and a tap outside the image gets
        // weight 0 (one compare per axis: no lane masks to combine).  With a branch per tap every load sat alone behind a
        // `s_waitcnt vmcnt(0)`: 48 dependent round trips per lane in the 96-channel ToRGB epilogue.
        float4 t[2][2];
        float wgt[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int yc = min(max(ys[a], 0), hs - 1), xc = min(max(xs[b], 0), ws - 1);
                t[a][b] = *reinterpret_cast<const float4*>(skip + (((long long)n * hs + yc) * ws + xc) * c + o);
                const float wye = (unsigned)ys[a] < (unsigned)hs ? wy[a] : 0.0f, wxe = (unsigned)xs[b] < (unsigned)ws ? wx[b] : 0.0f;
                wgt[a][b] = wye * wxe;
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                r.x = fmaf(wgt[a][b], t[a][b].x, r.x); r.y = fmaf(wgt[a][b], t[a][b].y, r.y);
                r.z = fmaf(wgt[a][b], t[a][b].z, r.z); r.w = fmaf(wgt[a][b], t[a][b].w, r.w);
            }
        return r;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (ys[a] < 0 || ys[a] >= hs || xs[b] < 0 || xs[b] >= ws) continue;
            const float* p = skip + (((long long)n * hs + ys[a]) * ws + xs[b]) * c + o;
            const float wgt = wy[a] * wx[b];
            if ((c & 3) == 0) {                 // o is a multiple of 4: one 16-byte load
                const float4 t = *reinterpret_cast<const float4*>(p);
                r.x = fmaf(wgt, t.x, r.x); r.y = fmaf(wgt, t.y, r.y); r.z = fmaf(wgt, t.z, r.z); r.w = fmaf(wgt, t.w, r.w);
                continue;
            }
            r.x = fmaf(wgt, p[0], r.x);
            if (o + 1 < c) r.y = fmaf(wgt, p[1], r.y);
            if (o + 2 < c) r.z = fmaf(wgt, p[2], r.z);
            if (o + 3 < c) r.w = fmaf(wgt, p[3], r.w);
        }
    return r;
}

template <int MODE, int TERMS>
__global__ __launch_bounds__(256, 2) void conv_kernel(ConvK P) {
    constexpr int TAPS = MODE == NFE_CONV_1X1 ? 1 : 9;
    constexpr int HALO = MODE == NFE_CONV_1X1 ? 0 : 1;
    constexpr int PUSED = MODE == NFE_CONV_3X3 ? 18 : (MODE == NFE_CONV_3X3_UP2 ? 17 : 16);
    constexpr int NACC = MODE == NFE_CONV_3X3_UP2 ? 4 : 1;
    __shared__ __attribute__((aligned(16))) unsigned char lds[TAPS * 2 * 1024 + 2 * PATCH_PART_BYTES];
    uint4* ldsA = reinterpret_cast<uint4*>(lds);
    unsigned char* ldsP = lds + TAPS * 2 * 1024;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    // tile grid: output pixels (modes 0,2) or the (H+1)x(W+1) extended input grid (mode 1)
    const int gh = MODE == NFE_CONV_3X3_UP2 ? P.H + 1 : P.H, gw = MODE == NFE_CONV_3X3_UP2 ? P.W + 1 : P.W;
    const int tiles_x = (gw + 15) >> 4;
    const int KS = P.ksplit > 1 ? P.ksplit : 1;                  // split-K: blockIdx.x = tile * KS + ks
    const int tile = blockIdx.x / KS, ks = blockIdx.x % KS;
    const int ty0 = (tile / tiles_x) * 16, tx0 = (tile % tiles_x) * 16;
    const int mb = blockIdx.y, n = blockIdx.z;
    const int G = (P.Cin + 15) >> 4;          // a ragged last K-group is zero-filled (Cin % 4 == 0)
    const int g_per = (G + KS - 1) / KS, g_lo = ks * g_per, g_hi = min(G, g_lo + g_per);

    f32x16 acc[NACC][2];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][nb][r] = 0.0f;

    const int q = tid & 3;                                    // staging: this thread's 4-channel quarter
    // Register double buffer (round 3): the weights and the input patch of K-group g+1 are requested while K-group g multiplies;
    // before, every K-group exposed a full global round trip between two barriers (79 us per 4^2..16^2 layer at 8 views).
    constexpr int A_PER = (TAPS * 2 * 64 + 255) / 256, P_PER = (PUSED * PUSED * 4 + 255) / 256;
    uint4 ra[A_PER];
    float4 rp[P_PER], rs4;
    bool rch_ok;
    auto request = [&](int g) {
        const uint4* src = P.packed + ((long long)mb * G + g) * (TAPS * 2 * 64);
#pragma unroll
        for (int k = 0; k < A_PER; ++k) { const int i = tid + 256 * k; ra[k] = i < TAPS * 2 * 64 ? src[i] : make_uint4(0, 0, 0, 0); }
        rch_ok = 16 * g + 4 * q < P.Cin;
        rs4 = rch_ok ? *reinterpret_cast<const float4*>(P.styles + (long long)n * P.Cin + 16 * g + 4 * q) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < P_PER; ++k) {
            const int idx = tid + 256 * k;
            const int pix = idx >> 2, py = pix / PUSED, px = pix % PUSED;
            const int y = ty0 - HALO + py, x = tx0 - HALO + px;
            rp[k] = make_float4(0, 0, 0, 0);
            if (idx < PUSED * PUSED * 4 && rch_ok && y >= 0 && y < P.H && x >= 0 && x < P.W)
                rp[k] = *reinterpret_cast<const float4*>(P.x + (((long long)n * P.H + y) * P.W + x) * P.Cin + 16 * g + 4 * q);
        }
    };
    if (g_lo < g_hi) request(g_lo);
    for (int g = g_lo; g < g_hi; ++g) {
        __syncthreads();
        // (a) weight fragments of this M-block / K-group: TAPS*2 KiB, contiguous in the packed image
#pragma unroll
        for (int k = 0; k < A_PER; ++k) { const int i = tid + 256 * k; if (i < TAPS * 2 * 64) ldsA[i] = ra[k]; }
        // (b) input patch * styles -> bf16 hi/lo
#pragma unroll
        for (int k = 0; k < P_PER; ++k) {
            const int idx = tid + 256 * k;
            if (idx >= PUSED * PUSED * 4) continue;
            const int pix = idx >> 2, py = pix / PUSED, px = pix % PUSED;
            float4 v = rp[k];
            v.x *= rs4.x; v.y *= rs4.y; v.z *= rs4.z; v.w *= rs4.w;
            unsigned h0, l0, h1, l1;
            split2<TERMS>(v.x, v.y, h0, l0);
            split2<TERMS>(v.z, v.w, h1, l1);
            const int off = ((py * 2 + (q >> 1)) * PATCH + px) * 16 + (q & 1) * 8;
            *reinterpret_cast<uint2*>(ldsP + off) = make_uint2(h0, h1);
            if (TERMS == 3) *reinterpret_cast<uint2*>(ldsP + PATCH_PART_BYTES + off) = make_uint2(l0, l1);
        }
        if (g + 1 < g_hi) request(g + 1);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int kh = t / 3, kw = t % 3;
            Frag8 ah, al;
            ah.q = ldsA[(t * 2 + 0) * 64 + lane];
            if (TERMS == 3) al.q = ldsA[(t * 2 + 1) * 64 + lane];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int ly = 4 * wave + 2 * nb + (j >> 4), lx = j & 15;
                int py, px, a = 0;
                if (MODE == NFE_CONV_3X3) { py = ly + kh; px = lx + kw; }
                else if (MODE == NFE_CONV_3X3_UP2) { py = ly + 1 - (kh >> 1); px = lx + 1 - (kw >> 1); a = (kh & 1) * 2 + (kw & 1); }
                else { py = ly; px = lx; }
                const int off = ((py * 2 + h) * PATCH + px) * 16;
                Frag8 bh, bl;
                bh.q = *reinterpret_cast<const uint4*>(ldsP + off);
                acc[a][nb] = mfma16<TERMS>(ah.v, bh.v, acc[a][nb], 0, 0, 0);
                if (TERMS == 3) {
                    bl.q = *reinterpret_cast<const uint4*>(ldsP + PATCH_PART_BYTES + off);
                    acc[a][nb] = mfma16<TERMS>(ah.v, bl.v, acc[a][nb], 0, 0, 0);
                    acc[a][nb] = mfma16<TERMS>(al.v, bh.v, acc[a][nb], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lane (j,h) register r holds out channel 32mb + (r&3) + 8(r>>2) + 4h of its pixel ----
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int ly = 4 * wave + 2 * nb + (j >> 4), lx = j & 15;
        const int y = ty0 + ly, x = tx0 + lx;
        if (MODE != NFE_CONV_3X3_UP2 && P.ksplit > 1) {          // raw partial sums of this K slice; epilogue in splitk_reduce_kernel
            if (y >= P.H || x >= P.W) continue;
            float* dst = P.partial + ((((long long)ks * P.N + n) * P.H + y) * P.W + x) * P.Cout + 32 * mb + 4 * h;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                if (32 * mb + 8 * qq + 4 * h < P.Cout)
                    *reinterpret_cast<float4*>(dst + 8 * qq) = make_float4(acc[0][nb][4 * qq], acc[0][nb][4 * qq + 1], acc[0][nb][4 * qq + 2], acc[0][nb][4 * qq + 3]);
            continue;
        }
        if (MODE == NFE_CONV_3X3_UP2) {
            if (y > P.H || x > P.W) continue;
            const int TH = 2 * P.H + 1, TW = 2 * P.W + 1;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int Y = 2 * y + (a >> 1), X = 2 * x + (a & 1);
                if (Y >= TH || X >= TW) continue;
                float* dst = (P.ksplit > 1 ? P.partial + (long long)ks * P.N * TH * TW * P.Cout : P.scratch) + (((long long)n * TH + Y) * TW + X) * P.Cout + 32 * mb + 4 * h;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    if (32 * mb + 8 * qq + 4 * h < P.Cout)       // Cout % 4 == 0 on up-conv layers (checked on host)
                        *reinterpret_cast<float4*>(dst + 8 * qq) = make_float4(acc[a][nb][4 * qq], acc[a][nb][4 * qq + 1],
                                                                                acc[a][nb][4 * qq + 2], acc[a][nb][4 * qq + 3]);
            }
        } else {
            if (y >= P.H || x >= P.W) continue;
            const float nz = P.noise ? P.noise[n * P.noise_n_stride + (long long)y * P.W + x] * P.noise_strength : 0.0f;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int o0 = 32 * mb + 8 * qq + 4 * h;
                if (o0 >= P.Cout) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = o0 + i;
                    float t = acc[0][nb][4 * qq + i];
                    if (o < P.Cout) {
                        if (P.dcoef) t *= P.dcoef[(long long)n * P.Cout + o];
                        t = epilogue_act(t + nz + P.bias[o], P.lrelu, P.act_gain, P.clamp);
                    }
                    v[i] = t;
                }
                if (MODE == NFE_CONV_1X1 && P.skip) {
                    const float4 s = skip_up2(P.skip, n, P.H >> 1, P.W >> 1, P.Cout, y, x, o0);
                    v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w;
                }
                float* dst;
                if (MODE == NFE_CONV_1X1 && P.out_planes)
                    dst = P.out + ((((long long)n * 3 + mb) * P.H + y) * P.W + x) * 32 + 8 * qq + 4 * h;
                else
                    dst = P.out + (((long long)n * P.H + y) * P.W + x) * P.Cout + o0;
                if (o0 + 3 < P.Cout && (P.Cout & 3) == 0) {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (o0 + i < P.Cout) dst[i] = v[i];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fast path of the plain 3x3 layers (W >= 32, Cin % 16 == 0, Cout % 64 == 0): the activations are modulated
// and split to bf16 ONCE (modsplit_kernel, an HBM-bound elementwise pass) instead of once per M-block
// workgroup, so the implicit-GEMM loop has no VALU staging left: weights and the input patch go
// global -> LDS by LDS-DMA (global_load_lds_dwordx4, double-buffered over the 16-channel K-groups), fragments
// LDS -> registers with immediate offsets, MFMA.  Workgroup = 4 waves = 64 output channels x (32 x 8) pixels;
// a wave owns two image rows of 32 pixels (two N-blocks) x two M-blocks.  A 32-pixel row read is conflict-free
// under the b128 lane groups with a compact patch layout.
// ------------------------------------------------------------------------------------------------
// Split-image layout ("group-major", round 3): the bf16 image a 3x3 layer consumes is stored as Cin/16 planes of
// [H][W][16 channels] per sample - element (n, y, x, c) at (((n * G + c/16) * H + y) * W + x) * 16 + c%16, G = Cin/16, hi plane set
// first, lo plane set (split-bf16) behind it.  A K-group of the implicit GEMM is 16 input channels, so a patch row of a K-group
// (34 pixels x 32 bytes) is one contiguous run and a 1-KiB LDS-DMA instruction is ~17 L1 requests.  With the NHWC order of
// round 2 every pixel's 32 bytes sat 2*Cin bytes apart: ~42 requests per instruction (TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD),
// the texture addresser 51 % busy and the waves stuck ~900 cycles in every LDS-DMA issue behind it (tools/c3_profile.py).
// uint2 index (4 channels) of channel-quad c4 of pixel (n, y, x):
__device__ __forceinline__ long long split_index(int n, int G, int H, int W, int y, int x, int c4) {
    return ((((long long)n * G + (c4 >> 2)) * H + y) * W + x) * 4 + (c4 & 3);
}

__global__ __launch_bounds__(256) void modsplit_kernel(const float4* __restrict__ x, const float* __restrict__ styles, long long n_vec,
                                                       long long hw_vec, int c4, uint2* __restrict__ hi, uint2* __restrict__ lo, int f16) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % c4);
        const long long n = i / hw_vec;
        const float4 s = *reinterpret_cast<const float4*>(styles + (n * c4 + q) * 4);
        float4 v = x[i];
        v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
        unsigned h0, l0, h1, l1;
        const long long pix = i / c4;                                  // (n, y, x) flattened; hw_vec / c4 pixels per sample
        const long long o = ((n * (c4 >> 2) + (q >> 2)) * (hw_vec / c4) + (pix - n * (hw_vec / c4))) * 4 + (q & 3);
        if (lo) { split2<3>(v.x, v.y, h0, l0); split2<3>(v.z, v.w, h1, l1); lo[o] = make_uint2(l0, l1); }
        else { split2_single(f16, v.x, v.y, h0); split2_single(f16, v.z, v.w, h1); }
        hi[o] = make_uint2(h0, h1);
    }
}

__device__ uint4 nfe_zero16[4];                                  // source of the zero padding for LDS-DMA
#define C3_LC_GENERIC_LOOP 0     // 1: the compute waves of the loader / compute split run the generic fragment pipeline (A/B)
#define C3_UP_DBUF 0                                             // 1: fused up-sampling epilogue with two FIR slice buffers (5 barriers per tile instead of 8): measured no gain (3.27 vs 3.27 ms FFHQ fp16 step), twice the LDS
// LDS of the fused up-sampling epilogue: C3_UP_DBUF + 1 slice buffers of [2 halves][2 ROWS][64] float4
constexpr int conv3_fused_t_bytes(int rows) { return (C3_UP_DBUF ? 2 : 1) * 2 * (2 * rows) * 64 * 16; }
#define C3_PATCH_SWZ 1                                           // 0: A/B - no XOR swizzle of the patch halves (ascending DMA addresses, 2-way conflicts on the fragment reads)
#define C3_UP_BCACHE 1                                           // up-sampling K loop: the six distinct patch fragments of a K-group in registers

struct Conv3K {
    const unsigned short* xh; const unsigned short* xl; const uint4* packed; const float* dcoef; const float* noise;
    long long noise_n_stride; float noise_strength; const float* bias; int N, H, W, Cin, Cout; int lrelu; float act_gain, clamp;
    float* out; float* scratch;
    int c3_tiles;       // real tile count (grid.x is padded to a multiple of 8 for the XCD-aware order)
    int up_fused;       // UP2: the 4 x 4 FIR, demodulation, noise, bias and activation run in this kernel's epilogue on overlapping tiles
                        // (30 x (ROWS - 2) new extended-input pixels per 32 x ROWS tile); no (2H+1)^2 scratch, no upfir_kernel
    const float* rgb_w; const float* rgb_s; float* rgb_partial; int rgb_c;     // fused ToRGB (plain 3x3, no split-K): see rgb_combine_kernel
    int ksplit;         // > 1: blockIdx.z = n * ksplit + ks; this workgroup sums K-groups [ks*G/ksplit, (ks+1)*G/ksplit) and writes
    float* partial;     //      raw partial sums [ksplit][N,H,W,Cout] that splitk_reduce_kernel adds in slice order (+ epilogue)
    const float* next_styles; uint2* split_hi; uint2* split_lo;     // plain 3x3, no split-K: the consuming layer's modulated bf16 image
    int f16;            // host-side only: launch the TERMS = 2 (fp16 operand) instantiation of the bf16 variant
                                                                    // (what modsplit_kernel would make of `out`), written by the epilogue
    // upconv_strip_kernel (round 6): a workgroup walks `seg_blocks` 8-row blocks down a 30-column strip of the extended input grid
    float* seam;        // [N][segs - 1][6][2W][Cout]: the three row-filtered T rows either side of every segment boundary
    int strips, segs, seg_blocks, blocks;
    int tyl;            // T rows a block adds (16: two image rows per wave, 8: one)
};

#define C3_XCD_ALL 0
__host__ __device__ constexpr bool c3_xcd_order(int terms) { return terms == 3 || C3_XCD_ALL; }

// WV waves per workgroup, each owning NBW image rows of 32 pixels: tile = 32 x (NBW * WV) pixels (ROWS rows).
constexpr int C3_TW = 32, C3_PW = C3_TW + 2;
template <int ROWS> struct C3Tile {
    static constexpr int TH = ROWS, PH = TH + 2;
    static constexpr int HALF_ITEMS = C3_PW * PH;                 // 16-byte items (8 bf16 channels of one pixel) per channel half
    static constexpr int B_CHUNKS = (2 * HALF_ITEMS + 63) / 64;   // 1-KiB LDS-DMA chunks per part
    static constexpr int B_BYTES = B_CHUNKS * 1024;
};

// One LDS-DMA instruction: 64 lanes x 16 bytes, per-lane global address -> 1 KiB of LDS at lds_dst (wave-uniform).
// Issued as inline asm on purpose.  Through __builtin_amdgcn_global_load_lds the compiler knows that an asynchronous write to
// LDS is in flight and, unable to prove that the fragment reads of the CURRENT stage do not alias the stage being filled (same
// extern __shared__ array), it puts `s_waitcnt vmcnt(0)` in front of the first ds_read of every MFMA phase - i.e. every K-group
// waited for the NEXT K-group's loads to land before computing, and the two-stage ring overlapped nothing (tools/c3_profile.py
// showed it as ~4 300 cycles of "issue" per K-group; round 3).  The waits this kernel needs are its own explicit
// `s_waitcnt vmcnt(N)` + s_barrier at the top of each K-group.
__device__ __forceinline__ void lds_dma16(const void* src, void* lds_dst) {
    const unsigned l = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds_dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(src) : "memory", "m0");
}

// The same with one dword per lane: 64 lanes x 4 bytes -> 256 bytes of LDS at lds_dst (the strip kernel's noise rows).
__device__ __forceinline__ void lds_dma4(const void* src, void* lds_dst) {
    const unsigned l = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds_dst;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(l), "v"(src) : "memory", "m0");
}

// UP2: the stride-2 transposed convolution of the up-sampling layers as four output phases over the (H+1)x(W+1)
// extended input grid (same tap -> phase map as conv_kernel<NFE_CONV_3X3_UP2>), written to the (2H+1)x(2W+1)
// scratch that upfir_kernel filters.
template <int TERMS, int MBW, int ROWS>
constexpr int conv3_stage_bytes() { return (MBW * 9 + C3Tile<ROWS>::B_CHUNKS) * (TERMS == 3 ? 2 : 1) * 1024; }
// LDS behind the ring for the epilogue constants ec[3][32 * MBW] (demodulation, bias, next styles): the kernel and its launch both
// size it from here (a fixed + 1024 was 512 bytes short for the MBW = 4 variants, whose third row lay past the allocation)
template <int MBW> constexpr int conv3_ec_bytes() { return 3 * 32 * MBW * 4 > 1024 ? 3 * 32 * MBW * 4 : 1024; }

// STAGES-deep ring of K-group buffers: the loads of K-group g+STAGES-1 are issued while g is computed, so a
// load has STAGES-1 K-groups of MFMA time to land.
// NBW = 2: up to four waves per SIMD (two workgroups per CU).  NBW = 4 with MBW = 4 is the big tile: 128 channels x
// 32 x 16 pixels on four waves, ONE wave per SIMD with 256 accumulator registers - half the LDS reads and half the
// L1->LDS bytes per MFMA of the small tiles.
// LW > 0 (round 3): LW extra LOADER waves per workgroup.  Waves 0..WV-1 only compute (fragment reads + MFMA + epilogue), waves
// WV..WV+LW-1 only stage operands: they keep STAGES-1 K-groups of LDS-DMA in flight, wait for the oldest one and meet the compute
// waves at the one barrier per K-group.  The load stream no longer stops while a wave is in its MFMA phase (and vice versa): in
// the symmetric form the two halves of the work overlapped by a quarter only (profiles/experiments/r03_conv3_negative.md).
// FU (round 5) > 0: the FUSED up-sampling form as a compile-time variant at FU waves per SIMD (no edge-column accumulators, branches
// and bookkeeping of the unfused form in its K loop); FU = 0: fused or not at run time (P.up_fused).  Three workgroups per CU (FU = 3,
// 168 registers) were measured and do not pay: profiles/experiments/r05_up_conv.md.
template <int TERMS, int MBW, bool UP2, int STAGES, int WV, int NBW = 2, int LW = 0, int FU = 0>
__global__ __launch_bounds__(64 * (WV + LW), FU ? FU : LW ? (WV + LW) / 4 : ((NBW * MBW > 8 || (UP2 && TERMS == 3 && WV == 8)) ? 1 : ((STAGES * conv3_stage_bytes<TERMS, MBW, NBW * WV>() > 80 * 1024) ? 1 : 2) * WV / 4)) void conv3_kernel(Conv3K P) {
    static_assert(FU == 0 || UP2, "FU: fused up-sampling variants only");
    constexpr int PARTS = TERMS == 3 ? 2 : 1;
    constexpr int NACC = UP2 ? 4 : 1;
    constexpr int A_CHUNKS = MBW * 9 * PARTS;
    constexpr int ROWS = NBW * WV;
    constexpr int STAGE_BYTES = conv3_stage_bytes<TERMS, MBW, ROWS>();
    constexpr int C3_TH = C3Tile<ROWS>::TH, C3_HALF_ITEMS = C3Tile<ROWS>::HALF_ITEMS, C3_B_CHUNKS = C3Tile<ROWS>::B_CHUNKS, C3_B_BYTES = C3Tile<ROWS>::B_BYTES;
    constexpr int IW = LW ? LW : WV;                                          // waves that issue the LDS-DMA of a stage
    constexpr int B_PER_WAVE = (C3_B_CHUNKS + IW - 1) / IW;
    constexpr int MIN_LOADS = A_CHUNKS / IW + (C3_B_CHUNKS / IW) * PARTS;    // fewest LDS-DMA instructions any issuing wave issues per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = LW > 0 && wave >= WV;              // wave-uniform role
    const bool issues = LW == 0 || loader;
    const int iw = LW ? wave - WV : wave;                  // index among the issuing waves
    // Tile grid: output pixels, or the (H+1) x (W+1) extended input grid of the transposed conv.  EDGE mode (W a multiple of
    // the tile width): the extra column x = W is not tiled (a 32-wide tile for one column); the right-most tile of every tile
    // row computes it as ONE extra N-block on wave 0 (lane = tile row): that column only sees input column W-1 through the
    // three kw = 2 taps, i.e. phases (a, b = 0), and its B fragments are column 32 of the patch the tile has staged anyway.
    const bool fusedup = UP2 && (FU > 0 || P.up_fused);    // wave-uniform
    const bool edge_mode = UP2 && FU == 0 && !fusedup && (P.W % C3_TW) == 0;
    const int gw = UP2 ? (edge_mode ? P.W : P.W + 1) : P.W;
    const int tiles_x = fusedup ? (P.W + 29) / 30 : (gw + C3_TW - 1) / C3_TW;
    // XCD-aware order (split-bf16 variants): workgroups reach the 8 XCDs round-robin in dispatch order and every XCD has its own
    // L2.  XCD x takes the tiles = x (mod 8) and walks the M-block groups of a tile back to back, so the tile's input patch is
    // fetched into that L2 once instead of once per M-block group (the single-buffered split-bf16 stage cannot hide the longer
    // fetch; measured -7 % on the SR head.  The double-buffered bf16 variants measured no gain and keep the plain order).
    int tile_ = blockIdx.x, mbg_ = blockIdx.y;
    if (c3_xcd_order(TERMS)) {
        const int MBG = gridDim.y;
        const int L = blockIdx.y * gridDim.x + blockIdx.x;
        const int k_ = L >> 3;
        tile_ = (k_ / MBG) * 8 + (L & 7); mbg_ = k_ % MBG;
        if (tile_ >= P.c3_tiles) return;              // grid.x is padded to a multiple of 8
    }
    // fused up-sampling tiles overlap by two extended-input pixels per axis and start at -1: tile (ky, kx) produces the output
    // pixels [2 ky (ROWS-2), 2 (ky+1)(ROWS-2)) x [60 kx, 60 kx + 60), whose FIR needs the scratch pixels one before and two after
    const int ty0 = fusedup ? (tile_ / tiles_x) * (C3_TH - 2) - 1 : (tile_ / tiles_x) * C3_TH;
    const int tx0 = fusedup ? (tile_ % tiles_x) * 30 - 1 : (tile_ % tiles_x) * C3_TW;
    const int KS = P.ksplit > 1 ? P.ksplit : 1;
    const int mb0 = mbg_ * MBW, n = blockIdx.z / KS, ks = blockIdx.z % KS;
    const int G_all = P.Cin >> 4;
    const int g_per = (G_all + KS - 1) / KS, g_base = ks * g_per;
    const int G = min(G_all, g_base + g_per) - g_base;        // K-groups of this workgroup: g_base + [0, G)

    // this wave's share of the patch chunks: chunk c = wave + WV k; per-lane element offset of the pixel (or -1 = padding)
    const long long plane16 = (long long)P.H * P.W * 16;      // elements of one 16-channel plane of the group-major split image
    long long boff[B_PER_WAVE];
#pragma unroll
    for (int k = 0; k < B_PER_WAVE; ++k) {
        const int item = ((issues ? iw : 0) + IW * k) * 64 + lane;
        // LDS item 2p + (hh ^ bit3(p)) holds channel half hh of patch pixel p: the two halves of a pixel (32 contiguous
        // bytes in NHWC) are fetched by adjacent lanes = one L1 request, and the XOR keeps the 32-byte-stride fragment
        // reads conflict-free (pixels p and p+8 share a bank pair, their halves are swapped).
        const int pp = item >> 1, hh = (item & 1) ^ (C3_PATCH_SWZ ? (pp >> 3) & 1 : 0), py = pp / C3_PW, px = pp % C3_PW;
        const int y = ty0 - 1 + py, x = tx0 - 1 + px;
        const bool ok = pp < C3_HALF_ITEMS && y >= 0 && y < P.H && x >= 0 && x < P.W;
        boff[k] = ok ? (((long long)n * G_all * P.H + y) * P.W + x) * 16 + 8 * hh : -1;     // group-major image: + g * H*W*16 per K-group
    }

    // byte offsets of this lane's B fragments inside the patch: rows NBW*wave + (0..NBW+1), columns j + (0..2)
    int brd[NBW + 2][3];
#pragma unroll
    for (int rr = 0; rr < NBW + 2; ++rr)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const int pp = (NBW * wave + rr) * C3_PW + j + cc;
            brd[rr][cc] = (2 * pp + (h ^ (C3_PATCH_SWZ ? (pp >> 3) & 1 : 0))) * 16;
        }

    const bool edge_tile = edge_mode && tx0 + C3_TW == P.W;
    int brde[2];                                               // patch offsets of the edge column for dy = 0, 1 (lane j = tile row j)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int pp = (min(j, C3_TH - 1) + dy) * C3_PW + C3_TW;
        brde[dy] = (2 * pp + (h ^ (C3_PATCH_SWZ ? (pp >> 3) & 1 : 0))) * 16;
    }
    f32x16 acce[UP2 ? 2 : 1][MBW];                             // edge column, phases a = 0, 1 (b = 0)
#pragma unroll
    for (int a = 0; a < (UP2 ? 2 : 1); ++a)
#pragma unroll
        for (int m = 0; m < MBW; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acce[a][m][r] = 0.0f;

    // Source pointers of this wave's LDS-DMA chunks, kept RUNNING (round 5): issue() is called for K-groups 0, 1, 2, ... in order, so a
    // pointer advances by one K-group per call (two vector adds) instead of being rebuilt from (M-block, K-group, tap, part) with
    // 64-bit scalar multiplies every time - the K loop of the up-sampling kernel spent 120 of its 198 instructions per K-group on that
    // (static census of the ISA), and a wave of this kernel issues instructions for 45 % of its life (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES).
    // Not for the register-tight variants (plain bf16 / fp16 at four waves per SIMD: 128 registers, the six pointer pairs spill and the
    // 8-view SR head got 4 % SLOWER): those rebuild their addresses as before (RUNPTR = false).
    constexpr bool RUNPTR = UP2 || TERMS == 3;
    constexpr int A_PER_WAVE = (A_CHUNKS + IW - 1) / IW;
    const uint4* aptr[A_PER_WAVE];
#pragma unroll
    for (int kk = 0; kk < A_PER_WAVE; ++kk) {
        const int c = min((issues ? iw : 0) + IW * kk, A_CHUNKS - 1);          // chunk (m, t, part) <- packed[((mb0+m)*G + g)*18 + t*2 + part]
        const int part = c % PARTS, t = (c / PARTS) % 9, m = c / (PARTS * 9);
        aptr[kk] = P.packed + (((long long)(mb0 + m) * G_all + g_base) * 18 + t * 2 + part) * 64 + lane;
    }
    const unsigned short* bptr[B_PER_WAVE][PARTS];
    long long bstep[B_PER_WAVE];                                // elements per K-group: 0 for a padding lane (it stays on the zero page)
#pragma unroll
    for (int k = 0; k < B_PER_WAVE; ++k) {
        bstep[k] = boff[k] >= 0 ? plane16 : 0;
#pragma unroll
        for (int part = 0; part < PARTS; ++part) {
            const unsigned short* xs = part ? P.xl : P.xh;
            bptr[k][part] = boff[k] >= 0 ? xs + boff[k] + (long long)g_base * plane16 : reinterpret_cast<const unsigned short*>(nfe_zero16);
        }
    }
    int g_next = 0;                                            // the K-group the next issue() call stages
    auto issue = [&](int stage) {
        unsigned char* base = lds + stage * STAGE_BYTES;
        if constexpr (!RUNPTR) {
            const int g = g_next++;
            for (int c = iw; c < A_CHUNKS; c += IW) {
                const int part = c % PARTS, t = (c / PARTS) % 9, m = c / (PARTS * 9);
                const uint4* src = P.packed + (((long long)(mb0 + m) * G_all + g_base + g) * 18 + t * 2 + part) * 64 + lane;
                lds_dma16(src, base + c * 1024);
            }
#pragma unroll
            for (int k = 0; k < B_PER_WAVE; ++k) {
                const int c = iw + IW * k;
                if (c < C3_B_CHUNKS) {
#pragma unroll
                    for (int part = 0; part < PARTS; ++part) {
                        const unsigned short* xs = part ? P.xl : P.xh;
                        const void* src = boff[k] >= 0 ? (const void*)(xs + boff[k] + (long long)(g_base + g) * plane16) : (const void*)nfe_zero16;
                        lds_dma16(src, base + A_CHUNKS * 1024 + part * C3_B_BYTES + c * 1024);
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < A_PER_WAVE; ++kk) {
            const int c = iw + IW * kk;
            if (c < A_CHUNKS) lds_dma16(aptr[kk], base + c * 1024);      // 1024: timing experiment, no weight staging
            aptr[kk] += 18 * 64;
        }
#pragma unroll
        for (int k = 0; k < B_PER_WAVE; ++k) {
            const int c = iw + IW * k;
#pragma unroll
            for (int part = 0; part < PARTS; ++part) {
                if (c < C3_B_CHUNKS)                                    // 2048: timing experiment, no patch staging
                    lds_dma16(bptr[k][part], base + A_CHUNKS * 1024 + part * C3_B_BYTES + c * 1024);
                bptr[k][part] += bstep[k];
            }
        }
    };

    f32x16 acc[NACC][MBW][NBW];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int m = 0; m < MBW; ++m)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][m][nb][r] = 0.0f;

    // Per-channel epilogue constants of this workgroup's 32 * MBW channels go to LDS now (1 KiB behind the ring): loaded in the
    // epilogue they were a dependent global round trip with nothing to hide it (8 % of the kernel, tools/r03_abm_stats.sh).
    // [0] demodulation coefficients, [1] bias, [2] the consuming layer's styles.  Visible after the first barrier of the K loop.
    constexpr int FUSED_T_BYTES = UP2 ? conv3_fused_t_bytes(ROWS) : 0;         // the fused up-sampling epilogue's scratch slices (8 channels each)
    constexpr int EC_OFFSET = STAGES * STAGE_BYTES > FUSED_T_BYTES ? STAGES * STAGE_BYTES : FUSED_T_BYTES;
    float* ec = reinterpret_cast<float*>(lds + EC_OFFSET);
    constexpr int EC = 32 * MBW;
    static_assert(3 * EC * 4 <= conv3_ec_bytes<MBW>(), "ec[] must fit the bytes launch_conv3 reserves behind the ring");
    const bool own_epilogue = !UP2 && KS == 1;
    if ((own_epilogue || fusedup) && tid < EC) {
        const int ch = 32 * mb0 + tid;
        ec[tid] = P.dcoef ? P.dcoef[(long long)n * P.Cout + ch] : 1.0f;
        ec[EC + tid] = P.bias[ch];
        ec[2 * EC + tid] = P.split_hi ? P.next_styles[(long long)n * P.Cout + ch] : 0.0f;
    }
    float nzv[NBW];                                            // noise of this lane's pixels: in flight during the K loop
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int y = min(ty0 + NBW * min(wave, WV - 1) + nb, P.H - 1), x = min(tx0 + j, P.W - 1);
        nzv[nb] = (own_epilogue && P.noise) ? P.noise[n * P.noise_n_stride + (long long)y * P.W + x] * P.noise_strength : 0.0f;
    }

    static_assert(LW == 0 || (STAGES >= 2 && !UP2), "loader waves need a ring of at least two stages; plain 3x3 only");
    if (issues)
        for (int pre = 0; pre < STAGES - 1; ++pre)
            if (pre < G) issue(pre);
    int stage = 0;
    if (loader) {                       // ---- loader waves: the whole K loop, then done (no epilogue, no further barriers) ----
        for (int g = 0; g < G; ++g) {
            // K-group g has landed once at most the loads of the STAGES-2 younger K-groups are outstanding (in-order return)
            if (STAGES <= 2 || g + STAGES - 2 >= G) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * MIN_LOADS) : "memory");
            __syncthreads();            // stage g is complete for everybody; the compute waves are done with stage g-1
            if (g + STAGES - 1 < G) issue(stage == 0 ? STAGES - 1 : stage - 1);
            stage = stage + 1 == STAGES ? 0 : stage + 1;
        }
        return;
    }
    for (int g = 0; g < G; ++g) {
        if (STAGES == 1) { __syncthreads(); issue(0); }
        // K-group g has landed once at most the loads of the STAGES-2 younger K-groups are outstanding (in-order return)
        if (LW > 0) {}                  // compute waves issue no loads: the loader waves wait for them
        else if (STAGES <= 2 || g + STAGES - 2 >= G) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * MIN_LOADS) : "memory");
        __syncthreads();
        const bool more = g + STAGES - 1 < G;
        const int nstage = stage == 0 ? STAGES - 1 : stage - 1;
        if (LW == 0 && STAGES >= 2 && more) issue(nstage);
        const unsigned char* base = lds + stage * STAGE_BYTES;
        const uint4* ldsA = reinterpret_cast<const uint4*>(base) + lane;
        const unsigned char* ldsB = base + A_CHUNKS * 1024;
        if constexpr (LW > 0 && TERMS != 3 && !UP2 && NBW == 4 && !C3_LC_GENERIC_LOOP) {
            // Compute wave of the loader / compute split, plain bf16: this wave is alone on its SIMD's matrix pipe, so the LDS latency
            // of a fragment read must be covered by its own MFMAs.  All 18 weight fragments of the K-group stay in registers (72
            // VGPRs; each is used by 4 rows), the 18 patch fragments (6 patch rows x 3 columns) stream through a ring of four,
            // read three fragments (>= 6 MFMAs = 192 cycles) ahead; a patch fragment (rr, dx) feeds every row nb with a tap
            // kh = rr - nb in 0..2: 36 reads per 72 MFMAs.
            Frag8 A[MBW][9], Bq[4];
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int m = 0; m < MBW; ++m) A[m][t].q = ldsA[(m * 9 + t) * 64];
            auto load_bf = [&](int f) { Bq[f & 3].q = *reinterpret_cast<const uint4*>(ldsB + brd[f / 3][f % 3]); };
            load_bf(0); load_bf(1); load_bf(2);
#pragma unroll
            for (int f = 0; f < 18; ++f) {
                if (f + 3 < 18) load_bf(f + 3);
                __builtin_amdgcn_sched_barrier(0);
                const int rr = f / 3, dx = f % 3;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    const int kh = rr - nb;
                    if (kh >= 0 && kh <= 2) {
#pragma unroll
                        for (int m = 0; m < MBW; ++m)
                            acc[0][m][nb] = mfma16<TERMS>(A[m][kh * 3 + dx].v, Bq[f & 3].v, acc[0][m][nb], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (UP2 && TERMS != 3 && C3_UP_BCACHE) {
            // Up-sampling layers, single-part operands (round 4).  The nine taps of the transposed convolution read only 2 x 2
            // input offsets (dy, dx = 1 - (k >> 1)), i.e. (NBW + 1) x 2 = 6 distinct patch fragments per K-group for the wave's NBW
            // rows - the tap loop above fetches one per (tap, row): 18.  With MBW = 1 (four phase accumulators fill the register
            // budget) every MFMA also needs its own weight fragment, so the loop ran at 1.5 LDS fragment reads per MFMA with all
            // four SIMDs sharing one 256 B/clk LDS: the K loop was LDS-read bound (matrix pipe 0.22-0.25 busy over the kernel).
            // Here the six patch fragments of the K-group sit in registers (24 VGPRs): 9 + 6 = 15 reads per 18 MFMAs.
            Frag8 Bc[NBW + 1][2];
#pragma unroll
            for (int rr = 0; rr <= NBW; ++rr)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) Bc[rr][dx].q = *reinterpret_cast<const uint4*>(ldsB + brd[rr][dx]);
            Frag8 ah[2][MBW];
#pragma unroll
            for (int m = 0; m < MBW; ++m) ah[0][m].q = ldsA[((m * 9 + 0) * PARTS) * 64];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kh = t / 3, kw = t % 3, dy = 1 - (kh >> 1), dx = 1 - (kw >> 1), a = (kh & 1) * 2 + (kw & 1);
                if (t + 1 < 9) {
#pragma unroll
                    for (int m = 0; m < MBW; ++m) ah[(t + 1) & 1][m].q = ldsA[((m * 9 + t + 1) * PARTS) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                    for (int m = 0; m < MBW; ++m) acc[a][m][nb] = mfma16<TERMS>(ah[t & 1][m].v, Bc[nb + dy][dx].v, acc[a][m][nb], 0, 0, 0);
                if (kw == 2 && edge_tile && wave == 0) {   // wave-uniform: the extra column of the unfused form
                    Frag8 eh;
                    eh.q = *reinterpret_cast<const uint4*>(ldsB + brde[dy]);
#pragma unroll
                    for (int m = 0; m < MBW; ++m) acce[kh & 1][m] = mfma16<TERMS>(ah[t & 1][m].v, eh.v, acce[kh & 1][m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        // Fragment reads run ONE (tap, N-block) step ahead of the MFMAs that use them (register double buffer, order pinned by
        // sched_barrier): the compiler's own schedule issues a read one or two MFMAs before its use, which leaves the matrix pipe
        // idle for most of the LDS latency some thirty times per K-group while this wave is the only one computing on its SIMD.
        auto load_a = [&](int t, Frag8 (&ah_)[MBW], Frag8 (&al_)[MBW]) {
#pragma unroll
            for (int m = 0; m < MBW; ++m) {
                ah_[m].q = ldsA[((m * 9 + t) * PARTS + 0) * 64];
                if (TERMS == 3) al_[m].q = ldsA[((m * 9 + t) * PARTS + 1) * 64];
            }
        };
        auto load_b = [&](int t, int nb, Frag8& bh_, Frag8& bl_) {
            const int kh = t / 3, kw = t % 3;
            const int dy = UP2 ? 1 - (kh >> 1) : kh, dx = UP2 ? 1 - (kw >> 1) : kw;
            bh_.q = *reinterpret_cast<const uint4*>(ldsB + brd[nb + dy][dx]);
            if (TERMS == 3) bl_.q = *reinterpret_cast<const uint4*>(ldsB + C3_B_BYTES + brd[nb + dy][dx]);
        };
        Frag8 ah[2][MBW], al[2][MBW], bh[2], bl[2];
        load_a(0, ah[0], al[0]);
        load_b(0, 0, bh[0], bl[0]);
#pragma unroll
        for (int s_ = 0; s_ < 9 * NBW; ++s_) {
            const int t = s_ / NBW, nb = s_ % NBW, kh = t / 3, kw = t % 3;
            const int a = UP2 ? (kh & 1) * 2 + (kw & 1) : 0;
            if (s_ + 1 < 9 * NBW) {
                const int t1 = (s_ + 1) / NBW, nb1 = (s_ + 1) % NBW;
                load_b(t1, nb1, bh[(s_ + 1) & 1], bl[(s_ + 1) & 1]);
                if (nb1 == 0) load_a(t1, ah[t1 & 1], al[t1 & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MBW; ++m) {
                acc[a][m][nb] = mfma16<TERMS>(ah[t & 1][m].v, bh[s_ & 1].v, acc[a][m][nb], 0, 0, 0);
                if (TERMS == 3) {
                    acc[a][m][nb] = mfma16<TERMS>(ah[t & 1][m].v, bl[s_ & 1].v, acc[a][m][nb], 0, 0, 0);
                    acc[a][m][nb] = mfma16<TERMS>(al[t & 1][m].v, bh[s_ & 1].v, acc[a][m][nb], 0, 0, 0);
                }
            }
            if (UP2 && nb == NBW - 1 && kw == 2 && edge_tile && wave == 0) {   // wave-uniform
                Frag8 eh, el;
                const int dy = 1 - (kh >> 1);
                eh.q = *reinterpret_cast<const uint4*>(ldsB + brde[dy]);
                if (TERMS == 3) el.q = *reinterpret_cast<const uint4*>(ldsB + C3_B_BYTES + brde[dy]);
#pragma unroll
                for (int m = 0; m < MBW; ++m) {
                    acce[kh & 1][m] = mfma16<TERMS>(ah[t & 1][m].v, eh.v, acce[kh & 1][m], 0, 0, 0);
                    if (TERMS == 3) {
                        acce[kh & 1][m] = mfma16<TERMS>(ah[t & 1][m].v, el.v, acce[kh & 1][m], 0, 0, 0);
                        acce[kh & 1][m] = mfma16<TERMS>(al[t & 1][m].v, eh.v, acce[kh & 1][m], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        stage = stage + 1 == STAGES ? 0 : stage + 1;
    }

    // ---- epilogue: lane (j,h) register r holds out channel 32mb + (r&3) + 8(r>>2) + 4h of pixel (row, j) ----
    if (UP2 && edge_tile && wave == 0 && j < C3_TH && ty0 + j <= P.H) {           // edge column: T[2y + a][2W], a = 0, 1
        const int TH2 = 2 * P.H + 1, TW2 = 2 * P.W + 1, y = ty0 + j;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int Y = 2 * y + a;
            if (Y >= TH2) continue;
#pragma unroll
            for (int m = 0; m < MBW; ++m) {
                float* dst = (KS > 1 ? P.partial + (long long)ks * P.N * TH2 * TW2 * P.Cout : P.scratch) +
                             (((long long)n * TH2 + Y) * TW2 + (TW2 - 1)) * P.Cout + 32 * (mb0 + m) + 4 * h;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    *reinterpret_cast<float4*>(dst + 8 * qq) = make_float4(acce[UP2 ? a : 0][m][4 * qq], acce[UP2 ? a : 0][m][4 * qq + 1],
                                                                            acce[UP2 ? a : 0][m][4 * qq + 2], acce[UP2 ? a : 0][m][4 * qq + 3]);
            }
        }
    }
    const bool fuse_rgb = !UP2 && P.rgb_w != nullptr;
    float* wmod = reinterpret_cast<float*>(lds);       // [rgb_c][32 * MBW]: ToRGB weight x style of this workgroup's channels
    // per-wave 32 px x 32 channel tile (row stride 36 floats) behind wmod: the activation goes through it so that every store
    // instruction writes full lines (8 pixels x 128 contiguous bytes of fp32, or 32 pixels x 32 bytes = 1 KiB of one 16-channel
    // plane of the consumer's bf16 image) instead of 64 separate 16-byte pieces
    constexpr int ST_STRIDE = 36;
    float* stile = reinterpret_cast<float*>(lds) + 4 * 32 * MBW + wave * (NBW * 32 * ST_STRIDE);      // NBW tiles per wave (one per image row it owns)
    static_assert(UP2 || (4 * 32 * MBW + WV * NBW * 32 * ST_STRIDE) * 4 <= STAGES * STAGE_BYTES, "epilogue tiles must fit the ring");
    if (KS == 1 || UP2) __syncthreads();               // every wave is done with the last K-group's fragments: LDS is free
    if (fuse_rgb) {
        for (int i = tid; i < P.rgb_c * 32 * MBW; i += 64 * WV) {
            const int c = i / (32 * MBW), ch = 32 * mb0 + i % (32 * MBW);
            wmod[i] = P.rgb_w[(long long)c * P.Cout + ch] * P.rgb_s[(long long)n * P.Cout + ch];
        }
        __syncthreads();
    }
    if constexpr (UP2) { if (fusedup) {
        // Fused FIR (round 3).  Per slice of 8 channels (accumulator registers 4 qq .. 4 qq + 3 of both lane halves) the tile's
        // transposed-conv result goes to LDS as Tl[h][2 ROWS][64] float4; thread = (row segment, lane half, output column) then
        // walks down its segment with a sliding window of row-filtered values (4 LDS reads + 8 FMA x 4 channels per output):
        // out[Y][X] = act(dcoef * sum_ab F[a] F[b] T[Y+a-1][X+b-1] + noise + bias), F = [1,3,3,1]/4 - upfir_kernel's arithmetic in
        // its order.  The tile yields the output rows / columns 2 .. 2 ROWS - 3 / 2 .. 61 of its 2 ROWS x 64 scratch pixels.
        static_assert(MBW == 1 && (WV == 4 || WV == 8), "fused up-sampling epilogue: one M-block, 4 or 8 waves");
        constexpr int TYL = 2 * ROWS, XS = 64, NSEG = WV / 2, SEG_ROWS = (2 * ROWS - 4) / NSEG;
        static_assert(SEG_ROWS * NSEG == 2 * ROWS - 4 && 120 * NSEG <= 64 * WV, "row segments");
        float4* Tl = reinterpret_cast<float4*>(lds);
        const int OH = 2 * P.H, OW = 2 * P.W;
        const int seg = tid / 120, hs = (tid % 120) / 60, xs = tid % 60;
        const bool active = seg < NSEG;
        const int Xl = 2 + xs, y0 = 2 + seg * SEG_ROWS;
        const int X = 2 * tx0 + Xl;
        const bool xok = active && X >= 0 && X < OW;
        const float F[4] = {0.25f, 0.75f, 0.75f, 0.25f};
        float nzs[SEG_ROWS];
#pragma unroll
        for (int r = 0; r < SEG_ROWS; ++r) {
            const int Y = 2 * ty0 + y0 + r;
            nzs[r] = (P.noise && xok && Y >= 0 && Y < OH) ? P.noise[n * P.noise_n_stride + (long long)Y * OW + X] * P.noise_strength : 0.0f;
        }
        // slice qq (accumulator registers 4 qq .. 4 qq + 3 of both lane halves) -> slice buffer qq & 1 (C3_UP_DBUF) or the only one
        constexpr int SLICE = 2 * TYL * XS;                    // float4 elements of one slice buffer
        auto put_slice = [&](int qq) {
            float4* dst = Tl + (C3_UP_DBUF ? (qq & 1) * SLICE : 0);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int a = 0; a < NACC; ++a)
                    dst[(h * TYL + 2 * (NBW * wave + nb) + (a >> 1)) * XS + 2 * j + (a & 1)] =
                        make_float4(acc[a][0][nb][4 * qq], acc[a][0][nb][4 * qq + 1], acc[a][0][nb][4 * qq + 2], acc[a][0][nb][4 * qq + 3]);
        };
        // Store addresses (round 5): element offsets of this thread's first output row at slice 0, advanced by one row per step and by
        // constants per slice - the per-row 64-bit index arithmetic (n, Y, X, channel quad -> offset, twice) was 100 of a slice's 690
        // instructions, in a kernel whose waves issue instructions for 45 % of their lives.
        const int Yf = 2 * ty0 + y0;
        const long long o_first = (((long long)n * OH + Yf) * OW + X) * P.Cout + 4 * (8 * mb0 + hs), o_row = (long long)OW * P.Cout;
        const long long s_first = split_index(n, P.Cout >> 4, OH, OW, Yf, X, 8 * mb0 + hs), s_row = (long long)OW * 4, s_plane = (long long)OH * OW * 4;
        if (C3_UP_DBUF) { put_slice(0); __syncthreads(); }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            if (C3_UP_DBUF) { if (qq + 1 < 4) put_slice(qq + 1); }      // the other buffer: its readers finished before the last barrier
            else { put_slice(qq); __syncthreads(); }
            if (active) {
                const float4 d = *reinterpret_cast<const float4*>(ec + 8 * qq + 4 * hs);
                const float4 b = *reinterpret_cast<const float4*>(ec + EC + 8 * qq + 4 * hs);
                const float4 s2 = *reinterpret_cast<const float4*>(ec + 2 * EC + 8 * qq + 4 * hs);
                const float4* col = Tl + (C3_UP_DBUF ? (qq & 1) * SLICE : 0) + (hs * TYL) * XS + Xl - 1;
                // (The FIR as packed fp32 FMAs - half the instructions, bit-identical - was measured in round 5: -6 % on the 32-channel layer,
                // whose launch is all epilogue, +2.5 / +6 % on the 256-channel layers: packed fp32 runs on the matrix pipe and contends with the
                // co-resident workgroup's MFMAs.  Scalar FMAs stay.)
                auto hrow = [&](int yl) {
                    const float4* rp = col + yl * XS;
                    const float4 t0 = rp[0], t1 = rp[1], t2 = rp[2], t3 = rp[3];
                    float4 a4 = make_float4(0, 0, 0, 0);
                    a4.x = fmaf(F[0], t0.x, a4.x); a4.y = fmaf(F[0], t0.y, a4.y); a4.z = fmaf(F[0], t0.z, a4.z); a4.w = fmaf(F[0], t0.w, a4.w);
                    a4.x = fmaf(F[1], t1.x, a4.x); a4.y = fmaf(F[1], t1.y, a4.y); a4.z = fmaf(F[1], t1.z, a4.z); a4.w = fmaf(F[1], t1.w, a4.w);
                    a4.x = fmaf(F[2], t2.x, a4.x); a4.y = fmaf(F[2], t2.y, a4.y); a4.z = fmaf(F[2], t2.z, a4.z); a4.w = fmaf(F[2], t2.w, a4.w);
                    a4.x = fmaf(F[3], t3.x, a4.x); a4.y = fmaf(F[3], t3.y, a4.y); a4.z = fmaf(F[3], t3.z, a4.z); a4.w = fmaf(F[3], t3.w, a4.w);
                    return a4;
                };
                float4 w0 = hrow(y0 - 1), w1 = hrow(y0), w2 = hrow(y0 + 1);
                long long oi = o_first + 8 * qq - o_row, si = s_first + (qq >> 1) * s_plane + 2 * (qq & 1) - s_row;   // channel quad 8 mb0 + 2 qq + hs
#pragma unroll
                for (int r = 0; r < SEG_ROWS; ++r) {
                    oi += o_row; si += s_row;
                    const float4 w3 = hrow(y0 + r + 2);
                    float4 sm = make_float4(0, 0, 0, 0);
                    sm.x = fmaf(F[0], w0.x, sm.x); sm.y = fmaf(F[0], w0.y, sm.y); sm.z = fmaf(F[0], w0.z, sm.z); sm.w = fmaf(F[0], w0.w, sm.w);
                    sm.x = fmaf(F[1], w1.x, sm.x); sm.y = fmaf(F[1], w1.y, sm.y); sm.z = fmaf(F[1], w1.z, sm.z); sm.w = fmaf(F[1], w1.w, sm.w);
                    sm.x = fmaf(F[2], w2.x, sm.x); sm.y = fmaf(F[2], w2.y, sm.y); sm.z = fmaf(F[2], w2.z, sm.z); sm.w = fmaf(F[2], w2.w, sm.w);
                    sm.x = fmaf(F[3], w3.x, sm.x); sm.y = fmaf(F[3], w3.y, sm.y); sm.z = fmaf(F[3], w3.z, sm.z); sm.w = fmaf(F[3], w3.w, sm.w);
                    w0 = w1; w1 = w2; w2 = w3;
                    const int Y = 2 * ty0 + y0 + r;
                    if (!(xok && Y >= 0 && Y < OH)) continue;
                    const float nz = nzs[r];
                    float4 o;
                    o.x = epilogue_act(sm.x * d.x + nz + b.x, P.lrelu, P.act_gain, P.clamp);
                    o.y = epilogue_act(sm.y * d.y + nz + b.y, P.lrelu, P.act_gain, P.clamp);
                    o.z = epilogue_act(sm.z * d.z + nz + b.z, P.lrelu, P.act_gain, P.clamp);
                    o.w = epilogue_act(sm.w * d.w + nz + b.w, P.lrelu, P.act_gain, P.clamp);
                    if (P.out) *reinterpret_cast<float4*>(P.out + oi) = o;
                    if (P.split_hi) {
                        unsigned h0, l0, h1, l1;
                        if (TERMS == 3) { split2<3>(o.x * s2.x, o.y * s2.y, h0, l0); split2<3>(o.z * s2.z, o.w * s2.w, h1, l1); P.split_lo[si] = make_uint2(l0, l1); }
                        else { split2<TERMS>(o.x * s2.x, o.y * s2.y, h0, l0); split2<TERMS>(o.z * s2.z, o.w * s2.w, h1, l1); }
                        P.split_hi[si] = make_uint2(h0, h1);
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // T[2y + ay][2x + bx][channels]: the 32 channels of an M-block are one 128-byte line of a scratch pixel; through the
        // wave's LDS tile a store instruction writes 8 such lines instead of 16 bytes of 64 (round 3: the scattered form cost
        // 64 address cycles per instruction, 46 % of the up-sampling kernel's time)
        const int TH2 = 2 * P.H + 1, TW2 = 2 * P.W + 1;
        float* tdst = (KS > 1 ? P.partial + (long long)ks * P.N * TH2 * TW2 * P.Cout : P.scratch) + (long long)n * TH2 * TW2 * P.Cout;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int y = ty0 + NBW * wave + nb;
            if (y > P.H) continue;                     // wave-uniform
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                const int Y = 2 * y + (a >> 1);
                if (Y >= TH2) continue;                // wave-uniform
#pragma unroll
                for (int m = 0; m < MBW; ++m) {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        *reinterpret_cast<float4*>(stile + j * ST_STRIDE + 8 * qq + 4 * h) =
                            make_float4(acc[a][m][nb][4 * qq], acc[a][m][nb][4 * qq + 1], acc[a][m][nb][4 * qq + 2], acc[a][m][nb][4 * qq + 3]);
                    float* row = tdst + ((long long)Y * TW2 + (a & 1)) * P.Cout + 32 * (mb0 + m);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {   // same-wave LDS operations execute in order: no barrier needed
                        const int p = 8 * it + (lane >> 3), c = lane & 7, x = tx0 + p;
                        const float4 v = *reinterpret_cast<const float4*>(stile + p * ST_STRIDE + 4 * c);
                        if (x <= P.W && 2 * x + (a & 1) < TW2) *reinterpret_cast<float4*>(row + (long long)(2 * x) * P.Cout + 4 * c) = v;
                    }
                }
            }
        }
    } } else if (KS > 1) {                              // raw partial sums of this K slice; the epilogue runs in splitk_reduce_kernel
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int y = ty0 + NBW * wave + nb, x = tx0 + j;
            if (y >= P.H || x >= P.W) continue;
            float* dst = P.partial + ((((long long)ks * P.N + n) * P.H + y) * P.W + x) * P.Cout + 32 * mb0 + 4 * h;
#pragma unroll
            for (int m = 0; m < MBW; ++m)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    *reinterpret_cast<float4*>(dst + 32 * m + 8 * qq) = make_float4(acc[0][m][nb][4 * qq], acc[0][m][nb][4 * qq + 1],
                                                                                      acc[0][m][nb][4 * qq + 2], acc[0][m][nb][4 * qq + 3]);
        }
    } else {
        float rgb[NBW][4];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int c = 0; c < 4; ++c) rgb[nb][c] = 0.0f;
        const int sp = lane >> 1, sh = lane & 1;        // consumer's image: lane = (pixel of 32, 8-channel half of a 16-channel plane)
        // Loop order m -> channel octet -> row: the per-channel constants of an octet (demodulation, bias, ToRGB weights: 5 LDS reads)
        // are read ONCE and serve the wave's NBW rows, whose arithmetic covers the next reads' latency; every row has its own
        // transposition tile.  (Row -> octet re-read them per row: 20 dependent LDS round trips per 32 x 32 block at two to four waves
        // per SIMD were most of the epilogue's 17 k - 36 k cycles.)
#pragma unroll
        for (int m = 0; m < MBW; ++m) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float4 d = make_float4(1, 1, 1, 1), b = make_float4(0.1f, 0.2f, 0.3f, 0.4f), wq[4];
                d = *reinterpret_cast<const float4*>(ec + 32 * m + 8 * qq + 4 * h);
                b = *reinterpret_cast<const float4*>(ec + EC + 32 * m + 8 * qq + 4 * h);
                if (fuse_rgb) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < P.rgb_c) wq[c] = *reinterpret_cast<const float4*>(wmod + c * 32 * MBW + 32 * m + 8 * qq + 4 * h);
                }
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    if (ty0 + NBW * wave + nb >= P.H) continue;      // wave-uniform; lanes beyond the image width compute along and are masked at the store
                    const float nz = nzv[nb];
                    float4 v;
                    v.x = epilogue_act(acc[0][m][nb][4 * qq + 0] * d.x + nz + b.x, P.lrelu, P.act_gain, P.clamp);
                    v.y = epilogue_act(acc[0][m][nb][4 * qq + 1] * d.y + nz + b.y, P.lrelu, P.act_gain, P.clamp);
                    v.z = epilogue_act(acc[0][m][nb][4 * qq + 2] * d.z + nz + b.z, P.lrelu, P.act_gain, P.clamp);
                    v.w = epilogue_act(acc[0][m][nb][4 * qq + 3] * d.w + nz + b.w, P.lrelu, P.act_gain, P.clamp);
                    if (P.out || P.split_hi) *reinterpret_cast<float4*>(stile + nb * (32 * ST_STRIDE) + j * ST_STRIDE + 8 * qq + 4 * h) = v;
                    if (fuse_rgb) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c < P.rgb_c) rgb[nb][c] = fmaf(v.x, wq[c].x, fmaf(v.y, wq[c].y, fmaf(v.z, wq[c].z, fmaf(v.w, wq[c].w, rgb[nb][c]))));
                    }
                }
            }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const int y = ty0 + NBW * wave + nb;
                if (y >= P.H) continue;
                const float* st_nb = stile + nb * (32 * ST_STRIDE);
                if (P.out) {                    // same-wave LDS operations execute in order: no barrier between the writes above and these reads
                    const long long o_row = (((long long)n * P.H + y) * P.W + tx0) * P.Cout + 32 * (mb0 + m);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int p = 8 * it + (lane >> 3), c = lane & 7;
                        const float4 v = *reinterpret_cast<const float4*>(st_nb + p * ST_STRIDE + 4 * c);
                        if (tx0 + p < P.W) *reinterpret_cast<float4*>(P.out + o_row + (long long)p * P.Cout + 4 * c) = v;
                    }
                }
                if (P.split_hi) {               // one store instruction = the 32 pixels x 16 channels of one plane of the group-major image:
#pragma unroll                                  // 1 KiB contiguous (round 3; was 16 pieces of 32 bytes per instruction)
                    for (int gI = 0; gI < 2; ++gI) {
                        const float* tp = st_nb + sp * ST_STRIDE + 16 * gI + 8 * sh;
                        const float* np_ = ec + 2 * EC + 32 * m + 16 * gI + 8 * sh;
                        const float4 v0 = *reinterpret_cast<const float4*>(tp), v1 = *reinterpret_cast<const float4*>(tp + 4);
                        float4 s0 = make_float4(1, 1, 1, 1), s1 = s0;
                        s0 = *reinterpret_cast<const float4*>(np_); s1 = *reinterpret_cast<const float4*>(np_ + 4);
                        uint4 hi4, lo4;
                        split2<TERMS>(v0.x * s0.x, v0.y * s0.y, hi4.x, lo4.x); split2<TERMS>(v0.z * s0.z, v0.w * s0.w, hi4.y, lo4.y);
                        split2<TERMS>(v1.x * s1.x, v1.y * s1.y, hi4.z, lo4.z); split2<TERMS>(v1.z * s1.z, v1.w * s1.w, hi4.w, lo4.w);
                        // uint2 units of split_index: 4 per (pixel, plane); this lane's 8 channels are units 2 sh, 2 sh + 1
                        const long long si = split_index(n, P.Cout >> 4, P.H, P.W, y, tx0 + sp, 4 * (2 * (mb0 + m) + gI) + 2 * sh);
                        if (tx0 + sp < P.W) {
                            *reinterpret_cast<uint4*>(P.split_hi + si) = hi4;
                            if (TERMS == 3) *reinterpret_cast<uint4*>(P.split_lo + si) = lo4;
                        }
                    }
                }
            }
        }
        if (fuse_rgb) {                                // the two lane halves hold complementary channels of pixel (y, x)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const int y = ty0 + NBW * wave + nb, x = tx0 + j;
#pragma unroll
                for (int c = 0; c < 4; ++c) rgb[nb][c] += __shfl_xor(rgb[nb][c], 32);
                if (h == 0 && y < P.H && x < P.W)
                    *reinterpret_cast<float4*>(P.rgb_partial + ((((long long)mbg_ * P.N + n) * P.H + y) * P.W + x) * 4) =
                        make_float4(rgb[nb][0], rgb[nb][1], rgb[nb][2], rgb[nb][3]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused up-sampling convolution, strip form (round 6).  conv3_kernel<.., UP2> with the FIR in its epilogue tiles the extended input
// grid with OVERLAPPING 32 x 8 tiles: the 4 x 4 FIR of an output needs the transposed-conv result T one pixel before and two after,
// so a tile yields 30 x 6 new pixels and every staged byte, every MFMA and the 1 080-instruction tile prologue are paid 1.42 times
// (profiles/experiments/r05_up_conv.md).  Here the vertical overlap is gone: a workgroup owns a 30-column STRIP and walks DOWN it in
// 8-row blocks; the FIR threads keep the last three row-filtered T rows of their column in registers from one block to the next
// (the sliding window upfir_kernel uses, across K loops), so block b emits the output rows [16 b - 2, 16 b + 14) and nothing is
// computed twice vertically (horizontally the strips still overlap by 2 of 32 columns).  A strip is cut into `segs` segments of
// `seg_blocks` blocks for parallelism; a segment cannot finish the three output rows across its upper boundary (their footprint
// needs rows of both segments), so each side of a boundary leaves its three row-filtered rows in `seam` and upconv_seam_kernel
// finishes those rows - 3 of every 16 x seg_blocks, an elementwise pass.  Same arithmetic in the same order as upfir_kernel and the
// overlapping form: bit-identical outputs (tests/test_dense_gpu.py::test_fused_up_layer_is_bit_identical_to_scratch_form).
//   workgroup = 4 waves, wave w owns the block's extended rows 2 w, 2 w + 1 (two N-blocks) x one M-block x four phase accumulators;
//   K loop    = conv3_kernel's (LDS-DMA ring over 16-channel K-groups, same patch swizzle and fragment order), with a 34 x 9 patch
//               (the transposed conv reads one row above, none below) and the K-group's chunks dealt round-robin to the waves:
//               19 chunks (bf16 / fp16) = 5,5,5,4 - no per-slot branches, running source pointers;
//   epilogue  = two passes of 16 channels: accumulators -> LDS slices [2][2][16][64] float4 (64 KB, over the ring), then wave
//               (slice, lane half), lane = output column: 16 rows x (row filter: 4 LDS reads + 16 FMA, column filter: 16 FMA,
//               demodulation / noise / bias / activation / consumer image).
// ------------------------------------------------------------------------------------------------
#define UPS_XCD_ORDER 1
// (The experiment switches of round 6 - UPS_ABLATE timing builds, UPS_PROFILE phase stamps, UPS_PRIO, UPS_DMA_SPREAD, the read-ahead depth -
// left this file at the end of the round; commit 74f5a14 has them, profiles/experiments/r06_up_conv.md what they measured.)
constexpr int UPS_A_AHEAD = 3;                                    // weight fragments read this many taps ahead of their MFMAs (1 measured equal)
// NBW = image rows per wave: 2 (32 x 8 blocks, two workgroups per CU) or 1 (32 x 4 blocks: 64 accumulator registers per wave instead of
// 128, three workgroups per CU - the occupancy experiment of profiles/experiments/r06_up_conv.md, section 5).
constexpr int UPS_PW = 34;
template <int NBW> struct UpsGeo {
    static constexpr int ROWS = 4 * NBW, PH = ROWS + 1, TYL = 2 * ROWS;      // extended rows per block, patch rows, T rows per block
    static constexpr int HALF_ITEMS = UPS_PW * PH, B_CHUNKS = (2 * HALF_ITEMS + 63) / 64, B_BYTES = B_CHUNKS * 1024;
};
// Slices Tl[slice 2][lane half 2][T row 16][UPS_XS] float4: a row holds its 32 even T columns at [0, 32) and its 32 odd ones at [40, 72).
// The accumulators of a lane are the columns 2 j + b: with the columns in order a wave's ds_write_b128 strides 32 bytes (2-way bank
// conflicts on every slice write: SQ_LDS_BANK_CONFLICT 1.5e7 per launch of the largest layer); split by parity the writes are contiguous,
// and the row filter's reads (columns X - 1 .. X + 2 of consecutive lanes) stay conflict-free because the odd half starts 8 slots (half a
// 256-byte bank row) past a multiple of 16: within each of ds_read_b128's 16-lane groups the even and the odd lanes take complementary slots.
constexpr int UPS_XS = 72, UPS_ODD = 40;
template <int NBW> constexpr int ups_slice_bytes() { return 2 * 2 * UpsGeo<NBW>::TYL * UPS_XS * 16; }
// KG = 16-channel K-groups per ring stage (1 or 2): one barrier, one wait and one round of first fragment reads per KG x 18 MFMAs of a wave.
// The one-row-per-wave experiment (NBW = 1: three workgroups per CU, 9 MFMAs per wave and K-group) took as long per K-group as two rows
// do - the K loop pays ~1 400 cycles per K-GROUP that are not MFMA (profiles/experiments/r06_up_conv.md) - so the stage is widened
// instead: two stages of 38 KiB; the block's noise rows then live in the ring's tail beyond the slices and are fetched after the K loop.
template <int TERMS, int NBW, int KG> constexpr int ups_stage_bytes() { return KG * (9 + UpsGeo<NBW>::B_CHUNKS) * (TERMS == 3 ? 2 : 1) * 1024; }
template <int TERMS, int STAGES, int NBW, int KG> constexpr bool ups_noise_late() {
    return STAGES * ups_stage_bytes<TERMS, NBW, KG>() >= ups_slice_bytes<NBW>() + UpsGeo<NBW>::TYL * 256;
}
template <int TERMS, int STAGES, int NBW, int KG> constexpr int ups_lds_bytes() {
    return (STAGES * ups_stage_bytes<TERMS, NBW, KG>() > ups_slice_bytes<NBW>() ? STAGES * ups_stage_bytes<TERMS, NBW, KG>() : ups_slice_bytes<NBW>()) + 512 +
           (ups_noise_late<TERMS, STAGES, NBW, KG>() ? 0 : UpsGeo<NBW>::TYL * 256);   // + epilogue constants (+ the block's noise rows)
}

template <int TERMS, int STAGES, int NBW = 2, int KG = 1>
__global__ __launch_bounds__(256, NBW == 1 ? 3 : 2) void upconv_strip_kernel(Conv3K P) {
    constexpr int PARTS = TERMS == 3 ? 2 : 1;
    constexpr int ROWS = UpsGeo<NBW>::ROWS, TYL = UpsGeo<NBW>::TYL, UPS_HALF_ITEMS = UpsGeo<NBW>::HALF_ITEMS, UPS_B_CHUNKS = UpsGeo<NBW>::B_CHUNKS,
                  UPS_B_BYTES = UpsGeo<NBW>::B_BYTES, UPS_SLICE_BYTES = ups_slice_bytes<NBW>();
    // chunks of one K-group, dealt round-robin to the four waves: chunk c = wave + 4 k of K-group `sub` of the stage sits at LDS chunk sub * CT1 + c.
    // The KG K-groups of a stage share the slots' geometry and pointers: K-group sub is fetched at pointer + sub x (one K-group's stride).
    constexpr int A_CHUNKS = 9 * PARTS, CT1 = A_CHUNKS + PARTS * UPS_B_CHUNKS, CT = CT1, SLOTS = (CT + 3) / 4;
    constexpr int STAGE_BYTES = ups_stage_bytes<TERMS, NBW, KG>();
    constexpr bool NOISE_LATE = ups_noise_late<TERMS, STAGES, NBW, KG>();
    static_assert(KG == 1 || (KG == 2 && TERMS != 3), "two K-groups per stage: single-part operands only");
    constexpr int EC_OFFSET = STAGES * STAGE_BYTES > UPS_SLICE_BYTES ? STAGES * STAGE_BYTES : UPS_SLICE_BYTES;
    static_assert(NBW == 1 || NBW == 2, "one or two image rows per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int item = blockIdx.x, mb0 = blockIdx.y;
    if (UPS_XCD_ORDER) {                                         // the M-blocks of an item back to back on one XCD: its patches are fetched into that L2 once
        const int MBG = gridDim.y, L = blockIdx.y * gridDim.x + blockIdx.x, k_ = L >> 3;
        item = (k_ / MBG) * 8 + (L & 7); mb0 = k_ % MBG;
        if (item >= P.c3_tiles) return;                          // grid.x is padded to a multiple of 8
    }
    const int seg = item / P.strips, strip = item % P.strips, n = blockIdx.z;
    const int by_first = seg * P.seg_blocks, by_end = min(by_first + P.seg_blocks, P.blocks);
    const int tx0 = strip * 30 - 1;
    const int G = P.Cin >> 4;
    const int OH = 2 * P.H, OW = 2 * P.W;

    // ---- per-lane constants of the staging slots: chunk c = wave + 4 k of the K-group's CT chunks; c < A_CHUNKS: weight chunk (tap, part),
    // else patch chunk: LDS item 2 pp + (hh ^ bit3(pp)) = channel half hh of patch pixel pp (conv3_kernel's layout)
    // patch slots: ((patch row * W + image column) * 16 + 8 * channel half) * 2 bytes - a multiple of 16 - with the patch row in the low
    // four bits, or bit 31 alone = padding lane.  A block adds its first row: no 64-bit arithmetic per slot and block.
    unsigned s_info[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int c = wave + 4 * k, cb = c - A_CHUNKS, bc = cb >= 0 ? cb % UPS_B_CHUNKS : 0;
        const int it = bc * 64 + lane, pp = it >> 1, hh = (it & 1) ^ ((pp >> 3) & 1), py = pp / UPS_PW, px = pp % UPS_PW;
        const int x = tx0 - 1 + px;
        s_info[k] = (cb >= 0 && c < CT && pp < UPS_HALF_ITEMS && x >= 0 && x < P.W) ? ((unsigned)((py * P.W + x) * 16 + 8 * hh) * 2u) | (unsigned)py : 0x80000000u;
    }
    const unsigned char* xbase[PARTS];                            // view n, K-group 0 of the split image (uniform)
    xbase[0] = reinterpret_cast<const unsigned char*>(P.xh + (long long)n * G * P.H * P.W * 16);
    if (PARTS == 2) xbase[PARTS - 1] = reinterpret_cast<const unsigned char*>(P.xl + (long long)n * G * P.H * P.W * 16);
    const unsigned char* sptr[SLOTS];                             // running source pointers, one K-group per issue()
    bool sadv[SLOTS];                                             // patch slots: does the lane advance (false: it stays on the zero page)
    const unsigned plane_bytes = (unsigned)P.H * (unsigned)P.W * 32u;
    auto issue_slot = [&](int stage, int k) {                     // k: compile-time after unrolling
        unsigned char* base = lds + stage * STAGE_BYTES;
        const int c = wave + 4 * k;                               // weight chunk if c < A_CHUNKS (wave-uniform)
        constexpr unsigned A_GROUP = 18u * 64u * 16u;             // bytes between the K-groups of the weight image
        const bool all_w = 4 * k + 3 < A_CHUNKS, all_p = 4 * k >= A_CHUNKS;      // compile time: the slot is a weight / a patch chunk in every wave
#pragma unroll
        for (int sub = 0; sub < KG; ++sub) {
            const unsigned char* src = sptr[k];
            if (sub) {
                if (all_w) src += A_GROUP;
                else if (all_p) src += sadv[k] ? plane_bytes : 0u;
                else if (c < A_CHUNKS) src += A_GROUP;
                else src += sadv[k] ? plane_bytes : 0u;
            }
            if (4 * k + 3 < CT || c < CT) lds_dma16(src, base + (sub * CT1 + c) * 1024);      // only the last slot can be past the end (wave-uniform)
        }
        if (all_w) sptr[k] += KG * A_GROUP;
        else if (all_p) sptr[k] += sadv[k] ? KG * plane_bytes : 0u;
        else if (c < A_CHUNKS) sptr[k] += KG * A_GROUP;           // mixed slot: wave-uniform branch
        else sptr[k] += sadv[k] ? KG * plane_bytes : 0u;
    };
    auto issue = [&](int stage) {
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) issue_slot(stage, k);
    };

    // fragment offsets inside the patch: rows NBW wave + (0..NBW), columns j + (0..1)
    int brd[NBW + 1][2];
#pragma unroll
    for (int rr = 0; rr < NBW + 1; ++rr)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int pp = (NBW * wave + rr) * UPS_PW + j + cc;
            brd[rr][cc] = (2 * pp + (h ^ ((pp >> 3) & 1))) * 16;
        }

    float* ec = reinterpret_cast<float*>(lds + EC_OFFSET);       // [0] demodulation, [1] bias, [2] the consuming layer's styles
    // [TYL output rows of the block][64 lanes = output columns]: raw noise - behind the constants, or (NOISE_LATE) in the ring's tail beyond the slices
    float* nzl = reinterpret_cast<float*>(NOISE_LATE ? lds + UPS_SLICE_BYTES : lds + EC_OFFSET + 512);
    if (tid < 32) {
        const int ch = 32 * mb0 + tid;
        ec[tid] = P.dcoef ? P.dcoef[(long long)n * P.Cout + ch] : 1.0f;
        ec[32 + tid] = P.bias[ch];
        ec[64 + tid] = P.split_hi ? P.next_styles[(long long)n * P.Cout + ch] : 0.0f;
    }

    // ---- the FIR role of this thread: wave -> (slice s, lane half hs), lane -> output column
    const int fs = wave >> 1, fhs = wave & 1, xs = lane;
    const int Xl = 2 + min(xs, 59), X = 2 * tx0 + Xl;
    const bool xok = xs < 60 && X >= 0 && X < OW;
    const float F[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    float4 win[2][3];                                             // row-filtered T rows Yt - 3 .. Yt - 1 of this thread's column, per pass
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) win[p][q] = make_float4(0, 0, 0, 0);
    float4* Tl = reinterpret_cast<float4*>(lds);
    const long long o_row = (long long)OW * P.Cout, s_row = (long long)OW * 4;
    const long long seam_row = (long long)OW * P.Cout;

    for (int by = by_first; by < by_end; ++by) {
        const int r0 = ROWS * by;                                // first extended row of the block; patch row py <-> image row r0 - 1 + py
        // staging pointers of this block, K-group 0
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int c = wave + 4 * k;
            if (c < A_CHUNKS) {                                   // weight chunk (t, part) <- packed[((mb0 * G + g) * 18 + t * 2 + part)]   (wave-uniform)
                const int part = c % PARTS, t = (c / PARTS) % 9;
                sptr[k] = reinterpret_cast<const unsigned char*>(P.packed + (((long long)mb0 * G) * 18 + t * 2 + part) * 64 + lane);
                sadv[k] = true;
            } else {
                const int cb = c - A_CHUNKS, part = cb / UPS_B_CHUNKS;
                // ONE compare decides the lane (tools/lint_lane_masks.py, S1: no select on a scalar-combined lane mask - written as two
                // conditions the compiler hoists the padding test out of the block loop as a lane mask and ANDs it with the row test on
                // the scalar unit): a padding lane's bit 31 becomes a row far below the image by arithmetic.
                const int y = r0 - 1 + (int)(s_info[k] & 15u) + (((int)s_info[k] >> 31) & 0x40000000);
                const bool ok = (unsigned)y < (unsigned)P.H;
                const long long off = (long long)(s_info[k] & 0x7ffffff0u) + (long long)(r0 - 1) * P.W * 32;      // (r0 - 1) may be -1: those lanes are not ok
                sptr[k] = ok ? xbase[PARTS == 2 ? part : 0] + off : reinterpret_cast<const unsigned char*>(nfe_zero16);
                sadv[k] = ok;
            }
        }
        // the block's TYL noise rows -> LDS by LDS-DMA (wave w: rows (TYL / 4) w ..; lane = output column; clamped addresses: a value that is
        // out of range belongs to an output that is not stored).  They land before the first K-group's wait.
        auto fetch_noise = [&]() {
            if (!P.noise) return;
            const int Xc = min(max(2 * tx0 + 2 + lane, 0), OW - 1);
#pragma unroll
            for (int rr = 0; rr < TYL / 4; ++rr) {
                const int r = (TYL / 4) * wave + rr, Y = min(max(TYL * by + r - 2, 0), OH - 1);
                lds_dma4(P.noise + n * P.noise_n_stride + (long long)Y * OW + Xc, nzl + r * 64);
            }
        };
        if (!NOISE_LATE) fetch_noise();

        f32x16 acc[4][NBW];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][nb][r] = 0.0f;

        // ---- K loop
        // STAGES-deep ring: the loads of K-group g + STAGES - 1 are issued while g is computed, so a load has STAGES - 1 K-groups to land.
        // A lone workgroup on a CU spends 1 900 cycles per K-group with a ring of two (576 of them MFMA): the LDS-DMA round trip, not
        // the matrix pipe, paces the loop (UPS_PROFILE, round 6) - and the ring fits under the epilogue's 64 KB of slices for free.
        constexpr int MIN_LOADS = KG * (CT / 4);                 // fewest LDS-DMA instructions a wave issues per stage
        const int GS = G / KG;                                   // stages of this block's K loop (the launch checks G % KG == 0)
#pragma unroll
        for (int pre = 0; pre < STAGES - 1; ++pre)
            if (pre < GS) issue(pre);
        int stage = 0;
        for (int g = 0; g < GS; ++g) {
            if (STAGES == 1) { __syncthreads(); issue(0); }
            // K-group g has landed once at most the loads of the STAGES - 2 younger K-groups are outstanding (in-order return)
            if (STAGES <= 2 || g + STAGES - 2 >= GS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * MIN_LOADS) : "memory");
            __syncthreads();
            // The LDS-DMA of K-group g + STAGES - 1 goes out as one burst here.  (Spread over the MFMA steps - one instruction after every
            // other step - it measured the same: an instruction costs its ~65 cycles of issue wherever it sits; r06_up_conv.md.)
            const bool more = STAGES >= 2 && g + STAGES - 1 < GS;     // wave-uniform
            const int nstage = stage == 0 ? STAGES - 1 : stage - 1;
            if (more) issue(nstage);
#pragma unroll
            for (int sub = 0; sub < KG; ++sub) {
            const unsigned char* base = lds + stage * STAGE_BYTES + sub * (CT1 * 1024);
            const uint4* ldsA = reinterpret_cast<const uint4*>(base) + lane;
            const unsigned char* ldsB = base + A_CHUNKS * 1024;
            if constexpr (TERMS != 3) {
                // the nine taps read 2 x 2 input offsets: (2 + 1) x 2 = 6 distinct patch fragments for the wave's two rows, kept in registers
                Frag8 Bc[NBW + 1][2];
#pragma unroll
                for (int rr = 0; rr < NBW + 1; ++rr)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) Bc[rr][dx].q = *reinterpret_cast<const uint4*>(ldsB + brd[rr][dx]);
                // weight fragments UPS_A_AHEAD taps (2 MFMAs = 64 matrix cycles each) ahead of their use, through a ring of four
                Frag8 ah[4];
#pragma unroll
                for (int t = 0; t < UPS_A_AHEAD; ++t) ah[t].q = ldsA[t * 64];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int kh = t / 3, kw = t % 3, dy = 1 - (kh >> 1), dx = 1 - (kw >> 1), a = (kh & 1) * 2 + (kw & 1);
                    if (t + UPS_A_AHEAD < 9) ah[(t + UPS_A_AHEAD) & 3].q = ldsA[(t + UPS_A_AHEAD) * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) acc[a][nb] = mfma16<TERMS>(ah[t & 3].v, Bc[nb + dy][dx].v, acc[a][nb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // split-bf16: fragment reads one (tap, row) step ahead of the three MFMAs that use them
                auto load_a = [&](int t, Frag8& ah_, Frag8& al_) { ah_.q = ldsA[(t * 2 + 0) * 64]; al_.q = ldsA[(t * 2 + 1) * 64]; };
                auto load_b = [&](int t, int nb, Frag8& bh_, Frag8& bl_) {
                    const int kh = t / 3, kw = t % 3, dy = 1 - (kh >> 1), dx = 1 - (kw >> 1);
                    bh_.q = *reinterpret_cast<const uint4*>(ldsB + brd[nb + dy][dx]);
                    bl_.q = *reinterpret_cast<const uint4*>(ldsB + UPS_B_BYTES + brd[nb + dy][dx]);
                };
                Frag8 ah[2], al[2], bh[2], bl[2];
                load_a(0, ah[0], al[0]);
                load_b(0, 0, bh[0], bl[0]);
#pragma unroll
                for (int s_ = 0; s_ < 9 * NBW; ++s_) {
                    const int t = s_ / NBW, nb = s_ % NBW, kh = t / 3, kw = t % 3, a = (kh & 1) * 2 + (kw & 1);
                    if (s_ + 1 < 9 * NBW) {
                        const int t1 = (s_ + 1) / NBW, nb1 = (s_ + 1) % NBW;
                        load_b(t1, nb1, bh[(s_ + 1) & 1], bl[(s_ + 1) & 1]);
                        if (nb1 == 0) load_a(t1, ah[t1 & 1], al[t1 & 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[a][nb] = mfma16<TERMS>(ah[t & 1].v, bh[s_ & 1].v, acc[a][nb], 0, 0, 0);
                    acc[a][nb] = mfma16<TERMS>(ah[t & 1].v, bl[s_ & 1].v, acc[a][nb], 0, 0, 0);
                    acc[a][nb] = mfma16<TERMS>(al[t & 1].v, bh[s_ & 1].v, acc[a][nb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
            stage = stage + 1 == STAGES ? 0 : stage + 1;
        }
        __syncthreads();                                         // every wave is done with the last K-group's fragments: LDS is free
        if (NOISE_LATE) fetch_noise();                           // into the ring's tail; they land under the first pass's slice writes

        // ---- epilogue: lane (j, h) register r holds out channel 32 mb0 + (r & 3) + 8 (r >> 2) + 4 h of the pixel (row, j)
        const bool seam_top = by == by_first && seg > 0;         // wave-uniform: the window is not valid for this block's first three rows
        const bool seam_bottom = by + 1 == by_end && seg + 1 < P.segs;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const int qq = 2 * p + sl;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        Tl[((sl * 2 + h) * TYL + 2 * (NBW * wave + nb) + (a >> 1)) * UPS_XS + j + UPS_ODD * (a & 1)] =
                            make_float4(acc[a][nb][4 * qq], acc[a][nb][4 * qq + 1], acc[a][nb][4 * qq + 2], acc[a][nb][4 * qq + 3]);
            }
            if (NOISE_LATE && p == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            {
                const int qq = 2 * p + fs;
                const float4 d = *reinterpret_cast<const float4*>(ec + 8 * qq + 4 * fhs);
                const float4 b = *reinterpret_cast<const float4*>(ec + 32 + 8 * qq + 4 * fhs);
                const float4 s2 = *reinterpret_cast<const float4*>(ec + 64 + 8 * qq + 4 * fhs);
                // columns Xl - 1 .. Xl + 2: c0, c2 in one parity half, c1, c3 in the other
                const int c0 = Xl - 1;
                const float4* col02 = Tl + ((fs * 2 + fhs) * TYL) * UPS_XS + (c0 >> 1) + UPS_ODD * (c0 & 1);
                const float4* col13 = Tl + ((fs * 2 + fhs) * TYL) * UPS_XS + ((c0 + 1) >> 1) + UPS_ODD * ((c0 + 1) & 1);
                // the four T pixels under a row's filter are read one row ahead of their use (alone on a CU the row loop took 570 cycles
                // per row against ~300 of issue: every row waited for its own LDS reads), the block's noise values all at once
                auto load_row = [&](int yl, float4 (&t)[4]) {
                    t[0] = col02[yl * UPS_XS]; t[1] = col13[yl * UPS_XS]; t[2] = col02[yl * UPS_XS + 1]; t[3] = col13[yl * UPS_XS + 1];
                };
                auto filt_row = [&](const float4 (&t)[4]) {
                    float4 a4 = make_float4(0, 0, 0, 0);
                    a4.x = fmaf(F[0], t[0].x, a4.x); a4.y = fmaf(F[0], t[0].y, a4.y); a4.z = fmaf(F[0], t[0].z, a4.z); a4.w = fmaf(F[0], t[0].w, a4.w);
                    a4.x = fmaf(F[1], t[1].x, a4.x); a4.y = fmaf(F[1], t[1].y, a4.y); a4.z = fmaf(F[1], t[1].z, a4.z); a4.w = fmaf(F[1], t[1].w, a4.w);
                    a4.x = fmaf(F[2], t[2].x, a4.x); a4.y = fmaf(F[2], t[2].y, a4.y); a4.z = fmaf(F[2], t[2].z, a4.z); a4.w = fmaf(F[2], t[2].w, a4.w);
                    a4.x = fmaf(F[3], t[3].x, a4.x); a4.y = fmaf(F[3], t[3].y, a4.y); a4.z = fmaf(F[3], t[3].z, a4.z); a4.w = fmaf(F[3], t[3].w, a4.w);
                    return a4;
                };
                float4 trow[2][4];
                load_row(0, trow[0]);
                float nzr[TYL];
#pragma unroll
                for (int r = 0; r < TYL; ++r) nzr[r] = P.noise ? nzl[r * 64 + xs] : 0.0f;
                const int ch = 32 * mb0 + 8 * qq + 4 * fhs;
                const int Y0 = TYL * by - 2;                     // the output row the block's T row 0 completes
                long long oi = (((long long)n * OH + Y0) * OW + X) * P.Cout + ch - o_row;
                long long si = split_index(n, P.Cout >> 4, OH, OW, Y0, X, 8 * mb0 + 2 * qq + fhs) - s_row;
                float* seam = P.seam ? P.seam + ((((long long)n * (P.segs - 1) + (seg - 1)) * 6 + 3) * OW + X) * P.Cout + ch : nullptr;   // this segment's upper boundary, rows 3..5
                float4 w0 = win[p][0], w1 = win[p][1], w2 = win[p][2];
#pragma unroll
                for (int r = 0; r < TYL; ++r) {
                    oi += o_row; si += s_row;
                    if (r + 1 < TYL) load_row(r + 1, trow[(r + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    const float4 w3 = filt_row(trow[r & 1]);
                    if (r < 3 && seam_top && xok) *reinterpret_cast<float4*>(seam + r * seam_row) = w3;
                    float4 sm = make_float4(0, 0, 0, 0);
                    sm.x = fmaf(F[0], w0.x, sm.x); sm.y = fmaf(F[0], w0.y, sm.y); sm.z = fmaf(F[0], w0.z, sm.z); sm.w = fmaf(F[0], w0.w, sm.w);
                    sm.x = fmaf(F[1], w1.x, sm.x); sm.y = fmaf(F[1], w1.y, sm.y); sm.z = fmaf(F[1], w1.z, sm.z); sm.w = fmaf(F[1], w1.w, sm.w);
                    sm.x = fmaf(F[2], w2.x, sm.x); sm.y = fmaf(F[2], w2.y, sm.y); sm.z = fmaf(F[2], w2.z, sm.z); sm.w = fmaf(F[2], w2.w, sm.w);
                    sm.x = fmaf(F[3], w3.x, sm.x); sm.y = fmaf(F[3], w3.y, sm.y); sm.z = fmaf(F[3], w3.z, sm.z); sm.w = fmaf(F[3], w3.w, sm.w);
                    w0 = w1; w1 = w2; w2 = w3;
                    const int Y = Y0 + r;
                    if (!(xok && Y >= 0 && Y < OH) || (r < 3 && seam_top)) continue;
                    const float nz = nzr[r] * P.noise_strength;
                    float4 o;
                    o.x = epilogue_act(sm.x * d.x + nz + b.x, P.lrelu, P.act_gain, P.clamp);
                    o.y = epilogue_act(sm.y * d.y + nz + b.y, P.lrelu, P.act_gain, P.clamp);
                    o.z = epilogue_act(sm.z * d.z + nz + b.z, P.lrelu, P.act_gain, P.clamp);
                    o.w = epilogue_act(sm.w * d.w + nz + b.w, P.lrelu, P.act_gain, P.clamp);
                    if (P.out) *reinterpret_cast<float4*>(P.out + oi) = o;
                    if (P.split_hi) {
                        // (Round 6 also built the coalesced form - the lane's 8 bytes parked in its dead slice row, whole 32-byte pixels stored by
                        // the workgroup after the pass: 4 x fewer texture-addresser cycles and 5 % SLOWER, one more barrier per pass and the stores
                        // bunched where nothing else runs: profiles/experiments/r06_up_conv.md.)
                        unsigned h0, l0, h1, l1;
                        if (TERMS == 3) { split2<3>(o.x * s2.x, o.y * s2.y, h0, l0); split2<3>(o.z * s2.z, o.w * s2.w, h1, l1); P.split_lo[si] = make_uint2(l0, l1); }
                        else { split2<TERMS>(o.x * s2.x, o.y * s2.y, h0, l0); split2<TERMS>(o.z * s2.z, o.w * s2.w, h1, l1); }
                        P.split_hi[si] = make_uint2(h0, h1);
                    }
                }
                win[p][0] = w0; win[p][1] = w1; win[p][2] = w2;
                if (seam_bottom && xok) {                         // the next segment's upper boundary, rows 0..2: T rows 16 by_end - 3 .. - 1
                    float* sb = P.seam + ((((long long)n * (P.segs - 1) + seg) * 6) * OW + X) * P.Cout + ch;
                    *reinterpret_cast<float4*>(sb) = w0; *reinterpret_cast<float4*>(sb + seam_row) = w1; *reinterpret_cast<float4*>(sb + 2 * seam_row) = w2;
                }
            }
            __syncthreads();
        }
    }
}

// The three output rows across every segment boundary of upconv_strip_kernel: rows 16 b0 - 2 .. 16 b0 of the boundary in front of
// block b0, from the six row-filtered T rows 16 b0 - 3 .. 16 b0 + 2 the two segments left in `seam`; column filter and layer epilogue
// exactly as in the strip kernel (and upfir_kernel).  One thread = (view, boundary, column, channel quad).
template <int TERMS>
__global__ __launch_bounds__(256) void upconv_seam_kernel(Conv3K P) {
    const int OH = 2 * P.H, OW = 2 * P.W, C4 = P.Cout >> 2, seams = P.segs - 1;
    const long long total = (long long)P.N * seams * OW * C4;
    const float F[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4); long long r_ = i / C4;
        const int X = (int)(r_ % OW); r_ /= OW;
        const int q = (int)(r_ % seams), n = (int)(r_ / seams);
        const float* sp = P.seam + ((((long long)n * seams + q) * 6) * OW + X) * P.Cout + 4 * c4;
        float4 hr[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) hr[k] = *reinterpret_cast<const float4*>(sp + (long long)k * OW * P.Cout);
        const float4 d = P.dcoef ? *reinterpret_cast<const float4*>(P.dcoef + (long long)n * P.Cout + 4 * c4) : make_float4(1, 1, 1, 1);
        const float4 b = *reinterpret_cast<const float4*>(P.bias + 4 * c4);
        float4 s2 = make_float4(0, 0, 0, 0);
        if (P.split_hi) s2 = *reinterpret_cast<const float4*>(P.next_styles + (long long)n * P.Cout + 4 * c4);
        const int Y0 = P.tyl * (q + 1) * P.seg_blocks - 2;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int Y = Y0 + r;
            float4 sm = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                sm.x = fmaf(F[aa], hr[r + aa].x, sm.x); sm.y = fmaf(F[aa], hr[r + aa].y, sm.y);
                sm.z = fmaf(F[aa], hr[r + aa].z, sm.z); sm.w = fmaf(F[aa], hr[r + aa].w, sm.w);
            }
            if (Y < 0 || Y >= OH) continue;
            const float nz = P.noise ? P.noise[n * P.noise_n_stride + (long long)Y * OW + X] * P.noise_strength : 0.0f;
            float4 o;
            o.x = epilogue_act(sm.x * d.x + nz + b.x, P.lrelu, P.act_gain, P.clamp);
            o.y = epilogue_act(sm.y * d.y + nz + b.y, P.lrelu, P.act_gain, P.clamp);
            o.z = epilogue_act(sm.z * d.z + nz + b.z, P.lrelu, P.act_gain, P.clamp);
            o.w = epilogue_act(sm.w * d.w + nz + b.w, P.lrelu, P.act_gain, P.clamp);
            if (P.out) *reinterpret_cast<float4*>(P.out + (((long long)n * OH + Y) * OW + X) * P.Cout + 4 * c4) = o;
            if (P.split_hi) {
                unsigned h0, l0, h1, l1;
                const long long si = split_index(n, C4 >> 2, OH, OW, Y, X, c4);
                if (TERMS == 3) { split2<3>(o.x * s2.x, o.y * s2.y, h0, l0); split2<3>(o.z * s2.z, o.w * s2.w, h1, l1); P.split_lo[si] = make_uint2(l0, l1); }
                else { split2<TERMS>(o.x * s2.x, o.y * s2.y, h0, l0); split2<TERMS>(o.z * s2.z, o.w * s2.w, h1, l1); }
                P.split_hi[si] = make_uint2(h0, h1);
            }
        }
    }
}

// Fused ToRGB, second half: add the M-block-group partial sums of every pixel in group order, then what torgb's own epilogue
// does (bias, clamp; networks_stylegan2.py:353-357) and the skip path img = upsample2d(img) + y (:453-456).
__global__ __launch_bounds__(256) void rgb_combine_kernel(const float* __restrict__ partial, int groups, int N, int H, int W, int C,
                                                          const float* __restrict__ bias, float clamp, const float* __restrict__ skip,
                                                          float* __restrict__ out) {
    const long long npix = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        float4 s = reinterpret_cast<const float4*>(partial)[i];
        for (int g = 1; g < groups; ++g) {
            const float4 t = reinterpret_cast<const float4*>(partial)[i + g * npix];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        float v[4] = {s.x, s.y, s.z, s.w};
        const int n = (int)(i / ((long long)H * W)); const long long yx = i % ((long long)H * W);
        const int y = (int)(yx / W), x = (int)(yx % W);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) v[c] = epilogue_act(v[c] + bias[c], 0, 1.0f, clamp);
        if (skip) {
            const float4 sk = skip_up2(skip, n, H >> 1, W >> 1, C, y, x, 0);
            v[0] += sk.x; v[1] += sk.y; v[2] += sk.z; v[3] += sk.w;
        }
        for (int c = 0; c < C; ++c) out[i * C + c] = v[c];
    }
}

// ------------------------------------------------------------------------------------------------
// ToRGB fast path (1x1, Cin % 16 == 0, the layer's weight fragments fit LDS): HBM-bound, so every input pixel
// is read exactly once.  The weight fragments of all M-blocks are staged in LDS once per workgroup; a wave walks
// over 32-pixel N-blocks, loads its B operand (8 channels per lane, fp32) straight from global memory,
// applies the styles, converts to bf16 (hi[+lo]) in registers and feeds the MFMAs of all M-blocks.
// ------------------------------------------------------------------------------------------------
template <int TERMS, int MB>
__global__ __launch_bounds__(256, 2) void torgb_kernel(ConvK P) {
    constexpr int PARTS = TERMS == 3 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint4* ldsA = reinterpret_cast<uint4*>(lds);                     // [mb][g][part][lane]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = P.Cin >> 4;
    for (int i = tid; i < MB * G * PARTS * 64; i += 256) {
        const int ln = i & 63, part = (i >> 6) % PARTS, r = (i >> 6) / PARTS, g = r % G, mb = r / G;
        ldsA[i] = P.packed[(((long long)mb * G + g) * 2 + part) * 64 + ln];      // packed image: [mb][g][tap 0][part 2][lane]
    }
    __syncthreads();
    const int HW = P.H * P.W, nblk = (HW + 31) >> 5;
    const long long total = (long long)P.N * nblk;
    for (long long blk = (long long)blockIdx.x * 4 + wave; blk < total; blk += (long long)gridDim.x * 4) {
        const int n = (int)(blk / nblk), p = (int)(blk % nblk) * 32 + j;
        const bool valid = p < HW;
        const float* xp = P.x + ((long long)n * HW + min(p, HW - 1)) * P.Cin + 8 * h;
        const float* sp = P.styles + (long long)n * P.Cin + 8 * h;
        f32x16 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.0f;
        // the next K-group's 8 input channels and styles are requested before this one's MFMAs (round 3: the loop was
        // load -> wait -> 3 MFMAs per K-group, one exposed round trip each)
        float4 nx0 = *reinterpret_cast<const float4*>(xp), nx1 = *reinterpret_cast<const float4*>(xp + 4);
        float4 ns0 = *reinterpret_cast<const float4*>(sp), ns1 = *reinterpret_cast<const float4*>(sp + 4);
#pragma unroll 1
        for (int g = 0; g < G; ++g) {
            const float4 x0 = nx0, x1 = nx1, s0 = ns0, s1 = ns1;
            if (g + 1 < G) {
                nx0 = *reinterpret_cast<const float4*>(xp + 16 * (g + 1)); nx1 = *reinterpret_cast<const float4*>(xp + 16 * (g + 1) + 4);
                ns0 = *reinterpret_cast<const float4*>(sp + 16 * (g + 1)); ns1 = *reinterpret_cast<const float4*>(sp + 16 * (g + 1) + 4);
            }
            Frag8 bh, bl;
            split2<TERMS>(x0.x * s0.x, x0.y * s0.y, bh.u[0], bl.u[0]);
            split2<TERMS>(x0.z * s0.z, x0.w * s0.w, bh.u[1], bl.u[1]);
            split2<TERMS>(x1.x * s1.x, x1.y * s1.y, bh.u[2], bl.u[2]);
            split2<TERMS>(x1.z * s1.z, x1.w * s1.w, bh.u[3], bl.u[3]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                Frag8 ah, al;
                ah.q = ldsA[((mb * G + g) * PARTS + 0) * 64 + lane];
                acc[mb] = mfma16<TERMS>(ah.v, bh.v, acc[mb], 0, 0, 0);
                if (TERMS == 3) {
                    al.q = ldsA[((mb * G + g) * PARTS + 1) * 64 + lane];
                    acc[mb] = mfma16<TERMS>(ah.v, bl.v, acc[mb], 0, 0, 0);
                    acc[mb] = mfma16<TERMS>(al.v, bh.v, acc[mb], 0, 0, 0);
                }
            }
        }
        if (!valid) continue;
        const int y = p / P.W, x = p % P.W;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int o0 = 32 * mb + 8 * qq + 4 * h;
                if (o0 >= P.Cout) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = o0 + i;
                    v[i] = o < P.Cout ? epilogue_act(acc[mb][4 * qq + i] + P.bias[o], 0, P.act_gain, P.clamp) : 0.0f;
                }
                if (P.skip) {
                    const float4 sk = skip_up2(P.skip, n, P.H >> 1, P.W >> 1, P.Cout, y, x, o0);
                    v[0] += sk.x; v[1] += sk.y; v[2] += sk.z; v[3] += sk.w;
                }
                float* dst = P.out_planes ? P.out + ((((long long)n * 3 + mb) * P.H + y) * P.W + x) * 32 + 8 * qq + 4 * h
                                          : P.out + ((long long)n * HW + p) * P.Cout + o0;
                if (o0 + 3 < P.Cout && (P.Cout & 3) == 0) {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (o0 + i < P.Cout) dst[i] = v[i];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// ToRGB, coalesced form (round 5; rows that are a multiple of 32 pixels wide, Cout a multiple of 32 - the 96-channel tri-plane image
// path of the backbone).  torgb_kernel above gives every lane its own operand bytes straight from memory: a lane's 32 bytes of a
// K-group, its four skip taps and its output piece are 16-byte accesses 128 to 512 bytes apart, and the texture addresser handles
// such an instruction at one lane per clock (63 cycles against 17 for four adjacent lanes x 16 contiguous bytes:
// profiles/r01_gather_rate_microbench.txt) - counters of the shipped kernel: TA 0.70 busy, 80 % of a wave's life in s_waitcnt, HBM at
// 2.0-2.3 TB/s (profiles/r04_pmc_dense_bf16.txt).  Here every global access is four adjacent lanes on 64 contiguous bytes and the
// re-ordering happens in LDS:
//   * inputs: one K-group of the wave's 32 pixels = 2 KiB = two instructions (lane = pixel l >> 2, piece l & 3), written to a
//     swizzled LDS tile and read back in MFMA-operand order (lane (j, h): bytes 32 h .. + 31 of pixel j), four K-groups in flight;
//   * skip image (upsample2d of the previous resolution, networks_stylegan2.py:453-456): the 2 rows x 18 half-resolution pixels a
//     32-pixel row segment touches are staged per M-block (eight lanes per 128-byte slice) and the four taps come from LDS;
//   * outputs: through a 32 pixel x 32 channel tile, so a store instruction writes eight whole 128-byte texels / pixel slices.
// Arithmetic and its order are torgb_kernel's (MFMA order over K-groups and parts, epilogue_act(acc + bias) + the taps in skip_up2's
// order): bit-identical outputs.  One 512-thread workgroup per CU (the weight fragments are shared by its eight waves).
// ------------------------------------------------------------------------------------------------
constexpr int TC_WAVES = 8, TC_PF = 4;
constexpr int TC_TILE_FLOATS = 32 * 36;                               // 32 pixels x 32 channels, row stride 36: the x tile (512 floats) lives here too
constexpr int TC_SKIP_FLOATS = 2 * 18 * 32;
__host__ __device__ constexpr int tc_wave_floats(int cin) { return TC_TILE_FLOATS + TC_SKIP_FLOATS + cin; }
template <int TERMS, int MB>
__global__ __launch_bounds__(64 * TC_WAVES, 2) void torgb_coalesced_kernel(ConvK P) {
    constexpr int PARTS = TERMS == 3 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint4* ldsA = reinterpret_cast<uint4*>(lds);                     // [mb][g][part][lane]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = P.Cin >> 4;
    for (int i = tid; i < MB * G * PARTS * 64; i += 64 * TC_WAVES) {
        const int ln = i & 63, part = (i >> 6) % PARTS, r = (i >> 6) / PARTS, g = r % G, mb = r / G;
        ldsA[i] = P.packed[(((long long)mb * G + g) * 2 + part) * 64 + ln];
    }
    float* wl = reinterpret_cast<float*>(lds + (size_t)MB * G * PARTS * 1024) + (size_t)wave * tc_wave_floats(P.Cin);
    float* tile = wl;                                    // x tile during the K loop, output tile in the epilogue
    float* skp = wl + TC_TILE_FLOATS;                    // the current M-block's skip slice (the next one waits in registers)
    float* sty = skp + TC_SKIP_FLOATS;                   // this view's styles
    __syncthreads();
    const int HW = P.H * P.W, nblk = HW >> 5, hs = P.H >> 1, ws = P.W >> 1;
    const long long total = (long long)P.N * nblk;
    const int pq = lane >> 2, pc = lane & 3;             // coalesced input loads: pixel pq (+16), 16-byte piece pc of the K-group
    int sty_n = -1;
    for (long long blk = (long long)blockIdx.x * TC_WAVES + wave; blk < total; blk += (long long)gridDim.x * TC_WAVES) {
        const int n = (int)(blk / nblk), p0 = (int)(blk % nblk) * 32;
        const int y = p0 / P.W, x0 = p0 % P.W;
        if (n != sty_n) {                                // wave-uniform; same-wave LDS operations execute in order
            for (int i = lane * 4; i < P.Cin; i += 256) *reinterpret_cast<float4*>(sty + i) = *reinterpret_cast<const float4*>(P.styles + (long long)n * P.Cin + i);
            sty_n = n;
        }
        // ---- skip slices: half-resolution rows ya, ya + 1 and pixels hx0 .. hx0 + 17 (clamped: skip_up2 gives such taps weight 0)
        const int ya = (y & 1) ? (y >> 1) : (y >> 1) - 1, hx0 = (x0 >> 1) - 1;
        auto skip_load = [&](int mb, f32x4 (&v)[5]) {
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int chunk = min(it * 8 + (lane >> 3), 35), a = chunk / 18, px = chunk % 18;   // 36 slices of 128 bytes, eight lanes each (the last instruction's upper lanes repeat slice 35)
                const int yc = min(max(ya + a, 0), hs - 1), xc = min(max(hx0 + px, 0), ws - 1);
                v[it] = *reinterpret_cast<const f32x4*>(P.skip + (((long long)n * hs + yc) * ws + xc) * P.Cout + 32 * mb + 4 * (lane & 7));
            }
        };
        auto skip_store = [&](const f32x4 (&v)[5]) {
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int chunk = min(it * 8 + (lane >> 3), 35), px = chunk % 18;                   // duplicates write the same value to the same place
                *reinterpret_cast<f32x4*>(skp + chunk * 32 + 4 * ((lane & 7) ^ (px & 7))) = v[it];
            }
        };
        f32x4 skv[5];
#pragma unroll
        for (int it = 0; it < 5; ++it) skv[it] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (P.skip) skip_load(0, skv);
        // ---- K loop: TC_PF K-groups of coalesced loads in flight
        const float* xrow = P.x + ((long long)n * HW + p0) * P.Cin;
        f32x16 acc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.0f;
        f32x4 ring[TC_PF][2];
#pragma unroll
        for (int g = 0; g < TC_PF; ++g) {                 // (a layer with fewer K-groups than TC_PF re-reads its last one: harmless)
            const int gc = min(g, G - 1);
            ring[g][0] = *reinterpret_cast<const f32x4*>(xrow + (long long)pq * P.Cin + 16 * gc + 4 * pc);
            ring[g][1] = *reinterpret_cast<const f32x4*>(xrow + (long long)(pq + 16) * P.Cin + 16 * gc + 4 * pc);
        }
        // x tile: 16-byte unit (pixel q, piece c) at q * 4 + (c ^ ((q >> 2) & 3)): the operand reads (16 lanes = 16 pixels, one piece) hit 16 bank groups
        const int wr0 = (pq * 4 + (pc ^ ((pq >> 2) & 3))) * 4, wr1 = ((pq + 16) * 4 + (pc ^ (((pq + 16) >> 2) & 3))) * 4;
        const int rdq = (j >> 2) & 3;
        const int rd0 = (j * 4 + ((2 * h) ^ rdq)) * 4, rd1 = (j * 4 + ((2 * h + 1) ^ rdq)) * 4;
#pragma unroll 1
        for (int g0 = 0; g0 < G; g0 += TC_PF) {
#pragma unroll
            for (int u = 0; u < TC_PF; ++u) {
                const int g = g0 + u;
                if (g < G) {                              // wave-uniform
                *reinterpret_cast<f32x4*>(tile + wr0) = ring[u][0];
                *reinterpret_cast<f32x4*>(tile + wr1) = ring[u][1];
                {
                    const int gn = min(g + TC_PF, G - 1);         // past the end: the last K-group once more, never used
                    ring[u][0] = *reinterpret_cast<const f32x4*>(xrow + (long long)pq * P.Cin + 16 * gn + 4 * pc);
                    ring[u][1] = *reinterpret_cast<const f32x4*>(xrow + (long long)(pq + 16) * P.Cin + 16 * gn + 4 * pc);
                }
                const float4 x0v = *reinterpret_cast<const float4*>(tile + rd0), x1v = *reinterpret_cast<const float4*>(tile + rd1);
                const float4 s0 = *reinterpret_cast<const float4*>(sty + 16 * g + 8 * h), s1 = *reinterpret_cast<const float4*>(sty + 16 * g + 8 * h + 4);
                Frag8 bh, bl;
                split2<TERMS>(x0v.x * s0.x, x0v.y * s0.y, bh.u[0], bl.u[0]);
                split2<TERMS>(x0v.z * s0.z, x0v.w * s0.w, bh.u[1], bl.u[1]);
                split2<TERMS>(x1v.x * s1.x, x1v.y * s1.y, bh.u[2], bl.u[2]);
                split2<TERMS>(x1v.z * s1.z, x1v.w * s1.w, bh.u[3], bl.u[3]);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    Frag8 ah, al;
                    ah.q = ldsA[((mb * G + g) * PARTS + 0) * 64 + lane];
                    acc[mb] = mfma16<TERMS>(ah.v, bh.v, acc[mb], 0, 0, 0);
                    if (TERMS == 3) {
                        al.q = ldsA[((mb * G + g) * PARTS + 1) * 64 + lane];
                        acc[mb] = mfma16<TERMS>(ah.v, bl.v, acc[mb], 0, 0, 0);
                        acc[mb] = mfma16<TERMS>(al.v, bh.v, acc[mb], 0, 0, 0);
                    }
                }
                }
            }
        }
        // ---- epilogue per M-block: act(acc + bias) + skip taps -> output tile -> whole-line stores
        const int x = x0 + j;
        const int xa = (x & 1) ? (x >> 1) : (x >> 1) - 1;
        const float wya = (y & 1) ? 0.75f : 0.25f, wyb = 1.0f - wya, wxa = (x & 1) ? 0.75f : 0.25f, wxb = 1.0f - wxa;
        const int ysv[2] = {ya, ya + 1}, xsv[2] = {xa, xa + 1};
        const float wyv[2] = {wya, wyb}, wxv[2] = {wxa, wxb};
        float wgt[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float wye = (unsigned)ysv[a] < (unsigned)hs ? wyv[a] : 0.0f, wxe = (unsigned)xsv[b] < (unsigned)ws ? wxv[b] : 0.0f;
                wgt[a][b] = wye * wxe;
            }
        if (P.skip) skip_store(skv);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            if (P.skip && mb + 1 < MB) skip_load(mb + 1, skv);          // in flight under this M-block's epilogue
            const float* sk = skp;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int o0 = 32 * mb + 8 * qq + 4 * h;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = epilogue_act(acc[mb][4 * qq + i] + P.bias[o0 + i], 0, P.act_gain, P.clamp);
                if (P.skip) {
                    float4 r = make_float4(0, 0, 0, 0);
                    float4 t[2][2];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const int px = xsv[b] - hx0;                 // 0 .. 17
                            t[a][b] = *reinterpret_cast<const float4*>(sk + (a * 18 + px) * 32 + 4 * ((2 * qq + h) ^ (px & 7)));
                        }
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            r.x = fmaf(wgt[a][b], t[a][b].x, r.x); r.y = fmaf(wgt[a][b], t[a][b].y, r.y);
                            r.z = fmaf(wgt[a][b], t[a][b].z, r.z); r.w = fmaf(wgt[a][b], t[a][b].w, r.w);
                        }
                    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                }
                *reinterpret_cast<float4*>(tile + j * 36 + 8 * qq + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
            }
            if (P.skip && mb + 1 < MB) skip_store(skv);            // after this M-block's taps were read (same-wave LDS operations execute in order)
            float* orow = P.out_planes ? P.out + ((((long long)n * 3 + mb) * P.H + y) * P.W + x0) * 32
                                       : P.out + ((long long)n * HW + p0) * P.Cout + 32 * mb;
            const long long ostride = P.out_planes ? 32 : P.Cout;
#pragma unroll
            for (int it = 0; it < 4; ++it) {             // eight lanes per pixel: 128 contiguous bytes each
                const int pp = 8 * it + (lane >> 3), c = lane & 7;
                const float4 v4 = *reinterpret_cast<const float4*>(tile + pp * 36 + 4 * c);
                *reinterpret_cast<float4*>(orow + (long long)pp * ostride + 4 * c) = v4;
            }
        }
    }
}

template <int TERMS, int MB>
static void launch_torgb_t(const ConvK& P, hipStream_t st);
template <int TERMS, int MB>
static void launch_torgb(const ConvK& P, hipStream_t st) {
    if constexpr (TERMS == 1) { if (P.f16) { launch_torgb_t<2, MB>(P, st); return; } }
    launch_torgb_t<TERMS, MB>(P, st);
}
template <int TERMS, int MB>
static void launch_torgb_t(const ConvK& P, hipStream_t st) {
    const int bytes = MB * (P.Cin / 16) * (TERMS == 3 ? 2 : 1) * 1024;
    static LdsOptInMax topt;                          // per device; a refused opt-in surfaces as the launch error the caller checks (NFE_ELAUNCH)
    if (topt.apply(torgb_kernel<TERMS, MB>, bytes) != hipSuccess) (void)hipGetLastError();
    // coalesced form (see torgb_coalesced_kernel): whole 32-pixel row segments, whole 32-channel output slices, LDS for eight waves
    static const bool coalesced_on = [] { const char* e = getenv("NFE_TORGB_COALESCED"); return !e || e[0] != '0'; }();
    const int cbytes = bytes + TC_WAVES * tc_wave_floats(P.Cin) * 4;
    if (coalesced_on && P.W % 32 == 0 && P.Cout == 32 * MB && P.Cin % 16 == 0 && (P.Cin & 3) == 0 && cbytes <= 160 * 1024 && !(P.skip && ((P.H | P.W) & 1))) {
        static LdsOptInMax copt;                      // per device (a function-static int was per process and unsynchronised)
        if (copt.apply(torgb_coalesced_kernel<TERMS, MB>, cbytes) != hipSuccess) {      // no opt-in on this device: the per-lane form needs none beyond `bytes`
            (void)hipGetLastError();
            const long long blocks = ((long long)P.N * ((P.H * P.W + 31) / 32) + 3) / 4;
            hipLaunchKernelGGL((torgb_kernel<TERMS, MB>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), bytes, st, P);
            return;
        }
        const long long cblocks = ((long long)P.N * (P.H * P.W / 32) + TC_WAVES - 1) / TC_WAVES;
        const long long cap = (long long)num_cus() * (cbytes <= 80 * 1024 ? 2 : 1);
        hipLaunchKernelGGL((torgb_coalesced_kernel<TERMS, MB>), dim3((unsigned)(cblocks < cap ? cblocks : cap)), dim3(64 * TC_WAVES), cbytes, st, P);
        return;
    }
    const long long blocks = ((long long)P.N * ((P.H * P.W + 31) / 32) + 3) / 4;
    hipLaunchKernelGGL((torgb_kernel<TERMS, MB>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), bytes, st, P);
}

// Split-K reduction (4^2..16^2 layers: one tile per sample, so the K loop is the only parallelism left).  Partial
// sums are added in slice order, so results do not depend on scheduling.  up: -> transposed-conv scratch (the FIR
// epilogue follows); else: demod + noise + bias + lrelu + clamp -> out.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvK P, long long n_vec, int up) {
    const int C4 = P.Cout >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long long)gridDim.x * blockDim.x) {
        float4 s = reinterpret_cast<const float4*>(P.partial)[i];
        for (int k = 1; k < P.ksplit; ++k) {
            const float4 t = reinterpret_cast<const float4*>(P.partial)[i + k * n_vec];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        if (up) { reinterpret_cast<float4*>(P.scratch)[i] = s; continue; }
        const int c4 = (int)(i % C4); const long long pix = i / C4;
        const int n = (int)(pix / ((long long)P.H * P.W)); const long long yx = pix % ((long long)P.H * P.W);
        const float nz = P.noise ? P.noise[n * P.noise_n_stride + yx] * P.noise_strength : 0.0f;
        const float4 d = P.dcoef ? *reinterpret_cast<const float4*>(P.dcoef + (long long)n * P.Cout + 4 * c4) : make_float4(1, 1, 1, 1);
        const float4 b = *reinterpret_cast<const float4*>(P.bias + 4 * c4);
        float4 o;
        o.x = epilogue_act(s.x * d.x + nz + b.x, P.lrelu, P.act_gain, P.clamp);
        o.y = epilogue_act(s.y * d.y + nz + b.y, P.lrelu, P.act_gain, P.clamp);
        o.z = epilogue_act(s.z * d.z + nz + b.z, P.lrelu, P.act_gain, P.clamp);
        o.w = epilogue_act(s.w * d.w + nz + b.w, P.lrelu, P.act_gain, P.clamp);
        if (P.skip) {                              // ToRGB: + upsample2d(previous image)
            const int y = (int)(yx / P.W), x = (int)(yx % P.W);
            const float4 sk = skip_up2(P.skip, n, P.H >> 1, P.W >> 1, P.Cout, y, x, 4 * c4);
            o.x += sk.x; o.y += sk.y; o.z += sk.z; o.w += sk.w;
        }
        reinterpret_cast<float4*>(P.out)[i] = o;
    }
}

// FIR + epilogue of the up-conv: out[Y][X] = act(dcoef * sum_ab F[a]F[b] T[Y+a-1][X+b-1] + noise + bias),
// F = [1,3,3,1]/4 per axis (setup_filter/64 * gain 4; conv2d_resample.py:127, upfirdn2d.py:169-207)
#define NFE_UPFIR_ROWS 4
constexpr int UPFIR_ROWS = NFE_UPFIR_ROWS;     // 2-row output blocks per thread: consecutive blocks share 3 of their 5 filtered rows

#define UPFIR_WAVES 2      // no register cap below what the kernel wants: at 3 waves per SIMD (168 registers) it spilled inside the row loop, and scratch traffic shares the vmcnt queue
// HAS_OUT / PARTS (0: no consumer image, 1: bf16 hi, 2: hi + lo) are compile-time on purpose: with the stores under run-time
// conditions the compiler cannot count them and waits with `s_waitcnt vmcnt(0)` for the prefetched rows - i.e. for the stores it has
// just issued as well (loads and stores retire in order on gfx9) - and the read and write streams serialise: 391 us for the SR
// layer's 1.08 GB, where reads alone take 140 and writes alone 108 (round 3, tools/r03_upfir_ablate.sh).
template <bool HAS_OUT, int PARTS>
__global__ __launch_bounds__(256, UPFIR_WAVES) void upfir_kernel(ConvK P) {
    // One thread = 4 channels x 2 output columns x 2*UPFIR_ROWS output rows, walking down the image with a sliding
    // window of row-filtered values: 10 loads of T per 4 outputs (separable 4-tap filter per axis) instead of 64,
    // and each T row is fetched by one workgroup instead of by the two that own the rows above and below it.
    // Lanes run over channels, so every load is a contiguous 512-byte row piece.
    const int OH = 2 * P.H, OW = 2 * P.W, TH = OH + 1, TW = OW + 1, C4 = P.Cout >> 2;
    const int groups = (P.H + UPFIR_ROWS - 1) / UPFIR_ROWS;
    const long long total = (long long)P.N * groups * P.W * C4;
    const float F[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4); long long r = i / C4;
        const int bx = (int)(r % P.W); r /= P.W;
        const int bg = (int)(r % groups); const int n = (int)(r / groups);
        const int X0 = 2 * bx;
        const float4 d = P.dcoef ? *reinterpret_cast<const float4*>(P.dcoef + (long long)n * P.Cout + 4 * c4) : make_float4(1, 1, 1, 1);
        const float4 b = *reinterpret_cast<const float4*>(P.bias + 4 * c4);
        float4 s2 = make_float4(0, 0, 0, 0);
        if (PARTS) s2 = *reinterpret_cast<const float4*>(P.next_styles + (long long)n * P.Cout + 4 * c4);
        // row-filtered T row ty: rf[dx] = sum_b F[b] T[ty][X0+dx+b-1].  The five loads are unconditional (clamped address, value
        // zeroed by a select afterwards): with the bounds tests as branches every load sat in its own exec-masked block behind an
        // `s_waitcnt vmcnt(0)` and a thread had one load in flight at a time (round 3: 126 -> see DESIGN 5).
        const float* __restrict__ tbase = P.scratch + (long long)n * TH * TW * P.Cout + 4 * c4;
        int toff[5]; float Fm[2][4];                       // filter taps with the column bounds folded in (a tap outside the scratch = 0)
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) toff[jj] = min(max(X0 - 1 + jj, 0), TW - 1) * P.Cout;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const int tx = X0 - 1 + dx + bb;
                Fm[dx][bb] = (unsigned)tx < (unsigned)TW ? F[bb] : 0.0f;      // one compare: no lane masks to combine (tools/lint_lane_masks.py)
            }
        auto load_row = [&](int ty, float4 (&t)[5]) {
            const float* __restrict__ trow = tbase + (long long)min(max(ty, 0), TH - 1) * TW * P.Cout;
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {
                t[jj] = *reinterpret_cast<const float4*>(trow + toff[jj]);
            }
        };
        auto reduce_row = [&](int ty, float4 (&t)[5], float4 (&rf)[2]) {
            const float rs = (unsigned)ty < (unsigned)TH ? 1.0f : 0.0f;      // the loads are clamped, so every t is a finite scratch value
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    a.x = fmaf(Fm[dx][bb], t[dx + bb].x, a.x); a.y = fmaf(Fm[dx][bb], t[dx + bb].y, a.y);
                    a.z = fmaf(Fm[dx][bb], t[dx + bb].z, a.z); a.w = fmaf(Fm[dx][bb], t[dx + bb].w, a.w);
                }
                rf[dx] = make_float4(a.x * rs, a.y * rs, a.z * rs, a.w * rs);
            }
        };
        const int by0 = bg * UPFIR_ROWS;
        float4 win[5][2];                                  // filtered rows 2by-1 .. 2by+3
        float4 ta[5], tb[5];                               // raw rows in flight: the NEXT block's two rows are requested before this
        {                                                  // block's outputs are computed and stored (round 3: the loads used to start
            float4 tc[5];                                  // only after the previous block's stores, half of the time nothing was in flight)
            load_row(2 * by0 - 1, ta); load_row(2 * by0, tb); load_row(2 * by0 + 1, tc);
            reduce_row(2 * by0 - 1, ta, win[0]); reduce_row(2 * by0, tb, win[1]); reduce_row(2 * by0 + 1, tc, win[2]);
        }
        load_row(2 * by0 + 2, ta); load_row(2 * by0 + 3, tb);
        // the block's four noise values travel with its rows: read where they are used they were four exposed global round trips
        // per block (a dependent load behind `s_waitcnt vmcnt(0)` in front of every output pixel)
        const float* nzp = P.noise ? P.noise + n * P.noise_n_stride + X0 : nullptr;
        float2 nza = make_float2(0, 0), nzb = make_float2(0, 0);
        auto load_noise = [&](int Y0_) {
            if (!nzp) return;
            nza = *reinterpret_cast<const float2*>(nzp + (long long)Y0_ * OW);          // X0 is even, OW is even: 8-byte aligned
            nzb = *reinterpret_cast<const float2*>(nzp + (long long)(Y0_ + 1) * OW);
        };
        load_noise(2 * by0);
#pragma unroll 1
        for (int rr = 0; rr < UPFIR_ROWS; ++rr) {
            const int by = by0 + rr;
            if (by >= P.H) break;
            const int Y0 = 2 * by;
            reduce_row(Y0 + 2, ta, win[3]); reduce_row(Y0 + 3, tb, win[4]);
            const float nzv4[2][2] = {{nza.x * P.noise_strength, nza.y * P.noise_strength}, {nzb.x * P.noise_strength, nzb.y * P.noise_strength}};
            if (rr + 1 < UPFIR_ROWS && by + 1 < P.H) { load_row(Y0 + 4, ta); load_row(Y0 + 5, tb); load_noise(Y0 + 2); }
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float4 sm = make_float4(0, 0, 0, 0);
#pragma unroll
                    for (int aa = 0; aa < 4; ++aa) {
                        sm.x = fmaf(F[aa], win[dy + aa][dx].x, sm.x); sm.y = fmaf(F[aa], win[dy + aa][dx].y, sm.y);
                        sm.z = fmaf(F[aa], win[dy + aa][dx].z, sm.z); sm.w = fmaf(F[aa], win[dy + aa][dx].w, sm.w);
                    }
                    const int Y = Y0 + dy, X = X0 + dx;
                    const float nz = nzv4[dy][dx];
                    float4 o;
                    o.x = epilogue_act(sm.x * d.x + nz + b.x, P.lrelu, P.act_gain, P.clamp);
                    o.y = epilogue_act(sm.y * d.y + nz + b.y, P.lrelu, P.act_gain, P.clamp);
                    o.z = epilogue_act(sm.z * d.z + nz + b.z, P.lrelu, P.act_gain, P.clamp);
                    o.w = epilogue_act(sm.w * d.w + nz + b.w, P.lrelu, P.act_gain, P.clamp);
                    const long long oi = (((long long)n * OH + Y) * OW + X) * C4 + c4;
                    if (HAS_OUT) reinterpret_cast<float4*>(P.out)[oi] = o;
                    if (PARTS) {                            // what modsplit_kernel would make of `o` for the next layer (group-major)
                        unsigned h0, l0, h1, l1;
                        const long long si = split_index(n, C4 >> 2, OH, OW, Y, X, c4);
                        if (PARTS == 2) { split2<3>(o.x * s2.x, o.y * s2.y, h0, l0); split2<3>(o.z * s2.z, o.w * s2.w, h1, l1); P.split_lo[si] = make_uint2(l0, l1); }
                        else { split2_single(P.f16, o.x * s2.x, o.y * s2.y, h0); split2_single(P.f16, o.z * s2.z, o.w * s2.w, h1); }
                        P.split_hi[si] = make_uint2(h0, h1);
                    }
                }
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) { win[0][dx] = win[2][dx]; win[1][dx] = win[3][dx]; win[2][dx] = win[4][dx]; }
        }
    }
}

// F.interpolate(bilinear, align_corners=False, antialias) — ATen's separable triangle filter
// (UpSampleKernel.cpp compute_indices_weights_aa) evaluated as one 2-D gather per output pixel.
struct AxisW { int lo, n; float scale, support, inv; };
__device__ __forceinline__ AxisW axis_setup(int i, int in, int out, int aa) {
    AxisW a;
    a.scale = (float)in / (float)out;
    if (aa) {
        a.support = a.scale >= 1.0f ? a.scale : 1.0f;
        a.inv = a.scale >= 1.0f ? 1.0f / a.scale : 1.0f;
        const float center = a.scale * ((float)i + 0.5f);
        a.lo = max((int)(center - a.support + 0.5f), 0);
        a.n = min((int)(center + a.support + 0.5f), in) - a.lo;
    } else {
        float src = a.scale * ((float)i + 0.5f) - 0.5f;
        if (src < 0.0f) src = 0.0f;
        a.lo = min((int)src, in - 1);
        a.n = (a.lo + 1 < in) ? 2 : 1;
        a.support = a.n == 2 ? src - (float)a.lo : 0.0f;   // lambda1 (weight of lo+1; at the far edge lo+1 clamps onto lo)
        a.inv = 0.0f;
    }
    return a;
}
__device__ __forceinline__ float axis_weight(const AxisW& a, int k, int i, int aa) {
    if (!aa) return k == 0 ? 1.0f - a.support : a.support;
    const float center = a.scale * ((float)i + 0.5f);
    const float xx = ((float)(k + a.lo) - center + 0.5f) * a.inv;
    return fmaxf(0.0f, 1.0f - fabsf(xx));
}
__global__ void resize_kernel(const float* __restrict__ in, int N, int H, int W, int C, int OH, int OW, int aa, float* __restrict__ out) {
    const long long total = (long long)N * OH * OW * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); long long r = i / C;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH); const int n = (int)(r / OH);
        const AxisW ay = axis_setup(oy, H, OH, aa), ax = axis_setup(ox, W, OW, aa);
        float sy = 0.0f, sx = 0.0f;
        for (int k = 0; k < ay.n; ++k) sy += axis_weight(ay, k, oy, aa);
        for (int k = 0; k < ax.n; ++k) sx += axis_weight(ax, k, ox, aa);
        // ATen resizes horizontally first, then vertically (separable passes): keep that order of products
        float acc = 0.0f;
        for (int ky = 0; ky < ay.n; ++ky) {
            float rowv = 0.0f;
            for (int kx = 0; kx < ax.n; ++kx)
                rowv = fmaf(axis_weight(ax, kx, ox, aa) / (aa ? sx : 1.0f), in[(((long long)n * H + ay.lo + ky) * W + ax.lo + kx) * C + c], rowv);
            acc = fmaf(axis_weight(ay, ky, oy, aa) / (aa ? sy : 1.0f), rowv, acc);
        }
        out[i] = acc;
    }
}

// Same arithmetic, same order of products, for C % 4 == 0 and at most RS_MAXT taps per axis (every down-scale up to 5.5x): one
// thread = one output pixel x 4 channels (16-byte loads, 8 lanes per 32-channel pixel = one 128-byte line); the normalised tap
// weights of both axes are computed once per thread instead of once per tap pair (the generic kernel spent its time on 64
// divisions per output: 239 us for the 512 -> 128 resize of 8 x 32-channel feature images).
constexpr int RS_MAXT = 12;
__global__ __launch_bounds__(256) void resize4_kernel(const float4* __restrict__ in, int N, int H, int W, int C4, int OH, int OW, int aa,
                                                     float4* __restrict__ out) {
    const long long total = (long long)N * OH * OW * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4); long long r = i / C4;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH); const int n = (int)(r / OH);
        const AxisW ay = axis_setup(oy, H, OH, aa), ax = axis_setup(ox, W, OW, aa);
        float sy = 0.0f, sx = 0.0f;
        for (int k = 0; k < ay.n; ++k) sy += axis_weight(ay, k, oy, aa);
        for (int k = 0; k < ax.n; ++k) sx += axis_weight(ax, k, ox, aa);
        float wx[RS_MAXT];
#pragma unroll
        for (int k = 0; k < RS_MAXT; ++k) wx[k] = k < ax.n ? axis_weight(ax, k, ox, aa) / (aa ? sx : 1.0f) : 0.0f;
        float4 acc = make_float4(0, 0, 0, 0);
        for (int ky = 0; ky < ay.n; ++ky) {
            const float4* row = in + (((long long)n * H + ay.lo + ky) * W + ax.lo) * C4 + c;
            float4 rv = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int kx = 0; kx < RS_MAXT; ++kx)
                if (kx < ax.n) {
                    const float4 v = row[(long long)kx * C4];
                    rv.x = fmaf(wx[kx], v.x, rv.x); rv.y = fmaf(wx[kx], v.y, rv.y); rv.z = fmaf(wx[kx], v.z, rv.z); rv.w = fmaf(wx[kx], v.w, rv.w);
                }
            const float wy = axis_weight(ay, ky, oy, aa) / (aa ? sy : 1.0f);
            acc.x = fmaf(wy, rv.x, acc.x); acc.y = fmaf(wy, rv.y, acc.y); acc.z = fmaf(wy, rv.z, acc.z); acc.w = fmaf(wy, rv.w, acc.w);
        }
        out[i] = acc;
    }
}

// [N,HW,C] -> [N,C,HW] for small C (the 3-channel images, the 15-channel segmentation image): one thread per pixel, C coalesced
// plane writes.  The 32 x 32 LDS-tile transpose above wastes 29 of 32 tile columns at C = 3 (75 us for 8 x 512^2 x 3).
template <int C>
__global__ __launch_bounds__(256) void nhwc_to_nchw_small_kernel(const float* __restrict__ in, int hw, float* __restrict__ out) {
    const int n = blockIdx.y;
    const float* src = in + (long long)n * hw * C;
    float* dst = out + (long long)n * hw * C;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) {
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = src[(long long)p * C + c];
#pragma unroll
        for (int c = 0; c < C; ++c) dst[(long long)c * hw + p] = v[c];
    }
}

// upfirdn2d with the path's one filter ([1,3,3,1] x [1,3,3,1] / 64, symmetric: flip_filter is moot), NHWC, C % 4 == 0 or any C:
// zero-insert by `up`, pad (pad0 before, pad1 after, both axes), 4 x 4 FIR, keep every `down`-th sample, times gain
// (torch_utils/ops/upfirdn2d.py:169-205 _upfirdn2d_ref).  The forward path never calls it (its FIRs are fused into the conv and
// ToRGB epilogues); it exists for the transposes the SR-head input gradient needs (sr_grad.py) and as the reference's op itself:
//   upsample2d(x)            = upfirdn2d(x, up=2, pad=(2,1), gain=4)         upfirdn2d.py:341-350
//   its transpose            = upfirdn2d(g, down=2, pad=(1,2), gain=4)
//   up-conv FIR (pad 1,1)^T  = upfirdn2d(g, pad=(2,2), gain=4)                conv2d_resample.py:114-128
// POLY (nfe_upfirdn2d_polyphase): the result leaves as the four polyphase images stacked along the channels,
// out[n][Y / 2][X / 2][((Y & 1) * 2 + (X & 1)) * C + c] over the even-sized grid (OH + 1 & ~1) x (OW + 1 & ~1), zeros beyond OH / OW - the
// operand layout of the up-sampling layer's backward-data convolution (sr_grad.py: four torch passes before).
template <bool POLY>
__global__ __launch_bounds__(256) void upfirdn_kernel(const float* __restrict__ in, int N, int H, int W, int C, int up, int down, int pad0,
                                                      float gain, int OH, int OW, float* __restrict__ out) {
    const float F[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    const int GH = POLY ? (OH + 1) & ~1 : OH, GW = POLY ? (OW + 1) & ~1 : OW;
    const long long total = (long long)N * GH * GW * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); long long r = i / C;
        const int X = (int)(r % GW); r /= GW;
        const int Y = (int)(r % GH); const int n = (int)(r / GH);
        float acc = 0.0f;
        if (POLY && (Y >= OH || X >= OW)) {
            out[(((long long)n * (GH / 2) + Y / 2) * (GW / 2) + X / 2) * (4 * C) + ((Y & 1) * 2 + (X & 1)) * C + c] = 0.0f;
            continue;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int u = Y * down + a - pad0;                 // row of the zero-inserted image
            if (u < 0 || u % up != 0 || u / up >= H) continue;
            float rowv = 0.0f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int v = X * down + b - pad0;
                if (v < 0 || v % up != 0 || v / up >= W) continue;
                rowv = fmaf(F[b], in[(((long long)n * H + u / up) * W + v / up) * C + c], rowv);
            }
            acc = fmaf(F[a], rowv, acc);
        }
        if (POLY) out[(((long long)n * (GH / 2) + Y / 2) * (GW / 2) + X / 2) * (4 * C) + ((Y & 1) * 2 + (X & 1)) * C + c] = acc * gain;
        else out[i] = acc * gain;
    }
}

// nfe_upfirdn2d_polyphase for C % 4 == 0 (round 5; the generic kernel above spends most of its 2 ms per 4 x 512^2 x 64 call on 64-bit
// index divisions and sixteen scalar loads per output): one thread = one polyphase quad (outputs (2 y, 2 x) .. (2 y + 1, 2 x + 1)) of four
// channels.  It loads the 5 x 5 input window once (25 x 16 bytes for 16 outputs), forms the ten horizontal sums its outputs share and then
// the four vertical ones - per output the same fmas in the same order as the generic kernel (taps outside the image contribute
// fma(F, 0, acc) = acc there, a skipped step here).  Grid: (quads of a row x channel groups, quad rows, views).
__global__ __launch_bounds__(256) void upfir_poly4_kernel(const float* __restrict__ in, int H, int W, int C, int pad0, float gain, int OH, int OW,
                                                          float* __restrict__ out) {
    const float F[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    const int GH2 = (OH + 1) >> 1, GW2 = (OW + 1) >> 1, C4 = C >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int qx = t / C4, c = (t - qx * C4) * 4;
    if (qx >= GW2) return;
    const int qy = blockIdx.y, n = blockIdx.z;
    const int u0 = 2 * qy - pad0, v0 = 2 * qx - pad0;          // window origin in the input
    const float* src = in + (long long)n * H * W * C + c;
    float4 rowv[5][2];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int u = u0 + r;
        const bool rin = (unsigned)u < (unsigned)H;
        float4 win[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int v = v0 + q;
            const bool ok = rin && (unsigned)v < (unsigned)W;
            const float4 x = *reinterpret_cast<const float4*>(src + ((long long)min(max(u, 0), H - 1) * W + min(max(v, 0), W - 1)) * C);
            win[q] = ok ? x : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                a.x = fmaf(F[b], win[dx + b].x, a.x); a.y = fmaf(F[b], win[dx + b].y, a.y);
                a.z = fmaf(F[b], win[dx + b].z, a.z); a.w = fmaf(F[b], win[dx + b].w, a.w);
            }
            rowv[r][dx] = a;
        }
    }
    float* dst = out + (((long long)n * GH2 + qy) * GW2 + qx) * (4 * C) + c;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a.x = fmaf(F[k], rowv[dy + k][dx].x, a.x); a.y = fmaf(F[k], rowv[dy + k][dx].y, a.y);
                a.z = fmaf(F[k], rowv[dy + k][dx].z, a.z); a.w = fmaf(F[k], rowv[dy + k][dx].w, a.w);
            }
            const bool live = 2 * qy + dy < OH && 2 * qx + dx < OW;          // beyond the odd-sized result: zeros
            *reinterpret_cast<float4*>(dst + (dy * 2 + dx) * C) = live ? make_float4(a.x * gain, a.y * gain, a.z * gain, a.w * gain) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
}

static unsigned grid1d(long long total, int per_block, long long cap = 1 << 16) {
    long long b = (total + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

}  // namespace nfe

using namespace nfe;

extern "C" int nfe_nchw_to_nhwc(const float* in, int n, int c, int h, int w, float* out, nfe_stream_t stream) {
    NFE_REQUIRE(in && out && n > 0 && c > 0 && h > 0 && w > 0, "nfe_nchw_to_nhwc: bad arguments");
    const int hw = h * w;
    hipLaunchKernelGGL((transpose_kernel<true>), dim3((hw + 31) / 32, (c + 31) / 32, n), dim3(256), 0, (hipStream_t)stream, in, c, hw, out);
    NFE_CHECK_LAUNCH("transpose_kernel");
    return NFE_OK;
}
extern "C" int nfe_nhwc_to_nchw(const float* in, int n, int c, int h, int w, float* out, nfe_stream_t stream) {
    NFE_REQUIRE(in && out && n > 0 && c > 0 && h > 0 && w > 0, "nfe_nhwc_to_nchw: bad arguments");
    const int hw = h * w;
    const dim3 gs((unsigned)std::min<long long>((hw + 255) / 256, 4096), n);
    if (c == 1) { if (hipMemcpyAsync(out, in, (size_t)n * hw * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return ::nfe::fail(NFE_ELAUNCH, "nfe_nhwc_to_nchw: copy failed"); return NFE_OK; }
    else if (c == 3) hipLaunchKernelGGL((nhwc_to_nchw_small_kernel<3>), gs, dim3(256), 0, (hipStream_t)stream, in, hw, out);
    else if (c == 15) hipLaunchKernelGGL((nhwc_to_nchw_small_kernel<15>), gs, dim3(256), 0, (hipStream_t)stream, in, hw, out);
    else hipLaunchKernelGGL((transpose_kernel<false>), dim3((hw + 31) / 32, (c + 31) / 32, n), dim3(256), 0, (hipStream_t)stream, in, c, hw, out);
    NFE_CHECK_LAUNCH("nhwc_to_nchw kernels");
    return NFE_OK;
}
extern "C" int nfe_nhwc_to_planes(const float* in, int n, int h, int w, float* out, nfe_stream_t stream) {
    NFE_REQUIRE(in && out && n > 0 && h > 0 && w > 0, "nfe_nhwc_to_planes: bad arguments");
    const long long npix = (long long)n * h * w;
    hipLaunchKernelGGL(nhwc_to_planes_kernel, dim3(grid1d(npix * 24, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(in), npix, h * w, reinterpret_cast<float4*>(out));
    NFE_CHECK_LAUNCH("nhwc_to_planes_kernel");
    return NFE_OK;
}
extern "C" int nfe_plane_stats_nhwc(const float* x, int n, int hw, int c, float* mean, float* std, void* scratch, nfe_stream_t stream) {
    NFE_REQUIRE(x && mean && std && scratch && n > 0 && hw > 1 && c > 0, "nfe_plane_stats_nhwc: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    double* sums = (double*)scratch;
    if (hipMemsetAsync(sums, 0, (size_t)n * c * 2 * sizeof(double), st) != hipSuccess) return fail(NFE_ELAUNCH, "memset failed");
    int split = hw / 128; if (split < 1) split = 1; if (split > 512) split = 512;       // >= 2 blocks per CU even for one view
    hipLaunchKernelGGL(stats_nhwc_kernel, dim3((c + 63) / 64, n, split), dim3(256), 0, st, x, hw, c, sums);
    hipLaunchKernelGGL(stats_finish_kernel, dim3((n * c + 255) / 256), dim3(256), 0, st, sums, n * c, hw, mean, std);
    NFE_CHECK_LAUNCH("stats_nhwc kernels");
    return NFE_OK;
}

extern "C" int nfe_fully_connected(const float* x, const float* w, const float* b, int n, int in_features, int out_features,
                                   float weight_gain, float bias_gain, int lrelu, float* y, int y_stride, nfe_stream_t stream) {
    NFE_REQUIRE(x && w && y && n > 0 && in_features > 0 && out_features > 0 && y_stride >= out_features, "nfe_fully_connected: bad arguments");
    const long long waves = (long long)n * out_features;
    hipLaunchKernelGGL(fc_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, w, b, n, in_features, out_features,
                       weight_gain, bias_gain, lrelu, y, y_stride);
    NFE_CHECK_LAUNCH("fc_kernel");
    return NFE_OK;
}
extern "C" int nfe_fully_connected_grouped(const nfe_fc_group* groups, int n_groups, int n, nfe_stream_t stream) {
    NFE_REQUIRE(groups && n_groups > 0 && n_groups <= NFE_MAX_GROUPS && n > 0, "nfe_fully_connected_grouped: bad arguments (at most %d groups)", NFE_MAX_GROUPS);
    FcGroups G{};
    int max_out = 0;
    for (int i = 0; i < n_groups; ++i) {
        const nfe_fc_group& g = groups[i];
        NFE_REQUIRE(g.x && g.w && g.y && g.in_features > 0 && g.out_features > 0 && g.x_stride >= g.in_features, "nfe_fully_connected_grouped: bad group %d", i);
        G.g[i] = g;
        if (g.out_features > max_out) max_out = g.out_features;
    }
    const long long waves = (long long)n * max_out;
    hipLaunchKernelGGL(fc_grouped_kernel, dim3((unsigned)((waves + 3) / 4), n_groups), dim3(256), 0, (hipStream_t)stream, G, n);
    NFE_CHECK_LAUNCH("fc_grouped_kernel");
    return NFE_OK;
}

extern "C" int nfe_conv_demod_grouped(const nfe_demod_group* groups, int n_groups, int n, nfe_stream_t stream) {
    NFE_REQUIRE(groups && n_groups > 0 && n_groups <= NFE_MAX_GROUPS && n > 0, "nfe_conv_demod_grouped: bad arguments (at most %d groups)", NFE_MAX_GROUPS);
    DemodGroups G{};
    int max_out = 0;
    for (int i = 0; i < n_groups; ++i) {
        const nfe_demod_group& g = groups[i];
        NFE_REQUIRE(g.styles && g.wsq && g.dcoef && g.cin > 0 && g.cout > 0, "nfe_conv_demod_grouped: bad group %d", i);
        G.g[i] = g;
        if (g.cout > max_out) max_out = g.cout;
    }
    const long long waves = (long long)n * max_out;
    hipLaunchKernelGGL(demod_grouped_kernel, dim3((unsigned)((waves + 3) / 4), n_groups), dim3(256), 0, (hipStream_t)stream, G, n);
    NFE_CHECK_LAUNCH("demod_grouped_kernel");
    return NFE_OK;
}

extern "C" int nfe_normalize_2nd_moment(const float* x, int n, int features, float* y, int y_stride, nfe_stream_t stream) {
    NFE_REQUIRE(x && y && n > 0 && features > 0 && y_stride >= features, "nfe_normalize_2nd_moment: bad arguments");
    hipLaunchKernelGGL(norm2_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, x, features, y, y_stride);
    NFE_CHECK_LAUNCH("norm2_kernel");
    return NFE_OK;
}
extern "C" int nfe_broadcast_truncate(const float* w, const float* w_avg, int n, int w_dim, int num_ws, float psi, int cutoff,
                                      float* ws, nfe_stream_t stream) {
    NFE_REQUIRE(w && ws && n > 0 && w_dim > 0 && num_ws > 0, "nfe_broadcast_truncate: bad arguments");
    NFE_REQUIRE(psi == 1.0f || w_avg, "nfe_broadcast_truncate: truncation_psi != 1 needs w_avg");
    hipLaunchKernelGGL(broadcast_truncate_kernel, dim3(grid1d((long long)n * num_ws * w_dim, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       w, w_avg, n, w_dim, num_ws, psi, cutoff, ws);
    NFE_CHECK_LAUNCH("broadcast_truncate_kernel");
    return NFE_OK;
}

extern "C" uint64_t nfe_conv_packed_words(int cout, int cin, int k) {
    if (cout <= 0 || cin <= 0 || cin % 4 != 0 || (k != 1 && k != 3)) return 0;
    return (uint64_t)((cout + 31) / 32) * ((cin + 15) / 16) * (k * k) * 2 * 256;
}
static int conv_pack_any(const float* weight, int cout, int cin, int k, float* packed, float* wsq, int f16, int prenormalize, nfe_stream_t stream) {
    NFE_REQUIRE(weight && packed && wsq, "nfe_conv_pack: null pointer");
    NFE_REQUIRE(cout > 0 && cin > 0 && cin % 4 == 0 && (k == 1 || k == 3), "nfe_conv_pack: need cin %% 4 == 0 and k in {1,3} (cout=%d cin=%d k=%d)", cout, cin, k);
    float* alpha = nullptr;
    if (f16 && prenormalize) {                           // the per-output-channel scales live behind the image (nfe_conv_pack_f16: + cout words)
        alpha = packed + nfe_conv_packed_words(cout, cin, k);
        hipLaunchKernelGGL(conv_wmax_kernel, dim3((unsigned)((cout + 3) / 4)), dim3(256), 0, (hipStream_t)stream, weight, cout, (long long)cin * k * k, alpha);
    }
    hipLaunchKernelGGL(conv_pack_kernel, dim3(grid1d((long long)nfe_conv_packed_words(cout, cin, k), 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       weight, cout, cin, k * k, packed, wsq, f16, alpha);
    NFE_CHECK_LAUNCH("conv_pack_kernel");
    return NFE_OK;
}
extern "C" int nfe_conv_pack(const float* weight, int cout, int cin, int k, float* packed, float* wsq, nfe_stream_t stream) {
    return conv_pack_any(weight, cout, cin, k, packed, wsq, 0, 0, stream);
}
extern "C" int nfe_conv_pack_f16(const float* weight, int cout, int cin, int k, int prenormalize, float* packed, float* wsq, nfe_stream_t stream) {
    return conv_pack_any(weight, cout, cin, k, packed, wsq, 1, prenormalize, stream);
}
extern "C" int nfe_conv_demod(const float* styles, const float* wsq, int n, int cin, int cout, float* dcoef, float* styles_norm, nfe_stream_t stream) {
    NFE_REQUIRE(styles && wsq && dcoef && n > 0 && cin > 0 && cout > 0, "nfe_conv_demod: bad arguments");
    const long long waves = (long long)n * cout;
    hipLaunchKernelGGL(demod_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, styles, wsq, n, cin, cout, dcoef, styles_norm);
    NFE_CHECK_LAUNCH("demod_kernel");
    return NFE_OK;
}

template <int MODE>
static void launch_conv(const ConvK& P, int math, dim3 grid, hipStream_t st) {
    if (math == NFE_CONV_F16) hipLaunchKernelGGL((conv_kernel<MODE, 2>), grid, dim3(256), 0, st, P);
    else if (math == NFE_CONV_BF16) hipLaunchKernelGGL((conv_kernel<MODE, 1>), grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL((conv_kernel<MODE, 3>), grid, dim3(256), 0, st, P);
}

#define C3_STAGES_BF16 2
#define C3_STAGES_BF16_UP 2
#define C3_STAGES_X3 1
#define C3_STAGES_X3_UP 1
#define C3_UP_FUSED_WAVES 0          // 0 = the run-time form (shipped).  > 0: waves per SIMD (= workgroups per CU) of a compile-time fused variant of the up-sampling kernel in bf16 / fp16: 2 measured the same as 0 (same box, three repetitions); 3 (168 registers: no patch-fragment cache, or 19 spilled dwords) measured 0 to -4 % on the 256-channel layers and +10 % on the 32-channel one, with the cache and its spills +40 %
#define C3_MID 0
#define C3_BIG 0          // measured slower (1 wave per SIMD, compiler-scheduled): bf16 SR 3.8 -> 4.0 ms, split-bf16 3.26 -> 3.45 ms
#define C3_TALL_MIN_TILES 4          // use the 8-wave 32x16 tile from 64 rows up
#define C3_WIDE8_DEFAULT 0
#define C3_LC_DEFAULT 0

static void launch_upfir(const ConvK& P, long long total, hipStream_t st) {
    const dim3 grid(grid1d(total, 256, 1 << 15)), block(256);
    const int parts = P.split_hi ? (P.split_lo ? 2 : 1) : 0;
#define NFE_UPFIR(O, S) hipLaunchKernelGGL((upfir_kernel<O, S>), grid, block, 0, st, P)
    if (P.out) { if (parts == 2) NFE_UPFIR(true, 2); else if (parts == 1) NFE_UPFIR(true, 1); else NFE_UPFIR(true, 0); }
    else { if (parts == 2) NFE_UPFIR(false, 2); else if (parts == 1) NFE_UPFIR(false, 1); else NFE_UPFIR(false, 0); }
#undef NFE_UPFIR
}

// Up-sampling layers on the conv3 fast path: FIR + layer epilogue fused into the transposed conv (no fp32 scratch round trip) where
// that pays, measured per math mode and input width (round 3: tools/r03_upfused_ab.sh; round 6: tools/r06_up_trace.sh).
//   strips (upconv_strip_kernel, round 6: no vertical overlap, 2 of 32 columns horizontally): bf16 / fp16 up to 256 input channels
//     (-12 / -17 % against the overlapping tiles on the two 256-channel layers, -2 % on the 32-channel one); split-bf16 from 64 to 256
//     input channels (-8 / -6 % against the unfused form, whose upfir pass it removes; the overlapping tiles LOSE to the unfused form
//     there: their 1.42 x MFMA work is what the split-bf16 K loop is bound by);
//   overlapping 32 x 8 tiles (conv3_kernel<.., UP2> with up_fused): split-bf16 with 32 input channels, where the launch is all
//     epilogue and the strips' longer row loop per thread measured 5 % slower; and everything above with NFE_UP_STRIP=0 (A/B).
// NFE_UP_FUSED=0 switches fusing off, NFE_UP_FUSED_CIN_BF16 / NFE_UP_FUSED_CIN_X3 move the thresholds (A/B knobs).
static bool up_strip_on() { static const bool on = [] { const char* e = getenv("NFE_UP_STRIP"); return !e || e[0] != '0'; }(); return on; }
static bool up_fused(int math, int ksplit, int cin) {
    if (math == NFE_CONV_F16) math = NFE_CONV_BF16;      // fp16 operands: same sizes, variants and thresholds as bf16
    static const bool on = [] { const char* e = getenv("NFE_UP_FUSED"); return !e || e[0] != '0'; }();
    static const int max_x3 = [] { const char* e = getenv("NFE_UP_FUSED_CIN_X3"); return e ? atoi(e) : (up_strip_on() ? 256 : 32); }();
    static const int max_bf16 = [] { const char* e = getenv("NFE_UP_FUSED_CIN_BF16"); return e ? atoi(e) : 256; }();
    return on && !ksplit && cin <= (math == NFE_CONV_BF16 ? max_bf16 : max_x3);
}
static bool up_strips(int math, int cin) {               // a fused layer: strips or overlapping tiles
    if (math == NFE_CONV_F16) math = NFE_CONV_BF16;
    return up_strip_on() && (math == NFE_CONV_BF16 || cin > 32);
}

// ---- strip form of the fused up-sampling layers (upconv_strip_kernel): geometry and launch
// NFE_UP_STRIP=0 keeps round 5's overlapping-tile kernel (A/B); NFE_UP_STRIP_SEGS=<n> fixes the segment count.
struct UpStripGeom { int strips, blocks, segs, seg_blocks; };
static UpStripGeom up_strip_geometry(int n, int h, int w, int cout, int tyl, int wgs_per_cu) {
    UpStripGeom g;
    g.strips = (w + 29) / 30;                                  // 60 output columns per strip
    g.blocks = (2 * h + 1 + tyl) / tyl;                        // block b emits the output rows [tyl b - 2, tyl b + tyl - 2)
    static const int forced = [] { const char* e = getenv("NFE_UP_STRIP_SEGS"); return e ? atoi(e) : 0; }();
    // Segments per strip: enough workgroups to keep every workgroup slot of the chip busy to the end, as few boundaries as that allows.  Estimated
    // utilisation of S segments = blocks of work / (rounds of slots x longest segment), charged 5 output rows' worth of time per
    // boundary (three rows leave as six fp32 row-filtered rows and come back: measured 22 us of 560 at 6 boundaries of a 512-row image).
    const long long slots = (long long)wgs_per_cu * num_cus(), wg0 = (long long)g.strips * (cout / 32) * n;
    int best = 1; double best_u = -1.0;
    for (int S = 1; S <= g.blocks && S <= 64; ++S) {
        const int L = (g.blocks + S - 1) / S, S2 = (g.blocks + L - 1) / L;
        if (S2 != S) continue;
        const long long wgs = wg0 * S;
        const double u = (double)g.blocks * wg0 / ((double)((wgs + slots - 1) / slots) * L * slots) / (1.0 + 5.0 * (S - 1) / (2.0 * h));
        if (u > best_u + 1e-9) { best_u = u; best = S; }
    }
    int S = forced > 0 ? (forced < g.blocks ? forced : g.blocks) : best;
    g.seg_blocks = (g.blocks + S - 1) / S;
    g.segs = (g.blocks + g.seg_blocks - 1) / g.seg_blocks;
    return g;
}
template <int TERMS, int STAGES, int NBW, int KG = 1>
static int launch_up_strip_t(const Conv3K& K0, hipStream_t st) {
    Conv3K K = K0;
    K.tyl = UpsGeo<NBW>::TYL;
    const UpStripGeom g = up_strip_geometry(K.N, K.H, K.W, K.Cout, K.tyl, NBW == 1 ? 3 : 2);
    K.strips = g.strips; K.blocks = g.blocks; K.segs = g.segs; K.seg_blocks = g.seg_blocks;
    K.c3_tiles = g.strips * g.segs;
    K.seam = g.segs > 1 ? K.scratch : nullptr;                 // the (2H+1) x (2W+1) scratch of the unfused form is free here and far larger
    constexpr int bytes = ups_lds_bytes<TERMS, STAGES, NBW, KG>();
    static LdsOptIn opt;
    const hipError_t e = opt.apply(upconv_strip_kernel<TERMS, STAGES, NBW, KG>, bytes);
    if (e != hipSuccess) return fail(NFE_ELAUNCH, "upconv_strip_kernel: LDS opt-in: %s", hipGetErrorString(e));
    const dim3 grid(UPS_XCD_ORDER ? (unsigned)((K.c3_tiles + 7) / 8 * 8) : (unsigned)K.c3_tiles, (unsigned)(K.Cout / 32), (unsigned)K.N);
    hipLaunchKernelGGL((upconv_strip_kernel<TERMS, STAGES, NBW, KG>), grid, dim3(256), bytes, st, K);
    if (g.segs > 1) {
        const long long total = (long long)K.N * (g.segs - 1) * 2 * K.W * (K.Cout / 4);
        hipLaunchKernelGGL((upconv_seam_kernel<TERMS>), dim3(grid1d(total, 256, 1 << 14)), dim3(256), 0, st, K);
    }
    return NFE_OK;
}
constexpr int UPS_STAGES_BF16 = 3, UPS_STAGES_X3 = 1;       // ring depth per operand mode (bf16 / fp16: 2 measured 6 % slower)
// NFE_UP_STRIP_ROWS=1 (A/B, read once): one image row per wave - 32 x 4 blocks, three workgroups per CU
static int launch_up_strip(const Conv3K& K, bool bf16, hipStream_t st) {
    static const int rows1 = [] { const char* e = getenv("NFE_UP_STRIP_ROWS"); return e && e[0] == '1'; }();
    // NFE_UP_STRIP_K32=1 (A/B): two K-groups per stage, ring of two - measured neutral on the 256-channel layers (523 vs 528 us) and 5 % slower
    // on the 32-channel one (a single stage: nothing to prefetch), so the ring of three single K-groups stays the default
    static const bool k32 = [] { const char* e = getenv("NFE_UP_STRIP_K32"); return e && e[0] == '1'; }();
    if (!bf16) return launch_up_strip_t<3, UPS_STAGES_X3, 2>(K, st);
    if (rows1) return K.f16 ? launch_up_strip_t<2, UPS_STAGES_BF16, 1>(K, st) : launch_up_strip_t<1, UPS_STAGES_BF16, 1>(K, st);
    if (k32 && K.Cin % 32 == 0) return K.f16 ? launch_up_strip_t<2, 2, 2, 2>(K, st) : launch_up_strip_t<1, 2, 2, 2>(K, st);
    if (K.f16) return launch_up_strip_t<2, UPS_STAGES_BF16, 2>(K, st);
    return launch_up_strip_t<1, UPS_STAGES_BF16, 2>(K, st);
}

static_assert(conv3_ec_bytes<2>() == 1024 && conv3_ec_bytes<4>() == 1536, "epilogue constants: three float rows of 32 * MBW channels");
template <int TERMS, int MBW, bool UP2, int STAGES, int WV, int NBW = 2, int LW = 0, int FU = 0>
static void launch_conv3_t(const Conv3K& K, int mode_h, int mode_w, hipStream_t st, unsigned tiles_override);
// the bf16 variant table serves fp16 too: same tiles, stages and thresholds, the kernel instantiated with TERMS = 2
template <int TERMS, int MBW, bool UP2, int STAGES, int WV, int NBW = 2, int LW = 0, int FU = 0>
static void launch_conv3(const Conv3K& K, int mode_h, int mode_w, hipStream_t st, unsigned tiles_override = 0) {
    if constexpr (TERMS == 1) { if (K.f16) { launch_conv3_t<2, MBW, UP2, STAGES, WV, NBW, LW, FU>(K, mode_h, mode_w, st, tiles_override); return; } }
    launch_conv3_t<TERMS, MBW, UP2, STAGES, WV, NBW, LW, FU>(K, mode_h, mode_w, st, tiles_override);
}
template <int TERMS, int MBW, bool UP2, int STAGES, int WV, int NBW, int LW, int FU>
static void launch_conv3_t(const Conv3K& K, int mode_h, int mode_w, hipStream_t st, unsigned tiles_override) {
    constexpr int ROWS = NBW * WV;
    constexpr int ring = STAGES * conv3_stage_bytes<TERMS, MBW, ROWS>(), fused_t = UP2 ? conv3_fused_t_bytes(ROWS) : 0;
    constexpr int bytes = (ring > fused_t ? ring : fused_t) + conv3_ec_bytes<MBW>();  // ring (or the fused FIR's tile) + the epilogue constants
    static LdsOptIn opt;                              // per device; a failed opt-in shows as the launch's own error (NFE_CHECK_LAUNCH at the call site)
    if (opt.apply(conv3_kernel<TERMS, MBW, UP2, STAGES, WV, NBW, LW, FU>, bytes) != hipSuccess) (void)hipGetLastError();
    const unsigned tiles = tiles_override ? tiles_override : ((mode_h + ROWS - 1) / ROWS) * ((mode_w + C3_TW - 1) / C3_TW);
    Conv3K K2 = K; K2.c3_tiles = (int)tiles;
    dim3 grid(c3_xcd_order(TERMS) ? (tiles + 7) / 8 * 8 : tiles, K.Cout / (32 * MBW), K.N * (K.ksplit > 1 ? K.ksplit : 1));
    hipLaunchKernelGGL((conv3_kernel<TERMS, MBW, UP2, STAGES, WV, NBW, LW, FU>), grid, dim3(64 * (WV + LW)), bytes, st, K2);
}

// Which layers take the LDS-DMA path.  The tiles are 32 pixels wide; split-bf16 wants images at least that wide (narrower ones waste
// the MFMA work its K loop is bound by: measured slower), plain bf16 takes everything down to 4 x 4 - the generic kernel has no
// operand pipeline and spent 79 us per 4^2..16^2 layer at 8 views (tools/r03_small_layers.sh: backbone 2.01 -> 1.83 ms).  The
// size-only queries of the API (no math argument) answer with the conservative rule.
static bool conv3_eligible(int mode, int h, int w, int cin, int cout, int math = NFE_CONV_BF16X3) {
    if (math == NFE_CONV_F16) math = NFE_CONV_BF16;      // fp16 operands: same sizes, variants and thresholds as bf16
    static const int min_w_env = [] { const char* e = getenv("NFE_C3_MIN_W"); return e ? atoi(e) : 0; }();      // A/B knobs
    static const int min_h_env = [] { const char* e = getenv("NFE_C3_MIN_H"); return e ? atoi(e) : 0; }();
    const int min_w = min_w_env ? min_w_env : (math == NFE_CONV_BF16 ? 4 : 32), min_h = min_h_env ? min_h_env : (math == NFE_CONV_BF16 ? 4 : 8);
    if (cin % 16 != 0 || w < min_w || h < min_h) return false;
    return mode == NFE_CONV_3X3 ? cout % 64 == 0 : (mode == NFE_CONV_3X3_UP2 && cout % 32 == 0);
}

// Plain 3x3 layers on the LDS-DMA path with too few (tile, M-block group, sample) workgroups to fill the chip twice (32^2 / 64^2
// layers at small batch): split the K loop over 2 or 4 workgroups, each keeping at least 8 K-groups; partial sums go through
// splitk_reduce_kernel (deterministic slice order).  0 = no split.
static int conv3_ksplit(int mode, int n, int h, int w, int cin, int cout, int math = NFE_CONV_BF16X3) {
    if (math == NFE_CONV_F16) math = NFE_CONV_BF16;      // fp16 operands: same sizes, variants and thresholds as bf16
    static const bool off = [] { const char* e = getenv("NFE_C3_KSPLIT"); return e && e[0] == '0'; }();
    if (off || mode == NFE_CONV_1X1 || !conv3_eligible(mode, h, w, cin, cout, math)) return 0;
    const int G = cin / 16;
    if (mode == NFE_CONV_3X3_UP2) {            // the kernel and the reduce pass support it (NFE_C3_KSPLIT_UP=1), measured without gain at
        static const bool on = [] { const char* e = getenv("NFE_C3_KSPLIT_UP"); return e && e[0] == '1'; }();      // batch 1 and 4: off
        if (!on) return 0;
        const long long wgs = (long long)((h + 1 + 7) / 8) * ((w % 32 == 0 ? w : w + 1 + 31) / 32) * (cout / 32) * n;
        int ks = 1;
        while (ks < 4 && wgs * ks < 4LL * num_cus() && G / (ks * 2) >= 4) ks *= 2;
        return ks > 1 ? ks : 0;
    }
    const int rows = (h >= 16 * C3_TALL_MIN_TILES && (long long)((h + 15) / 16) * ((w + 31) / 32) * ((cout + 63) / 64) * n >= num_cus()) ? 16 : 8;
    const long long wgs = (long long)((h + rows - 1) / rows) * ((w + 31) / 32) * (cout / 64) * n;
    int ks = 1;
    while (ks < 4 && wgs * ks < 2LL * num_cus() && G / (ks * 2) >= 8) ks *= 2;
    return ks > 1 ? ks : 0;
}

// Which conv3_kernel instantiation a fast-path layer runs (one place: the launcher and nfe_conv_describe both ask here).
enum { C3V_UP = 0, C3V_BIG, C3V_MID, C3V_X3_TALL4, C3V_TALL8, C3V_BASE, C3V_WIDE8, C3V_LC };
static int conv3_variant(int mode, int math, int n, int h, int w, int cout) {
    if (math == NFE_CONV_F16) math = NFE_CONV_BF16;      // fp16 operands: same sizes, variants and thresholds as bf16
    const bool bf16 = math == NFE_CONV_BF16;
    if (mode == NFE_CONV_3X3_UP2) return C3V_UP;
    const bool tall = h >= 16 * C3_TALL_MIN_TILES;
    const bool fills = (long long)((h + 15) / 16) * ((w + 31) / 32) * ((cout + 63) / 64) * n >= num_cus();
    if (C3_BIG && cout % 128 == 0 && h >= 16 && (long long)((h + 15) / 16) * ((w + 31) / 32) * (cout / 128) * n >= 2LL * num_cus()) return C3V_BIG;
    // round 3: 32x16 tile, four compute waves (4 rows x 2 M-blocks each) + four loader waves, one workgroup per CU, ring of 4 (bf16)
    // or 2 (split-bf16) stages
    static const int lc = [] { const char* e = getenv("NFE_C3_LC"); return e ? atoi(e) : C3_LC_DEFAULT; }();
    if (lc && tall && (long long)((h + 15) / 16) * ((w + 31) / 32) * ((cout + 63) / 64) * n >= num_cus()) return C3V_LC;
    if (C3_MID && tall && bf16) return C3V_MID;
    // round 3: 128 channels x 32x16 pixels on EIGHT waves (4 M-blocks x 2 rows per wave, 128 accumulator registers, one workgroup per
    // CU): 1.37x fewer staged bytes per MFMA than the 64-channel tile and 72 MFMAs per wave between barriers instead of 36
    static const int wide8 = [] { const char* e = getenv("NFE_C3_WIDE8"); return e ? atoi(e) : C3_WIDE8_DEFAULT; }();
    if (wide8 && bf16 && tall && cout % 128 == 0 && (long long)((h + 15) / 16) * ((w + 31) / 32) * (cout / 128) * n >= num_cus()) return C3V_WIDE8;
    if (!bf16 && tall && fills) return C3V_X3_TALL4;
    if (tall && fills) return C3V_TALL8;
    return C3V_BASE;
}

extern "C" int nfe_conv_splits_in_epilogue(int mode, int n, int h, int w, int cin, int cout) {
    return mode == NFE_CONV_3X3 && conv3_eligible(mode, h, w, cin, cout) && cout % 4 == 0 && !conv3_ksplit(mode, n, h, w, cin, cout) ? 1 : 0;
}

extern "C" int nfe_conv_fuses_rgb(int mode, int math, int n, int h, int w, int cin, int cout, int rgb_channels) {
    if (math == NFE_CONV_F16) math = NFE_CONV_BF16;      // fp16 operands: same sizes, variants and thresholds as bf16
    if (mode != NFE_CONV_3X3 || rgb_channels < 1 || rgb_channels > 4 || cout > 256 || C3_BIG) return 0;
    return conv3_eligible(mode, h, w, cin, cout, math) && conv3_ksplit(mode, n, h, w, cin, cout, math) == 0 ? 1 : 0;
}

extern "C" uint64_t nfe_conv_split_floats(int math, int n, int h, int w, int c) {
    if (math == NFE_CONV_F16) math = NFE_CONV_BF16;      // fp16 operands: same sizes, variants and thresholds as bf16
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return 0;
    const uint64_t elems = (uint64_t)n * h * w * c;
    return math == NFE_CONV_BF16 ? (elems + 1) / 2 : elems;
}

extern "C" int nfe_conv_accepts_split(int mode, int h, int w, int cin, int cout) { return conv3_eligible(mode, h, w, cin, cout) ? 1 : 0; }

// Small 3x3 layers (at most one 16x16 tile per sample) split their K loop over this many workgroups.  So do 1x1 (ToRGB)
// layers up to 128^2 whose weights do not fit the LDS-resident torgb_kernel (512 / 256 input channels) when the batch is too
// small to fill the chip with (tile, M-block) workgroups alone: 73 us -> 15 us for the 64^2 x 512 -> 96 layer at batch 1.
static int splitk_slices(int mode, int math, int n, int h, int w, int cin, int cout) {
    if (math == NFE_CONV_F16) math = NFE_CONV_BF16;      // fp16 operands: same sizes, variants and thresholds as bf16
    if (cout % 4 != 0) return 0;
    const int G = (cin + 15) / 16;
    const int ks = G >= 16 ? 8 : (G >= 8 ? 4 : 0);
    const int lim = mode == NFE_CONV_1X1 ? 32 : 16;            // ToRGB has 9x less work per K-group: worth it up to 32^2
    if (h <= lim && w <= lim) return ks;
    if (mode == NFE_CONV_1X1 && h <= 128 && w <= 128) {
        const int mb = (cout + 31) / 32, parts = math == NFE_CONV_BF16 ? 1 : 2;
        const bool lds_resident = cin % 16 == 0 && (mb == 1 || mb == 3) && (long long)mb * (cin / 16) * parts * 1024 <= 64 * 1024;
        const long long groups = (long long)n * ((h + 15) / 16) * ((w + 15) / 16) * mb;
        if (!lds_resident && groups < 256) return ks;
    }
    return 0;
}

extern "C" uint64_t nfe_conv_scratch_floats(int mode, int math, int n, int h, int w, int cin, int cout) {
    if (math == NFE_CONV_F16) math = NFE_CONV_BF16;      // fp16 operands: same sizes, variants and thresholds as bf16
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return 0;
    uint64_t fl = mode == NFE_CONV_3X3_UP2 ? (uint64_t)n * (2 * h + 1) * (2 * w + 1) * cout : 0;     // transposed-conv result
    if (conv3_eligible(mode, h, w, cin, cout, math)) {
        const uint64_t elems = (uint64_t)n * h * w * cin;           // bf16 hi (+ lo) image of the modulated input
        fl += math == NFE_CONV_BF16 ? (elems + 1) / 2 : elems;
        fl += (uint64_t)conv3_ksplit(mode, n, h, w, cin, cout, math) * n *
              (mode == NFE_CONV_3X3_UP2 ? (uint64_t)(2 * h + 1) * (2 * w + 1) : (uint64_t)h * w) * cout;          // split-K partial sums
        if (nfe_conv_fuses_rgb(mode, math, n, h, w, cin, cout, 3)) fl += (uint64_t)(cout / 64) * n * h * w * 4;   // fused-ToRGB partial sums
    } else if (const int ks = splitk_slices(mode, math, n, h, w, cin, cout)) {
        fl += (uint64_t)ks * n * (mode == NFE_CONV_3X3_UP2 ? (uint64_t)(2 * h + 1) * (2 * w + 1) : (uint64_t)h * w) * cout;   // partial sums
    }
    return fl;
}

// Text description of the kernels nfe_modulated_conv would launch for a layer of these sizes (tests log it so that a parity
// failure names the variant; batch-dependent: split-K, fused ToRGB, epilogue split and the tile shape all depend on n).
extern "C" int nfe_conv_describe(int mode, int math, int n, int h, int w, int cin, int cout, int rgb_channels, char* buf, int buf_len) {
    const bool f16_ = math == NFE_CONV_F16;
    if (f16_) math = NFE_CONV_BF16;
    NFE_REQUIRE(buf && buf_len > 0, "nfe_conv_describe: no buffer");
    const char* m = f16_ ? "fp16" : (math == NFE_CONV_BF16 ? "bf16" : "bf16x3");
    if (mode != NFE_CONV_1X1 && conv3_eligible(mode, h, w, cin, cout, math)) {
        static const char* names[] = {"up2 1x(32x8)/4w", "big 128ch 32x16/4w", "mid 32x16/4w", "x3 32x16/4w (2x4 blocks)", "32x16/8w", "32x8/4w", "128ch 32x16/8w", "32x16/4w compute + 4w loaders"};
        const int ks = conv3_ksplit(mode, n, h, w, cin, cout, math);
        snprintf(buf, (size_t)buf_len, "conv3[%s] %s ksplit=%d fuse_rgb=%d split_in_epilogue=%d%s", names[conv3_variant(mode, math, n, h, w, cout)], m, ks,
                 rgb_channels > 0 ? nfe_conv_fuses_rgb(mode, math, n, h, w, cin, cout, rgb_channels) : 0,
                 nfe_conv_splits_in_epilogue(mode, n, h, w, cin, cout),
                 mode != NFE_CONV_3X3_UP2 ? "" : (up_fused(math, ks, cin) ? (up_strips(math, cin) ? " fused FIR epilogue (strips)" : " fused FIR epilogue (overlapping tiles)") : " +upfir"));
        return NFE_OK;
    }
    const int ks = splitk_slices(mode, math, n, h, w, cin, cout);
    const int mb1 = (cout + 31) / 32, parts1 = math == NFE_CONV_BF16 ? 1 : 2;
    const bool torgb_fast = mode == NFE_CONV_1X1 && !ks && cin % 16 == 0 && (mb1 == 1 || mb1 == 3) &&
                            (long long)mb1 * (cin / 16) * parts1 * 1024 <= 64 * 1024 && (long long)h * w >= 1024;
    snprintf(buf, (size_t)buf_len, "%s %s splitk=%d%s", torgb_fast ? "torgb[lds-resident]" : (mode == NFE_CONV_1X1 ? "generic1x1" : "generic3x3"), m, ks,
             mode == NFE_CONV_3X3_UP2 ? " +upfir" : "");
    return NFE_OK;
}

extern "C" int nfe_modulated_conv(const nfe_conv_args* a, nfe_stream_t stream) {
    NFE_REQUIRE(a != nullptr, "nfe_modulated_conv: args is null");
    NFE_REQUIRE(a->struct_size == sizeof(nfe_conv_args), "nfe_modulated_conv: struct_size %u != %zu (ABI mismatch)", a->struct_size, sizeof(nfe_conv_args));
    NFE_REQUIRE(a->mode >= 0 && a->mode <= 2, "nfe_modulated_conv: unknown mode %d", a->mode);
    NFE_REQUIRE(a->math == NFE_CONV_BF16X3 || a->math == NFE_CONV_BF16 || a->math == NFE_CONV_F16, "nfe_modulated_conv: unknown math %d", a->math);
    const int f16 = a->math == NFE_CONV_F16 ? 1 : 0;
    const int math = f16 ? NFE_CONV_BF16 : a->math;      // fp16: bf16's sizes, variants and thresholds; the launches pick the TERMS = 2 kernels
    NFE_REQUIRE((a->x || a->x_split) && a->styles && a->packed && a->bias && (a->out || a->next_split || a->rgb_weight), "nfe_modulated_conv: null pointer");
    NFE_REQUIRE(!a->next_split || (a->next_styles && a->mode != NFE_CONV_1X1), "nfe_modulated_conv: next_split needs next_styles and a 3x3 mode");
    NFE_REQUIRE(a->n > 0 && a->h > 0 && a->w > 0 && a->cin > 0 && a->cout > 0 && a->cin % 4 == 0, "nfe_modulated_conv: bad sizes n=%d h=%d w=%d cin=%d cout=%d", a->n, a->h, a->w, a->cin, a->cout);
    NFE_REQUIRE(a->mode != NFE_CONV_3X3_UP2 || (a->scratch && a->cout % 4 == 0 && a->scratch_floats >= (uint64_t)a->n * (2 * a->h + 1) * (2 * a->w + 1) * a->cout),
                "nfe_modulated_conv: up-conv needs cout %% 4 == 0 and scratch of N*(2H+1)*(2W+1)*Cout floats");
    NFE_REQUIRE(!a->skip || (a->mode == NFE_CONV_1X1 && a->h % 2 == 0 && a->w % 2 == 0), "nfe_modulated_conv: skip needs mode 1x1 and even size");
    NFE_REQUIRE(!a->out_planes || (a->mode == NFE_CONV_1X1 && a->cout == 96), "nfe_modulated_conv: out_planes needs mode 1x1 and cout 96");
    ConvK P{};
    P.f16 = f16;
    P.x = a->x; P.styles = a->styles; P.packed = reinterpret_cast<const uint4*>(a->packed); P.dcoef = a->dcoef; P.noise = a->noise; P.noise_n_stride = a->noise_n_stride;
    P.noise_strength = a->noise_strength; P.bias = a->bias; P.N = a->n; P.H = a->h; P.W = a->w; P.Cin = a->cin; P.Cout = a->cout;
    P.lrelu = a->lrelu; P.act_gain = a->act_gain; P.clamp = a->clamp; P.skip = a->skip; P.out_planes = a->out_planes; P.out = a->out; P.scratch = a->scratch;
    hipStream_t st = (hipStream_t)stream;
    const int upf = a->mode == NFE_CONV_3X3_UP2 ? 2 : 1;
    const long long out_elems = (long long)a->n * a->h * upf * a->w * upf * a->cout;
    if (a->next_split && a->mode == NFE_CONV_3X3_UP2) {            // fused into the FIR epilogue
        P.next_styles = a->next_styles; P.split_hi = reinterpret_cast<uint2*>(a->next_split);
        P.split_lo = math == NFE_CONV_BF16X3 ? reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a->next_split) + out_elems) : nullptr;
    }
    // plain 3x3 with a consumer image: one extra elementwise pass over the fp32 output
    auto split_tail = [&]() -> int {
        if (!(a->next_split && a->mode == NFE_CONV_3X3)) return NFE_OK;
        NFE_REQUIRE(a->out && a->cout % 16 == 0, "nfe_modulated_conv: next_split on a 3x3 layer needs `out` and cout %% 16 == 0");
        unsigned short* sh = reinterpret_cast<unsigned short*>(a->next_split);
        hipLaunchKernelGGL(modsplit_kernel, dim3(grid1d(out_elems / 4, 256, 1 << 15)), dim3(256), 0, st, reinterpret_cast<const float4*>(a->out), a->next_styles,
                           out_elems / 4, (long long)a->h * a->w * (a->cout / 4), a->cout / 4, reinterpret_cast<uint2*>(sh),
                           math == NFE_CONV_BF16X3 ? reinterpret_cast<uint2*>(sh + out_elems) : nullptr, f16);
        NFE_CHECK_LAUNCH("modsplit_kernel");
        return NFE_OK;
    };
    const bool fast = a->mode != NFE_CONV_1X1 && a->scratch && conv3_eligible(a->mode, a->h, a->w, a->cin, a->cout, math) &&
                      a->scratch_floats >= nfe_conv_scratch_floats(a->mode, math, a->n, a->h, a->w, a->cin, a->cout);
    NFE_REQUIRE(!a->x_split || fast, "nfe_modulated_conv: x_split needs the fast path (eligible sizes and nfe_conv_scratch_floats() of scratch)");
    NFE_REQUIRE(a->mode != NFE_CONV_3X3_UP2 || a->out || a->next_split, "nfe_modulated_conv: no output requested");
    const bool fuse_rgb = a->rgb_weight != nullptr;
    if (fuse_rgb) {
        NFE_REQUIRE(a->rgb_styles && a->rgb_bias && a->rgb_out, "nfe_modulated_conv: rgb_weight needs rgb_styles, rgb_bias and rgb_out");
        NFE_REQUIRE(fast && nfe_conv_fuses_rgb(a->mode, math, a->n, a->h, a->w, a->cin, a->cout, a->rgb_channels),
                    "nfe_modulated_conv: this layer cannot evaluate ToRGB in its epilogue (ask nfe_conv_fuses_rgb first)");
        NFE_REQUIRE(!a->rgb_skip || (a->h % 2 == 0 && a->w % 2 == 0), "nfe_modulated_conv: rgb_skip needs even sizes");
    }
    const bool split_only = a->next_split && fast && nfe_conv_splits_in_epilogue(a->mode, a->n, a->h, a->w, a->cin, a->cout);   // the epilogue writes the consumer's image itself
    NFE_REQUIRE(a->mode == NFE_CONV_3X3_UP2 || a->out || fuse_rgb || split_only,
                "nfe_modulated_conv: `out` may only be NULL on up-sampling layers with next_split, with a fused ToRGB, or where nfe_conv_splits_in_epilogue()");
    if (fast) {
        // fast path: modulate + split once, then the LDS-DMA implicit GEMM
        const bool up2 = a->mode == NFE_CONV_3X3_UP2;
        const long long elems = (long long)a->n * a->h * a->w * a->cin;
        float* tail = a->scratch + (up2 ? (long long)a->n * (2 * a->h + 1) * (2 * a->w + 1) * a->cout : 0);   // split image sits after the FIR scratch
        unsigned short* xh = reinterpret_cast<unsigned short*>(a->x_split ? const_cast<float*>(a->x_split) : tail);
        unsigned short* xl = math == NFE_CONV_BF16X3 ? xh + elems : nullptr;
        if (!a->x_split)
            hipLaunchKernelGGL(modsplit_kernel, dim3(grid1d(elems / 4, 256, 1 << 15)), dim3(256), 0, st, reinterpret_cast<const float4*>(a->x), a->styles,
                               elems / 4, (long long)a->h * a->w * (a->cin / 4), a->cin / 4, reinterpret_cast<uint2*>(xh), reinterpret_cast<uint2*>(xl), f16);
        Conv3K K{};
        K.f16 = f16;
        K.xh = xh; K.xl = xl; K.packed = reinterpret_cast<const uint4*>(a->packed); K.dcoef = a->dcoef; K.noise = a->noise;
        K.noise_n_stride = a->noise_n_stride; K.noise_strength = a->noise_strength; K.bias = a->bias; K.N = a->n; K.H = a->h; K.W = a->w;
        K.Cin = a->cin; K.Cout = a->cout; K.lrelu = a->lrelu; K.act_gain = a->act_gain; K.clamp = a->clamp; K.out = a->out; K.scratch = a->scratch;
        const int ext = up2 ? 1 : 0;
        const bool bf16 = math == NFE_CONV_BF16;
        const int c3ks = conv3_ksplit(a->mode, a->n, a->h, a->w, a->cin, a->cout, math);
        if (fuse_rgb) {
            K.rgb_w = a->rgb_weight; K.rgb_s = a->rgb_styles; K.rgb_c = a->rgb_channels;
            K.rgb_partial = tail + (math == NFE_CONV_BF16 ? (elems + 1) / 2 : elems);   // behind the split-image area (no split-K here)
        }
        if (c3ks) {
            K.ksplit = c3ks;
            K.partial = tail + (math == NFE_CONV_BF16 ? (elems + 1) / 2 : elems);       // behind the (possibly unused) split-image area
        }
        const bool split_in_epilogue = a->next_split && !up2 && !c3ks && a->cout % 4 == 0;   // else: split_tail() re-reads `out`
        if (split_in_epilogue) {
            K.next_styles = a->next_styles; K.split_hi = reinterpret_cast<uint2*>(a->next_split);
            K.split_lo = math == NFE_CONV_BF16X3 ? reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a->next_split) + out_elems) : nullptr;
        }
        auto reduce_up = [&]() {                    // slices of the transposed-conv result -> a->scratch, in order, before the FIR
            if (!(c3ks && up2)) return;
            const long long slice = (long long)a->n * (2 * a->h + 1) * (2 * a->w + 1) * a->cout;
            P.ksplit = c3ks; P.partial = K.partial;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid1d(slice / 4, 256, 1 << 14)), dim3(256), 0, st, P, slice / 4, 1);
        };
        int rgb_groups = a->cout / 64;               // M-block groups (workgroups along Cout) that leave a fused-ToRGB partial sum
        switch (conv3_variant(a->mode, math, a->n, a->h, a->w, a->cout)) {
        case C3V_UP: {
            // (round 2 measured, without gain: a double-buffered stage (C3_STAGES_X3_UP = 2) and the 32 x 16 tile on 8 waves: DESIGN.md 5)
            if (up_fused(math, c3ks, a->cin)) {                 // FIR and layer epilogue inside the conv kernel, overlapping tiles
                K.up_fused = 1; K.next_styles = P.next_styles; K.split_hi = P.split_hi; K.split_lo = P.split_lo;
                if (up_strips(math, a->cin)) {                  // round 6: strips walked top to bottom, no vertical tile overlap
                    if (int rc = launch_up_strip(K, bf16, st)) return rc;
                    break;
                }
                const unsigned tiles = (unsigned)((a->h + 5) / 6) * (unsigned)((a->w + 29) / 30);      // ROWS = 8: 6 x 30 new extended-input pixels per tile
                if (bf16) launch_conv3<1, 1, true, C3_STAGES_BF16_UP, 4, 2, 0, C3_UP_FUSED_WAVES>(K, 0, 0, st, tiles);
                else launch_conv3<3, 1, true, C3_STAGES_X3_UP, 4>(K, 0, 0, st, tiles);
                break;
            }
            const int ext_w = (a->w % C3_TW) == 0 ? 0 : ext;       // EDGE mode: the extra column rides on the right-most tiles
            if (bf16) launch_conv3<1, 1, true, C3_STAGES_BF16_UP, 4>(K, a->h + ext, a->w + ext_w, st);
            else launch_conv3<3, 1, true, C3_STAGES_X3_UP, 4>(K, a->h + ext, a->w + ext_w, st);
            reduce_up();
            const long long total = (long long)a->n * ((a->h + UPFIR_ROWS - 1) / UPFIR_ROWS) * a->w * (a->cout / 4);
            launch_upfir(P, total, st);
            break;
        }
        case C3V_BIG:       // big tile, one wave per SIMD: 128 channels x 32x16 pixels on 4 waves, as long as the grid still fills the chip twice
            if (bf16) launch_conv3<1, 4, false, 2, 4, 4>(K, a->h, a->w, st);
            else launch_conv3<3, 2, false, 2, 4, 4>(K, a->h, a->w, st);
            break;
        case C3V_MID:       // 32 x 16 tiles on 4 waves (2 x 4 blocks per wave)
            launch_conv3<1, 2, false, 2, 4, 4>(K, a->h, a->w, st);
            break;
        case C3V_X3_TALL4:
            // split-bf16: the 32 x 16 tile on FOUR waves (2 x 4 blocks per wave): the kernel is co-limited by LDS fragment reads,
            // and a wave that owns four rows re-uses each weight fragment four times (0.5 instead of 0.67 reads per MFMA): +1.5 %
            launch_conv3<3, 2, false, C3_STAGES_X3, 4, 4>(K, a->h, a->w, st);
            break;
        case C3V_LC:
            if (bf16) launch_conv3<1, 2, false, 4, 4, 4, 4>(K, a->h, a->w, st);
            else launch_conv3<3, 2, false, 2, 4, 4, 4>(K, a->h, a->w, st);
            break;
        case C3V_WIDE8:
            launch_conv3<1, 4, false, C3_STAGES_BF16, 8>(K, a->h, a->w, st);
            rgb_groups = a->cout / 128;
            break;
        case C3V_TALL8:     // 32 x 16 tiles (8 waves): half the weight bytes per MFMA - as long as they still fill the chip
            if (bf16) launch_conv3<1, 2, false, C3_STAGES_BF16, 8>(K, a->h, a->w, st);
            else launch_conv3<3, 2, false, C3_STAGES_X3, 8>(K, a->h, a->w, st);
            break;
        default:
            if (bf16) launch_conv3<1, 2, false, C3_STAGES_BF16, 4>(K, a->h, a->w, st);
            else launch_conv3<3, 2, false, C3_STAGES_X3, 4>(K, a->h, a->w, st);
        }
        if (c3ks && !up2) {
            const long long slice = (long long)a->n * a->h * a->w * a->cout;
            P.ksplit = c3ks; P.partial = K.partial;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid1d(slice / 4, 256, 1 << 14)), dim3(256), 0, st, P, slice / 4, 0);
        }
        if (fuse_rgb)
            hipLaunchKernelGGL(rgb_combine_kernel, dim3(grid1d((long long)a->n * a->h * a->w, 256, 1 << 14)), dim3(256), 0, st, K.rgb_partial,
                               rgb_groups, a->n, a->h, a->w, a->rgb_channels, a->rgb_bias, a->rgb_clamp, a->rgb_skip, a->rgb_out);
        NFE_CHECK_LAUNCH("conv3 kernels");
        return split_in_epilogue ? NFE_OK : split_tail();
    }
    const int up = a->mode == NFE_CONV_3X3_UP2;
    const int gh = a->h + up, gw = a->w + up;
    dim3 grid(((gh + 15) / 16) * ((gw + 15) / 16), (a->cout + 31) / 32, a->n);
    const int ks = a->out_planes ? 0 : splitk_slices(a->mode, math, a->n, a->h, a->w, a->cin, a->cout);
    if (ks && a->scratch && a->scratch_floats >= nfe_conv_scratch_floats(a->mode, math, a->n, a->h, a->w, a->cin, a->cout)) {
        const long long slice = (long long)a->n * (up ? (long long)(2 * a->h + 1) * (2 * a->w + 1) : (long long)a->h * a->w) * a->cout;
        P.ksplit = ks; P.partial = a->scratch + (up ? slice : 0);      // after the transposed-conv scratch
        grid.x *= ks;
        if (up) launch_conv<NFE_CONV_3X3_UP2>(P, f16 ? NFE_CONV_F16 : math, grid, st);
        else if (a->mode == NFE_CONV_1X1) launch_conv<NFE_CONV_1X1>(P, f16 ? NFE_CONV_F16 : math, grid, st);
        else launch_conv<NFE_CONV_3X3>(P, f16 ? NFE_CONV_F16 : math, grid, st);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid1d(slice / 4, 256, 1 << 14)), dim3(256), 0, st, P, slice / 4, up);
        if (up) {
            const long long total = (long long)a->n * ((a->h + UPFIR_ROWS - 1) / UPFIR_ROWS) * a->w * (a->cout / 4);
            launch_upfir(P, total, st);
        }
        NFE_CHECK_LAUNCH("split-K conv kernels");
        return split_tail();
    }
    const int mb1 = (a->cout + 31) / 32, parts1 = math == NFE_CONV_BF16 ? 1 : 2;
    const bool torgb_fast = a->mode == NFE_CONV_1X1 && a->cin % 16 == 0 && (mb1 == 1 || mb1 == 3) && a->lrelu == 0 && !a->dcoef && !a->noise &&
                            (long long)mb1 * (a->cin / 16) * parts1 * 1024 <= 64 * 1024 && (long long)a->h * a->w >= 1024;
    if (a->mode == NFE_CONV_3X3) launch_conv<NFE_CONV_3X3>(P, f16 ? NFE_CONV_F16 : math, grid, st);
    else if (torgb_fast) {
        if (mb1 == 1) { if (parts1 == 1) launch_torgb<1, 1>(P, st); else launch_torgb<3, 1>(P, st); }
        else { if (parts1 == 1) launch_torgb<1, 3>(P, st); else launch_torgb<3, 3>(P, st); }
    }
    else if (a->mode == NFE_CONV_1X1) launch_conv<NFE_CONV_1X1>(P, f16 ? NFE_CONV_F16 : math, grid, st);
    else {
        launch_conv<NFE_CONV_3X3_UP2>(P, f16 ? NFE_CONV_F16 : math, grid, st);
        const long long total = (long long)a->n * ((a->h + UPFIR_ROWS - 1) / UPFIR_ROWS) * a->w * (a->cout / 4);
        launch_upfir(P, total, st);
    }
    NFE_CHECK_LAUNCH("conv kernels");
    return split_tail();
}

extern "C" int nfe_upfirdn2d(const float* in, int n, int h, int w, int c, int up, int down, int pad0, int pad1, float gain, float* out,
                             nfe_stream_t stream) {
    NFE_REQUIRE(in && out && n > 0 && h > 0 && w > 0 && c > 0, "nfe_upfirdn2d: bad arguments");
    NFE_REQUIRE((up == 1 || up == 2) && (down == 1 || down == 2) && pad0 >= 0 && pad1 >= 0, "nfe_upfirdn2d: up / down must be 1 or 2, pads >= 0");
    const int oh = (h * up + pad0 + pad1 - 4) / down + 1, ow = (w * up + pad0 + pad1 - 4) / down + 1;
    NFE_REQUIRE(oh > 0 && ow > 0, "nfe_upfirdn2d: empty output");
    hipLaunchKernelGGL(upfirdn_kernel<false>, dim3(grid1d((long long)n * oh * ow * c, 256, 1 << 15)), dim3(256), 0, (hipStream_t)stream,
                       in, n, h, w, c, up, down, pad0, gain, oh, ow, out);
    NFE_CHECK_LAUNCH("upfirdn_kernel");
    return NFE_OK;
}

extern "C" int nfe_upfirdn2d_polyphase(const float* in, int n, int h, int w, int c, int pad0, int pad1, float gain, float* out, nfe_stream_t stream) {
    NFE_REQUIRE(in && out && n > 0 && h > 0 && w > 0 && c > 0 && pad0 >= 0 && pad1 >= 0, "nfe_upfirdn2d_polyphase: bad arguments");
    const int oh = h + pad0 + pad1 - 3, ow = w + pad0 + pad1 - 3;
    NFE_REQUIRE(oh > 0 && ow > 0, "nfe_upfirdn2d_polyphase: empty output");
    if (c % 4 == 0 && n <= 65535 && (oh + 1) / 2 <= 65535) {
        const int gw2 = (ow + 1) / 2, gh2 = (oh + 1) / 2;
        hipLaunchKernelGGL(upfir_poly4_kernel, dim3((unsigned)(((long long)gw2 * (c / 4) + 255) / 256), (unsigned)gh2, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                           in, h, w, c, pad0, gain, oh, ow, out);
        NFE_CHECK_LAUNCH("upfir_poly4_kernel");
        return NFE_OK;
    }
    const long long total = (long long)n * ((oh + 1) & ~1) * ((ow + 1) & ~1) * c;
    hipLaunchKernelGGL(upfirdn_kernel<true>, dim3(grid1d(total, 256, 1 << 15)), dim3(256), 0, (hipStream_t)stream,
                       in, n, h, w, c, 1, 1, pad0, gain, oh, ow, out);
    NFE_CHECK_LAUNCH("upfirdn_kernel<polyphase>");
    return NFE_OK;
}

namespace nfe {
// Adjoint of resize_kernel (the input gradient of F.interpolate(bilinear, antialias=...)): g_in[y][x] = sum over the outputs whose
// window contains (y, x) of wy(oy, y) wx(ox, x) g_out[oy][ox], with exactly the forward's normalised weights.  Gather form: one
// thread per input element walks the few candidate outputs per axis (the window test is axis_setup's own lo / n), so no atomics
// and a fixed summation order.  The SR-head gradient at neural_rendering_resolution != 128 (sr_grad.py) is its only caller.
__device__ __forceinline__ int resize_bwd_candidates(int i, int in, int out, int aa, int& o0) {
    const float scale = (float)in / (float)out;
    const float sup = aa ? fmaxf(scale, 1.0f) : 1.0f;
    const float lo = ((float)i + 0.5f - sup) / scale - 0.5f, hi = ((float)i + 0.5f + sup) / scale - 0.5f;
    o0 = max((int)floorf(lo) - 1, 0);
    const int o1 = min((int)ceilf(hi) + 1, out - 1);
    return o1 - o0 + 1;
}
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ gout, int N, int H, int W, int C, int OH, int OW, int aa,
                                                          float* __restrict__ gin) {
    const long long total = (long long)N * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); long long r = i / C;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H); const int n = (int)(r / H);
        int oy0, ox0;
        const int ny = resize_bwd_candidates(y, H, OH, aa, oy0), nx = resize_bwd_candidates(x, W, OW, aa, ox0);
        float acc = 0.0f;
        for (int a = 0; a < ny; ++a) {
            const int oy = oy0 + a;
            const AxisW ay = axis_setup(oy, H, OH, aa);
            const int ky = y - ay.lo;
            if (ky < 0 || ky >= ay.n) continue;
            float sy = 0.0f;
            for (int k = 0; k < ay.n; ++k) sy += axis_weight(ay, k, oy, aa);
            const float wy = axis_weight(ay, ky, oy, aa) / (aa ? sy : 1.0f);
            float rowv = 0.0f;
            for (int b = 0; b < nx; ++b) {
                const int ox = ox0 + b;
                const AxisW ax = axis_setup(ox, W, OW, aa);
                const int kx = x - ax.lo;
                if (kx < 0 || kx >= ax.n) continue;
                float sx = 0.0f;
                for (int k = 0; k < ax.n; ++k) sx += axis_weight(ax, k, ox, aa);
                rowv = fmaf(axis_weight(ax, kx, ox, aa) / (aa ? sx : 1.0f), gout[(((long long)n * OH + oy) * OW + ox) * C + c], rowv);
            }
            acc = fmaf(wy, rowv, acc);
        }
        gin[i] = acc;
    }
}
}  // namespace nfe

namespace nfe {
// nfe_bias_act_backward: thread = (pixel, 4 channels); HBM-bound (reads out, grad, 3 floats of grad_rgb; writes dst)
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float4* __restrict__ out, const float4* __restrict__ grad, const float* __restrict__ grad_rgb,
                                                           const float* __restrict__ rgb_w, const float* __restrict__ rgb_s, int rgb_k,
                                                           const float* __restrict__ scale, float gain, float clamp, long long pixels, int c4, long long total,
                                                           float4* __restrict__ dst) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int q = (int)(i % c4);
        const long long p = i / c4;
        const int n = (int)(p / pixels);
        const int C = 4 * c4;
        float4 g = grad ? grad[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (grad_rgb) {
            const float4 s = *reinterpret_cast<const float4*>(rgb_s + (long long)n * C + 4 * q);
            float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int k = 0; k < rgb_k; ++k) {
                const float gy = grad_rgb[p * rgb_k + k];
                const float4 w = *reinterpret_cast<const float4*>(rgb_w + (long long)k * C + 4 * q);
                t.x = fmaf(gy, w.x, t.x); t.y = fmaf(gy, w.y, t.y); t.z = fmaf(gy, w.z, t.z); t.w = fmaf(gy, w.w, t.w);
            }
            g.x = fmaf(t.x, s.x, g.x); g.y = fmaf(t.y, s.y, g.y); g.z = fmaf(t.z, s.z, g.z); g.w = fmaf(t.w, s.w, g.w);
        }
        const float4 o = out[i];
        const float4 sc = scale ? *reinterpret_cast<const float4*>(scale + (long long)n * C + 4 * q) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        const float lim = clamp > 0.0f ? clamp : INFINITY;                 // one compare per decision (tools/lint_lane_masks.py)
        auto d = [&](float ov, float gv, float sv) {
            const float slope = ov < 0.0f ? 0.2f * gain : gain;
            const float keep = fabsf(ov) < lim ? 1.0f : 0.0f;
            return gv * slope * keep * sv;
        };
        dst[i] = make_float4(d(o.x, g.x, sc.x), d(o.y, g.y, sc.y), d(o.z, g.z, sc.z), d(o.w, g.w, sc.w));
    }
}
}  // namespace nfe

extern "C" int nfe_bias_act_backward(const float* out, const float* grad, const float* grad_rgb, const float* rgb_w, const float* rgb_s, int rgb_k,
                                     const float* scale, float gain, float clamp, int n, long long pixels, int c, float* dst, nfe_stream_t stream) {
    NFE_REQUIRE(out && dst && n > 0 && pixels > 0 && c > 0 && c % 4 == 0, "nfe_bias_act_backward: bad arguments (n=%d pixels=%lld c=%d)", n, pixels, c);
    NFE_REQUIRE(grad || grad_rgb, "nfe_bias_act_backward: no incoming gradient");
    NFE_REQUIRE(!grad_rgb || (rgb_w && rgb_s && rgb_k > 0 && rgb_k <= 4), "nfe_bias_act_backward: the ToRGB branch needs rgb_w, rgb_s and 1 <= rgb_k <= 4");
    const long long total = (long long)n * pixels * (c / 4);
    hipLaunchKernelGGL(nfe::bias_act_bwd_kernel, dim3(nfe::grid1d(total, 256, 1 << 15)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(out), reinterpret_cast<const float4*>(grad), grad_rgb, rgb_w, rgb_s, rgb_k, scale, gain, clamp,
                       pixels, c / 4, total, reinterpret_cast<float4*>(dst));
    NFE_CHECK_LAUNCH("bias_act_bwd_kernel");
    return NFE_OK;
}

extern "C" int nfe_resize_bilinear_backward(const float* grad_out, int n, int h, int w, int c, int oh, int ow, int antialias, float* grad_in,
                                            nfe_stream_t stream) {
    NFE_REQUIRE(grad_out && grad_in && n > 0 && h > 0 && w > 0 && c > 0 && oh > 0 && ow > 0, "nfe_resize_bilinear_backward: bad arguments");
    hipLaunchKernelGGL(nfe::resize_bwd_kernel, dim3(nfe::grid1d((long long)n * h * w * c, 256, 1 << 15)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, n, h, w, c, oh, ow, antialias, grad_in);
    NFE_CHECK_LAUNCH("resize_bwd_kernel");
    return NFE_OK;
}

extern "C" int nfe_resize_bilinear(const float* in, int n, int h, int w, int c, int oh, int ow, int antialias, float* out, nfe_stream_t stream) {
    NFE_REQUIRE(in && out && n > 0 && h > 0 && w > 0 && c > 0 && oh > 0 && ow > 0, "nfe_resize_bilinear: bad arguments");
    // taps per axis: ceil(2 * support) + 1 with support = max(in / out, 1) when antialiasing, 2 otherwise
    const float sup = antialias ? std::max(std::max((float)h / oh, (float)w / ow), 1.0f) : 1.0f;
    if (c % 4 == 0 && (int)(2.0f * sup) + 2 <= RS_MAXT)
        hipLaunchKernelGGL(resize4_kernel, dim3(grid1d((long long)n * oh * ow * (c / 4), 256, 1 << 15)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const float4*>(in), n, h, w, c / 4, oh, ow, antialias, reinterpret_cast<float4*>(out));
    else
        hipLaunchKernelGGL(resize_kernel, dim3(grid1d((long long)n * oh * ow * c, 256, 1 << 15)), dim3(256), 0, (hipStream_t)stream,
                           in, n, h, w, c, oh, ow, antialias, out);
    NFE_CHECK_LAUNCH("resize_kernel");
    return NFE_OK;
}

