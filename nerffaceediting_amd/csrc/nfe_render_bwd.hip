// nfe_render_backward: gradient of the renderer's final compositing pass w.r.t. the two plane sets (gfx950).
//
// What autograd does in the reference for DisentangledImportanceRenderer.forward (renderer.py:301-363) with the
// planes as leaves: backward of SegMipRayMarcher2.run_forward (ray_marcher.py:68-101), of the two decoder MLPs
// (triplane.py:249-270) and of F.grid_sample (renderer.py:64, a scatter-add).  Sample depths are constants
// (renderer.py:198,211 detach the importance depths), so everything is a function of the sorted depth buffer.
//
// Three passes, nothing per-sample wider than 3 floats is ever stored:
//   eval    one lane per sample   gather + both MLPs in fp32 -> sigma_i and a_i = <2 G_rgb, rgb_i> + <G_seg, seg_i>
//   ray     one lane per ray      forward march (T_j kept), cotangent of every weight, reverse recurrence
//                                 R_j = g_{j+1} alpha_{j+1} + (1 - alpha_{j+1} + 1e-10) R_{j+1};  dL/dalpha_j = T_j (g_j - R_j)
//                                 (no division by 1 - alpha, exact for opaque samples) -> dL/dsigma_i and
//                                 omega_i = (w_{i-1} + w_i)/2, the weight of sample i's colour in the outputs
//   scatter one lane per sample   gather + MLP forward again, MLP backward to the 32+32 feature gradients, then the
//                                 wave transposes them through LDS so that each half-wave adds one 128-byte texel
//                                 row per atomic instruction (global_atomic_add_f32, 12 taps x 2 sets per sample).
// Decoder weights are read with wave-uniform addresses (scalar loads); the gains of FullyConnectedLayer
// (networks_stylegan2.py:111-123) are applied once by prep_kernel.
#include "nfe_common.h"

#include <cstdlib>

// No implicit multiply-add fusion in this file: every fma here is written as one.  The split-bf16 MFMA stages re-quantise their inputs
// to 16 + 8 mantissa bits, which turns a one-ulp difference in an input (a product fused with the subtraction that forms its low
// half in one kernel and not in another) into 1e-5 of the output: with fusion left to the compiler the two decoder-backward kernels
// below - the same operations in the same order - differed by 4e-5 of the largest gradient, without it by the order of the
// accumulate pass's adds (6e-7; tests/test_render_backward_gpu.py).  Measured cost: none (1.77 ms either way).
#pragma clang fp contract(off)

namespace nfe {

// scaled decoder image in the workspace (floats)
constexpr int BW_G0 = 0;        // [64][32]   geo layer 0, weight * lr/sqrt(32)
constexpr int BB_G0 = 2048;     // [64]
constexpr int BW_G1T = 2112;    // [64][16]   geo layer 1 transposed (hidden-major), weight * lr/sqrt(64)
constexpr int BB_G1 = 3136;     // [16]
constexpr int BW_A0 = 3152;     // [64][32]
constexpr int BB_A0 = 5200;     // [64]
constexpr int BW_A1T = 5264;    // [64][32]
constexpr int BB_A1 = 7312;     // [32]
constexpr int BWD_DEC_FLOATS = 7344;
constexpr int BWD_DEC_BYTES = 32768;
#define BWD_WAVES 2      // waves per SIMD the sample kernels are compiled for; 3 and 4 (168 / 128 VGPRs, spills) measured slower

struct BwdK {
    const float* planes_g; const float* planes_a; long long plane_view_stride; int H, W;
    const float* aff[4];
    const float* dec;                  // scaled image above
    int N, M, R;
    const float* origins; const float* dirs; const float* cam2world; const float* intrinsics;
    int S; const float* depths; float coord_scale; int white_back;
    const float* g_rgb; const float* g_seg; const float* g_depth; const float* g_wsum; int channels_first;
    float* grad_g; float* grad_a; long long grad_view_stride;
    // per-sample records, each [view][64-ray tile][sample][lane of the tile] (bwd_slot_base, nfe_common.h): (sigma, a, T) from the
    // evaluation pass and the march, overwritten in place by (dL/dsigma, omega, -); rec_t: the depths in the same order
    float* rec_sig; float* rec_a; float* rec_T; float* rec_t; int T;      // T: ray tiles per view
    const uint4* bfrag;                // split-bf16 MFMA fragments of the decoder and its transposes (bwd_frag_kernel), or null
    // binned scatter (one chunk of views x 64-ray tiles): feature gradients, bin records and their sorted list
    float* df; uint2* rec_key; float4* rec_w; uint2* binrank; unsigned* counts; unsigned* offsets; unsigned* perm;
    int n0, t0, t_count, bins_x, bins_y;      // chunk origin (view, ray tile), ray tiles per view in the chunk, plane tiles
    unsigned* abort_word;                     // bwd_decoder_kernel: pairs that abandoned a hand-off wait; the accumulate pass writes NaN when it is not 0
};

struct PrepK { const float* w[8]; float lr_mul; float* out; };

__global__ void bwd_prep_kernel(PrepK P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BWD_DEC_FLOATS) return;
    const float g0 = P.lr_mul * 0.17677669529663687f /* 1/sqrt(32) */, g1 = P.lr_mul * 0.125f /* 1/sqrt(64) */;
    float v;
    if (i < BB_G0) v = P.w[0][i] * g0;
    else if (i < BW_G1T) v = P.w[1][i - BB_G0] * P.lr_mul;
    else if (i < BB_G1) { const int k = i - BW_G1T, j = k >> 4, o = k & 15; v = P.w[2][o * 64 + j] * g1; }
    else if (i < BW_A0) v = P.w[3][i - BB_G1] * P.lr_mul;
    else if (i < BB_A0) v = P.w[4][i - BW_A0] * g0;
    else if (i < BW_A1T) v = P.w[5][i - BB_A0] * P.lr_mul;
    else if (i < BB_A1) { const int k = i - BW_A1T, j = k >> 5, o = k & 31; v = P.w[6][o * 64 + j] * g1; }
    else v = P.w[7][i - BB_A1] * P.lr_mul;
    P.out[i] = v;
}

// ------------------------------------------------------------------------------------------------------------
// per-sample geometry and gather
// ------------------------------------------------------------------------------------------------------------
struct SampleGeo { int off[12]; float w[12]; };     // element offset of each tap's texel inside its view's plane set

__device__ __forceinline__ void ray_of(const BwdK& P, int n, int m, float (&o)[3], float (&d)[3]) {
    if (P.origins) {
        const float* po = P.origins + ((long long)n * P.M + m) * 3; const float* pd = P.dirs + ((long long)n * P.M + m) * 3;
        o[0] = po[0]; o[1] = po[1]; o[2] = po[2]; d[0] = pd[0]; d[1] = pd[1]; d[2] = pd[2];
    } else {    // RaySampler.forward, ray_sampler.py:35-61 (same arithmetic as render_kernel)
        const float* c = P.cam2world + n * 16; const float* K = P.intrinsics + n * 9;
        const float fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
        const float inv = 1.0f / (float)P.R;
        const int px = m % P.R, py = m / P.R;
        const float xc = (float)px * inv + 0.5f * inv, yc = (float)py * inv + 0.5f * inv;
        const float xl = (xc - cx + cy * sk / fy - sk * yc / fy) / fx;
        const float yl = (yc - cy) / fy;
        o[0] = c[3]; o[1] = c[7]; o[2] = c[11];
        const float wx = c[0] * xl + c[1] * yl + c[2] + c[3];
        const float wy = c[4] * xl + c[5] * yl + c[6] + c[7];
        const float wz = c[8] * xl + c[9] * yl + c[10] + c[11];
        d[0] = wx - o[0]; d[1] = wy - o[1]; d[2] = wz - o[2];
        const float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
        d[0] /= nrm; d[1] /= nrm; d[2] /= nrm;
    }
}

__device__ __forceinline__ void sample_geometry(const BwdK& P, int n, int m, float t, SampleGeo& g) {
    float o[3], d[3];
    ray_of(P, n, m, o, d);
    const float x = P.coord_scale * fmaf(t, d[0], o[0]);
    const float y = P.coord_scale * fmaf(t, d[1], o[1]);
    const float z = P.coord_scale * fmaf(t, d[2], o[2]);
    const int plane_elems = P.H * P.W * 32;
#pragma unroll
    for (int p = 0; p < 3; ++p) {     // plane axes of generate_planes (renderer.py:23-37): (x,y), (x,z), (z,x)
        const Taps tp = tap_geometry(P.H, P.W, p == 2 ? z : x, p == 0 ? y : (p == 1 ? z : x));
        g.off[4 * p + 0] = p * plane_elems + (tp.yc0 * P.W + tp.xc0) * 32;
        g.off[4 * p + 1] = p * plane_elems + (tp.yc0 * P.W + tp.xc1) * 32;
        g.off[4 * p + 2] = p * plane_elems + (tp.yc1 * P.W + tp.xc0) * 32;
        g.off[4 * p + 3] = p * plane_elems + (tp.yc1 * P.W + tp.xc1) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) g.w[4 * p + k] = tp.w[k];
    }
}

// Feature vector of one plane set: mean over planes of (bilinear sample * scale + in-bounds weight * shift)
// (scale/shift NULL = identity).  aff_* point at this view's [96] rows.  Channels are kept
// as pairs: the decoder below runs on v_pk_fma_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat(float x) { return f32x2{x, x}; }

__device__ __forceinline__ void gather_set(const float* __restrict__ planes, const SampleGeo& g,
                                           const float* __restrict__ scale, const float* __restrict__ shift, f32x2 (&f)[16]) {
#pragma unroll
    for (int c = 0; c < 16; ++c) f[c] = splat(0.0f);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        f32x2 s[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) s[c] = splat(0.0f);
        float wsum = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4* tx = reinterpret_cast<const float4*>(planes + g.off[4 * p + k]);
            const f32x2 w = splat(g.w[4 * p + k]);
            wsum += g.w[4 * p + k];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = tx[q];
                s[2 * q] = pk_fma(w, f32x2{v.x, v.y}, s[2 * q]);
                s[2 * q + 1] = pk_fma(w, f32x2{v.z, v.w}, s[2 * q + 1]);
            }
        }
        if (scale) {
            const f32x2* sc = reinterpret_cast<const f32x2*>(scale + p * 32);
            const f32x2* sh = reinterpret_cast<const f32x2*>(shift + p * 32);
#pragma unroll
            for (int c = 0; c < 16; ++c) f[c] += pk_fma(s[c], sc[c], splat(wsum) * sh[c]);
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) f[c] += s[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) f[c] *= splat(1.0f / 3.0f);
}

// ------------------------------------------------------------------------------------------------------------
// decoder heads in fp32, one sample per lane, weights at wave-uniform addresses (scalar loads, SGPR operands)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_t(float x) {      // torch.nn.Softplus(beta=1, threshold=20)
    const float r = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(x * LOG2E)) * LN2;
    return x > 20.0f ? x : r;
}
__device__ __forceinline__ float sigmoid_t(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * LOG2E)); }

__device__ __forceinline__ float hidden_pre(const float* __restrict__ w0, const float* __restrict__ b0, int j, const f32x2 (&f)[16]) {
    const f32x2* w = reinterpret_cast<const f32x2*>(w0 + j * 32);
    f32x2 acc = f32x2{b0[j], 0.0f};
#pragma unroll
    for (int c = 0; c < 16; ++c) acc = pk_fma(w[c], f[c], acc);
    return acc[0] + acc[1];
}

template <int NO>
__device__ __forceinline__ void head_forward(const float* __restrict__ w0, const float* __restrict__ b0,
                                             const float* __restrict__ w1t, const float* __restrict__ b1,
                                             const f32x2 (&f)[16], f32x2 (&out)[NO / 2]) {
#pragma unroll
    for (int o = 0; o < NO / 2; ++o) out[o] = f32x2{b1[2 * o], b1[2 * o + 1]};
#pragma unroll 2
    for (int j = 0; j < 64; ++j) {
        const f32x2 h = splat(softplus_t(hidden_pre(w0, b0, j, f)));
        const f32x2* w = reinterpret_cast<const f32x2*>(w1t + j * NO);
#pragma unroll
        for (int o = 0; o < NO / 2; ++o) out[o] = pk_fma(w[o], h, out[o]);
    }
}

// df = W0^T (softplus'(pre) * (W1^T dout)); the pre-activations are recomputed (cheaper than 64 live registers)
template <int NO>
__device__ __forceinline__ void head_backward(const float* __restrict__ w0, const float* __restrict__ b0,
                                              const float* __restrict__ w1t, const f32x2 (&f)[16],
                                              const f32x2 (&dout)[NO / 2], f32x2 (&df)[16]) {
#pragma unroll
    for (int c = 0; c < 16; ++c) df[c] = splat(0.0f);
#pragma unroll 2
    for (int j = 0; j < 64; ++j) {
        const float pre = hidden_pre(w0, b0, j, f);
        const f32x2* w1 = reinterpret_cast<const f32x2*>(w1t + j * NO);
        f32x2 dh = splat(0.0f);
#pragma unroll
        for (int o = 0; o < NO / 2; ++o) dh = pk_fma(w1[o], dout[o], dh);
        const f32x2 dpre = splat((dh[0] + dh[1]) * (pre > 20.0f ? 1.0f : sigmoid_t(pre)));
        const f32x2* w = reinterpret_cast<const f32x2*>(w0 + j * 32);
#pragma unroll
        for (int c = 0; c < 16; ++c) df[c] = pk_fma(w[c], dpre, df[c]);
    }
}

// (Round 4, profiles/experiments/r04_bwd_ab1.txt: the decoder-backward kernel without these loads is 0.34 ms faster per 4 views - they are
// ~500 of its ~4 700 cache-line requests per wave on a texture addresser that is 0.67 busy; issuing them before the gather instead of
// inside the MFMA chains made it 0.06 ms SLOWER, so it is their number, not their latency.)
__device__ __forceinline__ float cot_rgb(const BwdK& P, int n, int m, int c) {      // includes the *2 of rgb*2-1
    if (!P.g_rgb) return 0.0f;
    return 2.0f * (P.channels_first ? P.g_rgb[((long long)n * 32 + c) * P.M + m] : P.g_rgb[((long long)n * P.M + m) * 32 + c]);
}
__device__ __forceinline__ float cot_seg(const BwdK& P, int n, int m, int c) {
    if (!P.g_seg) return 0.0f;
    return P.channels_first ? P.g_seg[((long long)n * 15 + c) * P.M + m] : P.g_seg[((long long)n * P.M + m) * 15 + c];
}

// ------------------------------------------------------------------------------------------------------------
// pass 1: sigma_i and a_i
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, BWD_WAVES) void bwd_eval_kernel(BwdK P) {
    const int n = blockIdx.y;
    const long long per_view = (long long)P.M * P.S;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_view) return;
    const int m = (int)(i / P.S);
    const long long g = (long long)n * per_view + i;
    SampleGeo geo;
    sample_geometry(P, n, m, P.depths[g], geo);
    const long long pv = (long long)n * P.plane_view_stride;
    const float* dec = P.dec;
    f32x2 f[16];
    gather_set(P.planes_g + pv, geo, P.aff[0] ? P.aff[0] + n * 96 : nullptr, P.aff[1] ? P.aff[1] + n * 96 : nullptr, f);
    f32x2 og[8];
    head_forward<16>(dec + BW_G0, dec + BB_G0, dec + BW_G1T, dec + BB_G1, f, og);
    float a = 0.0f;
    if (P.g_seg) {
#pragma unroll
        for (int c = 0; c < 15; ++c) a = fmaf(cot_seg(P, n, m, c), og[(1 + c) >> 1][(1 + c) & 1], a);
    }
    const float sigma = og[0][0];
    if (P.g_rgb) {
        gather_set(P.planes_a + pv, geo, P.aff[2] ? P.aff[2] + n * 96 : nullptr, P.aff[3] ? P.aff[3] + n * 96 : nullptr, f);
        f32x2 oa[16];
        head_forward<32>(dec + BW_A0, dec + BB_A0, dec + BW_A1T, dec + BB_A1, f, oa);
#pragma unroll
        for (int c = 0; c < 32; ++c) a = fmaf(cot_rgb(P, n, m, c), sigmoid_t(oa[c >> 1][c & 1]) * 1.002f - 0.001f, a);    // triplane.py:269
    }
    const long long e = bwd_slot_base(P.R, P.M, P.S, n, m) + 64ll * (i - (long long)m * P.S);
    P.rec_sig[e] = sigma;
    P.rec_a[e] = a;
}

// ------------------------------------------------------------------------------------------------------------
// pass 2: the march and its reverse, one lane per ray
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void segment(float s0, float s1, float t0, float t1, float& alpha, float& dalpha_dsmid) {
    const float x = (s0 + s1) * 0.5f - 1.0f;                        // ray_marcher.py:72,76
    const float dens = softplus_t(x);
    const float dlt = t1 - t0;
    const float e = __builtin_amdgcn_exp2f(-dens * dlt * LOG2E);
    alpha = 1.0f - e;                                               // :80-82
    dalpha_dsmid = dlt * e * (x > 20.0f ? 1.0f : sigmoid_t(x));
}

// Ray m of lane `lane` of 64-ray tile t: an 8 x 8 pixel tile of a square image whose side is a multiple of 8, else 64 consecutive
// rays (dead lanes of the last tile repeat the view's last ray).  bwd_ray_kernel and bwd_scatter_sorted_kernel share it.
__device__ __forceinline__ int tile_ray(const BwdK& P, int t, int lane, bool& live) {
    live = true;
    if (P.R > 0 && (P.R & 7) == 0 && (long long)P.R * P.R == P.M) {
        const int tiles_x = P.R >> 3;
        return ((t / tiles_x) * 8 + (lane >> 3)) * P.R + (t % tiles_x) * 8 + (lane & 7);
    }
    const int m = t * 64 + lane;
    live = m < P.M;
    return min(m, P.M - 1);
}

constexpr int RAY_CH = 16;     // samples per batch of bwd_ray_kernel: their loads are issued together, one memory round trip per batch

__global__ __launch_bounds__(256) void bwd_ray_kernel(BwdK P) {
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);          // = (view, 64-ray tile)
    if (wave >= (long long)P.N * P.T) return;
    const int lane = threadIdx.x & 63;
    bool live;
    const int m_of_lane = tile_ray(P, (int)(wave % P.T), lane, live);
    if (!live) return;
    const long long ray = (wave / P.T) * P.M + m_of_lane;
    const int S = P.S;
    const float* __restrict__ t = P.depths + ray * S;
    // the records of this wave's tile: sample i of this lane is element 64 * i (256-byte rows per sample: coalesced)
    const size_t tb = (size_t)wave * S * 64 + lane;
    float* sig = P.rec_sig + tb; float* av = P.rec_a + tb; float* __restrict__ Tv = P.rec_T + tb; float* __restrict__ o_t = P.rec_t + tb;
    // forward: transmittance of every segment, sum of weights, weighted depth.  The depths come in [ray][sample] order (a lane's
    // loads are its own cache lines): the loop is batched by hand - a store per step kept the compiler from hoisting the next
    // step's loads, and the kernel spent one memory latency per sample.
    float T = 1.0f, wtot = 0.0f, dnum = 0.0f;
    float s0 = sig[0], t0 = t[0];
    o_t[0] = t0;
    for (int j0 = 0; j0 + 1 < S; j0 += RAY_CH) {
        float sc[RAY_CH], tc[RAY_CH], To[RAY_CH];
#pragma unroll
        for (int u = 0; u < RAY_CH; ++u) { const int i = min(j0 + 1 + u, S - 1); sc[u] = sig[(size_t)i * 64]; tc[u] = t[i]; }
#pragma unroll
        for (int u = 0; u < RAY_CH; ++u) {
            To[u] = T;
            if (j0 + 1 + u < S) {
                float alpha, dummy;
                segment(s0, sc[u], t0, tc[u], alpha, dummy);
                const float w = alpha * T;
                wtot += w; dnum = fmaf(w, (t0 + tc[u]) * 0.5f, dnum);
                T *= (1.0f - alpha + 1e-10f);                       // :85
                s0 = sc[u]; t0 = tc[u];
            }
        }
#pragma unroll
        for (int u = 0; u < RAY_CH; ++u) if (j0 + 1 + u < S) { Tv[(size_t)(j0 + u) * 64] = To[u]; o_t[(size_t)(j0 + 1 + u) * 64] = tc[u]; }
    }
    const float d0 = dnum / wtot;
    const bool ok = wtot != 0.0f && isfinite(d0);                   // nan_to_num + clamp (:93-94) pass nothing otherwise
    const float gd = (ok && P.g_depth) ? P.g_depth[ray] / wtot : 0.0f;
    float gconst = P.g_wsum ? P.g_wsum[ray] : 0.0f;
    if (P.white_back && P.g_rgb) {                                  // rgb + 1 - weight_total (:96-97)
        const int n = (int)(ray / P.M), m = (int)(ray % P.M);
        float s = 0.0f;
        for (int c = 0; c < 32; ++c) s += cot_rgb(P, n, m, c);
        gconst -= s;
    }
    // reverse: R_j = sum_{k>j} g_k alpha_k prod_{j<m<k} (1 - alpha_m + 1e-10).  Batches again: the loads of samples jh-7 .. jh first,
    // then the recurrence, then the stores (to samples jh-6 .. jh+1: never a sample a later batch still has to read)
    float R = 0.0f;
    float s1 = sig[(size_t)(S - 1) * 64], a1 = av[(size_t)(S - 1) * 64], t1 = t[S - 1];
    float gs_hi = 0.0f, om_hi = 0.0f;          // contributions of segment j to sample j+1
    for (int jh = S - 2; jh >= 0; jh -= RAY_CH) {
        float sc[RAY_CH], ac[RAY_CH], tc[RAY_CH], Tc[RAY_CH], gso[RAY_CH], omo[RAY_CH];
#pragma unroll
        for (int u = 0; u < RAY_CH; ++u) { const int i = max(jh - u, 0); sc[u] = sig[(size_t)i * 64]; ac[u] = av[(size_t)i * 64]; tc[u] = o_t[(size_t)i * 64]; Tc[u] = Tv[(size_t)i * 64]; }
#pragma unroll
        for (int u = 0; u < RAY_CH; ++u) {
            gso[u] = 0.0f; omo[u] = 0.0f;
            if (jh - u >= 0) {
                float alpha, dads;
                segment(sc[u], s1, tc[u], t1, alpha, dads);
                const float gw = 0.5f * (ac[u] + a1) + gconst + (ok ? gd * ((tc[u] + t1) * 0.5f - d0) : 0.0f);
                const float galpha = Tc[u] * (gw - R);
                R = fmaf(gw, alpha, (1.0f - alpha + 1e-10f) * R);
                const float gs = 0.5f * galpha * dads, om = 0.5f * alpha * Tc[u];
                gso[u] = gs + gs_hi; omo[u] = om + om_hi;           // sample j+1 is complete: segments j and j+1 seen
                gs_hi = gs; om_hi = om;
                s1 = sc[u]; a1 = ac[u]; t1 = tc[u];
            }
        }
#pragma unroll
        for (int u = 0; u < RAY_CH; ++u) if (jh - u >= 0) { sig[(size_t)(jh - u + 1) * 64] = gso[u]; av[(size_t)(jh - u + 1) * 64] = omo[u]; }
    }
    sig[0] = gs_hi; av[0] = om_hi;
}

// ------------------------------------------------------------------------------------------------------------
// pass 3: decoder backward and scatter
// ------------------------------------------------------------------------------------------------------------
constexpr int TILE_STRIDE = 33;                 // floats per sample row in the transpose tile (conflict-free both ways)
constexpr int SCATTER_LDS_FLOATS = 4 * 64 * TILE_STRIDE;

// Adds df (this lane's sample, 32 channels, already / 3) into the gradient plane set: half-wave h handles sample
// 2*it + h of the wave, lane&31 = channel, one 128-byte texel row per half-wave per atomic instruction.
__device__ __forceinline__ void scatter_set(float* __restrict__ tile, const f32x2 (&df)[16], const SampleGeo& geo, bool live,
                                            float* __restrict__ grad, const float* __restrict__ scale, int lane) {
#pragma unroll
    for (int c = 0; c < 32; ++c) tile[lane * TILE_STRIDE + c] = df[c >> 1][c & 1] * (1.0f / 3.0f);   // mean over planes, triplane.py:251
    __builtin_amdgcn_wave_barrier();
    const int ch = lane & 31, hh = lane >> 5;
    float sc[3] = {1.0f, 1.0f, 1.0f};
    if (scale) { sc[0] = scale[ch]; sc[1] = scale[32 + ch]; sc[2] = scale[64 + ch]; }
#pragma unroll 1
    for (int it = 0; it < 32; ++it) {
        const int src = 2 * it + hh;
        const float v = tile[src * TILE_STRIDE + ch];
        const int alive = __shfl((int)live, src);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int off = __shfl(geo.off[k], src);
            const float w = __shfl(geo.w[k], src);
            if ((alive ? w : 0.0f) != 0.0f) unsafeAtomicAdd(grad + off + ch, v * w * sc[k >> 2]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256, BWD_WAVES) void bwd_scatter_kernel(BwdK P) {
    __shared__ float tiles[SCATTER_LDS_FLOATS];
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63;
    float* tile = tiles + (threadIdx.x >> 6) * 64 * TILE_STRIDE;
    const long long per_view = (long long)P.M * P.S;
    const long long i_raw = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i_raw < per_view;
    const long long i = live ? i_raw : per_view - 1;              // dead lanes recompute the last sample, add nothing
    const int m = (int)(i / P.S);
    const long long g = (long long)n * per_view + i;
    SampleGeo geo;
    sample_geometry(P, n, m, P.depths[g], geo);
    const long long e = bwd_slot_base(P.R, P.M, P.S, n, m) + 64ll * (i - (long long)m * P.S);
    const float gsig = P.rec_sig[e], omega = P.rec_a[e];
    const long long pv = (long long)n * P.plane_view_stride;
    const long long gv = (long long)n * P.grad_view_stride;
    const float* dec = P.dec;
    f32x2 f[16], df[16];
    if (P.grad_g) {
        const float* scale = P.aff[0] ? P.aff[0] + n * 96 : nullptr;
        gather_set(P.planes_g + pv, geo, scale, P.aff[1] ? P.aff[1] + n * 96 : nullptr, f);
        f32x2 dout[8];
        dout[0][0] = gsig;                                         // sigma = channel 0, seg = 1..15 (triplane.py:260-261)
#pragma unroll
        for (int c = 0; c < 15; ++c) dout[(1 + c) >> 1][(1 + c) & 1] = omega * cot_seg(P, n, m, c);
        head_backward<16>(dec + BW_G0, dec + BB_G0, dec + BW_G1T, f, dout, df);
        scatter_set(tile, df, geo, live, P.grad_g + gv, scale, lane);
    }
    if (P.grad_a && P.g_rgb) {
        const float* scale = P.aff[2] ? P.aff[2] + n * 96 : nullptr;
        gather_set(P.planes_a + pv, geo, scale, P.aff[3] ? P.aff[3] + n * 96 : nullptr, f);
        f32x2 y[16];
        head_forward<32>(dec + BW_A0, dec + BB_A0, dec + BW_A1T, dec + BB_A1, f, y);
#pragma unroll
        for (int c = 0; c < 32; ++c) {                             // rgb = sigmoid(y) * 1.002 - 0.001 (triplane.py:269)
            const float s = sigmoid_t(y[c >> 1][c & 1]);
            y[c >> 1][c & 1] = omega * cot_rgb(P, n, m, c) * 1.002f * s * (1.0f - s);
        }
        head_backward<32>(dec + BW_A0, dec + BB_A0, dec + BW_A1T, f, y, df);
        scatter_set(tile, df, geo, live, P.grad_a + gv, scale, lane);
    }
}


// ------------------------------------------------------------------------------------------------------------
// Decoder forward + backward of the scatter pass on MFMA (v_mfma_f32_32x32x16_bf16, split-bf16: hi*hi + hi*lo + lo*hi,
// fp32-grade like the forward kernel).  The wave's 64 samples are two N-blocks of 32; lane = sample on the way
// in (that is how the gather leaves the features), lane (j, h) = (sample 32b + j, k-half h) inside the GEMMs.
//   F0  pre = W0 f + b0            A = W0 (2 M-blocks, K = 32 channels: 2 k-steps)            both heads
//   F1  y   = W1 softplus(pre) + b1 (appearance head only: its forward output enters the sigmoid derivative)
//   DH  dh  = W1^T dout            A = W1^T (rows = hidden units in pre's accumulator layout)
//       dpre = dh * sigmoid(pre)   elementwise, accumulator layout
//   DF  df  = W0^T dpre            A = W0^T (rows permuted so that register r of lane (j,h) is channel 16h + r)
// B operands never move between lanes after the first step: an accumulator IS the next B operand once the K order of the next
// A fragments is chosen to match ("hidden unit held by (M-block, register, half)" = "k index the lane supplies"), as in the
// forward kernel.  Only F0's B operand has to be made from the lane = sample layout: channels 16s..16s+7 (Y) and 16s+8..16s+15
// (X) of a k-step are swapped between the lane halves by v_permlane32_swap: Y becomes N-block 0's operand, X N-block 1's.
// Fragment image (bwd_frag_kernel, 52 fragments of 64 lanes x 8 bf16, hi and lo parts):
// ------------------------------------------------------------------------------------------------------------
constexpr int BF_F0 = 0;      // + ((head*2 + mblk)*2 + s)*2 + part
constexpr int BF_F1A = 16;    // + s*2 + part
constexpr int BF_DHG = 24;    // + mblk*2 + part
constexpr int BF_DHA = 28;    // + (mblk*2 + s)*2 + part
constexpr int BF_DF = 36;     // + (head*4 + s)*2 + part
constexpr int BF_COUNT = 52;
constexpr int BWD_FRAG_BYTES = BF_COUNT * 64 * 16;

typedef __bf16 bwd_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bwd_bf16x2 __attribute__((ext_vector_type(2)));
union BFrag { bwd_bf16x8 v; uint4 q; unsigned u[4]; };

__device__ __forceinline__ int bf_hidden_of(int s, int h, int e) { const int r = 8 * (s & 1) + e; return 32 * (s >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ int bf_channel_of_row(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }

__global__ void bwd_frag_kernel(const float* __restrict__ dec, unsigned* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= BF_COUNT * 64 * 4) return;
    const int w = idx & 3, lane = (idx >> 2) & 63, frag = idx >> 8;
    const int i = lane & 31, h = lane >> 5, part = frag & 1;
    unsigned bits[2];
    for (int k = 0; k < 2; ++k) {
        const int e = 2 * w + k;
        float v;
        if (frag < BF_F1A) {
            const int q = frag >> 1, s = q & 1, mblk = (q >> 1) & 1, head = q >> 2;
            v = dec[(head ? BW_A0 : BW_G0) + (32 * mblk + i) * 32 + 16 * s + 8 * h + e];
        } else if (frag < BF_DHG) {
            const int s = (frag - BF_F1A) >> 1;
            v = dec[BW_A1T + bf_hidden_of(s, h, e) * 32 + bf_channel_of_row(i)];
        } else if (frag < BF_DHA) {
            const int mblk = (frag - BF_DHG) >> 1;
            v = dec[BW_G1T + (32 * mblk + i) * 16 + 8 * h + e];
        } else if (frag < BF_DF) {
            const int q = (frag - BF_DHA) >> 1, s = q & 1, mblk = q >> 1;
            v = dec[BW_A1T + (32 * mblk + i) * 32 + 16 * h + 8 * s + e];
        } else {
            const int q = (frag - BF_DF) >> 1, s = q & 3, head = q >> 2;
            v = dec[(head ? BW_A0 : BW_G0) + bf_hidden_of(s, h, e) * 32 + bf_channel_of_row(i)];
        }
        bwd_bf16x2 ph = {(__bf16)v, (__bf16)0.0f};
        const unsigned hi = *reinterpret_cast<unsigned*>(&ph) & 0xffffu;
        bwd_bf16x2 pl = {(__bf16)(v - __uint_as_float(hi << 16)), (__bf16)0.0f};
        bits[k] = part == 0 ? hi : (*reinterpret_cast<unsigned*>(&pl) & 0xffffu);
    }
    out[idx] = bits[0] | (bits[1] << 16);
}

// fp32 pair -> packed bf16 hi word and lo word (hi = top 16 bits, lo = bf16(x - hi))
__device__ __forceinline__ void bsplit(float a, float b, unsigned& hi, unsigned& lo) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    bwd_bf16x2 p = {(__bf16)(a - __uint_as_float(ua & 0xffff0000u)), (__bf16)(b - __uint_as_float(ub & 0xffff0000u))};
    lo = *reinterpret_cast<unsigned*>(&p);
}
__device__ __forceinline__ f32x16 mfma3(const BFrag& ah, const BFrag& al, const BFrag& bh, const BFrag& bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bh.v, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bl.v, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bh.v, c, 0, 0, 0);
    return c;
}
// accumulator registers 8*(s&1) .. +7 -> B operand (hi, lo)
__device__ __forceinline__ void acc_operand(const f32x16& a, int s1, BFrag& bh, BFrag& bl) {
#pragma unroll
    for (int w = 0; w < 4; ++w) bsplit(a[8 * s1 + 2 * w], a[8 * s1 + 2 * w + 1], bh.u[w], bl.u[w]);
}

// One head for both N-blocks.  f: this lane's sample, 32 channels (lane = sample).  dout_of(b, r, acc y or nullptr) supplies the
// cotangent of the head's outputs in B-operand order.  Result: tile[(32b + j) * stride + col0 + 16h + r] = df / 3.
// Where the fragments and the biases come from: global memory (one wave per workgroup: every wave fetches all it uses, 178 KB per
// 64 samples) or the workgroup's LDS image (bwd_decoder_kernel).  frag(i): this lane's 16 bytes of fragment i; bias0(head, k) /
// bias1(k): four consecutive biases of layer 0 / of the appearance head's layer 1.
struct FragGlobal {
    const uint4* F; const float* dec;                 // F = fragments + lane
    __device__ __forceinline__ uint4 frag(int i) const { return F[i * 64]; }
    __device__ __forceinline__ float4 bias0(int head, int k) const { return *reinterpret_cast<const float4*>(dec + (head ? BB_A0 : BB_G0) + k); }
    __device__ __forceinline__ float4 bias1(int k) const { return *reinterpret_cast<const float4*>(dec + BB_A1 + k); }
    __device__ __forceinline__ void launder() {
        unsigned long long fp = reinterpret_cast<unsigned long long>(F);
        asm volatile("; nfe_launder %0" : "+v"(fp));
        F = reinterpret_cast<const uint4*>(fp);
    }
};
constexpr int BWD_LDS_BIAS = BWD_FRAG_BYTES;          // 160 floats: geometry layer 0 [64], appearance layer 0 [64], appearance layer 1 [32]
constexpr int BWD_LDS_TILES = BWD_LDS_BIAS + 1024;
struct FragLds {
    unsigned off;                                     // byte offset of this lane's 16 bytes inside a fragment
    __device__ __forceinline__ uint4 frag(int i) const {
        extern __shared__ __attribute__((aligned(16))) unsigned char nfe_bwd_lds[];
        return *reinterpret_cast<const uint4*>(nfe_bwd_lds + off + i * 1024);
    }
    __device__ __forceinline__ float4 bias0(int head, int k) const {
        extern __shared__ __attribute__((aligned(16))) unsigned char nfe_bwd_lds[];
        return *reinterpret_cast<const float4*>(nfe_bwd_lds + BWD_LDS_BIAS + (head * 64 + k) * 4);
    }
    __device__ __forceinline__ float4 bias1(int k) const {
        extern __shared__ __attribute__((aligned(16))) unsigned char nfe_bwd_lds[];
        return *reinterpret_cast<const float4*>(nfe_bwd_lds + BWD_LDS_BIAS + (128 + k) * 4);
    }
    __device__ __forceinline__ void launder() { asm volatile("; nfe_launder %0" : "+v"(off)); }
};

template <bool APP, typename Frag, typename DoutGeo, typename DoutApp>
__device__ __forceinline__ void head_mfma(Frag F, const f32x2 (&f)[16], int lane,
                                          float* __restrict__ tile, int stride, int col0, DoutGeo dout_geo, DoutApp dout_app) {
    const int j = lane & 31, h = lane >> 5;
    const int head = APP ? 1 : 0;
    // ---- B operands of F0 for both N-blocks: word w of k-step s holds channels 16s + 2w, +1 (Y) / 16s + 8 + 2w, +1 (X)
    BFrag bh[2][2], bl[2][2];                       // [block][k-step]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned yh, yl, xh, xl;
            bsplit(f[8 * s + w][0], f[8 * s + w][1], yh, yl);
            bsplit(f[8 * s + 4 + w][0], f[8 * s + 4 + w][1], xh, xl);
            auto sh = __builtin_amdgcn_permlane32_swap(yh, xh, false, false);
            auto sl = __builtin_amdgcn_permlane32_swap(yl, xl, false, false);
            bh[0][s].u[w] = sh[0]; bh[1][s].u[w] = sh[1];
            bl[0][s].u[w] = sl[0]; bl[1][s].u[w] = sl[1];
        }
#pragma unroll 1
    for (int b = 0; b < 2; ++b) {
        F.launder();          // the fragment loads are loop invariant: hoisted, all 52 of them would sit in registers (208 VGPRs) for the whole kernel
        // ---- F0: pre-activations of the 64 hidden units (2 M-blocks), bias first
        f32x16 pre[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = F.bias0(head, 32 * mb + 8 * q + 4 * h);
                pre[mb][4 * q] = bb.x; pre[mb][4 * q + 1] = bb.y; pre[mb][4 * q + 2] = bb.z; pre[mb][4 * q + 3] = bb.w;
            }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                BFrag ah, al;
                ah.q = F.frag(BF_F0 + ((head * 2 + mb) * 2 + s) * 2 + 0); al.q = F.frag(BF_F0 + ((head * 2 + mb) * 2 + s) * 2 + 1);
                pre[mb] = mfma3(ah, al, b == 0 ? bh[0][s] : bh[1][s], b == 0 ? bl[0][s] : bl[1][s], pre[mb]);
            }
        f32x16 dh[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dh[mb][r] = 0.0f;
        if (APP) {
            // ---- F1: y = W1 softplus(pre) + b1, register r = channel 16h + r
            f32x16 y;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = F.bias1(16 * h + 4 * q);
                y[4 * q] = bb.x; y[4 * q + 1] = bb.y; y[4 * q + 2] = bb.z; y[4 * q + 3] = bb.w;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x16 hv;
#pragma unroll
                for (int r = 0; r < 8; ++r) hv[8 * (s & 1) + r] = softplus_t(pre[s >> 1][8 * (s & 1) + r]);
                BFrag hh, hl, ah, al;
                acc_operand(hv, s & 1, hh, hl);
                ah.q = F.frag(BF_F1A + s * 2 + 0); al.q = F.frag(BF_F1A + s * 2 + 1);
                y = mfma3(ah, al, hh, hl, y);
            }
            dout_app(b, y);                                       // y[r] <- cotangent of app output channel 16h + r
            // ---- DH: dh = W1^T dout, K = 32 outputs: k-step s = registers 8s..8s+7
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                BFrag dhh, dhl;
                acc_operand(y, s, dhh, dhl);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    BFrag ah, al;
                    ah.q = F.frag(BF_DHA + (mb * 2 + s) * 2 + 0); al.q = F.frag(BF_DHA + (mb * 2 + s) * 2 + 1);
                    dh[mb] = mfma3(ah, al, dhh, dhl, dh[mb]);
                }
            }
        } else {
            // ---- DH, geometry head: K = 16 outputs (sigma, 15 seg): the lane supplies outputs 8h .. 8h+7
            float d[8];
            dout_geo(b, d);
            BFrag dhh, dhl;
#pragma unroll
            for (int w = 0; w < 4; ++w) bsplit(d[2 * w], d[2 * w + 1], dhh.u[w], dhl.u[w]);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                BFrag ah, al;
                ah.q = F.frag(BF_DHG + mb * 2 + 0); al.q = F.frag(BF_DHG + mb * 2 + 1);
                dh[mb] = mfma3(ah, al, dhh, dhl, dh[mb]);
            }
        }
        // ---- dpre = dh * softplus'(pre), then DF: df = W0^T dpre (K = 64 hidden units: 4 k-steps)
        f32x16 df;
#pragma unroll
        for (int r = 0; r < 16; ++r) df[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x16 dp;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float p = pre[s >> 1][8 * (s & 1) + r];
                dp[8 * (s & 1) + r] = dh[s >> 1][8 * (s & 1) + r] * (p > 20.0f ? 1.0f : sigmoid_t(p));
            }
            BFrag ph, pl, ah, al;
            acc_operand(dp, s & 1, ph, pl);
            ah.q = F.frag(BF_DF + (head * 4 + s) * 2 + 0); al.q = F.frag(BF_DF + (head * 4 + s) * 2 + 1);
            df = mfma3(ah, al, ph, pl, df);
        }
        float* row = tile + (32 * b + j) * stride + col0 + 16 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) row[r] = df[r] * (1.0f / 3.0f);          // mean over planes, triplane.py:251
    }
}

// ------------------------------------------------------------------------------------------------------------
// pass 3, sorted form: one wave = 8x8 neighbouring rays at ONE depth index.  Their 64 x 4 taps on a plane fall on
// ~75 distinct texels (3.4x fewer than taps at 128^2 rays on 256^2 planes), so the wave sorts its 256 (texel, sample,
// tap) keys per plane (bitonic network in registers: 4 keys per lane, lane exchanges by ds_bpermute), walks the sorted list summing weight * feature-gradient over each run of equal
// texels, and issues ONE atomic row per run and plane set (lanes 0..31: geometry set, 32..63: appearance set).
// The atomic unit - the bound of the direct form - sees 3.4x fewer operations.
// ------------------------------------------------------------------------------------------------------------
constexpr int SORT_TILE_STRIDE = 68;      // floats per sample row: rows stay 16-byte aligned (the gather exchange uses ds_*_b128)
constexpr unsigned KEY_INVALID = 0xFFFFFFFFu;
constexpr int BIN_SHIFT = 3, BIN_MASK = 7, BIN_TEXELS = 9;       // plane tiles of the binned form: 8 x 8 texels + the far taps' row/column
#define NFE_BIN_SEGMENT 4096
#define NFE_BWD_COT_STAGED 1       // decoder-backward kernel: output cotangents fetched coalesced and handed over through the tile (0: per-lane loads)
#define NFE_BWD_DEPTH_FAST 0       // A/B: block order of the decoder-backward launch (1 = the depths of one ray tile are neighbours)
constexpr int BIN_SEGMENT = NFE_BIN_SEGMENT;                                // records per workgroup before a bin is split (at most BIN_SPLIT ways)
constexpr int BIN_SPLIT = 4;
constexpr int BIN_BATCH = 16;                                    // records (256-byte row loads) a wave keeps in flight
constexpr uint64_t BWD_CHUNK_SAMPLES = 1ull << 23;               // 2 GiB of feature gradients per chunk
constexpr uint64_t BWD_MAX_BINS = 1ull << 19;

// Both plane sets of the wave's 64 samples, gathered with EIGHT LANES PER TEXEL ROW (lane = (sample within a group of 8, 4
// channels)): a 16-byte load instruction covers 8 rows of 128 contiguous bytes instead of 64 rows, which is what the texture
// path is priced by (the lane = sample form of gather_set spends more than half of this kernel there).  Tap offsets and weights
// live in the lane = sample layout and are fetched by ds_bpermute.  Results go through the tile: row = sample, columns 0..31 the
// geometry set, 32..63 the appearance set, already scaled by 1/3; same operation order per channel as gather_set.
// "pointer is not null" as an integer in an SGPR, opaque to the optimiser: branches on it are s_cmp + s_cbranch_scc instead of the
// lane-mask form (s_and_b64 vcc, exec, mask; s_cbranch_vccz).  Written in rounds 2-3 as a guard against the run-dependent results
// of this kernel; their cause turned out to be a packed-fp32 operand form (profiles/experiments/r04_pk_opsel_hazard.md, removed by
// the build's assembly pass), not lane masks.  Kept: it costs nothing.
__device__ __forceinline__ int sgpr_nonnull(const void* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    int f;
    asm volatile("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(f) : "s"(v) : "scc");
    return f;
}

template <bool DO_G, bool DO_A>
__device__ __forceinline__ void gather_pair_coop(const BwdK& P, int n, const SampleGeo& geo, int lane, float* tile) {
    const int s8 = lane >> 3, c4 = (lane & 7) * 4;
    const long long pv = (long long)n * P.plane_view_stride;
    const float* pg = P.planes_g + pv + c4;
    const float* pa = P.planes_a + pv + c4;
    float4 sc[2][3], sh[2][3];
#pragma unroll
    for (int set = 0; set < 2; ++set)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            sc[set][p] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            sh[set][p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
    for (int set = 0; set < 2; ++set)
        if (sgpr_nonnull(P.aff[2 * set]) != 0) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                sc[set][p] = *reinterpret_cast<const float4*>(P.aff[2 * set] + n * 96 + p * 32 + c4);
                sh[set][p] = *reinterpret_cast<const float4*>(P.aff[2 * set + 1] + n * 96 + p * 32 + c4);
            }
        }
    auto fma4 = [](float w, const float4& v, const float4& a) { return make_float4(fmaf(w, v.x, a.x), fmaf(w, v.y, a.y), fmaf(w, v.z, a.z), fmaf(w, v.w, a.w)); };
    auto affine = [](const float4& s, const float4& c, float wsum, const float4& h, const float4& f) {
        return make_float4(f.x + fmaf(s.x, c.x, wsum * h.x), f.y + fmaf(s.y, c.y, wsum * h.y), f.z + fmaf(s.z, c.z, wsum * h.z), f.w + fmaf(s.w, c.w, wsum * h.w));
    };
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
        const int src = 8 * i + s8;
        float4 fg = make_float4(0.0f, 0.0f, 0.0f, 0.0f), fa = fg;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            float4 sg = make_float4(0.0f, 0.0f, 0.0f, 0.0f), sa = sg;
            float wsum = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int off = __shfl(geo.off[4 * p + k], src);
                const float w = __shfl(geo.w[4 * p + k], src);
                wsum += w;
                if (DO_G) sg = fma4(w, *reinterpret_cast<const float4*>(pg + off), sg);
                if (DO_A) sa = fma4(w, *reinterpret_cast<const float4*>(pa + off), sa);
            }
            fg = affine(sg, sc[0][p], wsum, sh[0][p], fg);
            fa = affine(sa, sc[1][p], wsum, sh[1][p], fa);
        }
        const float third = 1.0f / 3.0f;        // mean over planes, triplane.py:251
        *reinterpret_cast<float4*>(tile + src * SORT_TILE_STRIDE + c4) = make_float4(fg.x * third, fg.y * third, fg.z * third, fg.w * third);
        *reinterpret_cast<float4*>(tile + src * SORT_TILE_STRIDE + 32 + c4) = make_float4(fa.x * third, fa.y * third, fa.z * third, fa.w * third);
    }
    __builtin_amdgcn_wave_barrier();
}
// this lane's sample (lane = sample) back from the tile
template <int STRIDE>
__device__ __forceinline__ void tile_row(const float* tile, int lane, int col0, f32x2 (&f)[16]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(tile + lane * STRIDE + col0 + 4 * q);
        f[2 * q] = f32x2{v.x, v.y}; f[2 * q + 1] = f32x2{v.z, v.w};
    }
}

template <bool MFMA, bool BINNED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void bwd_scatter_sorted_kernel(BwdK P) {   // 256 registers
    __shared__ __attribute__((aligned(16))) float tile[64 * SORT_TILE_STRIDE];       // [sample][0..31 geometry-set gradient, 32..63 appearance-set]
    __shared__ float wtab[BINNED ? 1 : 64 * 12];        // tap weights [sample][plane*4 + tap]
    const int lane = threadIdx.x;
    // grid = (ray tiles, depths, views), or with NFE_BWD_DEPTH_FAST (depths, ray tiles, views): which neighbours run together
    const int bx = NFE_BWD_DEPTH_FAST ? blockIdx.y : blockIdx.x;
    const int n = blockIdx.z + (BINNED ? P.n0 : 0), kdepth = NFE_BWD_DEPTH_FAST ? blockIdx.x : blockIdx.y, t = bx + (BINNED ? P.t0 : 0);
    int m; bool live = true;
    if (P.R > 0 && (P.R & 7) == 0 && (long long)P.R * P.R == P.M) {       // 8x8 pixel tile
        const int tiles_x = P.R >> 3;
        m = ((t / tiles_x) * 8 + (lane >> 3)) * P.R + (t % tiles_x) * 8 + (lane & 7);
    } else {
        m = t * 64 + lane; live = m < P.M; m = min(m, P.M - 1);
    }
    // dL/dsigma, omega and the depth of this wave's 64 samples: one 256-byte row each (bwd_ray_kernel wrote them in this order)
    // (dead lanes of a view's last tile repeat its last ray, whose slot is m & 63 of the same tile: nobody wrote theirs)
    const size_t rec = (((size_t)n * P.T + t) * P.S + kdepth) * 64 + (live ? lane : (m & 63));
    const float t_sample = P.rec_t[rec];
    SampleGeo geo;
    sample_geometry(P, n, m, t_sample, geo);
    const float gsig = P.rec_sig[rec], omega = P.rec_a[rec];
    const long long pv = (long long)n * P.plane_view_stride;
    const long long gv = (long long)n * P.grad_view_stride;
    const float* dec = P.dec;
    const bool do_g = P.grad_g != nullptr, do_a = P.grad_a != nullptr && P.g_rgb != nullptr;
    if (MFMA) {       // decoder forward + backward on the matrix cores (head_mfma): 64 samples = two N-blocks of 32
        const FragGlobal F{P.bfrag + lane, dec};
        const int jj = lane & 31, hh = lane >> 5;
        f32x2 f[16];
        const int sets = sgpr_nonnull(P.grad_g) | (sgpr_nonnull(P.grad_a) & sgpr_nonnull(P.g_rgb)) << 1;
        if (sets == 3) gather_pair_coop<true, true>(P, n, geo, lane, tile);          // (no run-time switch inside the pipelined loop)
        else if (sets == 1) gather_pair_coop<true, false>(P, n, geo, lane, tile);
        else gather_pair_coop<false, true>(P, n, geo, lane, tile);
        // Output cotangents through the tile (round 4).  Lane (jj, hh) of N-block b needs 8 / 16 consecutive cotangents of sample
        // 32b + jj; fetched in that layout every load instruction touched 32 rays' cache lines, ~500 line requests per wave on a texture
        // addresser that is the kernel's busiest unit (without them the kernel is 0.34 ms faster per 4 views).  For the [view, ray,
        // channel] layout they are now fetched with 8 (4) consecutive lanes per ray row - every line requested once, ~110 requests - into
        // the columns of the tile the head's features have just left (sample s: row s; N-block b overwrites rows 32b.. only after
        // it has read them), and the lanes pick their values up with two / four ds_read_b128.
        const bool staged = NFE_BWD_COT_STAGED && !P.channels_first;
        if (do_g) {
            tile_row<SORT_TILE_STRIDE>(tile, lane, 0, f);
            if (staged) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {                 // column o = 1 + seg channel (column 0: sigma's slot, not read)
                    const int ss = 4 * i + (lane >> 4), c = lane & 15;
                    const int ms = __shfl(m, ss);
                    const float v = (P.g_seg && c < 15) ? P.g_seg[((long long)n * P.M + ms) * 15 + c] : 0.0f;
                    if (c < 15) tile[ss * SORT_TILE_STRIDE + 1 + c] = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
            head_mfma<false>(F, f, lane, tile, SORT_TILE_STRIDE, 0,
                             [&](int b, float (&d)[8]) {          // outputs 8h..8h+7 of sample 32b + j: sigma = 0, seg = 1..15 (triplane.py:260-261)
                                 const int src = 32 * b + jj;
                                 const float gs = __shfl(gsig, src), om = __shfl(omega, src);
                                 if (staged) {
                                     const float4 c0 = *reinterpret_cast<const float4*>(tile + src * SORT_TILE_STRIDE + 8 * hh);
                                     const float4 c1 = *reinterpret_cast<const float4*>(tile + src * SORT_TILE_STRIDE + 8 * hh + 4);
                                     const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                                     for (int e = 0; e < 8; ++e) d[e] = (8 * hh + e) == 0 ? gs : om * cv[e];
                                     return;
                                 }
                                 const int mm = __shfl(m, src);
#pragma unroll
                                 for (int e = 0; e < 8; ++e) {
                                     const int o = 8 * hh + e;
                                     d[e] = o == 0 ? gs : om * cot_seg(P, n, mm, o - 1);
                                 }
                             },
                             [](int, f32x16&) {});
        } else {
#pragma unroll
            for (int c = 0; c < 32; ++c) tile[lane * SORT_TILE_STRIDE + c] = 0.0f;
        }
        if (do_a) {
            tile_row<SORT_TILE_STRIDE>(tile, lane, 32, f);
            if (staged) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll 4
                for (int i = 0; i < 8; ++i) {                  // eight lanes per ray row of 128 bytes; the * 2 of rgb * 2 - 1 here (cot_rgb)
                    const int ss = 8 * i + (lane >> 3), c4 = (lane & 7) * 4;
                    const int ms = __shfl(m, ss);
                    const float4 v = *reinterpret_cast<const float4*>(P.g_rgb + ((long long)n * P.M + ms) * 32 + c4);      // do_a: g_rgb is not null
                    *reinterpret_cast<float4*>(tile + ss * SORT_TILE_STRIDE + 32 + c4) = make_float4(2.0f * v.x, 2.0f * v.y, 2.0f * v.z, 2.0f * v.w);
                }
                __builtin_amdgcn_wave_barrier();
            }
            head_mfma<true>(F, f, lane, tile, SORT_TILE_STRIDE, 32, [](int, float (&)[8]) {},
                            [&](int b, f32x16& y) {               // rgb = sigmoid(y) * 1.002 - 0.001 (triplane.py:269), channel 16h + r
                                const int src = 32 * b + jj;
                                const float om = __shfl(omega, src);
                                if (staged) {
                                    const float* cr = tile + src * SORT_TILE_STRIDE + 32 + 16 * hh;
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        const float4 cv = *reinterpret_cast<const float4*>(cr + 4 * q);
                                        const float c4[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                                        for (int u = 0; u < 4; ++u) {
                                            const float sg = sigmoid_t(y[4 * q + u]);
                                            y[4 * q + u] = om * c4[u] * 1.002f * sg * (1.0f - sg);
                                        }
                                    }
                                    return;
                                }
                                const int mm = __shfl(m, src);
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float sg = sigmoid_t(y[r]);
                                    y[r] = om * cot_rgb(P, n, mm, 16 * hh + r) * 1.002f * sg * (1.0f - sg);
                                }
                            });
        } else {
#pragma unroll
            for (int c = 0; c < 32; ++c) tile[lane * SORT_TILE_STRIDE + 32 + c] = 0.0f;
        }
    } else {
        f32x2 f[16], df[16];
        if (do_g) {
            gather_set(P.planes_g + pv, geo, P.aff[0] ? P.aff[0] + n * 96 : nullptr, P.aff[1] ? P.aff[1] + n * 96 : nullptr, f);
            f32x2 dout[8];
            dout[0][0] = gsig;                                         // sigma = channel 0, seg = 1..15 (triplane.py:260-261)
#pragma unroll
            for (int c = 0; c < 15; ++c) dout[(1 + c) >> 1][(1 + c) & 1] = omega * cot_seg(P, n, m, c);
            head_backward<16>(dec + BW_G0, dec + BB_G0, dec + BW_G1T, f, dout, df);
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) df[c] = splat(0.0f);
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) tile[lane * SORT_TILE_STRIDE + c] = df[c >> 1][c & 1] * (1.0f / 3.0f);   // mean over planes, triplane.py:251
        if (do_a) {
            gather_set(P.planes_a + pv, geo, P.aff[2] ? P.aff[2] + n * 96 : nullptr, P.aff[3] ? P.aff[3] + n * 96 : nullptr, f);
            f32x2 y[16];
            head_forward<32>(dec + BW_A0, dec + BB_A0, dec + BW_A1T, dec + BB_A1, f, y);
#pragma unroll
            for (int c = 0; c < 32; ++c) {                             // rgb = sigmoid(y) * 1.002 - 0.001 (triplane.py:269)
                const float sg = sigmoid_t(y[c >> 1][c & 1]);
                y[c >> 1][c & 1] = omega * cot_rgb(P, n, m, c) * 1.002f * sg * (1.0f - sg);
            }
            head_backward<32>(dec + BW_A0, dec + BB_A0, dec + BW_A1T, f, y, df);
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) df[c] = splat(0.0f);
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) tile[lane * SORT_TILE_STRIDE + 32 + c] = df[c >> 1][c & 1] * (1.0f / 3.0f);
    }
    if (BINNED) {       // feature gradients to the chunk buffer, one bin record per (sample, plane); bwd_accumulate_kernel adds them up
        __builtin_amdgcn_wave_barrier();
        const unsigned wave = ((unsigned)blockIdx.z * (unsigned)P.t_count + (unsigned)bx) * (unsigned)P.S + (unsigned)kdepth;
        float* dst = P.df + (size_t)wave * 4096 + lane;
#pragma unroll 8
        for (int sidx = 0; sidx < 64; ++sidx) dst[sidx * 64] = tile[sidx * SORT_TILE_STRIDE + lane];
        const unsigned idx = wave * 64 + (unsigned)lane;
        const unsigned bins_per_plane = (unsigned)(P.bins_x * P.bins_y);
        float ro[3], rd[3];                 // tap geometry again (cheaper than 24 registers kept across the decoder)
        ray_of(P, n, m, ro, rd);
        const float tt = t_sample;
        const float cx = P.coord_scale * fmaf(tt, rd[0], ro[0]), cy = P.coord_scale * fmaf(tt, rd[1], ro[1]), cz = P.coord_scale * fmaf(tt, rd[2], ro[2]);
        // Three phases so that the planes' rank atomics (a round trip to memory each) are in flight together:
        // records and their position inside the wave's share of each bin, then the three atomics, then the stores.
        unsigned bin3[3], rank3[3], group3[3], loc3[3];
        int first3[3];
        float4 w3[3];
        bool any3[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const Taps tp = tap_geometry(P.H, P.W, p == 2 ? cz : cx, p == 0 ? cy : (p == 1 ? cz : cx));      // as sample_geometry
            const int x0 = tp.xc0, x1 = tp.xc1, y0 = tp.yc0, y1 = tp.yc1;
            // (one compare per decision: a round-2 precaution, see sgpr_nonnull)
            const bool any = (live ? (tp.w[0] + tp.w[1]) + (tp.w[2] + tp.w[3]) : 0.0f) != 0.0f;        // the weights are >= 0
            const unsigned bin = any ? ((unsigned)blockIdx.z * 3u + (unsigned)p) * bins_per_plane + (unsigned)((y0 >> BIN_SHIFT) * P.bins_x + (x0 >> BIN_SHIFT))
                                     : KEY_INVALID;
            // rank inside the bin: for every distinct bin of the wave its first lane, the lanes' position among the wave's records
            // of that bin and their number (registers only); ONE returning atomic instruction per plane, executed by the first lanes
            // of all its bins together, fetches the bases.  The loop condition is made with scalar compares (a round-2 precaution, see
            // sgpr_nonnull; the failures it was meant to avoid were profiles/experiments/r04_pk_opsel_hazard.md).
            unsigned rank = 0, group = 0;
            int first_lane = -1;                      // stays -1 on lanes without a record
            const unsigned long long have = __ballot(any);
            unsigned todo_lo = (unsigned)have, todo_hi = (unsigned)(have >> 32);
            for (;;) {
                todo_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)todo_lo); todo_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)todo_hi);
                asm volatile("; nfe_launder %0 %1" : "+s"(todo_lo), "+s"(todo_hi));
                if ((todo_lo | todo_hi) == 0u) break;
                const int leader = todo_lo ? __builtin_ctz(todo_lo) : 32 + __builtin_ctz(todo_hi);
                const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
                const unsigned long long same = __ballot(bin == b);          // b is a live record's bin, dead lanes hold KEY_INVALID
                if (bin == b) { rank = (unsigned)__popcll(same & ((1ull << lane) - 1ull)); group = (unsigned)__popcll(same); first_lane = leader; }
                todo_lo &= ~(unsigned)same; todo_hi &= ~(unsigned)(same >> 32);
            }
            // The accumulate pass adds the four taps of a record at local texels l0, l0 + 1, l0 + row, l0 + row + 1 with plain
            // read-modify-writes, so they must be four different texels.  Where clamping makes two taps the same texel (x1 == x0
            // at a plane edge: one of the pair is out of range and carries weight 0) their weights are folded into the first.
            const int sx = x1 - x0, sy = y1 - y0;
            float w0 = tp.w[0], w1 = tp.w[1], w2 = tp.w[2], w3v = tp.w[3];
            if (sx == 0) { w0 += w1; w2 += w3v; w1 = 0.0f; w3v = 0.0f; }
            if (sy == 0) { w0 += w2; w1 += w3v; w2 = 0.0f; w3v = 0.0f; }
            bin3[p] = bin; rank3[p] = rank; group3[p] = group; first3[p] = first_lane; any3[p] = any;
            loc3[p] = (unsigned)((y0 & BIN_MASK) * BIN_TEXELS + (x0 & BIN_MASK));
            w3[p] = make_float4(w0, w1, w2, w3v);
        }
        unsigned base3[3] = {0u, 0u, 0u};
#pragma unroll
        for (int p = 0; p < 3; ++p)
            if (first3[p] == lane) base3[p] = __hip_atomic_fetch_add(P.counts + bin3[p], group3[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned rank = rank3[p] + (unsigned)__shfl((int)base3[p], first3[p] & 63);
            const size_t slot = (size_t)p * ((size_t)gridDim.z * P.t_count * P.S * 64) + idx;
            P.binrank[slot] = make_uint2(bin3[p], rank);
            if (any3[p]) {
                P.rec_key[slot] = make_uint2(idx, loc3[p]);
                P.rec_w[slot] = w3[p];
            }
        }
        return;
    }
    const int plane_elems = P.H * P.W * 32;
    const float livef = live ? 1.0f : 0.0f;
    unsigned key[3][4];                                 // element e = lane * 4 + r of each plane's list
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const int p = q >> 2;
        wtab[lane * 12 + q] = geo.w[q];
        const unsigned texel = (unsigned)(geo.off[q] - p * plane_elems) >> 5;
        key[p][q & 3] = geo.w[q] * livef != 0.0f ? (texel << 8 | (unsigned)lane << 2 | (unsigned)(q & 3)) : KEY_INVALID;   // (a product: the compiler turns `(live ? w : 0) != 0` back into two compares and an s_and_b64)
    }
    __threadfence_block();
    // bitonic sort of the three 256-key lists in registers (ascending; invalid keys end up last): partners at distance
    // 1, 2 are in the same lane, larger distances are lane ^ (distance / 4)
#pragma unroll
    for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 4) {
                const int lj = j >> 2;
                const bool keep_min = ((((unsigned)lane / (unsigned)lj) ^ ((unsigned)lane / (unsigned)(k >> 2))) & 1u) == 0u;   // one compare, no mask logic on the scalar unit (r02_lane_mask.md)
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned other = (unsigned)__shfl_xor((int)key[p][r], lj);
                        key[p][r] = keep_min ? min(key[p][r], other) : max(key[p][r], other);
                    }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r & j) continue;
                    const bool up = k == 2 ? (r & 2) == 0 : (k == 4 ? (lane & 1) == 0 : (lane & (k >> 2)) == 0);
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const unsigned lo = min(key[p][r], key[p][r | j]), hi = max(key[p][r], key[p][r | j]);
                        key[p][r] = up ? lo : hi; key[p][r | j] = up ? hi : lo;
                    }
                }
            }
        }
    }
    // walk the sorted lists: element e is register e & 3 of lane e >> 2
    const int ch = lane & 31, set = lane >> 5;
    float* gbase = set ? (do_a ? P.grad_a + gv : nullptr) : (do_g ? P.grad_g + gv : nullptr);
    const float* scale = P.aff[2 * set] ? P.aff[2 * set] + n * 96 : nullptr;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float wreg[4]; unsigned wk[4];              // walk key: texel << 8 | sample << 2 | (1 if the run of this texel ends here)
        unsigned nxt0 = (unsigned)__shfl_down((int)key[p][0], 1);
        if (lane == 63) nxt0 = KEY_INVALID;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned kk = key[p][r];
            const unsigned nxt = r < 3 ? key[p][r < 3 ? r + 1 : 3] : nxt0;
            wreg[r] = kk != KEY_INVALID ? wtab[((kk >> 2) & 63) * 12 + p * 4 + (kk & 3)] : 0.0f;
            wk[r] = (kk & ~3u) | ((kk >> 8) != (nxt >> 8) ? 1u : 0u);
        }
        const float sc = scale ? scale[p * 32 + ch] : 1.0f;
        float* base = gbase ? gbase + (long long)p * plane_elems + ch : nullptr;
        float acc = 0.0f;
#pragma unroll 1
        for (int i = 0; i < 64; i += 2) {
            if ((unsigned)__builtin_amdgcn_readlane((int)key[p][0], i) == KEY_INVALID) break;      // sorted: nothing valid from here on
            unsigned kk[8]; float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                kk[u] = (unsigned)__builtin_amdgcn_readlane((int)wk[u & 3], i + (u >> 2));
                v[u] = tile[((kk[u] >> 2) & 63) * SORT_TILE_STRIDE + lane];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wreg[u & 3]), i + (u >> 2)));
                acc = fmaf(v[u], w, acc);
                if (kk[u] & 1u) {
                    if (base) unsafeAtomicAdd(base + (long long)(kk[u] >> 8) * 32, acc * sc);
                    acc = 0.0f;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Decoder-backward pass of the binned form, wave-specialised (round 5).  The single-wave kernel above leaves every phase exposed:
// per 64 samples a wave walks ~8 dependent memory round trips (records, two gather batches per plane set, cotangents, rank atomics)
// and ~5 700 vector instructions, two waves per SIMD at 256 registers - the SIMD issues vector work 47 % of the time, the texture
// addresser is 0.49-0.67 busy, the matrix pipe 0.09 (profiles/r04_backward_counters.json, r05_pmc_backward.txt).
// Here a persistent workgroup of eight waves (one per CU) is four PAIRS, producer wave w and consumer wave w + 4 on the same SIMD:
//   producer  item -> sample geometry -> gather of one plane set (eight lanes per texel row, sums in registers) + that set's output
//             cotangents (already multiplied by omega; dL/dsigma in column 0 of the geometry set's) -> waits until the consumer has
//             released the pair's tile -> writes both tiles, publishes; after both sets: the item's bin records and rank atomics.
//   consumer  waits for a published set -> its sample's 32 features from the tile -> head forward / backward on the matrix cores
//             (fragments and biases from the workgroup's LDS image, staged once: the single-wave kernel fetched 178 KB of them per
//             item) -> feature gradients through the tile into registers -> releases the tile -> 64 half rows (128 B) of df.
// The producer's loads for the next set are in flight (its sums sit in registers) while the consumer still owns the tile, so one tile
// pair per wave pair is enough: 52 KB image + 4 x 18 KB.  Hand-off: two LDS counters per pair (ws_wait / ws_signal of
// render_ws_kernel); an abandoned wait marks the pair, the launch ends, and P.abort_word makes the accumulate pass write NaN.
// Arithmetic per channel is that of the single-wave kernel, operation for operation (tests compare the two forms).
// ------------------------------------------------------------------------------------------------------------
constexpr int DEC_PAIRS = 4;
constexpr int DEC_RING = 3;                                      // stage buffers of a producer: two stages in flight while one is consumed (a ring of two measured the same, 1.91 vs 1.88 ms)
constexpr int DEC_TILE_STRIDE = 36;                              // floats per sample row (32 channels of ONE plane set + pad; rows 16-byte aligned)
constexpr int DEC_TILE_BYTES = 64 * DEC_TILE_STRIDE * 4;
constexpr int DEC_GEO_BYTES = 64 * 24 * 4;                       // the item's tap geometry, [sample][12 byte offsets, 12 weights]: read back with lane = (sample of a group, ...)
constexpr int DEC_AFF_BYTES = 2 * 3 * 64 * 4;                    // the view's appearance statistics, [set][plane][32 scales, 32 shifts] (1 / 0 without)
constexpr int DEC_PAIR_BYTES = 2 * DEC_TILE_BYTES + 16 + DEC_GEO_BYTES + DEC_AFF_BYTES;      // feature tile, cotangent tile, {published, released, abort, -}, geometry, statistics
constexpr int DEC_LDS_BYTES = BWD_LDS_TILES + DEC_PAIRS * DEC_PAIR_BYTES;
static_assert(DEC_LDS_BYTES <= 160 * 1024 && DEC_PAIR_BYTES % 16 == 0, "fragment image + four tile pairs must fit a CU's LDS");
constexpr int DEC_SPIN_LIMIT = 1 << 19;                          // polls (s_sleep 2 + an LDS read each, ~100 ms) before a hand-off wait is abandoned
__device__ int g_dec_spin_limit = DEC_SPIN_LIMIT;                // read only on the slow path of a wait; NFE_WS_SPIN_LIMIT (tests/test_handoff_abort_gpu.py) shortens it

__device__ __forceinline__ bool dec_wait(unsigned* flags, int which, unsigned need) {
    int spins = 0;
    while (true) {
        const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flags + which, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(v - need) >= 0) return true;
        const unsigned ab = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (ab != 0u) return false;
        if (++spins > __builtin_nontemporal_load(&g_dec_spin_limit)) {
            __hip_atomic_store(flags + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
__device__ __forceinline__ void dec_signal(unsigned* flags, int which, unsigned value, int lane) {      // release: this wave's LDS accesses are complete
    if (lane == 0) __hip_atomic_store(flags + which, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The producer's work on one item as a software pipeline of STAGES: per plane set eight gather stages (one group of eight samples:
// lane = (sample of the group, four channels), twelve 16-byte loads = the twelve taps of the group's rows) and one cotangent stage
// (the set's output cotangents, coalesced).  A stage is ISSUED two stages before it is CONSUMED, into a ring of three 48-register
// buffers, so 24-36 loads are in flight while the tap sums of an older group are formed - and across the hand-off: the first groups
// of the appearance set are in flight while the producer waits for the consumer to release the tile.  (Two batches of 48 loads per
// set, each waited for in full before anything else happened, made the producer the slower role: 2.8 ms per launch against the
// consumer's 1.4; profiles/experiments/r05_backward_ws.md.)  Same arithmetic per channel as gather_pair_coop, operation for operation.
struct ProdCtx {
    const BwdK& P;
    int n, m, lane; float gsig, omega;
    float* feat; float* cot; unsigned* flags; const float* geo; const float* aff;      // geo / aff: the pair's LDS copies (DEC_GEO_BYTES, DEC_AFF_BYTES)
};
__device__ __forceinline__ float& comp(float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }      // k is a constant after unrolling
__device__ __forceinline__ float comp(const float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }
__device__ __forceinline__ int pinned(int v) { asm volatile("; nfe_pin %0" : "+v"(v)); return v; }       // not before the statement in front of it
#define NFE_STAGE_FENCE() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

template <int SET, int G>
__device__ __forceinline__ void prod_issue(const ProdCtx& c, float4 (&b)[12]) {
    const BwdK& P = c.P;
    const int lane = c.lane, s8 = lane >> 3, c4 = (lane & 7) * 4;
    if constexpr (G < 8) {
        const int src = pinned(8 * G + s8);
        // wave-uniform base + a 32-bit byte offset per lane (a plane set the binned form accepts is below 4 GiB): one add per load
        const char* base = reinterpret_cast<const char*>((SET ? P.planes_a : P.planes_g) + (long long)c.n * P.plane_view_stride);
        const unsigned lane_off = (unsigned)c4 * 4u;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint4 off = *reinterpret_cast<const uint4*>(c.geo + src * 24 + 4 * p);        // byte offsets of the row's four taps
            b[4 * p + 0] = *reinterpret_cast<const float4*>(base + (off.x + lane_off));
            b[4 * p + 1] = *reinterpret_cast<const float4*>(base + (off.y + lane_off));
            b[4 * p + 2] = *reinterpret_cast<const float4*>(base + (off.z + lane_off));
            b[4 * p + 3] = *reinterpret_cast<const float4*>(base + (off.w + lane_off));
        }
    } else if constexpr (SET == 0) {         // seg cotangent c - 1 of sample 4 i + (lane >> 4) in column c = lane & 15 (column 0: dL/dsigma, from a register)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ms = __shfl(c.m, pinned(4 * i + (lane >> 4)));
            comp(b[i >> 2], i & 3) = cot_seg(P, c.n, ms, max((lane & 15) - 1, 0));
        }
    } else {                                  // rgb cotangents, eight lanes per ray row of 128 bytes; the * 2 of rgb * 2 - 1 here (cot_rgb)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ms = __shfl(c.m, pinned(8 * i + s8));
            if (!P.channels_first) {
                const float4 g4 = *reinterpret_cast<const float4*>(P.g_rgb + ((long long)c.n * P.M + ms) * 32 + c4);      // this set is differentiated: g_rgb is not null
                b[i] = make_float4(2.0f * g4.x, 2.0f * g4.y, 2.0f * g4.z, 2.0f * g4.w);
            } else {
                b[i] = make_float4(cot_rgb(P, c.n, ms, c4), cot_rgb(P, c.n, ms, c4 + 1), cot_rgb(P, c.n, ms, c4 + 2), cot_rgb(P, c.n, ms, c4 + 3));
            }
        }
    }
}

// returns false when the hand-off wait was abandoned
template <int SET, int G, bool AFF>
__device__ __forceinline__ bool prod_consume(const ProdCtx& c, const float4 (&b)[12], float4 (&acc)[8], unsigned& q, bool alive) {
    const int lane = c.lane, s8 = lane >> 3, c4 = (lane & 7) * 4;
    if constexpr (G < 8) {
        auto fma4 = [](float w, const float4& v, const float4& a) { return make_float4(fmaf(w, v.x, a.x), fmaf(w, v.y, a.y), fmaf(w, v.z, a.z), fmaf(w, v.w, a.w)); };
        auto affine = [](const float4& s, const float4& cc, float wsum, const float4& h, const float4& f) {
            return make_float4(f.x + fmaf(s.x, cc.x, wsum * h.x), f.y + fmaf(s.y, cc.y, wsum * h.y), f.z + fmaf(s.z, cc.z, wsum * h.z), f.w + fmaf(s.w, cc.w, wsum * h.w));
        };
        const int src = pinned(8 * G + s8);
        float4 fs = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            float4 sm = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float wsum = 0.0f;
            const float4 w4 = *reinterpret_cast<const float4*>(c.geo + src * 24 + 12 + 4 * p);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float w = comp(w4, k);
                if (AFF) wsum += w;
                sm = fma4(w, b[4 * p + k], sm);
            }
            if (AFF) {
                const float4 scp = *reinterpret_cast<const float4*>(c.aff + (SET * 3 + p) * 64 + c4), shp = *reinterpret_cast<const float4*>(c.aff + (SET * 3 + p) * 64 + 32 + c4);
                fs = affine(sm, scp, wsum, shp, fs);
            } else {          // no appearance statistics at all (scale 1, shift 0): f + fma(s, 1, wsum * 0) is f + s, to the bit
                fs = make_float4(fs.x + sm.x, fs.y + sm.y, fs.z + sm.z, fs.w + sm.w);
            }
        }
        // (formed HERE: without this the optimiser sinks the whole sum behind the hand-off wait, where acc is first read, and every tap
        // load of the set stays live until then - 2.5 KB of spills per lane)
        asm volatile("; nfe_pin %0 %1 %2 %3" : "+v"(fs.x), "+v"(fs.y), "+v"(fs.z), "+v"(fs.w));
        acc[G] = fs;
        return alive;
    } else {
        // the set's tiles: features / 3 (mean over planes, triplane.py:251) and the cotangents already multiplied by omega
        float4 cv[8];
        if constexpr (SET == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ss = pinned(4 * i + (lane >> 4));
                const float gs = __shfl(c.gsig, ss), om = __shfl(c.omega, ss);
                comp(cv[i >> 2], i & 3) = (lane & 15) == 0 ? gs : om * comp(b[i >> 2], i & 3);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float om = __shfl(c.omega, pinned(8 * i + s8));
                cv[i] = make_float4(om * b[i].x, om * b[i].y, om * b[i].z, om * b[i].w);
            }
        }
        if (alive) alive = dec_wait(c.flags, 1, q);
        const float third = 1.0f / 3.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<float4*>(c.feat + (8 * i + s8) * DEC_TILE_STRIDE + c4) = make_float4(acc[i].x * third, acc[i].y * third, acc[i].z * third, acc[i].w * third);
        if constexpr (SET == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c.cot[(4 * i + (lane >> 4)) * DEC_TILE_STRIDE + (lane & 15)] = comp(cv[i >> 2], i & 3);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(c.cot + (8 * i + s8) * DEC_TILE_STRIDE + c4) = cv[i];
        }
        dec_signal(c.flags, 0, ++q, lane);
        return alive;
    }
}

// stage t of an item's sequence: nine per differentiated plane set
template <bool DO_G, bool DO_A, int T> struct ProdStage {
    static constexpr int SET = (DO_G && DO_A) ? (T >= 9 ? 1 : 0) : (DO_A ? 1 : 0);
    static constexpr int G = T - ((DO_G && DO_A && T >= 9) ? 9 : 0);
};
template <bool DO_G, bool DO_A, bool AFF, int T>
__device__ __forceinline__ bool prod_run(const ProdCtx& c, float4 (&buf)[DEC_RING][12], float4 (&acc)[8], unsigned& q, bool alive) {
    constexpr int NST = 9 * ((DO_G ? 1 : 0) + (DO_A ? 1 : 0));
    if constexpr (T < NST) {
        if constexpr (T + DEC_RING - 1 < NST) prod_issue<ProdStage<DO_G, DO_A, T + DEC_RING - 1>::SET, ProdStage<DO_G, DO_A, T + DEC_RING - 1>::G>(c, buf[(T + DEC_RING - 1) % DEC_RING]);
        NFE_STAGE_FENCE();
        alive = prod_consume<ProdStage<DO_G, DO_A, T>::SET, ProdStage<DO_G, DO_A, T>::G, AFF>(c, buf[T % DEC_RING], acc, q, alive);
        NFE_STAGE_FENCE();
        return prod_run<DO_G, DO_A, AFF, T + 1>(c, buf, acc, q, alive);
    } else {
        return alive;
    }
}
// an item's place: ray tiles fastest, then depths, then views - the single-wave kernel's block order (neighbours in the plane run together)
struct DecItem { int n, t, kdepth, bx; unsigned bz; int m; bool live; size_t rec; unsigned slot; };
__device__ __forceinline__ DecItem dec_item(const BwdK& P, unsigned item, int lane) {
    DecItem it;
    const unsigned per_depth = (unsigned)P.t_count, per_view = per_depth * (unsigned)P.S;
    it.bz = item / per_view;
    const unsigned rest = item - it.bz * per_view;
    it.kdepth = (int)(rest / per_depth); it.bx = (int)(rest - (unsigned)it.kdepth * per_depth);
    it.n = (int)it.bz + P.n0; it.t = it.bx + P.t0;
    it.live = true;
    if (P.R > 0 && (P.R & 7) == 0 && (long long)P.R * P.R == P.M) {       // 8x8 pixel tile
        const int tiles_x = P.R >> 3;
        it.m = ((it.t / tiles_x) * 8 + (lane >> 3)) * P.R + (it.t % tiles_x) * 8 + (lane & 7);
    } else {
        it.m = it.t * 64 + lane; it.live = it.m < P.M; it.m = min(it.m, P.M - 1);
    }
    // (dead lanes of a view's last tile repeat its last ray, whose slot is m & 63 of the same tile: nobody wrote theirs)
    it.rec = (((size_t)it.n * P.T + it.t) * P.S + it.kdepth) * 64 + (it.live ? lane : (it.m & 63));
    it.slot = (it.bz * (unsigned)P.t_count + (unsigned)it.bx) * (unsigned)P.S + (unsigned)it.kdepth;      // the item's 64 sample slots in the chunk
    return it;
}

// An item's tap geometry into the pair's LDS copy ([sample][12 byte offsets, 12 weights]; for a new view also its appearance
// statistics, or 1 / 0) and the first half of its bin records (one per sample and plane, as in the single-wave kernel): bins, the lanes'
// ranks among the wave's records of a bin, and ONE returning atomic per plane for the bases.  The bases are picked up by
// dec_records_end a whole item's gathers later, so their round trip costs nothing.
struct DecRecords { unsigned bin[3], rank[3], loc[3], base[3]; int first[3]; unsigned idx; };
__device__ __forceinline__ void dec_geometry(const BwdK& P, const DecItem& it, float t_sample, int lane, float* geo_lds, float* aff_lds, int& aff_view, DecRecords& R) {
    if (it.n != aff_view) {
        aff_view = it.n;
#pragma unroll
        for (int r = 0; r < 6; ++r) {      // [set = r / 3][plane = r % 3][lane: 32 scales, 32 shifts]
            const float* src = P.aff[2 * (r / 3) + (lane >> 5)];
            aff_lds[r * 64 + lane] = src ? src[it.n * 96 + (r % 3) * 32 + (lane & 31)] : (lane < 32 ? 1.0f : 0.0f);
        }
    }
    float ro[3], rd[3];
    ray_of(P, it.n, it.m, ro, rd);
    const float cx = P.coord_scale * fmaf(t_sample, rd[0], ro[0]), cy = P.coord_scale * fmaf(t_sample, rd[1], ro[1]), cz = P.coord_scale * fmaf(t_sample, rd[2], ro[2]);
    const unsigned plane_elems = (unsigned)(P.H * P.W * 32);
    const unsigned bins_per_plane = (unsigned)(P.bins_x * P.bins_y);
    R.idx = it.slot * 64 + (unsigned)lane;
#pragma unroll
    for (int p = 0; p < 3; ++p) {     // plane axes of generate_planes (renderer.py:23-37): (x,y), (x,z), (z,x) - sample_geometry's arithmetic
        const Taps tp = tap_geometry(P.H, P.W, p == 2 ? cz : cx, p == 0 ? cy : (p == 1 ? cz : cx));
        const int x0 = tp.xc0, x1 = tp.xc1, y0 = tp.yc0, y1 = tp.yc1;
        const unsigned pb = (unsigned)p * plane_elems;
        *reinterpret_cast<uint4*>(geo_lds + lane * 24 + 4 * p) = make_uint4((pb + (unsigned)(y0 * P.W + x0) * 32u) * 4u, (pb + (unsigned)(y0 * P.W + x1) * 32u) * 4u,
                                                                             (pb + (unsigned)(y1 * P.W + x0) * 32u) * 4u, (pb + (unsigned)(y1 * P.W + x1) * 32u) * 4u);
        *reinterpret_cast<float4*>(geo_lds + lane * 24 + 12 + 4 * p) = make_float4(tp.w[0], tp.w[1], tp.w[2], tp.w[3]);
        const bool any = (it.live ? (tp.w[0] + tp.w[1]) + (tp.w[2] + tp.w[3]) : 0.0f) != 0.0f;        // the weights are >= 0
        const unsigned bin = any ? (it.bz * 3u + (unsigned)p) * bins_per_plane + (unsigned)((y0 >> BIN_SHIFT) * P.bins_x + (x0 >> BIN_SHIFT))
                                 : KEY_INVALID;
        unsigned rank = 0, group = 0;
        int first_lane = -1;                      // stays -1 on lanes without a record
        const unsigned long long have = __ballot(any);
        unsigned todo_lo = (unsigned)have, todo_hi = (unsigned)(have >> 32);
        for (;;) {
            todo_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)todo_lo); todo_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)todo_hi);
            asm volatile("; nfe_launder %0 %1" : "+s"(todo_lo), "+s"(todo_hi));
            if ((todo_lo | todo_hi) == 0u) break;
            const int leader = todo_lo ? __builtin_ctz(todo_lo) : 32 + __builtin_ctz(todo_hi);
            const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
            const unsigned long long same = __ballot(bin == b);          // b is a live record's bin, dead lanes hold KEY_INVALID
            if (bin == b) { rank = (unsigned)__popcll(same & ((1ull << lane) - 1ull)); group = (unsigned)__popcll(same); first_lane = leader; }
            todo_lo &= ~(unsigned)same; todo_hi &= ~(unsigned)(same >> 32);
        }
        R.bin[p] = bin; R.rank[p] = rank; R.first[p] = first_lane;
        R.loc[p] = (unsigned)((y0 & BIN_MASK) * BIN_TEXELS + (x0 & BIN_MASK));
        R.base[p] = 0u;
        if (first_lane == lane) R.base[p] = __hip_atomic_fetch_add(P.counts + bin, group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// second half: the records' ranks and weights to memory.  The weights come back from the LDS geometry (still this item's); where
// clamping makes two taps the same texel (equal offsets) their weights are folded into the first, as in the single-wave kernel.
__device__ __forceinline__ void dec_records_end(const BwdK& P, const DecRecords& R, int lane, const float* geo_lds, unsigned n_views) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const unsigned rank = R.rank[p] + (unsigned)__shfl((int)R.base[p], R.first[p] & 63);
        const size_t slot = (size_t)p * ((size_t)n_views * P.t_count * P.S * 64) + R.idx;
        P.binrank[slot] = make_uint2(R.bin[p], rank);
        if (R.bin[p] != KEY_INVALID) {
            const uint4 off = *reinterpret_cast<const uint4*>(geo_lds + lane * 24 + 4 * p);
            const float4 w = *reinterpret_cast<const float4*>(geo_lds + lane * 24 + 12 + 4 * p);
            float w0 = w.x, w1 = w.y, w2 = w.z, w3v = w.w;
            if (off.y == off.x) { w0 += w1; w2 += w3v; w1 = 0.0f; w3v = 0.0f; }
            if (off.z == off.x) { w0 += w2; w1 += w3v; w2 = 0.0f; w3v = 0.0f; }
            P.rec_key[slot] = make_uint2(R.idx, R.loc[p]);
            P.rec_w[slot] = make_float4(w0, w1, w2, w3v);
        }
    }
}

// The producer wave of a pair.  The stage pipeline does not drain between items: once the last stage of an item is consumed and its
// records are written (its geometry in LDS is dead then) the next item's geometry is staged, its rank atomics and its first two
// stages are issued.  Returns false when a
// hand-off wait was abandoned.
template <bool DO_G, bool DO_A, bool AFF>
__device__ __forceinline__ bool dec_producer(const BwdK& P, unsigned first, unsigned step, unsigned n_items, unsigned n_views, int lane,
                                             float* feat, float* cot, unsigned* flags, float* geo_lds, float* aff_lds) {
    if (first >= n_items) return true;
    unsigned q = 0;                      // plane sets handed over so far
    bool alive = true;
    int aff_view = -1;
    float4 buf[DEC_RING][12], acc[8];
    DecItem it = dec_item(P, first, lane);
    float t_sample = P.rec_t[it.rec], gsig = P.rec_sig[it.rec], omega = P.rec_a[it.rec];
    float nx_t = 0.0f, nx_sig = 0.0f, nx_om = 0.0f;          // the next item's records, one item ahead
    if (first + step < n_items) {
        const DecItem n1 = dec_item(P, first + step, lane);
        nx_t = P.rec_t[n1.rec]; nx_sig = P.rec_sig[n1.rec]; nx_om = P.rec_a[n1.rec];
    }
    DecRecords R;
    dec_geometry(P, it, t_sample, lane, geo_lds, aff_lds, aff_view, R);
    NFE_STAGE_FENCE();
    {
        const ProdCtx c0{P, it.n, it.m, pinned(lane), gsig, omega, feat, cot, flags, geo_lds, aff_lds};
        prod_issue<ProdStage<DO_G, DO_A, 0>::SET, 0>(c0, buf[0]);
        if constexpr (DEC_RING > 2) prod_issue<ProdStage<DO_G, DO_A, 1>::SET, 1>(c0, buf[1]);
    }
#pragma unroll 1
    for (unsigned item = first; item < n_items; item += step) {
        // (the lane index laundered per item: the unrolled stages derive some two hundred per-lane LDS and tile addresses from it, which
        // loop-invariant code motion would otherwise compute once in front of the item loop - and spill, 2 KB per lane)
        const ProdCtx ctx{P, it.n, it.m, pinned(lane), gsig, omega, feat, cot, flags, geo_lds, aff_lds};
        alive = prod_run<DO_G, DO_A, AFF, 0>(ctx, buf, acc, q, alive);
        dec_records_end(P, R, lane, geo_lds, n_views);
        NFE_STAGE_FENCE();
        if (item + step < n_items) {
            it = dec_item(P, item + step, lane);
            t_sample = nx_t; gsig = nx_sig; omega = nx_om;
            if (item + 2 * step < n_items) {
                const DecItem n2 = dec_item(P, item + 2 * step, lane);
                nx_t = P.rec_t[n2.rec]; nx_sig = P.rec_sig[n2.rec]; nx_om = P.rec_a[n2.rec];
            }
            dec_geometry(P, it, t_sample, lane, geo_lds, aff_lds, aff_view, R);
            NFE_STAGE_FENCE();
            const ProdCtx cn{P, it.n, it.m, pinned(lane), gsig, omega, feat, cot, flags, geo_lds, aff_lds};
            prod_issue<ProdStage<DO_G, DO_A, 0>::SET, 0>(cn, buf[0]);
            if constexpr (DEC_RING > 2) prod_issue<ProdStage<DO_G, DO_A, 1>::SET, 1>(cn, buf[1]);
            NFE_STAGE_FENCE();
        }
    }
    return alive;
}

__global__ __launch_bounds__(128 * DEC_PAIRS) __attribute__((amdgpu_waves_per_eu(2, 2))) void bwd_decoder_kernel(BwdK P, unsigned n_items, unsigned n_views) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nfe_bwd_lds[];
    {   // the fragment image and the biases, once per workgroup; hand-off counters to zero
        uint4* sf = reinterpret_cast<uint4*>(nfe_bwd_lds);
        for (int i = threadIdx.x; i < BF_COUNT * 64; i += 128 * DEC_PAIRS) sf[i] = P.bfrag[i];
        float* sb = reinterpret_cast<float*>(nfe_bwd_lds + BWD_LDS_BIAS);
        const int i = threadIdx.x;
        if (i < 160) sb[i] = P.dec[i < 64 ? BB_G0 + i : (i < 128 ? BB_A0 + i - 64 : BB_A1 + i - 128)];
        if (i < 4 * DEC_PAIRS) *reinterpret_cast<unsigned*>(nfe_bwd_lds + BWD_LDS_TILES + (i >> 2) * DEC_PAIR_BYTES + 2 * DEC_TILE_BYTES + (i & 3) * 4) = 0u;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave_in_wg & (DEC_PAIRS - 1);
    const bool producer = wave_in_wg < DEC_PAIRS;                 // waves w and w + 4 share a SIMD (MI355X_MICROARCH.md)
    float* feat = reinterpret_cast<float*>(nfe_bwd_lds + BWD_LDS_TILES + pair * DEC_PAIR_BYTES);
    float* cot = feat + 64 * DEC_TILE_STRIDE;
    unsigned* flags = reinterpret_cast<unsigned*>(cot + 64 * DEC_TILE_STRIDE);       // {published, released, abort, -}
    float* geo_lds = reinterpret_cast<float*>(flags + 4);
    float* aff_lds = geo_lds + DEC_GEO_BYTES / 4;
    const bool do_g = P.grad_g != nullptr, do_a = P.grad_a != nullptr && P.g_rgb != nullptr;
    const unsigned first = blockIdx.x * DEC_PAIRS + (unsigned)pair, step = gridDim.x * DEC_PAIRS;
    unsigned q = 0;                      // plane sets handed over so far
    bool alive = true;
    if (producer) {
        // AFF = false: no appearance statistics on either plane set (planes edited directly: the common case of the backward) - the
        // tap sums then skip the scale / shift arithmetic, a fifth of the producer's vector instructions
        const bool aff_any = sgpr_nonnull(P.aff[0]) != 0 || sgpr_nonnull(P.aff[2]) != 0;
#define NFE_DEC_PRODUCER(G_, A_) (aff_any ? dec_producer<G_, A_, true>(P, first, step, n_items, n_views, lane, feat, cot, flags, geo_lds, aff_lds) \
                                          : dec_producer<G_, A_, false>(P, first, step, n_items, n_views, lane, feat, cot, flags, geo_lds, aff_lds))
        if (do_g && do_a) alive = NFE_DEC_PRODUCER(true, true);
        else if (do_g) alive = NFE_DEC_PRODUCER(true, false);
        else alive = NFE_DEC_PRODUCER(false, true);
#undef NFE_DEC_PRODUCER
    } else {
        const FragLds F{(unsigned)lane * 16u};
        const int jj = lane & 31, hh = lane >> 5;
        const int s8 = lane >> 3, c4 = (lane & 7) * 4;
        // a plane set's half of the item's 64 df rows (eight rows of 128 B per store instruction); the tile is released before the stores
        auto half_rows = [&](float* __restrict__ dst, int col, bool have) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = have ? *reinterpret_cast<const float4*>(feat + (8 * i + s8) * DEC_TILE_STRIDE + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (have) dec_signal(flags, 1, ++q, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(dst + (8 * i + s8) * 64 + col + c4) = v[i];
        };
#pragma unroll 1
        for (unsigned item = first; item < n_items; item += step) {
            const DecItem it = dec_item(P, item, lane);
            float* dst = P.df + (size_t)it.slot * 4096;
            f32x2 f[16];
            if (do_g) {
                if (alive) alive = dec_wait(flags, 0, q + 1);
                tile_row<DEC_TILE_STRIDE>(feat, lane, 0, f);
                head_mfma<false>(F, f, lane, feat, DEC_TILE_STRIDE, 0,
                                 [&](int b, float (&d)[8]) {          // outputs 8h..8h+7 of sample 32b + j, as the producer laid them out
                                     const float4 c0 = *reinterpret_cast<const float4*>(cot + (32 * b + jj) * DEC_TILE_STRIDE + 8 * hh);
                                     const float4 c1 = *reinterpret_cast<const float4*>(cot + (32 * b + jj) * DEC_TILE_STRIDE + 8 * hh + 4);
                                     d[0] = c0.x; d[1] = c0.y; d[2] = c0.z; d[3] = c0.w; d[4] = c1.x; d[5] = c1.y; d[6] = c1.z; d[7] = c1.w;
                                 },
                                 [](int, f32x16&) {});
                __builtin_amdgcn_wave_barrier();
                half_rows(dst, 0, true);
            } else {
                half_rows(dst, 0, false);
            }
            if (do_a) {
                if (alive) alive = dec_wait(flags, 0, q + 1);
                tile_row<DEC_TILE_STRIDE>(feat, lane, 0, f);
                head_mfma<true>(F, f, lane, feat, DEC_TILE_STRIDE, 0, [](int, float (&)[8]) {},
                                [&](int b, f32x16& y) {               // rgb = sigmoid(y) * 1.002 - 0.001 (triplane.py:269), channel 16h + r
                                    const float* cr = cot + (32 * b + jj) * DEC_TILE_STRIDE + 16 * hh;
#pragma unroll
                                    for (int qq = 0; qq < 4; ++qq) {
                                        const float4 cvv = *reinterpret_cast<const float4*>(cr + 4 * qq);
                                        const float c4v[4] = {cvv.x, cvv.y, cvv.z, cvv.w};
#pragma unroll
                                        for (int u = 0; u < 4; ++u) {
                                            const float sg = sigmoid_t(y[4 * qq + u]);
                                            y[4 * qq + u] = c4v[u] * 1.002f * sg * (1.0f - sg);
                                        }
                                    }
                                });
                __builtin_amdgcn_wave_barrier();
                half_rows(dst, 32, true);
            } else {
                half_rows(dst, 32, false);
            }
        }
    }
    if (!alive && lane == 0) atomicAdd(P.abort_word, 1u);       // either role: a wave that abandoned a wait (or found its pair marked) reports it
}

// ------------------------------------------------------------------------------------------------------------
// pass 3, binned form (default).  The memory-side float atomics bound the sorted form (4.6 M atomic texel rows per 1.6 M
// samples); here ONE WAVE owns an 8 x 8 texel tile of one plane of one view and adds every tap that lands on it into its own
// LDS tile (both plane sets: 9 x 9 texels x 64 channels = 20 KiB, seven waves per CU) with plain read-modify-writes - LDS
// executes a wave's operations in order and nobody else writes the tile; ds_add_f32 is not an option: measured at one lane per
// 12 cycles (768 cycles per wave instruction, tools/microbench/valu_rate.hip) - then adds the tile to the gradient planes once.
//   bwd_scatter_sorted_kernel<MFMA, BINNED>  decoder backward per sample -> df[sample][64] (256 B rows) + per (sample, plane) a
//                                            record (sample, four tile-local texels, 4 weights), its bin and its rank in the bin
//                                            (one returning atomic per distinct bin of a wave)
//   bwd_bin_scan_kernel                      exclusive scan of the bin counts
//   bwd_bin_fill_kernel                      sorted list of record indices (offset of the bin + rank)
//   bwd_accumulate_kernel                    one wave per (bin, segment): 64 records at a time (one per lane), then per record
//                                            one 256-byte row load (lane = channel of both sets, 16 in flight) and 4 LDS updates
// Work is cut into chunks (whole views, or ray tiles of one view) so that df stays within BWD_CHUNK_SAMPLES * 256 B.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bwd_bin_scan_kernel(const unsigned* __restrict__ counts, unsigned* __restrict__ offsets, int nbins) {
    __shared__ unsigned part[1024];
    const int per = (nbins + 1023) / 1024;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, nbins);
    unsigned sum = 0;
    for (int b = b0; b < b1; ++b) sum += counts[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - sum;
    for (int b = b0; b < b1; ++b) { offsets[b] = run; run += counts[b]; }
}

__global__ __launch_bounds__(256) void bwd_bin_fill_kernel(BwdK P, unsigned long long slots) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= slots) return;
    const uint2 br = P.binrank[i];
    if (br.x == KEY_INVALID) return;
    P.perm[P.offsets[br.x] + br.y] = (unsigned)i;      // the records stay where they were written; only this index list is sorted
}

__global__ __launch_bounds__(64) void bwd_accumulate_kernel(BwdK P) {
    __shared__ float acc[BIN_TEXELS * BIN_TEXELS * 64];      // [texel][channel of both sets]; this wave is the only writer: no LDS atomics
    const unsigned bin = blockIdx.x;                         // (ds_add_f32 runs at one lane per 12 cycles on gfx950, profiles/r02_valu_rate.txt)
    const unsigned count = P.counts[bin];
    const unsigned seg_len = max((unsigned)BIN_SEGMENT, (count + gridDim.y - 1) / gridDim.y);
    const unsigned seg0 = blockIdx.y * seg_len;
    if (seg0 >= count) return;
    const unsigned seg1 = min(seg0 + seg_len, count);
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < BIN_TEXELS * BIN_TEXELS; ++i) acc[i * 64 + lane] = 0.0f;
    const unsigned first = P.offsets[bin] + seg0, last = P.offsets[bin] + seg1;
    const float* __restrict__ df = P.df + lane;
    float* accl = acc + lane;
    // 64 records at a time, one per lane (lanes past the end: the last record with zero weights); per record one 256-byte row
    // load, lane = channel.  Two-deep pipeline: the next 64 records and the next 16 rows are in flight while 16 rows are added.
    auto load_keys = [&](unsigned r0, uint2& key, float4& wq) {
        const bool in = r0 + (unsigned)lane < last;
        const unsigned slot = P.perm[in ? r0 + (unsigned)lane : last - 1];
        key = P.rec_key[slot];
        wq = P.rec_w[slot];
        if (!in) wq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    };
    uint2 key; float4 wq;
    load_keys(first, key, wq);
    for (unsigned r0 = first; r0 < last; r0 += 64) {
        uint2 nkey; float4 nwq;
        load_keys(min(r0 + 64, last - 1), nkey, nwq);
        float va[BIN_BATCH], vb[BIN_BATCH];      // ping-pong (no copies: a register move would wait for the load it copies)
        auto fetch = [&](float (&v)[BIN_BATCH], int i0) {
#pragma unroll
            for (int u = 0; u < BIN_BATCH; ++u) v[u] = df[(size_t)(unsigned)__builtin_amdgcn_readlane((int)key.x, i0 + u) * 64];
        };
        auto add = [&](const float (&v)[BIN_BATCH], int i0) {
#pragma unroll
            for (int u = 0; u < BIN_BATCH; ++u) {
                const int i = i0 + u;
                float* a = accl + ((unsigned)__builtin_amdgcn_readlane((int)key.y, i) << 6);
                const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wq.x), i));
                const float w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wq.y), i));
                const float w2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wq.z), i));
                const float w3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wq.w), i));
                const float c0 = a[0], c1 = a[64], c2 = a[BIN_TEXELS * 64], c3 = a[BIN_TEXELS * 64 + 64];      // four different texels
                a[0] = fmaf(w0, v[u], c0);
                a[64] = fmaf(w1, v[u], c1);
                a[BIN_TEXELS * 64] = fmaf(w2, v[u], c2);
                a[BIN_TEXELS * 64 + 64] = fmaf(w3, v[u], c3);
            }
        };
        static_assert(BIN_BATCH == 16, "the 64 records of a group are four batches");
        const int cnt = (int)min(64u, last - r0);
        fetch(va, 0);
        if (cnt > 16) fetch(vb, 16);
        __builtin_amdgcn_sched_barrier(0);      // keep the loads ahead of the batch they overlap with
        add(va, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (cnt > 32) fetch(va, 32);
        __builtin_amdgcn_sched_barrier(0);
        if (cnt > 16) add(vb, 16);
        __builtin_amdgcn_sched_barrier(0);
        if (cnt > 48) fetch(vb, 48);
        __builtin_amdgcn_sched_barrier(0);
        if (cnt > 32) add(va, 32);
        __builtin_amdgcn_sched_barrier(0);
        if (cnt > 48) add(vb, 48);
        key = nkey; wq = nwq;
    }
    // tile -> gradient planes: lanes 0..31 the geometry set, 32..63 the appearance set, one 128-byte row each per atomic
    const unsigned bins_per_plane = (unsigned)(P.bins_x * P.bins_y);
    const unsigned vp = bin / bins_per_plane, tb = bin - vp * bins_per_plane;
    const int n = P.n0 + (int)(vp / 3), p = (int)(vp % 3);
    const int ty = (int)(tb / P.bins_x), tx = (int)(tb - ty * P.bins_x);
    const int ch = lane & 31, set = lane >> 5;
    float* g = set ? ((P.grad_a && P.g_rgb) ? P.grad_a : nullptr) : P.grad_g;
    const float* scale = P.aff[2 * set] ? P.aff[2 * set] + n * 96 : nullptr;
    const float sc = *P.abort_word != 0u ? __builtin_nanf("") : (scale ? scale[p * 32 + ch] : 1.0f);       // see bwd_accumulate_reg_kernel
    if (g) g += (long long)n * P.grad_view_stride + (long long)p * P.H * P.W * 32 + ch;
#pragma unroll 1
    for (int ly = 0; ly < BIN_TEXELS; ++ly) {
        const int y = (ty << BIN_SHIFT) + ly;
        if (y >= P.H) break;
#pragma unroll
        for (int lx = 0; lx < BIN_TEXELS; ++lx) {
            const int x = (tx << BIN_SHIFT) + lx;
            if (x >= P.W) break;
            const float v = g ? acc[((ly * BIN_TEXELS + lx) << 6) + lane] : 0.0f;
            if (v != 0.0f) unsafeAtomicAdd(g + ((long long)y * P.W + x) * 32, v * sc);
        }
    }
}

// The same pass with the tile in REGISTERS (default).  The LDS form above spends ~330 cycles per record on a dependent chain
// (LDS read latency -> fma -> write, per record, seven waves per CU).  With lane = channel the 9 x 9 tile is 81 values per lane:
// it lives in v[ACC_TB .. ACC_TB+80], a register range the compiler cannot allocate (amdgpu_num_vgpr caps its own use below
// ACC_TB), and a record's four taps are four v_fma_f32 whose destination / addend registers are indexed by the wave-uniform
// texel through the VGPR index mode (s_set_gpr_idx_on: M0 = tile-local texel, applied to vdst and src2 only): the taps sit at
// fixed offsets 0, 1, 9, 10 from the first, so one index serves all four.  No LDS, no memory latency in the chain: six
// instructions per record after the five v_readlane that fetch its key and weights.
#define ACC_TB 80                      // first tile register
#define ACC_TOP "v160"                 // ACC_TB + 80: named once as a clobber so that the kernel's register count covers the tile
#define ACC_STR2(x) #x
#define ACC_STR(x) ACC_STR2(x)
#define ACC_V(k) "v[" ACC_STR(ACC_TB) "+" k "]"
static_assert(BIN_TEXELS == 9, "tap offsets 0, 1, 9, 10 below");

// M0 discipline: s_set_gpr_idx_on / _idx write M0[7:0] and the asm statements below consume it inside the SAME
// statement, so nothing of ours is live in M0 across a statement boundary; the "m0" clobber tells the compiler that whatever IT
// kept there is gone (hipcc treats M0 as reserved and answers with -Winline-asm "clobber list contains reserved registers", hence
// the pragmas).  tests/test_lint_cpu.py::test_accumulate_reg_kernel_register_contract checks the kernel's ISA: no M0 use and no
// index-mode instruction outside these statements, no scratch, no spills, and the register count that covers the tile.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void acc_tile_zero() {
    // "clobber list contains reserved registers": reserving them is the point
    asm volatile(".set nfe_i, 0\n.rept 81\nv_mov_b32 v[" ACC_STR(ACC_TB) "+nfe_i], 0\n.set nfe_i, nfe_i+1\n.endr" ::: ACC_TOP);
}
__device__ __forceinline__ void acc_tile_add(unsigned texel, float w0, float w1, float w2, float w3, float row) {
    asm volatile("s_set_gpr_idx_on %0, 0xc\n" "s_nop 3\n"                       // 0xc: index vdst and src2; wait states: see ACC_NOP_A
                 "v_fma_f32 " ACC_V("0") ", %1, %5, " ACC_V("0") "\n"
                 "v_fma_f32 " ACC_V("1") ", %2, %5, " ACC_V("1") "\n"
                 "v_fma_f32 " ACC_V("9") ", %3, %5, " ACC_V("9") "\n"
                 "v_fma_f32 " ACC_V("10") ", %4, %5, " ACC_V("10") "\n"
                 "s_set_gpr_idx_off" :: "s"(texel), "s"(w0), "s"(w1), "s"(w2), "s"(w3), "v"(row) : "m0");
}
// four records under one index-mode window (s_set_gpr_idx_idx moves the index; the mode switch is paid once per four)
__device__ __forceinline__ void acc_tile_add4(const unsigned (&t)[4], const float (&w)[4][4], const float (&r)[4]) {
#define ACC_FMA4(T, A, B, C, D, R) \
                 "v_fma_f32 " ACC_V("0") ", " A ", " R ", " ACC_V("0") "\n" \
                 "v_fma_f32 " ACC_V("1") ", " B ", " R ", " ACC_V("1") "\n" \
                 "v_fma_f32 " ACC_V("9") ", " C ", " R ", " ACC_V("9") "\n" \
                 "v_fma_f32 " ACC_V("10") ", " D ", " R ", " ACC_V("10") "\n"
#define ACC_WC "s"                 // experiment: "v" keeps SGPR operands out of the index-mode VOP3
#define ACC_NOP_A "s_nop 3\n"    // MODE) and the first indexed VALU.  Without them that VALU occasionally ran with the stale index or
                             // mode: `bench.py --workload editstep` died with "Memory access fault" in 2 of 14 runs (a write through
                                 // a stale M0 lands outside the wave's registers); 44 of 44 clean with them, 4 of 48 failing with the
                                 // wait states placed before the change or after s_set_gpr_idx_off instead (tools/r03_edit_rep.sh).
                                 // The compiler inserts such wait states around its own M0 users, not inside inline asm.
#define ACC_NOP_B ""            // experiment: wait states between the last indexed VALU and the next index-mode change
#define ACC_NOP_C ""            // experiment: wait states after s_set_gpr_idx_off
    asm volatile("s_set_gpr_idx_on %0, 0xc\n" ACC_NOP_A
                 ACC_FMA4("%0", "%4", "%5", "%6", "%7", "%20")
                 ACC_NOP_B "s_set_gpr_idx_idx %1\n" ACC_NOP_A
                 ACC_FMA4("%1", "%8", "%9", "%10", "%11", "%21")
                 ACC_NOP_B "s_set_gpr_idx_idx %2\n" ACC_NOP_A
                 ACC_FMA4("%2", "%12", "%13", "%14", "%15", "%22")
                 ACC_NOP_B "s_set_gpr_idx_idx %3\n" ACC_NOP_A
                 ACC_FMA4("%3", "%16", "%17", "%18", "%19", "%23")
                 ACC_NOP_B "s_set_gpr_idx_off\n" ACC_NOP_C
                 :: "s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]),
                    ACC_WC(w[0][0]), ACC_WC(w[0][1]), ACC_WC(w[0][2]), ACC_WC(w[0][3]), ACC_WC(w[1][0]), ACC_WC(w[1][1]), ACC_WC(w[1][2]), ACC_WC(w[1][3]),
                    ACC_WC(w[2][0]), ACC_WC(w[2][1]), ACC_WC(w[2][2]), ACC_WC(w[2][3]), ACC_WC(w[3][0]), ACC_WC(w[3][1]), ACC_WC(w[3][2]), ACC_WC(w[3][3]),
                    "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]) : "m0");
#undef ACC_FMA4
}
#pragma clang diagnostic pop
template <int K> __device__ __forceinline__ float acc_tile_get() {
    float v;
    asm volatile("v_mov_b32 %0, v[" ACC_STR(ACC_TB) "+%1]" : "=v"(v) : "n"(K));
    return v;
}
template <int K> __device__ __forceinline__ void acc_tile_flush(float* g, float sc, int y0, int x0, int H, int W) {
    constexpr int ly = K / BIN_TEXELS, lx = K % BIN_TEXELS;
    const float v = acc_tile_get<K>();
    if (y0 + ly < H && x0 + lx < W && v != 0.0f) unsafeAtomicAdd(g + ((long long)(y0 + ly) * W + (x0 + lx)) * 32, v * sc);
    if constexpr (K + 1 < BIN_TEXELS * BIN_TEXELS) acc_tile_flush<K + 1>(g, sc, y0, x0, H, W);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(ACC_TB))) void bwd_accumulate_reg_kernel(BwdK P) {
    const unsigned bin = blockIdx.x;
    const unsigned count = P.counts[bin];
    const unsigned seg_len = max((unsigned)BIN_SEGMENT, (count + gridDim.y - 1) / gridDim.y);
    const unsigned seg0 = blockIdx.y * seg_len;
    if (seg0 >= count) return;
    const unsigned seg1 = min(seg0 + seg_len, count);
    const int lane = threadIdx.x;
    acc_tile_zero();
    const unsigned first = P.offsets[bin] + seg0, last = P.offsets[bin] + seg1;
    const float* __restrict__ df = P.df + lane;
    auto load_keys = [&](unsigned r0, uint2& key, float4& wq) {
        const bool in = r0 + (unsigned)lane < last;
        const unsigned slot = P.perm[in ? r0 + (unsigned)lane : last - 1];
        key = P.rec_key[slot];
        wq = P.rec_w[slot];
        if (!in) wq = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    };
    uint2 key; float4 wq;
    load_keys(first, key, wq);
    for (unsigned r0 = first; r0 < last; r0 += 64) {
        uint2 nkey; float4 nwq;
        load_keys(min(r0 + 64, last - 1), nkey, nwq);
        float va[BIN_BATCH], vb[BIN_BATCH];
        auto fetch = [&](float (&v)[BIN_BATCH], int i0) {
#pragma unroll
            for (int u = 0; u < BIN_BATCH; ++u) {
                v[u] = df[(size_t)(unsigned)__builtin_amdgcn_readlane((int)key.x, i0 + u) * 64];
            }
        };
        auto rl = [&](float x, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), i)); };
        auto add = [&](const float (&v)[BIN_BATCH], int i0) {
#pragma unroll
            for (int u = 0; u < BIN_BATCH; u += 4) {
                unsigned t[4]; float w[4][4], r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + u + k;
                    t[k] = (unsigned)__builtin_amdgcn_readlane((int)key.y, i);
                    w[k][0] = rl(wq.x, i); w[k][1] = rl(wq.y, i); w[k][2] = rl(wq.z, i); w[k][3] = rl(wq.w, i);
                    r[k] = v[u + k];
                }
                acc_tile_add4(t, w, r);
            }
        };
        const int cnt = (int)min(64u, last - r0);
        fetch(va, 0);
        if (cnt > 16) fetch(vb, 16);
        __builtin_amdgcn_sched_barrier(0);
        add(va, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (cnt > 32) fetch(va, 32);
        __builtin_amdgcn_sched_barrier(0);
        if (cnt > 16) add(vb, 16);
        __builtin_amdgcn_sched_barrier(0);
        if (cnt > 48) fetch(vb, 48);
        __builtin_amdgcn_sched_barrier(0);
        if (cnt > 32) add(va, 32);
        __builtin_amdgcn_sched_barrier(0);
        if (cnt > 48) add(vb, 48);
        key = nkey; wq = nwq;
    }
    const unsigned bins_per_plane = (unsigned)(P.bins_x * P.bins_y);
    const unsigned vp = bin / bins_per_plane, tb = bin - vp * bins_per_plane;
    const int n = P.n0 + (int)(vp / 3), p = (int)(vp % 3);
    const int ty = (int)(tb / P.bins_x), tx = (int)(tb - ty * P.bins_x);
    const int ch = lane & 31, set = lane >> 5;
    float* g = set ? ((P.grad_a && P.g_rgb) ? P.grad_a : nullptr) : P.grad_g;
    const float* scale = P.aff[2 * set] ? P.aff[2 * set] + n * 96 : nullptr;
    // a decoder-backward launch that lost a hand-off (bwd_decoder_kernel) must not pass for a result: every gradient it touches becomes NaN
    const float sc = *P.abort_word != 0u ? __builtin_nanf("") : (scale ? scale[p * 32 + ch] : 1.0f);
    if (!g) return;
    g += (long long)n * P.grad_view_stride + (long long)p * P.H * P.W * 32 + ch;
    acc_tile_flush<0>(g, sc, ty << BIN_SHIFT, tx << BIN_SHIFT, P.H, P.W);
}

static uint64_t align256(uint64_t x) { return (x + 255) & ~uint64_t(255); }

// Chunking of the binned scatter: sample slots (64 per wave, whole 64-ray tiles x all depths) of the largest chunk
static uint64_t chunk_limit() {
    static const uint64_t v = [] { const char* e = getenv("NFE_BWD_CHUNK"); const long long x = e ? atoll(e) : 0; return x > 0 ? (uint64_t)x : BWD_CHUNK_SAMPLES; }();
    return v;
}
static uint64_t chunk_slots(int n_views, int n_rays, int n_samples) {
    if (n_views <= 0 || n_rays <= 0 || n_samples <= 0) return 0;
    const uint64_t tiles = ((uint64_t)n_rays + 63) / 64, per_tile = 64ull * (uint64_t)n_samples, per_view = tiles * per_tile;
    const uint64_t cap = chunk_limit() > per_tile ? chunk_limit() : per_tile;
    if (per_view >= cap) return (cap / per_tile) * per_tile;                            // ray tiles of one view
    const uint64_t views = cap / per_view < (uint64_t)n_views ? cap / per_view : (uint64_t)n_views;
    return views * per_view;
}
// floats of one array of BwdK::til: whole 64-ray tiles
static uint64_t tiled_floats(int n_views, int n_rays, int n_samples) {
    if (n_views <= 0 || n_rays <= 0 || n_samples <= 0) return 0;
    return (uint64_t)n_views * (((uint64_t)n_rays + 63) / 64) * 64ull * (uint64_t)n_samples;
}
// df rows + (key, weights) records + (bin, rank) + sorted record indices + counts and offsets
static uint64_t binned_bytes(uint64_t slots) {
    return align256(slots * 256) + align256(slots * 3 * 8) + align256(slots * 3 * 16) + align256(slots * 3 * 8) + align256(slots * 3 * 4) +
           2 * align256(BWD_MAX_BINS * 4);
}

// Closes nfe_render_backward's binned form (the one with a hand-off): see the launch.
struct BwdPoison { const unsigned* abort_word; unsigned long long* host_status; float* g[2]; int sets; long long view_stride; unsigned long long per_set; };
__global__ __launch_bounds__(256) void bwd_poison_kernel(BwdPoison Z) {
    const unsigned lost = *Z.abort_word;
    if (lost == 0u) return;
    const float bad = __uint_as_float(0x7fc00000u);
    const unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (unsigned long long)gridDim.x * blockDim.x;
    for (int b = 0; b < 2; ++b)
        if (Z.g[b])
            for (int v = 0; v < Z.sets; ++v)
                for (unsigned long long i = i0; i < Z.per_set; i += stride) Z.g[b][(long long)v * Z.view_stride + (long long)i] = bad;
    if (i0 == 0 && Z.host_status)
        __hip_atomic_fetch_add(Z.host_status, (1ull << 32) | (unsigned long long)lost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace nfe

using namespace nfe;

// NFE_WS_SPIN_LIMIT (read once; the same variable as the render kernel's) into g_dec_spin_limit, once per device and only when set
static int apply_dec_spin_limit() {
    static const int v = [] { const char* e = getenv("NFE_WS_SPIN_LIMIT"); const long long x = e ? atoll(e) : 0; return x > 0 && x < (1ll << 30) ? (int)x : 0; }();
    if (!v) return NFE_OK;
    static std::atomic<unsigned long long> done{0};
    const int dev = current_device();
    if (dev < MAX_DEVICES && (done.load(std::memory_order_acquire) >> dev & 1ull)) return NFE_OK;
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_dec_spin_limit), &v, sizeof(int));
    if (e != hipSuccess) return fail(NFE_ELAUNCH, "NFE_WS_SPIN_LIMIT: hipMemcpyToSymbol: %s", hipGetErrorString(e));
    if (dev < MAX_DEVICES) done.fetch_or(1ull << dev, std::memory_order_release);
    return NFE_OK;
}

extern "C" uint64_t nfe_render_backward_workspace_bytes(int n_views, int n_rays, int n_samples) {
    return 256 /* word 0: the call's hand-off abort count (nfe_render_backward_call_status) */ + BWD_DEC_BYTES + 4 * align256(tiled_floats(n_views, n_rays, n_samples) * 4) + align256((uint64_t)NFE_DECODER_PACKED_FLOATS * 4) + align256(BWD_FRAG_BYTES) +
           binned_bytes(chunk_slots(n_views, n_rays, n_samples));
}

extern "C" int nfe_render_backward(const nfe_render_backward_args* a, nfe_stream_t stream) {
    NFE_REQUIRE(a != nullptr, "nfe_render_backward: args is null");
    NFE_REQUIRE(a->struct_size == sizeof(nfe_render_backward_args), "nfe_render_backward: struct_size %u != %zu (ABI mismatch)",
                a->struct_size, sizeof(nfe_render_backward_args));
    NFE_REQUIRE(a->planes_geo && a->planes_app, "nfe_render_backward: plane pointers are null");
    NFE_REQUIRE(a->geo_w0 && a->geo_b0 && a->geo_w1 && a->geo_b1 && a->app_w0 && a->app_b0 && a->app_w1 && a->app_b1,
                "nfe_render_backward: decoder parameter pointers are null");
    NFE_REQUIRE(a->plane_h > 0 && a->plane_w > 0 && (long long)a->plane_h * a->plane_w * 96 < (1ll << 31),
                "nfe_render_backward: bad plane size %dx%d (element offsets inside a view's plane set are 32-bit)", a->plane_h, a->plane_w);
    NFE_REQUIRE(a->n_views > 0 && a->n_views <= 65535 && a->n_rays > 0, "nfe_render_backward: n_views=%d n_rays=%d out of range", a->n_views, a->n_rays);
    NFE_REQUIRE(a->n_samples >= 2 && a->n_samples <= 2 * NFE_MAX_SAMPLES, "nfe_render_backward: n_samples=%d out of [2,%d]", a->n_samples, 2 * NFE_MAX_SAMPLES);
    NFE_REQUIRE(a->depths != nullptr, "nfe_render_backward: depths is null");
    NFE_REQUIRE((a->origins != nullptr) == (a->dirs != nullptr), "nfe_render_backward: origins and dirs must both be given or both null");
    if (!a->origins) {
        NFE_REQUIRE(a->cam2world && a->intrinsics, "nfe_render_backward: need origins/dirs or cam2world/intrinsics");
        NFE_REQUIRE(a->resolution > 0 && (long long)a->resolution * a->resolution == a->n_rays,
                    "nfe_render_backward: resolution^2 (%d^2) != n_rays (%d)", a->resolution, a->n_rays);
    }
    NFE_REQUIRE((a->geo_scale != nullptr) == (a->geo_shift != nullptr) && (a->app_scale != nullptr) == (a->app_shift != nullptr),
                "nfe_render_backward: affine scale/shift must come in pairs");
    NFE_REQUIRE(a->box_warp > 0.0f, "nfe_render_backward: box_warp must be positive");
    NFE_REQUIRE(a->grad_planes_geo || a->grad_planes_app, "nfe_render_backward: no gradient output requested");
    NFE_REQUIRE(a->workspace != nullptr, "nfe_render_backward: workspace is null");
    const uint64_t need = nfe_render_backward_workspace_bytes(a->n_views, a->n_rays, a->n_samples);
    if (a->workspace_bytes < need) return fail(NFE_EWORKSPACE, "nfe_render_backward: workspace %llu < %llu bytes",
                                               (unsigned long long)a->workspace_bytes, (unsigned long long)need);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)a->workspace;
    unsigned* abort_word = (unsigned*)ws; ws += 256;       // zeroed now, whichever form runs: the per-call status reads it
    if (hipMemsetAsync(abort_word, 0, 4, st) != hipSuccess) return fail(NFE_ELAUNCH, "nfe_render_backward: hipMemsetAsync failed");
    float* dec = (float*)ws; ws += BWD_DEC_BYTES;
    PrepK Q{};
    Q.w[0] = a->geo_w0; Q.w[1] = a->geo_b0; Q.w[2] = a->geo_w1; Q.w[3] = a->geo_b1;
    Q.w[4] = a->app_w0; Q.w[5] = a->app_b0; Q.w[6] = a->app_w1; Q.w[7] = a->app_b1;
    Q.lr_mul = a->lr_mul; Q.out = dec;
    hipLaunchKernelGGL(bwd_prep_kernel, dim3((BWD_DEC_FLOATS + 255) / 256), dim3(256), 0, st, Q);
    NFE_CHECK_LAUNCH("bwd_prep_kernel");

    BwdK P{};
    P.planes_g = a->planes_geo; P.planes_a = a->planes_app; P.plane_view_stride = a->plane_view_stride;
    P.H = a->plane_h; P.W = a->plane_w;
    P.aff[0] = a->geo_scale; P.aff[1] = a->geo_shift; P.aff[2] = a->app_scale; P.aff[3] = a->app_shift;
    P.dec = dec;
    P.N = a->n_views; P.M = a->n_rays; P.R = a->resolution;
    P.origins = a->origins; P.dirs = a->dirs; P.cam2world = a->cam2world; P.intrinsics = a->intrinsics;
    P.S = a->n_samples; P.depths = a->depths; P.coord_scale = 2.0f / a->box_warp; P.white_back = a->white_back;
    P.g_rgb = a->grad_rgb; P.g_seg = a->grad_seg; P.g_depth = a->grad_depth; P.g_wsum = a->grad_wsum;
    P.channels_first = a->channels_first;
    P.grad_g = a->grad_planes_geo; P.grad_a = a->grad_planes_app; P.grad_view_stride = a->grad_view_stride;
    const uint64_t rec_bytes = align256(tiled_floats(a->n_views, a->n_rays, a->n_samples) * 4);
    P.rec_sig = (float*)ws; ws += rec_bytes;
    P.rec_a = (float*)ws; ws += rec_bytes;
    P.rec_T = (float*)ws; ws += rec_bytes;
    P.rec_t = (float*)ws;
    P.T = (a->n_rays + 63) / 64;
    static const char scatter_mode = [] { const char* e = getenv("NFE_BWD_SCATTER"); return e ? e[0] : 'b'; }();     // A/B knob: "direct", "sorted", default binned
    const bool direct = scatter_mode == 'd' || (long long)a->plane_h * a->plane_w > (1ll << 24);                    // sort keys carry a 24-bit texel index

    const long long per_view = (long long)a->n_rays * a->n_samples;
    const dim3 sgrid((unsigned)((per_view + 255) / 256), (unsigned)a->n_views);
    // pass 1: on the forward kernel's machinery (quad gather + split-bf16 MFMA decoder, nfe_render.hip) unless NFE_BWD_EVAL=valu
    static const bool valu_eval = [] { const char* e = getenv("NFE_BWD_EVAL"); return e && e[0] == 'v'; }();
    if (a->sample_colors) {             // the forward kept the decoders' outputs: no gather, no decoder here (ABI v11)
        int rc = render_color_dot_pass(a, P.rec_sig, P.rec_a, st);
        if (rc) return rc;
    } else if (valu_eval || (long long)a->plane_h * a->plane_w > (1ll << 25)) {
        hipLaunchKernelGGL(bwd_eval_kernel, sgrid, dim3(256), 0, st, P);
        NFE_CHECK_LAUNCH("bwd_eval_kernel");
    } else {
        float* packed = (float*)((char*)P.rec_t + rec_bytes);
        int rc = nfe_decoder_pack(a->geo_w0, a->geo_b0, a->geo_w1, a->geo_b1, a->app_w0, a->app_b0, a->app_w1, a->app_b1, a->lr_mul, packed, stream);
        if (rc) return rc;
        rc = render_eval_pass(a, packed, P.rec_sig, P.rec_a, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(bwd_ray_kernel, dim3((unsigned)(((long long)a->n_views * P.T + 3) / 4)), dim3(256), 0, st, P);
    NFE_CHECK_LAUNCH("bwd_ray_kernel");
    static const bool valu_dec = [] { const char* e = getenv("NFE_BWD_DECODER"); return e && e[0] == 'v'; }();      // A/B knob: "valu"
    static const bool single_wave = [] { const char* e = getenv("NFE_BWD_DECODER"); return e && e[0] == 's'; }();   // A/B knob: "single" = round 4's one-wave workgroups
    static const bool acc_lds = [] { const char* e = getenv("NFE_BWD_ACC"); return e && e[0] == 'l'; }();            // A/B knob: "lds" = tile in LDS
    unsigned* frags = (unsigned*)((char*)P.rec_t + rec_bytes + align256((uint64_t)NFE_DECODER_PACKED_FLOATS * 4));
    if (direct) {
        hipLaunchKernelGGL(bwd_scatter_kernel, sgrid, dim3(256), 0, st, P);
        NFE_CHECK_LAUNCH("bwd_scatter_kernel");
        return NFE_OK;
    }
    if (!valu_dec) {
        hipLaunchKernelGGL(bwd_frag_kernel, dim3((BF_COUNT * 64 * 4 + 255) / 256), dim3(256), 0, st, dec, frags);
        P.bfrag = reinterpret_cast<const uint4*>(frags);
    }
    const unsigned ray_tiles = (unsigned)((a->n_rays + 63) / 64);
    P.bins_x = (a->plane_w + BIN_MASK) >> BIN_SHIFT; P.bins_y = (a->plane_h + BIN_MASK) >> BIN_SHIFT;
    const uint64_t bins_per_view = 3ull * P.bins_x * P.bins_y;
    if (scatter_mode == 's' || bins_per_view > BWD_MAX_BINS) {
        const dim3 tgrid = NFE_BWD_DEPTH_FAST ? dim3((unsigned)a->n_samples, ray_tiles, (unsigned)a->n_views) : dim3(ray_tiles, (unsigned)a->n_samples, (unsigned)a->n_views);
        if (valu_dec) hipLaunchKernelGGL((bwd_scatter_sorted_kernel<false, false>), tgrid, dim3(64), 0, st, P);
        else hipLaunchKernelGGL((bwd_scatter_sorted_kernel<true, false>), tgrid, dim3(64), 0, st, P);
        NFE_CHECK_LAUNCH("bwd_scatter_sorted_kernel");
        return NFE_OK;
    }
    // binned form, chunk by chunk
    const uint64_t slots_max = chunk_slots(a->n_views, a->n_rays, a->n_samples);
    char* bw = (char*)frags + align256(BWD_FRAG_BYTES);
    P.df = (float*)bw; bw += align256(slots_max * 256);
    P.rec_key = (uint2*)bw; bw += align256(slots_max * 3 * 8);
    P.rec_w = (float4*)bw; bw += align256(slots_max * 3 * 16);
    P.binrank = (uint2*)bw; bw += align256(slots_max * 3 * 8);
    P.perm = (unsigned*)bw; bw += align256(slots_max * 3 * 4);
    P.counts = (unsigned*)bw; bw += align256(BWD_MAX_BINS * 4);
    P.offsets = (unsigned*)bw; bw += align256(BWD_MAX_BINS * 4);
    P.abort_word = abort_word;
    const uint64_t per_tile = 64ull * (uint64_t)a->n_samples, view_slots = (uint64_t)ray_tiles * per_tile;
    unsigned views_per_chunk = 1, tiles_per_chunk = ray_tiles;
    if (view_slots >= slots_max) tiles_per_chunk = (unsigned)(slots_max / per_tile);
    else views_per_chunk = (unsigned)(slots_max / view_slots);
    while ((uint64_t)views_per_chunk * bins_per_view > BWD_MAX_BINS) --views_per_chunk;
    for (unsigned n0 = 0; n0 < (unsigned)a->n_views; n0 += views_per_chunk) {
        const unsigned nv = min(views_per_chunk, (unsigned)a->n_views - n0);
        for (unsigned t0 = 0; t0 < ray_tiles; t0 += tiles_per_chunk) {
            const unsigned nt = min(tiles_per_chunk, ray_tiles - t0);
            P.n0 = (int)n0; P.t0 = (int)t0; P.t_count = (int)nt;
            const unsigned nbins = nv * (unsigned)bins_per_view;
            const unsigned long long slots = (unsigned long long)nv * nt * per_tile;
            if (hipMemsetAsync(P.counts, 0, (size_t)nbins * 4, st) != hipSuccess) return fail(NFE_ELAUNCH, "nfe_render_backward: hipMemsetAsync failed");
            const dim3 tgrid = NFE_BWD_DEPTH_FAST ? dim3((unsigned)a->n_samples, nt, nv) : dim3(nt, (unsigned)a->n_samples, nv);
            if (valu_dec) hipLaunchKernelGGL((bwd_scatter_sorted_kernel<false, true>), tgrid, dim3(64), 0, st, P);
            else if (single_wave) hipLaunchKernelGGL((bwd_scatter_sorted_kernel<true, true>), tgrid, dim3(64), 0, st, P);
            else {      // persistent workgroups of four producer / consumer wave pairs, the fragment image in LDS; items in the block order of tgrid
                if (int rc = apply_dec_spin_limit()) return rc;
                static LdsOptIn opt;
                const hipError_t e = opt.apply(bwd_decoder_kernel, DEC_LDS_BYTES);
                if (e != hipSuccess) return fail(NFE_ELAUNCH, "bwd_decoder_kernel: LDS opt-in: %s", hipGetErrorString(e));
                const unsigned long long n_items = (unsigned long long)nv * nt * (unsigned)a->n_samples;       // <= slots_max / 64 < 2^32
                const unsigned wgs = (unsigned)min((unsigned long long)num_cus(), (n_items + DEC_PAIRS - 1) / DEC_PAIRS);
                hipLaunchKernelGGL(bwd_decoder_kernel, dim3(wgs), dim3(128 * DEC_PAIRS), DEC_LDS_BYTES, st, P, (unsigned)n_items, nv);
            }
            NFE_CHECK_LAUNCH("decoder-backward kernel (binned form)");
            hipLaunchKernelGGL(bwd_bin_scan_kernel, dim3(1), dim3(1024), 0, st, P.counts, P.offsets, (int)nbins);
            hipLaunchKernelGGL(bwd_bin_fill_kernel, dim3((unsigned)((slots * 3 + 255) / 256)), dim3(256), 0, st, P, slots * 3);
            if (acc_lds) hipLaunchKernelGGL(bwd_accumulate_kernel, dim3(nbins, BIN_SPLIT), dim3(64), 0, st, P);
            else hipLaunchKernelGGL(bwd_accumulate_reg_kernel, dim3(nbins, BIN_SPLIT), dim3(64), 0, st, P);
            NFE_CHECK_LAUNCH("bwd_accumulate_kernel");
        }
    }
    // closes the call: a lost hand-off in ANY chunk poisons BOTH gradient buffers entirely (the accumulate passes of the chunks
    // before the aborting one wrote finite numbers) and is counted in the sticky status word; nothing to do otherwise
    BwdPoison Z{};
    Z.abort_word = abort_word; Z.host_status = handoff_status_word();
    Z.g[0] = a->grad_planes_geo; Z.g[1] = a->grad_planes_app == a->grad_planes_geo ? nullptr : a->grad_planes_app;
    Z.sets = a->grad_view_stride ? a->n_views : 1; Z.view_stride = a->grad_view_stride; Z.per_set = 96ull * (unsigned long long)a->plane_h * a->plane_w;
    hipLaunchKernelGGL(bwd_poison_kernel, dim3(256), dim3(256), 0, st, Z);
    NFE_CHECK_LAUNCH("bwd_poison_kernel");
    return NFE_OK;
}
