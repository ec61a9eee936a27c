// Memory-bound helper kernels around the render core (gfx950): ray sampler, plane statistics,
// per-channel affine, NCHW -> gather-layout pack, decoder weight packing.
#include "nfe_common.h"

namespace nfe {

// ---- a2: RaySampler.forward (ray_sampler.py:24-62) ------------------------------------------
__global__ void ray_sampler_kernel(const float* __restrict__ cam2world, const float* __restrict__ intrinsics,
                                   int N, int R, float* __restrict__ origins, float* __restrict__ dirs) {
    const long long M = (long long)R * R, total = (long long)N * M;
    const float inv = 1.0f / (float)R;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / M); const int m = (int)(i % M);
        const int py = m / R, px = m % R;
        const float* c = cam2world + n * 16; const float* K = intrinsics + n * 9;
        const float fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
        const float xc = (float)px * inv + 0.5f * inv, yc = (float)py * inv + 0.5f * inv;   // :41
        const float xl = (xc - cx + cy * sk / fy - sk * yc / fy) / fx;                      // :50
        const float yl = (yc - cy) / fy;                                                    // :51
        const float ox = c[3], oy = c[7], oz = c[11];
        float dx = (c[0] * xl + c[1] * yl + c[2] + c[3]) - ox;                              // :55-57
        float dy = (c[4] * xl + c[5] * yl + c[6] + c[7]) - oy;
        float dz = (c[8] * xl + c[9] * yl + c[10] + c[11]) - oz;
        const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);                // F.normalize :59
        origins[i * 3 + 0] = ox; origins[i * 3 + 1] = oy; origins[i * 3 + 2] = oz;
        dirs[i * 3 + 0] = dx / nrm; dirs[i * 3 + 1] = dy / nrm; dirs[i * 3 + 2] = dz / nrm;
    }
}

// ---- 'auto' ray limits: math_utils.get_ray_limits_box (math_utils.py:46-98) ----------------------
__global__ void ray_limits_kernel(const float* __restrict__ origins, const float* __restrict__ dirs, long long n,
                                  float half, float* __restrict__ rs, float* __restrict__ re, unsigned* minmax) {
    float lo = INFINITY, hi = -INFINITY;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float ox = origins[i * 3], oy = origins[i * 3 + 1], oz = origins[i * 3 + 2];
        const float ix = 1.0f / dirs[i * 3], iy = 1.0f / dirs[i * 3 + 1], iz = 1.0f / dirs[i * 3 + 2];
        // validity as a chain of selects, each fed by its own compare (no lane masks combined on the scalar unit: lint shape S1)
        float tmin = ((ix < 0 ? half : -half) - ox) * ix, tmax = ((ix < 0 ? -half : half) - ox) * ix;
        const float tymin = ((iy < 0 ? half : -half) - oy) * iy, tymax = ((iy < 0 ? -half : half) - oy) * iy;
        auto flag = [](bool c) { int v = c ? 1 : 0; asm volatile("; nfe_launder %0" : "+v"(v)); return v; };     // opaque: stays a per-lane integer
        int invalid = flag(tmin > tymax) | flag(tymin > tmax);
        tmin = fmaxf(tmin, tymin); tmax = fminf(tmax, tymax);          // torch.max/min propagate like fmax here
        const float tzmin = ((iz < 0 ? half : -half) - oz) * iz, tzmax = ((iz < 0 ? -half : half) - oz) * iz;
        invalid |= flag(tmin > tzmax) | flag(tzmin > tmax);
        tmin = fmaxf(tmin, tzmin); tmax = fminf(tmax, tzmax);
        tmin = invalid != 0 ? -1.0f : tmin; tmax = invalid != 0 ? -2.0f : tmax;
        rs[i] = tmin; re[i] = tmax;
        if (tmax > tmin) { lo = fminf(lo, tmin); hi = fmaxf(hi, tmin); }  // is_ray_valid = ray_end > ray_start
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off)); hi = fmaxf(hi, __shfl_xor(hi, off)); }
    if ((threadIdx.x & 63) == 0 && lo <= hi) { atomicMin(minmax, f2ord(lo)); atomicMax(minmax + 1, f2ord(hi)); }
}

__global__ void ray_limits_fix_kernel(long long n, float* __restrict__ rs, float* __restrict__ re, const unsigned* minmax) {
    if (minmax[0] == 0xFFFFFFFFu && minmax[1] == 0u) return;          // torch.any(is_ray_valid) is False
    const float lo = ord2f(minmax[0]), hi = ord2f(minmax[1]);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        if (!(re[i] > rs[i])) { rs[i] = lo; re[i] = hi; }
}

__global__ void limits_init_kernel(unsigned* minmax) { minmax[0] = 0xFFFFFFFFu; minmax[1] = 0u; }

// ---- a4: compute_mean_var (triplane.py:56-60): one block per (n,c) row of HW elements -------
__global__ __launch_bounds__(256) void plane_stats_kernel(const float* __restrict__ planes, int hw,
                                                          float* __restrict__ mean, float* __restrict__ stdv) {
    __shared__ double sh[2][4];
    const float* row = planes + (long long)blockIdx.x * hw;
    double s = 0.0, ss = 0.0;
    const int n4 = ((reinterpret_cast<uintptr_t>(row) & 15) == 0) ? hw / 4 : 0;
    for (int i = threadIdx.x; i < n4; i += 256) {
        float4 v = reinterpret_cast<const float4*>(row)[i];
        s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    for (int i = n4 * 4 + threadIdx.x; i < hw; i += 256) { double v = row[i]; s += v; ss += v * v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); ss += __shfl_xor(ss, off); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh[0][wave] = s; sh[1][wave] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        ss = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        const double mu = s / hw;
        double var = (ss - s * mu) / (double)(hw - 1);       // unbiased (torch.var default)
        if (var < 0.0) var = 0.0;
        mean[blockIdx.x] = (float)mu;
        stdv[blockIdx.x] = (float)sqrt(var);
    }
}

// ---- a4: out = in*scale + shift per (n,c) row (normalize_plane / denormalize_plane) ----------
__global__ __launch_bounds__(256) void plane_affine_kernel(const float* __restrict__ in, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int c, int hw, int n_affine,
                                                           float* __restrict__ out) {
    const int row = blockIdx.x;                 // n*c + ch
    const int n = row / c, ch = row % c;
    const int ai = (n_affine == 1 ? 0 : n) * c + ch;
    const float a = scale[ai], b = shift[ai];
    const float* src = in + (long long)row * hw; float* dst = out + (long long)row * hw;
    const bool al = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const int n4 = al ? hw / 4 : 0;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n4; i += gridDim.y * 256) {
        float4 v = reinterpret_cast<const float4*>(src)[i];
        v.x = fmaf(v.x, a, b); v.y = fmaf(v.y, a, b); v.z = fmaf(v.z, a, b); v.w = fmaf(v.w, a, b);
        reinterpret_cast<float4*>(dst)[i] = v;
    }
    for (int i = n4 * 4 + blockIdx.y * 256 + threadIdx.x; i < hw; i += gridDim.y * 256) dst[i] = fmaf(src[i], a, b);
}

// ---- a4 -> a5 affines -----------------------------------------------------------
__global__ void make_affine_kernel(const float* __restrict__ mean, const float* __restrict__ stdv,
                                   const float* __restrict__ new_mean, const float* __restrict__ new_std,
                                   int n, int c, int n_override, float* gs, float* gb, float* as, float* ab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * c) return;
    const float inv = 1.0f / (stdv[i] + 1e-8f);          // normalize_plane: (x - mean)/(std + 1e-8)
    gs[i] = inv; gb[i] = -mean[i] * inv;
    if (new_mean) {                                       // denormalize_plane(norm, new_mean, new_std)
        const int o = (n_override == 1 ? 0 : i / c) * c + i % c;
        const float sc = inv * new_std[o];
        as[i] = sc; ab[i] = new_mean[o] - mean[i] * sc;
    } else {
        as[i] = 1.0f; ab[i] = 0.0f;
    }
}

// ---- NCHW [N,96,H,W] -> [N,3,H,W,32]: LDS-tiled transpose, 32 channels x 64 pixels per block --
__global__ __launch_bounds__(256) void plane_pack_kernel(const float* __restrict__ src, int hw, float* __restrict__ dst) {
    __shared__ float tile[32][65];
    const int np = blockIdx.y;                    // n*3 + plane
    const int p0 = blockIdx.x * 64;
    const float* s = src + (long long)np * 32 * hw;
    float* d = dst + (long long)np * hw * 32;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // read: 64 pixels x 4 channels per pass
#pragma unroll
    for (int cc = 0; cc < 32; cc += 4) {
        const int pix = p0 + tx;
        tile[cc + ty][tx] = pix < hw ? s[(long long)(cc + ty) * hw + pix] : 0.0f;
    }
    __syncthreads();
    const int c = threadIdx.x & 31, pr = threadIdx.x >> 5;            // write: 32 channels x 8 pixels per pass
#pragma unroll
    for (int pp = 0; pp < 64; pp += 8) {
        const int pix = p0 + pp + pr;
        if (pix < hw) d[(long long)pix * 32 + c] = tile[c][pp + pr];
    }
}

// ---- decoder weights -> MFMA A-operand layouts ----------------------------------
// hidden unit held by (layer-0 M-block mb, accumulator register r, lane half h)
__device__ __forceinline__ int hidden_unit(int mb, int r, int h) { return 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * h; }

// MFMA output row i of the geometry head -> geo_net.2 output index (0 = sigma, 1..15 = seg), -1 = unused.
// Row i is register ri of lane half hi; sigma is duplicated into both halves.
// (as a table: the compound range tests it stands for would be combined on the scalar unit into one lane mask feeding a select -
// shape S1 of tools/lint_lane_masks.py, kept out of every kernel of this library)
__device__ const signed char GEO_ROW_TO_OUT[32] = {0, -1, 1, 2, 0, -1, 9, 10, 3, 4, 5, 6, 11, 12, 13, 14, 7, 8, -1, -1, 15, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
__device__ __forceinline__ int geo_row_to_out(int i) {
    // ri = (i & 3) + 4 * (i >> 3), hi = (i >> 2) & 1:  ri == 0 -> 0 (sigma, register 0 of both halves);
    // hi == 0, ri in 2..9 -> ri - 1 (seg 0..7, even start: packed pairs);  hi == 1, ri in 2..8 -> ri + 7 (seg 8..14);  else -1
    return GEO_ROW_TO_OUT[i & 31];
}
__device__ __forceinline__ int app_row_to_out(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }

// The hidden activation is evaluated as log2(1 + 2^y) (v_exp_f32/v_log_f32 are base 2): log2(e) is
// folded into layer 0, ln(2) into layer 1; the appearance head additionally carries log2(e) for its
// sigmoid, which cancels the ln(2).  All foldings are exact algebra on softplus/sigmoid.
struct DecSrc { const float *gw0, *gb0, *gw1, *gb1, *aw0, *ab0, *aw1, *ab1; float lr_mul; };

__device__ __forceinline__ float dec_w0(const DecSrc& S, int net, int unit, int ch) {
    return (net ? S.aw0 : S.gw0)[unit * 32 + ch] * (S.lr_mul / sqrtf(32.0f)) * LOG2E;   // weight_gain, networks_stylegan2.py:111
}
__device__ __forceinline__ float dec_w1(const DecSrc& S, int net, int row, int unit) {
    const float g1 = S.lr_mul / sqrtf(64.0f);
    if (net == 0) { const int o = geo_row_to_out(row); return o >= 0 ? S.gw1[o * 64 + unit] * g1 * LN2 : 0.0f; }
    return S.aw1[app_row_to_out(row) * 64 + unit] * g1;       // * LN2 * LOG2E == 1
}
__device__ __forceinline__ unsigned bf16_rne_bits(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__global__ void decoder_pack_kernel(DecSrc S, float* out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= DEC_TOTAL) return;
    float v = 0.0f;
    if (e < DEC_A_G1) {                       // fp32 layer-0 A fragments: [net][mb][ks4][lane][kk]
        const int net = e / 2048, r_ = e % 2048;
        const int mb = r_ / 1024, ks4 = (r_ % 1024) / 256, lane = (r_ % 256) / 4, kk = r_ % 4;
        v = dec_w0(S, net, 32 * mb + (lane & 31), 16 * (lane >> 5) + ks4 * 4 + kk);
    } else if (e < DEC_B_G0) {                // fp32 layer-1 A fragments: [net][ks4][lane][kk]
        const int net = (e - DEC_A_G1) / 2048, r_ = (e - DEC_A_G1) % 2048;
        const int ks4 = r_ / 256, lane = (r_ % 256) / 4, kk = r_ % 4, ks = ks4 * 4 + kk;
        v = dec_w1(S, net, lane & 31, hidden_unit(ks >> 4, ks & 15, lane >> 5));
    } else if (e < DEC_B_A0) {
        v = S.gb0[e - DEC_B_G0] * S.lr_mul * LOG2E;
    } else if (e < DEC_B_G1) {
        v = S.ab0[e - DEC_B_A0] * S.lr_mul * LOG2E;
    } else if (e < DEC_B_A1) {
        const int o = geo_row_to_out(e - DEC_B_G1);
        v = o >= 0 ? S.gb1[o] * S.lr_mul : 0.0f;
    } else if (e < DEC_FLOATS) {
        v = S.ab1[app_row_to_out(e - DEC_B_A1)] * S.lr_mul * LOG2E;
    } else {                                  // split-bf16 fragments: [frag][lane][word], 2 bf16 per word
        const int x = e - DEC_BF16;
        const int frag = x / 256, lane = (x % 256) / 4, word = x % 4;
        const int i = lane & 31, h = lane >> 5, part = frag & 1;
        unsigned bits[2];
        for (int t = 0; t < 2; ++t) {
            const int el = 2 * word + t;      // element of the 8-vector: k = 8h + el within the k-step
            float w;
            if (frag < 16) {
                const int net = frag >> 3, mb = (frag >> 2) & 1, ks = (frag >> 1) & 1;
                w = dec_w0(S, net, 32 * mb + i, 16 * h + 8 * ks + el);
            } else {
                const int f = frag - 16, net = f >> 3, ks = (f >> 1) & 3;
                w = dec_w1(S, net, i, hidden_unit(ks >> 1, 8 * (ks & 1) + el, h));
            }
            const unsigned hi = bf16_rne_bits(w);
            bits[t] = part == 0 ? hi : bf16_rne_bits(w - __uint_as_float(hi << 16));
        }
        out[e] = __uint_as_float((bits[0] & 0xffffu) | (bits[1] << 16));
        return;
    }
    out[e] = v;
}

}  // namespace nfe

using namespace nfe;

extern "C" int nfe_ray_sampler(const float* cam2world, const float* intrinsics, int n_views, int resolution,
                               float* origins, float* dirs, nfe_stream_t stream) {
    NFE_REQUIRE(cam2world && intrinsics && origins && dirs, "nfe_ray_sampler: null pointer");
    NFE_REQUIRE(n_views > 0 && resolution > 0 && resolution <= 16384, "nfe_ray_sampler: bad sizes N=%d R=%d", n_views, resolution);
    const long long total = (long long)n_views * resolution * resolution;
    long long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ray_sampler_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       cam2world, intrinsics, n_views, resolution, origins, dirs);
    NFE_CHECK_LAUNCH("ray_sampler_kernel");
    return NFE_OK;
}

extern "C" int nfe_ray_limits_box(const float* origins, const float* dirs, int64_t n_rays, float box_side_length,
                                  float* ray_start, float* ray_end, void* scratch, nfe_stream_t stream) {
    NFE_REQUIRE(origins && dirs && ray_start && ray_end && scratch, "nfe_ray_limits_box: null pointer");
    NFE_REQUIRE(n_rays > 0 && box_side_length > 0.0f, "nfe_ray_limits_box: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    long long blocks = (n_rays + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(limits_init_kernel, dim3(1), dim3(1), 0, st, (unsigned*)scratch);
    hipLaunchKernelGGL(ray_limits_kernel, dim3((unsigned)blocks), dim3(256), 0, st, origins, dirs, (long long)n_rays,
                       box_side_length * 0.5f, ray_start, ray_end, (unsigned*)scratch);
    hipLaunchKernelGGL(ray_limits_fix_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (long long)n_rays, ray_start, ray_end,
                       (const unsigned*)scratch);
    NFE_CHECK_LAUNCH("ray_limits kernels");
    return NFE_OK;
}

extern "C" int nfe_plane_stats(const float* planes, int n, int c, int hw, float* mean, float* std, nfe_stream_t stream) {
    NFE_REQUIRE(planes && mean && std, "nfe_plane_stats: null pointer");
    NFE_REQUIRE(n > 0 && c > 0 && hw > 1, "nfe_plane_stats: bad sizes n=%d c=%d hw=%d", n, c, hw);
    hipLaunchKernelGGL(plane_stats_kernel, dim3((unsigned)(n * c)), dim3(256), 0, (hipStream_t)stream, planes, hw, mean, std);
    NFE_CHECK_LAUNCH("plane_stats_kernel");
    return NFE_OK;
}

extern "C" int nfe_plane_affine(const float* in, const float* scale, const float* shift, int n, int c, int hw,
                                int n_affine, float* out, nfe_stream_t stream) {
    NFE_REQUIRE(in && scale && shift && out, "nfe_plane_affine: null pointer");
    NFE_REQUIRE(n > 0 && c > 0 && hw > 0, "nfe_plane_affine: bad sizes n=%d c=%d hw=%d", n, c, hw);
    NFE_REQUIRE(n_affine == 1 || n_affine == n, "nfe_plane_affine: n_affine=%d must be 1 or n=%d", n_affine, n);
    int gy = (hw / 4 + 1023) / 1024; if (gy < 1) gy = 1; if (gy > 64) gy = 64;
    hipLaunchKernelGGL(plane_affine_kernel, dim3((unsigned)(n * c), (unsigned)gy), dim3(256), 0, (hipStream_t)stream,
                       in, scale, shift, c, hw, n_affine, out);
    NFE_CHECK_LAUNCH("plane_affine_kernel");
    return NFE_OK;
}

extern "C" int nfe_make_affine(const float* mean, const float* std, const float* new_mean, const float* new_std,
                               int n, int c, int n_override, float* geo_scale, float* geo_shift,
                               float* app_scale, float* app_shift, nfe_stream_t stream) {
    NFE_REQUIRE(mean && std && geo_scale && geo_shift && app_scale && app_shift, "nfe_make_affine: null pointer");
    NFE_REQUIRE((new_mean != nullptr) == (new_std != nullptr), "nfe_make_affine: override mean/std must come in pairs");
    NFE_REQUIRE(n > 0 && c > 0, "nfe_make_affine: bad sizes");
    NFE_REQUIRE(!new_mean || n_override == 1 || n_override == n, "nfe_make_affine: n_override=%d must be 1 or n=%d", n_override, n);
    hipLaunchKernelGGL(make_affine_kernel, dim3((unsigned)((n * c + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       mean, std, new_mean, new_std, n, c, n_override, geo_scale, geo_shift, app_scale, app_shift);
    NFE_CHECK_LAUNCH("make_affine_kernel");
    return NFE_OK;
}

extern "C" int nfe_plane_pack(const float* planes_nchw, int n, int h, int w, float* packed, nfe_stream_t stream) {
    NFE_REQUIRE(planes_nchw && packed, "nfe_plane_pack: null pointer");
    NFE_REQUIRE(n > 0 && h > 0 && w > 0, "nfe_plane_pack: bad sizes n=%d h=%d w=%d", n, h, w);
    const int hw = h * w;
    hipLaunchKernelGGL(plane_pack_kernel, dim3((unsigned)((hw + 63) / 64), (unsigned)(n * 3)), dim3(256), 0, (hipStream_t)stream,
                       planes_nchw, hw, packed);
    NFE_CHECK_LAUNCH("plane_pack_kernel");
    return NFE_OK;
}

// Cross fragments: the layer-1 geometry-head layout (frag = ks*2 + part) with weights that multiply the APPEARANCE head's
// hidden units (which carry 1/ln2, like the geometry head's: same ln2 folding as dec_w1(net 0)).
__global__ void decoder_pack_cross_kernel(const float* cross_w1, float lr_mul, float* out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= NFE_DECODER_CROSS_FLOATS) return;
    const int frag = x / 256, lane = (x % 256) / 4, word = x % 4;
    const int i = lane & 31, h = lane >> 5, part = frag & 1, ks = frag >> 1;
    const float g1 = lr_mul / sqrtf(64.0f);
    unsigned bits[2];
    for (int t = 0; t < 2; ++t) {
        const int el = 2 * word + t;
        const int o = geo_row_to_out(i);
        const float w = o >= 0 ? cross_w1[o * 64 + hidden_unit(ks >> 1, 8 * (ks & 1) + el, h)] * g1 * LN2 : 0.0f;
        const unsigned hi = bf16_rne_bits(w);
        bits[t] = part == 0 ? hi : bf16_rne_bits(w - __uint_as_float(hi << 16));
    }
    out[x] = __uint_as_float((bits[0] & 0xffffu) | (bits[1] << 16));
}

extern "C" int nfe_decoder_pack_cross(const float* cross_w1, float lr_mul, float* packed_cross, nfe_stream_t stream) {
    NFE_REQUIRE(cross_w1 && packed_cross, "nfe_decoder_pack_cross: null pointer");
    hipLaunchKernelGGL(decoder_pack_cross_kernel, dim3(NFE_DECODER_CROSS_FLOATS / 256), dim3(256), 0, (hipStream_t)stream,
                       cross_w1, lr_mul, packed_cross);
    NFE_CHECK_LAUNCH("decoder_pack_cross_kernel");
    return NFE_OK;
}

extern "C" int nfe_decoder_pack(const float* geo_w0, const float* geo_b0, const float* geo_w1, const float* geo_b1,
                                const float* app_w0, const float* app_b0, const float* app_w1, const float* app_b1,
                                float lr_mul, float* packed, nfe_stream_t stream) {
    NFE_REQUIRE(geo_w0 && geo_b0 && geo_w1 && geo_b1 && app_w0 && app_b0 && app_w1 && app_b1 && packed,
                "nfe_decoder_pack: null pointer");
    DecSrc S{geo_w0, geo_b0, geo_w1, geo_b1, app_w0, app_b0, app_w1, app_b1, lr_mul};
    hipLaunchKernelGGL(decoder_pack_kernel, dim3((DEC_TOTAL + 255) / 256), dim3(256), 0, (hipStream_t)stream, S, packed);
    NFE_CHECK_LAUNCH("decoder_pack_kernel");
    return NFE_OK;
}
