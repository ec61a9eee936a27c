// Fused tri-plane volume renderer for MI355X (gfx950): stratified depths -> quad-cooperative tri-plane bilinear
// gather (pipelined over the planes) -> dual MLP decoder on MFMA (split-bf16 or exact fp32) -> mid-point alpha
// compositing, one kernel; importance sampling + merge in a second kernel between the two passes.
//
// Replaces DisentangledImportanceRenderer.forward (training/volumetric_rendering/renderer.py:301-363)
// and everything it calls: sample_stratified (:169-192), sample_from_planes (:55-65),
// DisentangledOSGDecoder.forward (training/triplane.py:249-270), SegMipRayMarcher2.run_forward
// (ray_marcher.py:68-101), sample_importance/sample_pdf (renderer.py:194-253), unify_samples
// (:288-300).  See DESIGN.md §4 for the lane mapping and the MFMA operand layouts.
#include "nfe_common.h"

#include <atomic>
#include <cstring>

#define NFE_RENDER_WS_DEFAULT 1     // environment NFE_RENDER_WS: 0 = always the fused render_kernel, otherwise render_ws_kernel<4, 2> where it applies

namespace nfe {

// ------------------------------------------------------------------------------------------
// kernel parameters
// ------------------------------------------------------------------------------------------
enum DepthMode : int { DEPTH_STRATIFIED = 0, DEPTH_PER_RAY = 1, DEPTH_DISPARITY = 2, DEPTH_BUFFER = 3 };

struct RenderK {
    const float* planes_g; const float* planes_a; long long plane_view_stride;
    int H, W;
    const float* aff[4];          // geo_scale, geo_shift, app_scale, app_shift: [N,96] or null
    const float* dec;
    int N, M, R, tiled;
    const float* origins; const float* dirs; const float* cam2world; const float* intrinsics;
    int S;                         // samples marched per ray in this pass
    int depth_mode;
    float ray_start, ray_end;
    const float* rs_ray; const float* re_ray;
    const float* u; unsigned long long seed; const unsigned long long* seed_dev;
    const float* depth_buf;        // DEPTH_BUFFER: [N*M, S]
    float coord_scale;             // 2 / box_warp
    int white_back;
    float* rgb; float* seg; float* depth; float* wsum; int channels_first;
    float* out_depths;             // optional [N*M, S]
    float* out_weights;            // optional [N*M, S-1]
    unsigned* depth_minmax;        // ordered-uint {min, max}, and [2] = wave pairs of render_ws_kernel that abandoned a hand-off wait
    float density_noise;           // std of the Gaussian added to sigma (renderer.py:285-286), NOISE variants only
    const int* src_buf;            // DEPTH_BUFFER + NOISE: [N*M, S] which draw each merged sample is (k, or D + fine rank)
    const float* noise_buf; int noise_stride;      // optional: the normals themselves, [N*M, noise_stride] indexed by draw (nfe_render_args.density_noise_values)
    const float* dec_cross;        // CROSS variants: packed cross fragments (nfe_decoder_pack_cross)
    // EVAL variants (first pass of nfe_render_backward): per sample sigma and a = <2 g_rgb, rgb> + <g_seg, seg> instead of a march
    const float* ev_g_rgb; const float* ev_g_seg; int ev_channels_first; float* ev_sig; float* ev_a;
    float* tap_colors;     // STORE variants: [N * blocks_per_view][S][48][32 lanes] decoder outputs of every sample (rgb 0..31, seg 32..46, sigma 47)
    int seg_count;                 // SPLIT variants: depth segments per ray block (each marched by its own wave)
    float* partials;               // SPLIT variants: [N*M, seg_count, PARTIAL_FLOATS] segment composites, see render_combine_kernel
    unsigned long long* clock_probe;   // optional [4]: {s_memtime, s_memrealtime} of workgroup 0 / wave 0 at kernel start and end
};

// LDS map (floats): [0, DEC_FLOATS) decoder image shared by the block's 4 waves, then per wave AFF_FLOATS of
// view affines and a 32-point x 32-channel exchange tile (gather layout -> MFMA operand layout, eval_point).
constexpr int LDS_AFF = DEC_FLOATS;
constexpr int AFF_FLOATS = 4 * 96 + 2 * 32;   // + per-channel sums over planes of the two shifts
constexpr int AFF_BSUM = 4 * 96;
constexpr int XCHG_FLOATS = 32 * 32;
constexpr int WAVE_LDS_FLOATS = AFF_FLOATS + XCHG_FLOATS;
constexpr int RENDER_LDS_BYTES = (DEC_FLOATS + 4 * WAVE_LDS_FLOATS) * 4;
// Depth-split launches (few rays: not enough ray blocks to fill the SIMDs): a ray block's S samples are cut into seg_count
// contiguous segments marched by different waves; a segment starts at transmittance 1 and leaves (sum of w*rgb, sum of
// w*seg, sum of w*t, sum of w, its transmittance) = 32 + 15 + 3 floats, composited in order by render_combine_kernel
// (alpha compositing is associative: out = sum_s (prod_{s'<s} T_s') * partial_s).
constexpr int PARTIAL_FLOATS = 52;
constexpr int SPLIT_MAX_ITEMS = 2048;            // wave-items (ray block x segment) a split launch may have: 2 per SIMD
constexpr int SPLIT_PARTIAL_BYTES = SPLIT_MAX_ITEMS * 32 * PARTIAL_FLOATS * 4;
// CROSS variants (SegmentationOSGDecoder, triplane.py:192-230: sigma comes from the OTHER net's hidden layer): 8 more
// split-bf16 layer-1 fragments behind the per-wave regions, geometry-head rows fed by the appearance head's hidden units.
constexpr int LDS_CROSS = DEC_FLOATS + 4 * WAVE_LDS_FLOATS;
constexpr int CROSS_WORDS = 8 * 64 * 4;
constexpr int RENDER_LDS_BYTES_CROSS = RENDER_LDS_BYTES + CROSS_WORDS * 4;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
union Frag { bf16x8 v; uint4 q; unsigned u[4]; };

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    return __builtin_elementwise_fma(a, b, c);
}
__device__ __forceinline__ f32x2 splat(float x) { return f32x2{x, x}; }

// Raw v_exp_f32 / v_log_f32 (base 2, 1 ulp, no denormal-range fix-up code: the differences are far
// below the 1e-3 parity budget).
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float log2_fast(float x) { return __builtin_amdgcn_logf(x); }

// torch Softplus(beta=1, threshold=20) in natural units (used by the ray marcher, ray_marcher.py:76)
__device__ __forceinline__ float softplus_f(float x) {
    float r = log2_fast(1.0f + exp2_fast(x * LOG2E)) * LN2;
    return x > 20.0f ? x : r;
}
// The same function on y = x*log2(e), returning softplus(x)/ln(2): the scale factors live in the packed
// decoder weights (nfe_decoder_pack).  threshold 20 -> 20*log2(e).
__device__ __forceinline__ float softplus_log2(float y) {
    float r = log2_fast(1.0f + exp2_fast(y));
    return y > 20.0f * LOG2E ? y : r;
}

// max(y, 0) as one integer max on the bit pattern (negative floats are negative ints; fmaxf would add a
// canonicalising v_max before the real one)
__device__ __forceinline__ float relu_bits(float y) { return __int_as_float(max(__float_as_int(y), 0)); }

// softplus_log2 over a whole accumulator.  Default: max(y,0) + log2(1 + 2^-|y|) - 4 instructions per value (v_exp with a free
// -|y| source modifier, half a v_pk_add, v_log, v_max_i32, half a v_pk_add), no overflow, no threshold select, and exact for
// every y: torch's Softplus returns x itself above its threshold and so does this form (the log term is 0 there).
// (The 3.5-instruction form log2(1 + 2^min(y, 126)) of round 2 saturates at x = 87.3 where the reference returns x; not kept.)
// vector ALU and transcendental instructions stay on their side of this fence; MFMA, LDS, scalar and memory instructions may cross
#define NFE_VALU_FENCE() __builtin_amdgcn_sched_barrier(0x4 | 0x8 | 0x10 | 0x80)
__device__ __forceinline__ void softplus_log2_x16(f32x16& a) {
    // Four phases of INDEPENDENT instructions (16 exps, 8 packed adds, 16 logs, 8 packed adds + 16 integer max): left to itself the
    // scheduler sometimes emits the per-pair dependent chain exp, exp -> add -> log, log -> add with a hazard nop between every two
    // instructions (+9 % kernel cycles, found in round 5 when an unrelated edit flipped it); the fences pin the batched order.
    f32x2 e[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) e[r] = f32x2{exp2_fast(-__builtin_fabsf(a[2 * r])), exp2_fast(-__builtin_fabsf(a[2 * r + 1]))};
    NFE_VALU_FENCE();
#pragma unroll
    for (int r = 0; r < 8; ++r) e[r] = e[r] + splat(1.0f);
    NFE_VALU_FENCE();
#pragma unroll
    for (int r = 0; r < 8; ++r) e[r] = f32x2{log2_fast(e[r][0]), log2_fast(e[r][1])};
    NFE_VALU_FENCE();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const f32x2 l = e[r] + f32x2{relu_bits(a[2 * r]), relu_bits(a[2 * r + 1])};
        a[2 * r] = l[0]; a[2 * r + 1] = l[1];
    }
}

// fp32 -> (hi, lo) bf16 pairs: hi = top 16 bits (truncation), lo = bf16(x - hi); x - hi - lo <= 2^-17 |x|.
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    bf16x2 p = {(__bf16)(a - __uint_as_float(ua & 0xffff0000u)), (__bf16)(b - __uint_as_float(ub & 0xffff0000u))};
    lo = *reinterpret_cast<unsigned*>(&p);
}

// Launder a value through an empty asm: the optimiser can no longer prove it loop-invariant, so addresses
// and constants derived from it are recomputed at the point of use (a few VALU ops) instead of being hoisted
// to the kernel prologue and spilled to scratch (which is what happens at 256 VGPRs otherwise).
// The comment inside the statement names the register in the ISA: tools/asm_audit.py finds the value's real producer and its
// consumers around the (empty) statement - the hazard recogniser sees the STATEMENT as the producer and pads nothing for it.
__device__ __forceinline__ int launder(int v) { asm volatile("; nfe_launder %0" : "+v"(v)); return v; }

// ---- quad-cooperative gather ------------------------------------------------------------------------
// The vector L1 retires one 64-byte request per clock and merges only ADJACENT lanes (4 lanes x 16 contiguous
// bytes = one request; anything else is one request per lane: profiles/r01_gather_rate_microbench.txt).  So
// a texel is never read by "its" lane alone: the 4 lanes of a quad read one 64-byte half texel together, for
// each of the quad's 4 points in turn.  Lane l = (quad Q = l>>2, c = l&3) owns point l&31 (as the MFMA
// layout wants); quads Q and Q+8 own the same 4 points and read the two halves (h = l>>5) of their texels.
// Load i of a tap: every lane of the quad fetches bytes [64h+16c, +16) of point (4(Q&7)+i)'s texel; the
// address and the bilinear weight come from lane i of the quad by DPP quad-broadcast.  A lane therefore
// accumulates channels 16h+4c..+3 of FOUR points; exchange_to_own() moves them to the owners through LDS.
template <int I> __device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_mov_dpp(v, I * 0x55, 0xf, 0xf, true); }
template <int I> __device__ __forceinline__ float quad_bcast(float v) { return __int_as_float(quad_bcast<I>(__float_as_int(v))); }

// One 16-byte piece of a texel: uniform base (SGPR pair) + 32-bit byte offset (the saddr form of global_load).
__device__ __forceinline__ float4 texel_piece(const float* __restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// Bilinear weights travel through the LDS crossbar (ds_swizzle, quad-perm mode: no LDS memory, no VALU issue
// slot) while the texel loads are in flight; only the address broadcast stays on DPP (it feeds the load).
template <int I> __device__ __forceinline__ float quad_swizzle(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x8000 | (I * 0x55)));
}
// ---- the gather, software-pipelined over the three planes (one plane set) ---------------------------------
// Six batches of 8 loads (plane p, tap pair): two batches are in flight; the loads of batch b+2 are issued right
// after batch b has been consumed, so only the first batch's latency is exposed per sample instead of one per plane.
#define NFE_PIPE_ISSUE_I(S, K, I)                                                                          \
    vg[S][(K) * 4 + I] = texel_piece(base_, (unsigned)quad_bcast<I>((int)off_) + qoff_bytes);                \
    wq[S][(K) * 4 + I] = quad_swizzle<I>(wk_);
#define NFE_PIPE_ISSUE(S, PL, K0)                                                                          \
    {                                                                                                      \
        const float* base_ = pg + (PL) * plane_elems;                                                      \
        _Pragma("unroll") for (int K = 0; K < 2; ++K) {                                                    \
            const unsigned off_ = offs[PL][(K0) + K];                                                      \
            const float wk_ = tp[PL].w[(K0) + K];                                                          \
            NFE_PIPE_ISSUE_I(S, K, 0) NFE_PIPE_ISSUE_I(S, K, 1) NFE_PIPE_ISSUE_I(S, K, 2) NFE_PIPE_ISSUE_I(S, K, 3)  \
        }                                                                                                  \
    }
#define NFE_PIPE_FMA(S, K, I)                                                                              \
    {                                                                                                      \
        const f32x2 w2 = splat(wq[S][(K) * 4 + I]);                                                        \
        sg[2 * I + 0] = pk_fma(w2, f32x2{vg[S][(K) * 4 + I].x, vg[S][(K) * 4 + I].y}, sg[2 * I + 0]);      \
        sg[2 * I + 1] = pk_fma(w2, f32x2{vg[S][(K) * 4 + I].z, vg[S][(K) * 4 + I].w}, sg[2 * I + 1]);      \
    }
#define NFE_PIPE_CONSUME(S)                                                                                \
    NFE_PIPE_FMA(S, 0, 0) NFE_PIPE_FMA(S, 0, 1) NFE_PIPE_FMA(S, 0, 2) NFE_PIPE_FMA(S, 0, 3)                \
    NFE_PIPE_FMA(S, 1, 0) NFE_PIPE_FMA(S, 1, 1) NFE_PIPE_FMA(S, 1, 2) NFE_PIPE_FMA(S, 1, 3)

template <bool SIGMA_ONLY, int PL, bool DUAL = false, bool INB = false>
__device__ __forceinline__ void plane_affine_acc(const float* __restrict__ aff, int qoff, const Taps& tp, f32x2 (&sg)[8],
                                                 f32x2 (&qn)[8], f32x2 (&qd)[8], f32x2* sa = nullptr) {
    {
        const float4 sc = *reinterpret_cast<const float4*>(aff + 0 * 96 + PL * 32 + qoff);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qn[2 * i + 0] = pk_fma(sg[2 * i + 0], f32x2{sc.x, sc.y}, qn[2 * i + 0]);
            qn[2 * i + 1] = pk_fma(sg[2 * i + 1], f32x2{sc.z, sc.w}, qn[2 * i + 1]);
        }
    }
    if (!SIGMA_ONLY) {
        const float4 sc = *reinterpret_cast<const float4*>(aff + 2 * 96 + PL * 32 + qoff);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qd[2 * i + 0] = pk_fma(DUAL ? sa[2 * i + 0] : sg[2 * i + 0], f32x2{sc.x, sc.y}, qd[2 * i + 0]);
            qd[2 * i + 1] = pk_fma(DUAL ? sa[2 * i + 1] : sg[2 * i + 1], f32x2{sc.z, sc.w}, qd[2 * i + 1]);
        }
    }
    if (!INB && __builtin_amdgcn_ballot_w64(tp.wdef != 0.0f) != 0) {     // rare: some sample of the wave left the plane
        const float wd[4] = {quad_bcast<0>(tp.wdef), quad_bcast<1>(tp.wdef), quad_bcast<2>(tp.wdef), quad_bcast<3>(tp.wdef)};
        const float4 b = *reinterpret_cast<const float4*>(aff + 1 * 96 + PL * 32 + qoff);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qn[2 * i + 0] = pk_fma(splat(wd[i]), f32x2{b.x, b.y}, qn[2 * i + 0]);
            qn[2 * i + 1] = pk_fma(splat(wd[i]), f32x2{b.z, b.w}, qn[2 * i + 1]);
        }
        if (!SIGMA_ONLY) {
            const float4 e = *reinterpret_cast<const float4*>(aff + 3 * 96 + PL * 32 + qoff);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                qd[2 * i + 0] = pk_fma(splat(wd[i]), f32x2{e.x, e.y}, qd[2 * i + 0]);
                qd[2 * i + 1] = pk_fma(splat(wd[i]), f32x2{e.z, e.w}, qd[2 * i + 1]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) { sg[c] = splat(0.0f); if (DUAL) sa[c] = splat(0.0f); }
}

#define NFE_PIPE_GEOM(PL, U, V)                                                                            \
    tp[PL] = tap_geometry(H, W, U, V);                                                                     \
    offs[PL][0] = (unsigned)(tp[PL].yc0 * W + tp[PL].xc0) * 128u; offs[PL][1] = (unsigned)(tp[PL].yc0 * W + tp[PL].xc1) * 128u; \
    offs[PL][2] = (unsigned)(tp[PL].yc1 * W + tp[PL].xc0) * 128u; offs[PL][3] = (unsigned)(tp[PL].yc1 * W + tp[PL].xc1) * 128u;

// SQUARE planes: the axes are shared between the projections (compile-time variant: a run-time branch here, with loads in
// flight around it, produced run-dependent wrong results and is avoided on purpose).
#define NFE_PIPE_GEOM_AX(PL, AU, AV)                                                                       \
    tp[PL] = taps_from_axes(AU, AV);                                                                       \
    offs[PL][0] = (unsigned)(tp[PL].yc0 * W + tp[PL].xc0) * 128u; offs[PL][1] = (unsigned)(tp[PL].yc0 * W + tp[PL].xc1) * 128u; \
    offs[PL][2] = (unsigned)(tp[PL].yc1 * W + tp[PL].xc0) * 128u; offs[PL][3] = (unsigned)(tp[PL].yc1 * W + tp[PL].xc1) * 128u;

// Two plane sets (norm_planes != normalised denorm_planes): twelve batches of one tap = 4 + 4 loads, two in flight.
#define NFE_PIPE2_ISSUE_I(S, I)                                                                            \
    {                                                                                                      \
        const unsigned o_ = (unsigned)quad_bcast<I>((int)off_) + qoff_bytes;                               \
        vg[S][I] = texel_piece(bg_, o_); vg[S][4 + I] = texel_piece(ba_, o_);                              \
        wq[S][I] = quad_swizzle<I>(wk_);                                                                   \
    }
#define NFE_PIPE2_ISSUE(S, PL, K)                                                                          \
    {                                                                                                      \
        const float* bg_ = pg + (PL) * plane_elems; const float* ba_ = pa + (PL) * plane_elems;            \
        const unsigned off_ = offs[PL][K];                                                                 \
        const float wk_ = tp[PL].w[K];                                                                     \
        NFE_PIPE2_ISSUE_I(S, 0) NFE_PIPE2_ISSUE_I(S, 1) NFE_PIPE2_ISSUE_I(S, 2) NFE_PIPE2_ISSUE_I(S, 3)    \
    }
#define NFE_PIPE2_CONSUME(S)                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                     \
        const f32x2 w2 = splat(wq[S][i_]);                                                                 \
        sg[2 * i_ + 0] = pk_fma(w2, f32x2{vg[S][i_].x, vg[S][i_].y}, sg[2 * i_ + 0]);                      \
        sg[2 * i_ + 1] = pk_fma(w2, f32x2{vg[S][i_].z, vg[S][i_].w}, sg[2 * i_ + 1]);                      \
        sa[2 * i_ + 0] = pk_fma(w2, f32x2{vg[S][4 + i_].x, vg[S][4 + i_].y}, sa[2 * i_ + 0]);              \
        sa[2 * i_ + 1] = pk_fma(w2, f32x2{vg[S][4 + i_].z, vg[S][4 + i_].w}, sa[2 * i_ + 1]);              \
    }

// ---- in-bounds fast path of the gather (square planes; round 4) ---------------------------------------------
// ~90 % of the wave-steps of an FFHQ-like camera have the 2 x 2 footprints of all their 32 samples inside all three planes.  Then
// nothing of the zeros-padding machinery is needed: no clamped coordinates, no validity selects, no weight deficit (`wdef` == 0
// exactly), and the four taps of a point sit at fixed distances from the first - +128 bytes (next texel: the load's immediate
// offset) and +W * 128 bytes (next row: a second scalar base) - so ONE byte offset per (plane, point) goes through the DPP
// quad-broadcast instead of four.  Saves ~125 vector + 36 DPP instructions of the ~1 000 a wave issues per 32 samples; the kernel
// is bound by exactly that count.  The arithmetic that produces weights, offsets and sums is the general
// path's, in the same order: results are bit-identical.  The decision is one ballot -> s_cmp -> s_cbranch_scc per sample (the
// stable branch form of profiles/experiments/r02_square_branch.md), taken before any load of the sample is in flight.
struct InbAxes { int x0, y0, z0; float fx, fy, fz; };
// floor coordinate and fraction of the three axes exactly as axis_geometry computes them; true when every lane is inside
__device__ __forceinline__ bool inb_axes(int size, float gx, float gy, float gz, InbAxes& a) {
    const float hs = 0.5f * (float)size;
    const float ix = (gx + 1.0f) * hs - 0.5f, iy = (gy + 1.0f) * hs - 0.5f, iz = (gz + 1.0f) * hs - 0.5f;
    const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
    a.fx = ix - fx0; a.fy = iy - fy0; a.fz = iz - fz0;
    a.x0 = (int)fx0; a.y0 = (int)fy0; a.z0 = (int)fz0;                   // saturating conversion: far-away samples fail the test below
    const unsigned worst = max(max((unsigned)a.x0, (unsigned)a.y0), (unsigned)a.z0);        // negative -> huge
    return __builtin_amdgcn_ballot_w64(worst >= (unsigned)(size - 1)) == 0;
}
// one 16-byte piece at (uniform base) + (32-bit byte offset) + (compile-time immediate)
template <int IMM>
__device__ __forceinline__ float4 texel_piece_imm(const char* __restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float4*>(base + byte_off + IMM);
}
#define NFE_INB_PLANE(PL, A0, B0, FA, FB)          /* plane PL: axis a indexes W (taps a0, a0+1), axis b indexes H */         \
    off0[PL] = (unsigned)((B0) * W + (A0)) * 128u;                                                                             \
    { const float ea_ = 1.0f - (FA), eb_ = 1.0f - (FB);                                                                        \
      wt[PL][0] = ea_ * eb_; wt[PL][1] = (FA) * eb_; wt[PL][2] = ea_ * (FB); wt[PL][3] = (FA) * (FB); }
#define NFE_INB_VOFF(PL)                                                                                                       \
    vo[0] = (unsigned)quad_bcast<0>((int)off0[PL]) + qoff_bytes; vo[1] = (unsigned)quad_bcast<1>((int)off0[PL]) + qoff_bytes;  \
    vo[2] = (unsigned)quad_bcast<2>((int)off0[PL]) + qoff_bytes; vo[3] = (unsigned)quad_bcast<3>((int)off0[PL]) + qoff_bytes;
#define NFE_INB_ISSUE_I(S, K, I)                                                                                               \
    vg[S][(K) * 4 + I] = texel_piece_imm<(K) * 128>(b_, vo[I]); wq[S][(K) * 4 + I] = quad_swizzle<I>(wt_[K]);
#define NFE_INB_ISSUE(S, PL, ROW)                  /* the two taps of row ROW (0 / 1) of plane PL, four points each */          \
    {                                                                                                                          \
        const char* b_ = reinterpret_cast<const char*>(pg + (PL) * plane_elems) + ((ROW) ? row_bytes : 0);                     \
        const float wt_[2] = {wt[PL][2 * (ROW)], wt[PL][2 * (ROW) + 1]};                                                       \
        NFE_INB_ISSUE_I(S, 0, 0) NFE_INB_ISSUE_I(S, 0, 1) NFE_INB_ISSUE_I(S, 0, 2) NFE_INB_ISSUE_I(S, 0, 3)                    \
        NFE_INB_ISSUE_I(S, 1, 0) NFE_INB_ISSUE_I(S, 1, 1) NFE_INB_ISSUE_I(S, 1, 2) NFE_INB_ISSUE_I(S, 1, 3)                    \
    }
template <bool SIGMA_ONLY>
__device__ __forceinline__ void gather_pipelined_inb(const float* __restrict__ pg, int W, long long plane_elems,
                                                     const float* __restrict__ aff, int lane, const InbAxes& a,
                                                     f32x2 (&qn)[8], f32x2 (&qd)[8]) {
    unsigned off0[3];
    float wt[3][4];
    unsigned vo[4];
    const long long row_bytes = (long long)W * 128;
    const int ll = launder(lane);
    const int qoff = (ll >> 5) * 16 + (ll & 3) * 4;
    const unsigned qoff_bytes = (unsigned)qoff * 4u;
    float4 vg[2][8];
    float wq[2][8];
    f32x2 sg[8];
    const Taps none{};                                      // plane_affine_acc<.., INB> does not look at it
#pragma unroll
    for (int c = 0; c < 8; ++c) sg[c] = splat(0.0f);
    NFE_INB_PLANE(0, a.x0, a.y0, a.fx, a.fy)                // project_onto_planes (renderer.py:39-53): p0 = (x, y)
    NFE_INB_VOFF(0)
    NFE_INB_ISSUE(0, 0, 0) NFE_INB_ISSUE(1, 0, 1)
    NFE_INB_PLANE(1, a.x0, a.z0, a.fx, a.fz)                // p1 = (x, z)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(0) NFE_INB_VOFF(1) NFE_INB_ISSUE(0, 1, 0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(1) NFE_INB_ISSUE(1, 1, 1)
    plane_affine_acc<SIGMA_ONLY, 0, false, true>(aff, qoff, none, sg, qn, qd);
    NFE_INB_PLANE(2, a.z0, a.x0, a.fz, a.fx)                // p2 = (z, x)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(0) NFE_INB_VOFF(2) NFE_INB_ISSUE(0, 2, 0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(1) NFE_INB_ISSUE(1, 2, 1)
    plane_affine_acc<SIGMA_ONLY, 1, false, true>(aff, qoff, none, sg, qn, qd);
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(1)
    plane_affine_acc<SIGMA_ONLY, 2, false, true>(aff, qoff, none, sg, qn, qd);
}

// the same for two plane sets (norm_planes != normalised denorm_planes): twelve batches of one tap = 4 + 4 loads, two in flight
#define NFE_INB2_ISSUE_I(S, I, IMM)                                                                                            \
    vg[S][I] = texel_piece_imm<IMM>(bg_, vo[I]); vg[S][4 + I] = texel_piece_imm<IMM>(ba_, vo[I]); wq[S][I] = quad_swizzle<I>(wk_);
#define NFE_INB2_ISSUE(S, PL, K)                                                                                               \
    {                                                                                                                          \
        const char* bg_ = reinterpret_cast<const char*>(pg + (PL) * plane_elems) + (((K) >> 1) ? row_bytes : 0);               \
        const char* ba_ = reinterpret_cast<const char*>(pa + (PL) * plane_elems) + (((K) >> 1) ? row_bytes : 0);               \
        const float wk_ = wt[PL][K];                                                                                           \
        NFE_INB2_ISSUE_I(S, 0, ((K) & 1) * 128) NFE_INB2_ISSUE_I(S, 1, ((K) & 1) * 128)                                        \
        NFE_INB2_ISSUE_I(S, 2, ((K) & 1) * 128) NFE_INB2_ISSUE_I(S, 3, ((K) & 1) * 128)                                        \
    }
__device__ __forceinline__ void gather_pipelined_dual_inb(const float* __restrict__ pg, const float* __restrict__ pa, int W, long long plane_elems,
                                                          const float* __restrict__ aff, int lane, const InbAxes& a,
                                                          f32x2 (&qn)[8], f32x2 (&qd)[8]) {
    unsigned off0[3];
    float wt[3][4];
    unsigned vo[4];
    const long long row_bytes = (long long)W * 128;
    const int ll = launder(lane);
    const int qoff = (ll >> 5) * 16 + (ll & 3) * 4;
    const unsigned qoff_bytes = (unsigned)qoff * 4u;
    float4 vg[2][8];
    float wq[2][4];
    f32x2 sg[8], sa[8];
    const Taps none{};
#pragma unroll
    for (int c = 0; c < 8; ++c) { sg[c] = splat(0.0f); sa[c] = splat(0.0f); }
    NFE_INB_PLANE(0, a.x0, a.y0, a.fx, a.fy)
    NFE_INB_VOFF(0)
    NFE_INB2_ISSUE(0, 0, 0) NFE_INB2_ISSUE(1, 0, 1)
    NFE_INB_PLANE(1, a.x0, a.z0, a.fx, a.fz)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0) NFE_INB2_ISSUE(0, 0, 2)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1) NFE_INB2_ISSUE(1, 0, 3)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0) NFE_INB_VOFF(1) NFE_INB2_ISSUE(0, 1, 0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1) NFE_INB2_ISSUE(1, 1, 1)
    plane_affine_acc<false, 0, true, true>(aff, qoff, none, sg, qn, qd, sa);
    NFE_INB_PLANE(2, a.z0, a.x0, a.fz, a.fx)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0) NFE_INB2_ISSUE(0, 1, 2)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1) NFE_INB2_ISSUE(1, 1, 3)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0) NFE_INB_VOFF(2) NFE_INB2_ISSUE(0, 2, 0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1) NFE_INB2_ISSUE(1, 2, 1)
    plane_affine_acc<false, 1, true, true>(aff, qoff, none, sg, qn, qd, sa);
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0) NFE_INB2_ISSUE(0, 2, 2)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1) NFE_INB2_ISSUE(1, 2, 3)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1)
    plane_affine_acc<false, 2, true, true>(aff, qoff, none, sg, qn, qd, sa);
}

template <bool SQUARE>
__device__ __forceinline__ void gather_pipelined_dual(const float* __restrict__ pg, const float* __restrict__ pa, int H, int W,
                                                      long long plane_elems, const float* __restrict__ aff, int lane,
                                                      float gx, float gy, float gz, f32x2 (&qn)[8], f32x2 (&qd)[8]) {
    if (SQUARE) {
        InbAxes ia;
        if (inb_axes(W, gx, gy, gz, ia)) { gather_pipelined_dual_inb(pg, pa, W, plane_elems, aff, lane, ia, qn, qd); return; }
    }
    Taps tp[3];
    unsigned offs[3][4];
    Axis ax_xw, ax_zh;
    if (SQUARE) { ax_xw = axis_geometry(W, gx); NFE_PIPE_GEOM_AX(0, ax_xw, axis_geometry(H, gy)) } else { NFE_PIPE_GEOM(0, gx, gy) }
    const int ll = launder(lane);
    const int qoff = (ll >> 5) * 16 + (ll & 3) * 4;
    const unsigned qoff_bytes = (unsigned)qoff * 4u;
    float4 vg[2][8];
    float wq[2][4];
    f32x2 sg[8], sa[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { sg[c] = splat(0.0f); sa[c] = splat(0.0f); }
    NFE_PIPE2_ISSUE(0, 0, 0) NFE_PIPE2_ISSUE(1, 0, 1)
    if (SQUARE) { ax_zh = axis_geometry(H, gz); NFE_PIPE_GEOM_AX(1, ax_xw, ax_zh) } else { NFE_PIPE_GEOM(1, gx, gz) }
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0) NFE_PIPE2_ISSUE(0, 0, 2)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1) NFE_PIPE2_ISSUE(1, 0, 3)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0) NFE_PIPE2_ISSUE(0, 1, 0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1) NFE_PIPE2_ISSUE(1, 1, 1)
    plane_affine_acc<false, 0, true>(aff, qoff, tp[0], sg, qn, qd, sa);
    if (SQUARE) { NFE_PIPE_GEOM_AX(2, ax_zh, ax_xw) } else { NFE_PIPE_GEOM(2, gz, gx) }
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0) NFE_PIPE2_ISSUE(0, 1, 2)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1) NFE_PIPE2_ISSUE(1, 1, 3)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0) NFE_PIPE2_ISSUE(0, 2, 0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1) NFE_PIPE2_ISSUE(1, 2, 1)
    plane_affine_acc<false, 1, true>(aff, qoff, tp[1], sg, qn, qd, sa);
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0) NFE_PIPE2_ISSUE(0, 2, 2)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1) NFE_PIPE2_ISSUE(1, 2, 3)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE2_CONSUME(1)
    plane_affine_acc<false, 2, true>(aff, qoff, tp[2], sg, qn, qd, sa);
}

// experiment helper (NFE_SQUARE_RUNTIME == 6): `H == W` as an integer in an SGPR, opaque to the optimiser
__device__ __forceinline__ int sq_flag_scc(int H, int W) {
    int f;
    asm volatile("s_cmp_eq_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(f) : "s"(H), "s"(W) : "scc");
    return f;
}

template <bool SIGMA_ONLY, bool SQUARE>
__device__ __forceinline__ void gather_pipelined(const float* __restrict__ pg, int H, int W, long long plane_elems,
                                                 const float* __restrict__ aff, int lane, float gx, float gy, float gz,
                                                 f32x2 (&qn)[8], f32x2 (&qd)[8]) {
    // project_onto_planes (renderer.py:39-53): p0=(x,y), p1=(x,z), p2=(z,x); first coord indexes W.
    if (SQUARE) {
        InbAxes ia;
        if (inb_axes(W, gx, gy, gz, ia)) { gather_pipelined_inb<SIGMA_ONLY>(pg, W, plane_elems, aff, lane, ia, qn, qd); return; }
    }
    Taps tp[3];
    unsigned offs[3][4];
    Axis ax_xw, ax_zh;
    // Experiment switch (profiles/experiments/r02_square_branch.md; never defined in the shipped build): decide SQUARE at run time.
    //   1: `H == W` as the compiler lowers it: run-dependent wrong pixels at two workgroups per CU FROM HIPCC'S ASSEMBLY - round 2 read
    //      that as the branch going the wrong way; it is the packed multiplies of the tap_geometry arm (56 x v_pk_mul_f32 ... op_sel:[0,1],
    //      profiles/experiments/r04_pk_opsel_hazard.md): through the build's assembly pass the same source is stable;
    //   6: the condition re-made by s_cmp at every site;   10: as 1, but both arms compute the square geometry (both stable either way).
#define NFE_ELSE_GEOM(PL, U, V, AU, AV) NFE_PIPE_GEOM(PL, U, V)
    if (SQUARE) { ax_xw = axis_geometry(W, gx); NFE_PIPE_GEOM_AX(0, ax_xw, axis_geometry(H, gy)) }
    else { ax_xw = axis_geometry(W, gx); NFE_ELSE_GEOM(0, gx, gy, ax_xw, axis_geometry(H, gy)) }
    const int ll = launder(lane);
    const int qoff = (ll >> 5) * 16 + (ll & 3) * 4;
    const unsigned qoff_bytes = (unsigned)qoff * 4u;
    float4 vg[2][8];
    float wq[2][8];
    f32x2 sg[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) sg[c] = splat(0.0f);
    NFE_PIPE_ISSUE(0, 0, 0) NFE_PIPE_ISSUE(1, 0, 2)
    // the next plane's tap geometry runs under the loads in flight
    if (SQUARE) { ax_zh = axis_geometry(H, gz); NFE_PIPE_GEOM_AX(1, ax_xw, ax_zh) }
    else { ax_zh = axis_geometry(H, gz); NFE_ELSE_GEOM(1, gx, gz, ax_xw, ax_zh) }
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(0) NFE_PIPE_ISSUE(0, 1, 0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(1) NFE_PIPE_ISSUE(1, 1, 2)
    plane_affine_acc<SIGMA_ONLY, 0>(aff, qoff, tp[0], sg, qn, qd);
    if (SQUARE) { NFE_PIPE_GEOM_AX(2, ax_zh, ax_xw) } else { NFE_ELSE_GEOM(2, gz, gx, ax_zh, ax_xw) }
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(0) NFE_PIPE_ISSUE(0, 2, 0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(1) NFE_PIPE_ISSUE(1, 2, 2)
    plane_affine_acc<SIGMA_ONLY, 1>(aff, qoff, tp[1], sg, qn, qd);
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(0)
    __builtin_amdgcn_sched_barrier(0);
    NFE_PIPE_CONSUME(1)
    plane_affine_acc<SIGMA_ONLY, 2>(aff, qoff, tp[2], sg, qn, qd);
#undef NFE_ELSE_GEOM
}

// Exchange tile: row = point (0..31), 8 granules of 4 channels per row, granule g of row r stored at position
// g ^ swz(r).  swz makes both sides conflict-free under the b128 lane groupings (MI355X guide, LDS table):
// writers = 8 consecutive lanes (2 quads, rows r and r+4) on 32 banks, readers = 16 rows per cycle on 64 banks.
__device__ __forceinline__ int xchg_swz(int row) {
    const int x = (row >> 1) & 7;
    return (x & 1) | ((x & 2) << 1) | ((x & 4) >> 1);
}

// f_quad[2i], f_quad[2i+1] (channels 16h+4c.. of quad point i)  ->  f_own[2q], f_own[2q+1] (channels 16h+4q.. of
// this lane's own point).  LDS operations of one wave execute in order, so no wait is needed between the
// stores and the loads; the fences only pin the compiler's ordering.
__device__ __forceinline__ void exchange_to_own(float* __restrict__ xp, int lane, const f32x2 (&fq)[8], f32x2 (&fo)[8]) {
    lane = launder(lane);
    const int Q = lane >> 2, c = lane & 3, h = lane >> 5, j = lane & 31;
    const int row0 = 4 * (Q & 7);
    float* wr = xp + row0 * 32 + 4 * ((4 * h + c) ^ xchg_swz(row0));       // rows row0, row0+1 (same swizzle)
    float* wr2 = xp + row0 * 32 + 4 * ((4 * h + c) ^ xchg_swz(row0 + 2));  // rows row0+2, row0+3
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    *reinterpret_cast<float4*>(wr) = make_float4(fq[0][0], fq[0][1], fq[1][0], fq[1][1]);
    *reinterpret_cast<float4*>(wr + 32) = make_float4(fq[2][0], fq[2][1], fq[3][0], fq[3][1]);
    *reinterpret_cast<float4*>(wr2 + 64) = make_float4(fq[4][0], fq[4][1], fq[5][0], fq[5][1]);
    *reinterpret_cast<float4*>(wr2 + 96) = make_float4(fq[6][0], fq[6][1], fq[7][0], fq[7][1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int rd = j * 32 + 4 * ((4 * h) ^ xchg_swz(j));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(xp + (rd ^ (4 * q)));
        fo[2 * q + 0] = f32x2{v.x, v.y};
        fo[2 * q + 1] = f32x2{v.z, v.w};
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- decoder, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) ---------------------------------------------
// FC 32->64, softplus, FC 64->32 rows; weights are the A operand (LDS), the per-point vectors are the
// B operand and never leave registers.  Register discipline: the machine scheduler
// would otherwise hoist every LDS weight read of both nets to the top (>150 VGPRs); each group of
// k-steps prefetches the next group's A fragments and ends in a scheduling fence.
__device__ __forceinline__ void mlp_fp32(const float* __restrict__ lds, const f32x2 (&f)[8], int net, int lane, f32x16& out) {
    lane = launder(lane);
    const int h = lane >> 5;
    __builtin_amdgcn_sched_barrier(0);
    f32x16 a0, a1;
    const float4* b0 = reinterpret_cast<const float4*>(lds + (net ? DEC_B_A0 : DEC_B_G0) + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 x = b0[2 * q], y = b0[8 + 2 * q];
        a0[4 * q + 0] = x.x; a0[4 * q + 1] = x.y; a0[4 * q + 2] = x.z; a0[4 * q + 3] = x.w;
        a1[4 * q + 0] = y.x; a1[4 * q + 1] = y.y; a1[4 * q + 2] = y.z; a1[4 * q + 3] = y.w;
    }
    const float4* A0 = reinterpret_cast<const float4*>(lds + (net ? DEC_A_A0 : DEC_A_G0)) + lane;
    float4 w0 = A0[0], w1 = A0[4 * 64];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        float4 n0 = w0, n1 = w1;
        if (k4 < 3) { n0 = A0[(k4 + 1) * 64]; n1 = A0[(4 + k4 + 1) * 64]; }
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, f[2 * k4][0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.x, f[2 * k4][0], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, f[2 * k4][1], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.y, f[2 * k4][1], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, f[2 * k4 + 1][0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.z, f[2 * k4 + 1][0], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, f[2 * k4 + 1][1], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1.w, f[2 * k4 + 1][1], a1, 0, 0, 0);
        w0 = n0; w1 = n1;
        __builtin_amdgcn_sched_barrier(0);
    }
    softplus_log2_x16(a0); softplus_log2_x16(a1);
    const float4* b1 = reinterpret_cast<const float4*>(lds + (net ? DEC_B_A1 : DEC_B_G1) + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 x = b1[2 * q];
        out[4 * q + 0] = x.x; out[4 * q + 1] = x.y; out[4 * q + 2] = x.z; out[4 * q + 3] = x.w;
    }
    const float4* A1 = reinterpret_cast<const float4*>(lds + (net ? DEC_A_A1 : DEC_A_G1)) + lane;
    float4 w = A1[0];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
        float4 nw = w;
        if (k4 < 7) nw = A1[(k4 + 1) * 64];
        const int kb = 4 * (k4 & 3);
        if (k4 < 4) {
            out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, a0[kb + 0], out, 0, 0, 0);
            out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, a0[kb + 1], out, 0, 0, 0);
            out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, a0[kb + 2], out, 0, 0, 0);
            out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, a0[kb + 3], out, 0, 0, 0);
        } else {
            out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, a1[kb + 0], out, 0, 0, 0);
            out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, a1[kb + 1], out, 0, 0, 0);
            out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, a1[kb + 2], out, 0, 0, 0);
            out = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, a1[kb + 3], out, 0, 0, 0);
        }
        w = nw;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- decoder, split-bf16 MFMA (v_mfma_f32_32x32x16_bf16, 3 per product: hi*hi + hi*lo + lo*hi) ------
// LDS holds the bf16 fragment image at word 0 (DEC_BF16 area of the blob) and the fp32 biases at
// DEC_B_*.  Fragment f of lane l is the uint4 at (f*64 + l).
#define NFE_MFMA_BF16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).v, (B).v, (C), 0, 0, 0)
__device__ __forceinline__ void mlp_bf16(const float* __restrict__ lds, const f32x2 (&f)[8], int net, int lane, f32x16& out) {
    lane = launder(lane);
    const int h = lane >> 5;
    const uint4* F = reinterpret_cast<const uint4*>(lds) + lane;
    __builtin_amdgcn_sched_barrier(0);
    Frag fh[2], fl[2];                       // B operand of k-step s: channels 16h + 8s + (0..7)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int w = 0; w < 4; ++w) split_pair(f[4 * s + w][0], f[4 * s + w][1], fh[s].u[w], fl[s].u[w]);
    f32x16 a0, a1;
    const float4* b0 = reinterpret_cast<const float4*>(lds + (net ? DEC_B_A0 : DEC_B_G0) + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 x = b0[2 * q], y = b0[8 + 2 * q];
        a0[4 * q + 0] = x.x; a0[4 * q + 1] = x.y; a0[4 * q + 2] = x.z; a0[4 * q + 3] = x.w;
        a1[4 * q + 0] = y.x; a1[4 * q + 1] = y.y; a1[4 * q + 2] = y.z; a1[4 * q + 3] = y.w;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        Frag h0, l0, h1, l1;
        h0.q = F[(((net * 2 + 0) * 2 + s) * 2 + 0) * 64]; l0.q = F[(((net * 2 + 0) * 2 + s) * 2 + 1) * 64];
        h1.q = F[(((net * 2 + 1) * 2 + s) * 2 + 0) * 64]; l1.q = F[(((net * 2 + 1) * 2 + s) * 2 + 1) * 64];
        a0 = NFE_MFMA_BF16(h0, fh[s], a0); a1 = NFE_MFMA_BF16(h1, fh[s], a1);
        a0 = NFE_MFMA_BF16(h0, fl[s], a0); a1 = NFE_MFMA_BF16(h1, fl[s], a1);
        a0 = NFE_MFMA_BF16(l0, fh[s], a0); a1 = NFE_MFMA_BF16(l1, fh[s], a1);
        __builtin_amdgcn_sched_barrier(0);
    }
    softplus_log2_x16(a0); softplus_log2_x16(a1);
    const float4* b1 = reinterpret_cast<const float4*>(lds + (net ? DEC_B_A1 : DEC_B_G1) + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 x = b1[2 * q];
        out[4 * q + 0] = x.x; out[4 * q + 1] = x.y; out[4 * q + 2] = x.z; out[4 * q + 3] = x.w;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {            // k-step s: hidden registers 8(s&1)..+7 of M-block s>>1
        Frag hh, hl, wh, wl;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int r = 8 * (s & 1) + 2 * w;
            if (s < 2) split_pair(a0[r], a0[r + 1], hh.u[w], hl.u[w]);
            else split_pair(a1[r], a1[r + 1], hh.u[w], hl.u[w]);
        }
        wh.q = F[(16 + (net * 4 + s) * 2 + 0) * 64]; wl.q = F[(16 + (net * 4 + s) * 2 + 1) * 64];
        out = NFE_MFMA_BF16(wh, hh, out);
        out = NFE_MFMA_BF16(wh, hl, out);
        out = NFE_MFMA_BF16(wl, hh, out);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- both decoder heads, split-bf16, software-pipelined against each other ---------------------------
// One wave runs MFMAs and VALU work in order, so a head evaluated on its own alternates between the two
// pipes (MFMA chain -> softplus -> split -> MFMA chain).  The two heads are independent: each stage below
// pairs the MFMAs of one head with the VALU work of the other, so the matrix pipe runs under the vector work.
//   S2: geometry layer 0 (12 MFMA)    | appearance features -> bf16 pairs
//   S3: appearance layer 0 (12 MFMA)  | softplus of the geometry hidden layer
//   S4: geometry layer 1 (12 MFMA)    | geometry hidden -> bf16 pairs, softplus of the appearance hidden layer
//   S5: appearance layer 1 (12 MFMA)  | appearance hidden -> bf16 pairs
#define NFE_L0_FRAG(NET, MB, S, PART) F[((((NET) * 2 + (MB)) * 2 + (S)) * 2 + (PART)) * 64]
#define NFE_L1_FRAG(NET, S, PART) F[(16 + ((NET) * 4 + (S)) * 2 + (PART)) * 64]

__device__ __forceinline__ void load_bias0(const float* __restrict__ lds, int net, int h, f32x16& a0, f32x16& a1) {
    const float4* b0 = reinterpret_cast<const float4*>(lds + (net ? DEC_B_A0 : DEC_B_G0) + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 x = b0[2 * q], y = b0[8 + 2 * q];
        a0[4 * q + 0] = x.x; a0[4 * q + 1] = x.y; a0[4 * q + 2] = x.z; a0[4 * q + 3] = x.w;
        a1[4 * q + 0] = y.x; a1[4 * q + 1] = y.y; a1[4 * q + 2] = y.z; a1[4 * q + 3] = y.w;
    }
}
__device__ __forceinline__ void load_bias1(const float* __restrict__ lds, int net, int h, f32x16& out) {
    const float4* b1 = reinterpret_cast<const float4*>(lds + (net ? DEC_B_A1 : DEC_B_G1) + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 x = b1[2 * q];
        out[4 * q + 0] = x.x; out[4 * q + 1] = x.y; out[4 * q + 2] = x.z; out[4 * q + 3] = x.w;
    }
}

// ---- sigma only (the coarse pass of a two-pass render): the geometry head's second layer reduced to its first row ----
// Of the head's 16 outputs the coarse pass uses sigma, i.e. row 0 of layer 1: 64 multiply-adds per point instead of a 32-row MFMA
// block (12 MFMAs = 384 matrix cycles) plus the hi / lo split of the 32 hidden values a lane holds (96 vector instructions).  A lane
// of half h holds the hidden units of K positions 16 s + 8 h + e (s = k-step 0..3, e = 0..7: mlp_bf16's operand order), whose row-0
// weights are the hi + lo bf16 pairs in lane 32 h of the layer-1 fragments - constants of the launch, fetched once per wave
// (sigma_row_weights) and kept in 32 registers.  fp32 products of unsplit activations: not less exact than the split-bf16 form.
__device__ __forceinline__ void sigma_row_weights(const float* __restrict__ lds, int lane, float (&w)[32]) {
    const uint4* F = reinterpret_cast<const uint4*>(lds) + 32 * (lane >> 5);         // row 0 lives in lane 32 h of every fragment
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint4 hi = F[(16 + s * 2 + 0) * 64], lo = F[(16 + s * 2 + 1) * 64];
        const unsigned hw[4] = {hi.x, hi.y, hi.z, hi.w}, lw[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w[8 * s + 2 * q] = __uint_as_float(hw[q] << 16) + __uint_as_float(lw[q] << 16);
            w[8 * s + 2 * q + 1] = __uint_as_float(hw[q] & 0xffff0000u) + __uint_as_float(lw[q] & 0xffff0000u);
        }
    }
}
// geometry head of one point, sigma only: out[0] = sigma (both lane halves), the other outputs are not computed
__device__ __forceinline__ void mlp_bf16_sigma(const float* __restrict__ lds, const f32x2 (&f)[8], int lane, const float (&w)[32], f32x16& out) {
    lane = launder(lane);
    const int h = lane >> 5;
    const uint4* F = reinterpret_cast<const uint4*>(lds) + lane;
    __builtin_amdgcn_sched_barrier(0);
    Frag fh[2], fl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) split_pair(f[4 * s + q][0], f[4 * s + q][1], fh[s].u[q], fl[s].u[q]);
    f32x16 a0, a1;
    load_bias0(lds, 0, h, a0, a1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        Frag h0, l0, h1, l1;
        h0.q = NFE_L0_FRAG(0, 0, s, 0); l0.q = NFE_L0_FRAG(0, 0, s, 1);
        h1.q = NFE_L0_FRAG(0, 1, s, 0); l1.q = NFE_L0_FRAG(0, 1, s, 1);
        a0 = NFE_MFMA_BF16(h0, fh[s], a0); a1 = NFE_MFMA_BF16(h1, fh[s], a1);
        a0 = NFE_MFMA_BF16(h0, fl[s], a0); a1 = NFE_MFMA_BF16(h1, fl[s], a1);
        a0 = NFE_MFMA_BF16(l0, fh[s], a0); a1 = NFE_MFMA_BF16(l1, fh[s], a1);
        __builtin_amdgcn_sched_barrier(0);
    }
    softplus_log2_x16(a0); softplus_log2_x16(a1);
    float p0 = 0.0f, p1 = 0.0f;               // two chains: K positions of k-steps 0, 1 (M-block 0) and 2, 3 (M-block 1)
#pragma unroll
    for (int r = 0; r < 16; ++r) { p0 = fmaf(w[r], a0[r], p0); p1 = fmaf(w[16 + r], a1[r], p1); }
    const float part = p0 + p1;
    out[0] = lds[DEC_B_G1] + (part + __shfl_xor(part, 32));
}

// hidden registers 8(s&1)..+7 of M-block s>>1 -> B operand of layer-1 k-step s
__device__ __forceinline__ void split_hidden(const f32x16& a0, const f32x16& a1, int s, Frag& hh, Frag& hl) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int r = 8 * (s & 1) + 2 * w;
        if (s < 2) split_pair(a0[r], a0[r + 1], hh.u[w], hl.u[w]);
        else split_pair(a1[r], a1[r + 1], hh.u[w], hl.u[w]);
    }
}

template <bool CROSS = false>
__device__ __forceinline__ void mlp_pair_bf16(const float* __restrict__ lds, const f32x2 (&fn)[8], const f32x2 (&fd)[8],
                                              int lane, f32x16& og, f32x16& oa) {
    lane = launder(lane);
    const int h = lane >> 5;
    const uint4* F = reinterpret_cast<const uint4*>(lds) + lane;
    __builtin_amdgcn_sched_barrier(0);
    // ---- S1
    Frag gh[2], gl[2], ah[2], al[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int w = 0; w < 4; ++w) split_pair(fn[4 * s + w][0], fn[4 * s + w][1], gh[s].u[w], gl[s].u[w]);
    f32x16 g0, g1, p0, p1;
    load_bias0(lds, 0, h, g0, g1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- S2
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        Frag h0, l0, h1, l1;
        h0.q = NFE_L0_FRAG(0, 0, s, 0); l0.q = NFE_L0_FRAG(0, 0, s, 1);
        h1.q = NFE_L0_FRAG(0, 1, s, 0); l1.q = NFE_L0_FRAG(0, 1, s, 1);
        g0 = NFE_MFMA_BF16(h0, gh[s], g0); g1 = NFE_MFMA_BF16(h1, gh[s], g1);
        g0 = NFE_MFMA_BF16(h0, gl[s], g0); g1 = NFE_MFMA_BF16(h1, gl[s], g1);
        g0 = NFE_MFMA_BF16(l0, gh[s], g0); g1 = NFE_MFMA_BF16(l1, gh[s], g1);
#pragma unroll
        for (int w = 0; w < 4; ++w) split_pair(fd[4 * s + w][0], fd[4 * s + w][1], ah[s].u[w], al[s].u[w]);
    }
    load_bias0(lds, 1, h, p0, p1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- S3
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        Frag h0, l0, h1, l1;
        h0.q = NFE_L0_FRAG(1, 0, s, 0); l0.q = NFE_L0_FRAG(1, 0, s, 1);
        h1.q = NFE_L0_FRAG(1, 1, s, 0); l1.q = NFE_L0_FRAG(1, 1, s, 1);
        p0 = NFE_MFMA_BF16(h0, ah[s], p0); p1 = NFE_MFMA_BF16(h1, ah[s], p1);
        p0 = NFE_MFMA_BF16(h0, al[s], p0); p1 = NFE_MFMA_BF16(h1, al[s], p1);
        p0 = NFE_MFMA_BF16(l0, ah[s], p0); p1 = NFE_MFMA_BF16(l1, ah[s], p1);
    }
    softplus_log2_x16(g0); softplus_log2_x16(g1);
    load_bias1(lds, 0, h, og);
    __builtin_amdgcn_sched_barrier(0);
    // ---- S4
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        Frag hh, hl, wh, wl;
        split_hidden(g0, g1, s, hh, hl);
        wh.q = NFE_L1_FRAG(0, s, 0); wl.q = NFE_L1_FRAG(0, s, 1);
        og = NFE_MFMA_BF16(wh, hh, og);
        og = NFE_MFMA_BF16(wh, hl, og);
        og = NFE_MFMA_BF16(wl, hh, og);
    }
    softplus_log2_x16(p0); softplus_log2_x16(p1);
    load_bias1(lds, 1, h, oa);
    __builtin_amdgcn_sched_barrier(0);
    // ---- S5
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        Frag hh, hl, wh, wl;
        split_hidden(p0, p1, s, hh, hl);
        wh.q = NFE_L1_FRAG(1, s, 0); wl.q = NFE_L1_FRAG(1, s, 1);
        oa = NFE_MFMA_BF16(wh, hh, oa);
        oa = NFE_MFMA_BF16(wh, hl, oa);
        oa = NFE_MFMA_BF16(wl, hh, oa);
        if (CROSS) {        // geometry-head rows that read the appearance head's hidden layer
            const uint4* X = reinterpret_cast<const uint4*>(lds + LDS_CROSS) + lane;
            Frag xh, xl;
            xh.q = X[(2 * s) * 64]; xl.q = X[(2 * s + 1) * 64];
            og = NFE_MFMA_BF16(xh, hh, og);
            og = NFE_MFMA_BF16(xh, hl, og);
            og = NFE_MFMA_BF16(xl, hh, og);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// The decoder proper (DisentangledOSGDecoder.forward after the mean over planes, triplane.py:254-270) on features that
// are already in "own" layout: fn / fd = channels 16h..16h+15 of this lane's point of the geometry / appearance set.
// Outputs as eval_point documents them.
template <bool SIGMA_ONLY, int MATH, bool CROSS>
__device__ __forceinline__ void decode_features(const float* __restrict__ lds, const f32x2 (&fn)[8], const f32x2 (&fd)[8],
                                                int lane, f32x16& og, f32x16& oa, const float (&wsig)[32]) {
    constexpr bool PAIR = !SIGMA_ONLY && MATH == NFE_MATH_BF16X3;
    static_assert(!CROSS || PAIR, "the cross term lives in the paired split-bf16 decoder");
    if (PAIR) mlp_pair_bf16<CROSS>(lds, fn, fd, lane, og, oa);
    else if (MATH == NFE_MATH_FP32) mlp_fp32(lds, fn, 0, lane, og);
    else if (SIGMA_ONLY) mlp_bf16_sigma(lds, fn, lane, wsig, og);      // wsig: sigma_row_weights(), once per wave
    else mlp_bf16(lds, fn, 0, lane, og);
    if (!SIGMA_ONLY) {
        if (MATH == NFE_MATH_FP32) mlp_fp32(lds, fd, 1, lane, oa);
        else if (!PAIR) mlp_bf16(lds, fd, 1, lane, oa);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {    // oa carries log2(e): sigmoid(x) = 1/(1 + 2^-(x log2 e))
            const f32x2 d = f32x2{exp2_fast(-oa[r]), exp2_fast(-oa[r + 1])} + splat(1.0f);
            const f32x2 sg = pk_fma(f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])}, splat(1.002f), splat(-0.001f));
            oa[r] = sg[0]; oa[r + 1] = sg[1];    // sigmoid(x)*(1+2*0.001) - 0.001, triplane.py:269
        }
    }
}

// Evaluate the implicit model at one point per lane PAIR (lanes j and j+32 share a point; lane half
// h holds channels [16h,16h+16) of every 32-vector).  Returns, for this lane:
//   og[0] = sigma; og[2..] = seg channels (h=0: seg 0..7 in og[2..9]; h=1: seg 8..14 in og[2..8]); og[1] unused
//   (seg starts on an even register so packed-fp32 pairs need no realigning moves)
//   oa[r] = rgb channel 16h + r   (after the sigmoid clamp, triplane.py:269)
// All 64 lanes must be active (quad broadcasts and the LDS exchange involve the whole wave).
template <bool DUAL, bool SIGMA_ONLY, int MATH, bool CROSS = false, bool SQUARE = false>
__device__ __forceinline__ void eval_point(const float* __restrict__ pg, const float* __restrict__ pa,
                                           int H, int W, const float* __restrict__ lds,
                                           const float* __restrict__ aff, float* __restrict__ xp,
                                           float gx, float gy, float gz,
                                           int lane, f32x16& og, f32x16& oa, const float (&wsig)[32]) {
    f32x2 qn[8], qd[8];          // quad layout: [2i], [2i+1] = channels 16h+4c.. of quad point i
    const int l0 = launder(lane);
    const int qoff0 = (l0 >> 5) * 16 + (l0 & 3) * 4;
    {   // Start from the plane-summed shift (all taps inside: sum of weights == 1); out-of-range samples are
        // corrected below.  sample(norm) = scale * sample(raw) + shift * sum(weights), DESIGN.md section 3.
        const float4 bn = *reinterpret_cast<const float4*>(aff + AFF_BSUM + qoff0);
        const float4 bd = *reinterpret_cast<const float4*>(aff + AFF_BSUM + 32 + qoff0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qn[2 * i] = f32x2{bn.x, bn.y}; qn[2 * i + 1] = f32x2{bn.z, bn.w};
            qd[2 * i] = f32x2{bd.x, bd.y}; qd[2 * i + 1] = f32x2{bd.z, bd.w};
        }
    }
    const long long plane_elems = (long long)H * W * 32;
    if (DUAL && !SIGMA_ONLY) gather_pipelined_dual<SQUARE>(pg, pa, H, W, plane_elems, aff, lane, gx, gy, gz, qn, qd);
    else gather_pipelined<SIGMA_ONLY, SQUARE>(pg, H, W, plane_elems, aff, lane, gx, gy, gz, qn, qd);
    f32x2 fn[8], fd[8];          // own layout: channels 16h..16h+15 of this lane's point
    exchange_to_own(xp, lane, qn, fn);
    if (!SIGMA_ONLY) exchange_to_own(xp, lane, qd, fd);
    decode_features<SIGMA_ONLY, MATH, CROSS>(lds, fn, fd, lane, og, oa, wsig);
}

// Copy the decoder image for this math mode into LDS words [0, DEC_FLOATS): fragments then biases.
template <int MATH>
__device__ __forceinline__ void stage_decoder(const float* __restrict__ dec, float* lds) {
    const float4* frag = reinterpret_cast<const float4*>(dec + (MATH == NFE_MATH_FP32 ? 0 : DEC_BF16));
    for (int i = threadIdx.x; i < DEC_B_G0 / 4; i += 256) reinterpret_cast<float4*>(lds)[i] = frag[i];
    const float4* bias = reinterpret_cast<const float4*>(dec + DEC_B_G0);
    for (int i = threadIdx.x; i < (DEC_FLOATS - DEC_B_G0) / 4; i += 256) reinterpret_cast<float4*>(lds + DEC_B_G0)[i] = bias[i];
}

__device__ __forceinline__ void stage_cross(const float* __restrict__ cross, float* lds) {
    for (int i = threadIdx.x; i < CROSS_WORDS / 4; i += 256)
        reinterpret_cast<float4*>(lds + LDS_CROSS)[i] = reinterpret_cast<const float4*>(cross)[i];
}

// Stage this wave's view affines into its LDS region, folding in the 1/3 of the mean over planes
// (triplane.py:251-252).
__device__ __forceinline__ void stage_affine(const float* const (&src)[4], int n, float* aff, int lane) {
#pragma unroll
    for (int arr = 0; arr < 4; ++arr) {
        const float* p = src[arr];
        const float dflt = (arr & 1) ? 0.0f : 1.0f;
        for (int c = lane; c < 96; c += 64) {
            float v = p ? p[(long long)n * 96 + c] : dflt;
            aff[arr * 96 + c] = v * (1.0f / 3.0f);
        }
    }
    if (lane < 64) {   // plane-summed shifts: [0,32) geometry set, [32,64) appearance set
        const float* p = src[(lane < 32) ? 1 : 3];
        const int c = lane & 31;
        float b = 0.0f;
        if (p) b = (p[(long long)n * 96 + c] * (1.0f / 3.0f) + p[(long long)n * 96 + 32 + c] * (1.0f / 3.0f)) + p[(long long)n * 96 + 64 + c] * (1.0f / 3.0f);
        aff[AFF_BSUM + lane] = b;
    }
    __threadfence_block();
}

// N(0,1) attached to a sample by (seed, ray, draw index): coarse sample k -> k, fine sample of ascending rank r -> D + r.
// The coarse samples re-evaluated in the final pass of a two-pass render therefore get the very noise they had in the
// coarse pass, as in the reference, where the coarse sigmas (noise included) are the ones merged (renderer.py:333-358).
// Box-Muller on two Philox words (stream 2).
__device__ __forceinline__ float sample_gaussian(unsigned long long seed, unsigned ray, unsigned draw) {
    const u32x4 r = philox4x32_10(ray, draw, 2u, 0u, (unsigned)seed, (unsigned)(seed >> 32));
    const float u1 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = u01(r.y);
    return sqrtf(-2.0f * LN2 * log2_fast(u1)) * __builtin_amdgcn_cosf(u2);       // v_cos_f32 takes revolutions
}

template <bool DUAL, bool SIGMA_ONLY, int MATH, bool NOISE = false, bool CROSS = false, bool SPLIT = false, bool SQUARE = false, bool EVAL = false, bool STORE = false>
__global__ __launch_bounds__(256, 2) void render_kernel(RenderK P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_decoder<MATH>(P.dec, lds);
    if (CROSS) stage_cross(P.dec_cross, lds);
    // wave id in an SGPR: everything derived from it (ray block, view, plane base) stays scalar
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    float* aff = lds + LDS_AFF + wave * WAVE_LDS_FLOATS;
    float* xp = aff + AFF_FLOATS;
    __syncthreads();
    // In-run shader clock (bench.py): shader-cycle counter against the 100 MHz reference counter, stamped by one lane at the two
    // ends of the launch (the grid is persistent: workgroup 0 lives as long as the kernel).  Nothing is stamped inside the loop.
    const bool probe = P.clock_probe != nullptr && blockIdx.x == 0 && wave == 0;
    if (probe && lane == 0) { P.clock_probe[0] = __builtin_amdgcn_s_memtime(); P.clock_probe[1] = __builtin_amdgcn_s_memrealtime(); }
    float wsig[32] = {};                     // sigma-only pass, split-bf16: row 0 of the geometry head's second layer (mlp_bf16_sigma)
    if (SIGMA_ONLY && MATH == NFE_MATH_BF16X3) sigma_row_weights(lds, lane, wsig);

    const int S = P.S;
    const unsigned long long seed = P.seed_dev ? *P.seed_dev : P.seed;
    const int blocks_per_view = (P.M + 31) >> 5;
    const long long total_rb = (long long)P.N * blocks_per_view;
    const long long n_waves = (long long)gridDim.x * 4;
    int cur_view = -1;
    float tmin = INFINITY, tmax = -INFINITY;

    const int nseg = SPLIT ? P.seg_count : 1;
#pragma unroll 1
    for (long long item = (long long)blockIdx.x * 4 + wave; item < total_rb * nseg; item += n_waves) {
        const long long rb = SPLIT ? item / nseg : item;
        const int seg = SPLIT ? (int)(item % nseg) : 0;
        const int n = (int)(rb / blocks_per_view), b = (int)(rb % blocks_per_view);
        if (n != cur_view) { stage_affine(P.aff, n, aff, lane); cur_view = n; }

        // ---- ray for this lane pair -------------------------------------------------------
        int m, px = 0, py = 0;
        if (P.tiled) {   // 8x4 pixel tile per wave: neighbouring rays hit neighbouring texels
            const int tiles_x = P.R >> 3;
            px = (b % tiles_x) * 8 + (j & 7);
            py = (b / tiles_x) * 4 + (j >> 3);
            m = py * P.R + px;
        } else {
            m = b * 32 + j;
            if (P.R > 0) { py = min(m, P.M - 1) / P.R; px = min(m, P.M - 1) % P.R; }
        }
        const bool valid = m < P.M;
        m = min(m, P.M - 1);
        const long long ray = (long long)n * P.M + m;
        float ox, oy, oz, dx, dy, dz;
        if (P.origins) {
            const float* o = P.origins + ray * 3; const float* d = P.dirs + ray * 3;
            ox = o[0]; oy = o[1]; oz = o[2]; dx = d[0]; dy = d[1]; dz = d[2];
        } else {
            // RaySampler.forward, ray_sampler.py:35-61
            const float* c = P.cam2world + n * 16; const float* K = P.intrinsics + n * 9;
            const float fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
            const float inv = 1.0f / (float)P.R;
            const float xc = (float)px * inv + 0.5f * inv, yc = (float)py * inv + 0.5f * inv;
            const float xl = (xc - cx + cy * sk / fy - sk * yc / fy) / fx;
            const float yl = (yc - cy) / fy;
            ox = c[3]; oy = c[7]; oz = c[11];
            float wx = c[0] * xl + c[1] * yl + c[2] + c[3];
            float wy = c[4] * xl + c[5] * yl + c[6] + c[7];
            float wz = c[8] * xl + c[9] * yl + c[10] + c[11];
            dx = wx - ox; dy = wy - oy; dz = wz - oz;
            float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
            dx /= nrm; dy /= nrm; dz /= nrm;
        }
        const float* pg = P.planes_g + (long long)n * P.plane_view_stride;
        const float* pa = P.planes_a + (long long)n * P.plane_view_stride;

        float ev_cr[16], ev_cs[8];            // EVAL: this lane half's cotangents (2 * g_rgb[16h..], g_seg[8h..])
        if (EVAL) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ev_cr[r] = P.ev_g_rgb ? 2.0f * (P.ev_channels_first ? P.ev_g_rgb[((long long)n * 32 + 16 * h + r) * P.M + m]
                                                                      : P.ev_g_rgb[ray * 32 + 16 * h + r]) : 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                ev_cs[c] = (P.ev_g_seg && 8 * h + c < 15) ? (P.ev_channels_first ? P.ev_g_seg[((long long)n * 15 + 8 * h + c) * P.M + m]
                                                                                   : P.ev_g_seg[ray * 15 + 8 * h + c]) : 0.0f;
        }
        // ---- depth schedule (sample_stratified, renderer.py:169-192) -------------------------
        float rs = P.ray_start, re = P.ray_end;
        if (P.depth_mode == DEPTH_PER_RAY) { rs = P.rs_ray[ray]; re = P.re_ray[ray]; }
        const float inv_dm1 = 1.0f / (float)(S - 1);
        const float delta = (re - rs) / (float)(S - 1);

        // ---- march state (SegMipRayMarcher2.run_forward, ray_marcher.py:68-101) --------------
        f32x2 acc_rgb[8], acc_seg[4], prev_rgb[8], prev_seg[4];
        float acc_d = 0.0f, acc_w = 0.0f, T = 1.0f, prev_t = 0.0f, prev_sig = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) { acc_rgb[c] = splat(0.0f); prev_rgb[c] = splat(0.0f); }
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc_seg[c] = splat(0.0f); prev_seg[c] = splat(0.0f); }
        u32x4 rnd = {0, 0, 0, 0};

        // samples [k0, k1) belong to this wave; a later segment first evaluates sample k0-1 (no compositing) to have the
        // left end of its first mid-point interval
        const int k0 = SPLIT ? (int)((long long)seg * S / nseg) : 0, k1 = SPLIT ? (int)((long long)(seg + 1) * S / nseg) : S;
        const int kfirst = EVAL ? k0 : ((SPLIT && k0 > 0) ? k0 - 1 : 0);     // EVAL: samples are independent, nothing to prime
#pragma unroll 1
        for (int k = kfirst; k < k1; ++k) {
            float t;
            if (P.depth_mode == DEPTH_BUFFER) {
                t = P.depth_buf[ray * S + k];
            } else {
                float u;
                if (P.u) {
                    u = P.u[ray * S + k];
                } else {
                    if ((k & 3) == 0 || (SPLIT && k == kfirst))
                        rnd = philox4x32_10((unsigned)ray, (unsigned)(k >> 2), 0u, 0u,
                                            (unsigned)seed, (unsigned)(seed >> 32));
                    unsigned bits = (k & 3) == 0 ? rnd.x : (k & 3) == 1 ? rnd.y : (k & 3) == 2 ? rnd.z : rnd.w;
                    u = u01(bits);
                }
                if (P.depth_mode == DEPTH_DISPARITY) {
                    float s = (float)k * inv_dm1 + u * inv_dm1;
                    t = 1.0f / (1.0f / rs * (1.0f - s) + 1.0f / re * s);
                } else if (P.depth_mode == DEPTH_PER_RAY) {
                    t = rs + ((float)k / (float)(S - 1)) * (re - rs) + u * delta;
                } else {
                    t = fmaf((float)k, delta, rs) + u * delta;
                }
            }
            if (P.out_depths && valid && h == 0) P.out_depths[ray * S + k] = t;
            tmin = fminf(tmin, t); tmax = fmaxf(tmax, t);

            const float gx = P.coord_scale * fmaf(t, dx, ox);
            const float gy = P.coord_scale * fmaf(t, dy, oy);
            const float gz = P.coord_scale * fmaf(t, dz, oz);
            f32x16 og, oa;
            // Opaque zero: keeps the (loop-invariant) LDS weight reads inside the loop; hoisted, they
            // would pin >200 VGPRs per lane and spill.
            int opq;
            asm volatile("s_mov_b32 %0, 0" : "=s"(opq));
            eval_point<DUAL, SIGMA_ONLY, MATH, CROSS, SQUARE>(pg, pa, P.H, P.W, lds + opq, aff + opq, xp + opq, gx, gy, gz, lane, og, oa, wsig);

            if (NOISE) {
                const unsigned draw = (P.depth_mode == DEPTH_BUFFER && P.src_buf) ? (unsigned)P.src_buf[ray * S + k] : (unsigned)k;
                const float nv = P.noise_buf ? P.noise_buf[(long long)ray * P.noise_stride + draw] : sample_gaussian(seed, (unsigned)ray, draw);
                og[0] = fmaf(P.density_noise, nv, og[0]);
            }
            if (STORE) {      // keep what the decoders returned (nfe_render_args.tap_sample_colors): lane = ray of the block, 128-byte rows
                float* cb = P.tap_colors + (((long long)rb * S + k) * 48) * 32 + j;
#pragma unroll
                for (int r = 0; r < 16; ++r) cb[(16 * h + r) * 32] = oa[r];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (8 * h + c < 15) cb[(32 + 8 * h + c) * 32] = og[2 + c];
                if (h == 0) cb[47 * 32] = og[0];
            }
            if (EVAL) {
                float a = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) a = fmaf(ev_cr[r], oa[r], a);
#pragma unroll
                for (int c = 0; c < 8; ++c) a = fmaf(ev_cs[c], og[2 + c], a);
                a += __shfl_xor(a, 32);                                   // the two channel halves of the point
                if (valid && h == 0) { const long long e = bwd_slot_base(P.R, P.M, S, n, m) + 64ll * k; P.ev_sig[e] = og[0]; P.ev_a[e] = a; }
                continue;
            }
            {   // Branch-free: the first sample of a march (or of a depth segment) composites with a zero-length interval, i.e.
                // alpha = 0, w = 0, T unchanged (1 + 1e-10 == 1 in fp32), every accumulator += 0.  A uniform `if (k > kfirst)`
                // here made every loop-carried accumulator a phi and cost ~57 register copies per step at the back edge.
                const bool first = k == kfirst;
                const float dlt = t - (first ? t : prev_t);
                const float dens = softplus_f((prev_sig + og[0]) * 0.5f - 1.0f);
                // `first` forces alpha = 0 by a select on the (scalar) condition: with dlt = 0 a non-finite density (sigma = +inf,
                // or a huge value + density_noise) would give 0 * inf = NaN and poison T and every accumulator of the ray; the
                // reference's first interval starts at the NEXT sample, where inf gives alpha = 1 (ray_marcher.py:79-83).
                const float alpha = first ? 0.0f : 1.0f - exp2_fast(-(dens * dlt) * LOG2E);
                const float w = alpha * T;
                T = T * (1.0f - alpha + 1e-10f);
                if (P.out_weights && valid && h == 0 && !first) P.out_weights[ray * (S - 1) + (k - 1)] = w;
                if (!SIGMA_ONLY) {
                    const f32x2 wh = splat(w * 0.5f);      // w * (a + b)/2 == (w/2) * (a + b), exactly
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        acc_rgb[c] = pk_fma(wh, prev_rgb[c] + f32x2{oa[2 * c], oa[2 * c + 1]}, acc_rgb[c]);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc_seg[c] = pk_fma(wh, prev_seg[c] + f32x2{og[2 + 2 * c], og[3 + 2 * c]}, acc_seg[c]);
                    acc_d = fmaf(w, (prev_t + t) * 0.5f, acc_d);
                    acc_w += w;
                }
            }
            prev_t = t; prev_sig = og[0];
            if (!SIGMA_ONLY) {
#pragma unroll
                for (int c = 0; c < 8; ++c) prev_rgb[c] = f32x2{oa[2 * c], oa[2 * c + 1]};
#pragma unroll
                for (int c = 0; c < 4; ++c) prev_seg[c] = f32x2{og[2 + 2 * c], og[3 + 2 * c]};
            }
        }

        // ---- outputs -------------------------------------------------------------------------
        if (EVAL) continue;
        if (SPLIT) {
            if (valid) {
                float* pp = P.partials + (ray * nseg + seg) * PARTIAL_FLOATS;
                if (!SIGMA_ONLY) {
                    float4* o = reinterpret_cast<float4*>(pp + 16 * h);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[q] = make_float4(acc_rgb[2 * q][0], acc_rgb[2 * q][1], acc_rgb[2 * q + 1][0], acc_rgb[2 * q + 1][1]);
                    const int ns = h ? 7 : 8;
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c < ns) pp[32 + 8 * h + c] = acc_seg[c >> 1][c & 1];
                }
                if (h == 0) { pp[47] = acc_d; pp[48] = acc_w; pp[49] = T; }
            }
            continue;
        }
        if (!SIGMA_ONLY && valid) {
            const float wb = P.white_back ? (1.0f - acc_w) : 0.0f;
            float rgbv[16], segv[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {                                                  // :96-99
                rgbv[2 * c] = (acc_rgb[c][0] + wb) * 2.0f - 1.0f;
                rgbv[2 * c + 1] = (acc_rgb[c][1] + wb) * 2.0f - 1.0f;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) { segv[2 * c] = acc_seg[c][0]; segv[2 * c + 1] = acc_seg[c][1]; }
            const int nseg = h ? 7 : 8;
            if (P.channels_first) {
#pragma unroll
                for (int c = 0; c < 16; ++c) P.rgb[((long long)n * 32 + 16 * h + c) * P.M + m] = rgbv[c];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < nseg) P.seg[((long long)n * 15 + 8 * h + c) * P.M + m] = segv[c];
            } else {
                float4* o = reinterpret_cast<float4*>(P.rgb + ray * 32 + 16 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q] = make_float4(rgbv[4 * q], rgbv[4 * q + 1], rgbv[4 * q + 2], rgbv[4 * q + 3]);
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < nseg) P.seg[ray * 15 + 8 * h + c] = segv[c];
            }
            if (h == 0) {
                P.depth[ray] = acc_d / acc_w;     // NaN when acc_w == 0; fixed by depth_clamp_kernel
                P.wsum[ray] = acc_w;
            }
        }
    }

    // ---- whole-tensor depth bounds for the clamp (ray_marcher.py:94) -----------------------------
    if (P.depth_minmax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            tmin = fminf(tmin, __shfl_xor(tmin, off));
            tmax = fmaxf(tmax, __shfl_xor(tmax, off));
        }
        if (lane == 0 && tmin <= tmax) {
            atomicMin(P.depth_minmax + 0, f2ord(tmin));
            atomicMax(P.depth_minmax + 1, f2ord(tmax));
        }
    }
    if (probe && lane == 0) { P.clock_probe[2] = __builtin_amdgcn_s_memtime(); P.clock_probe[3] = __builtin_amdgcn_s_memrealtime(); }
}


// ------------------------------------------------------------------------------------------
// Wave-specialised form of the single-set, split-bf16 render launch (round 4).
//
// render_kernel above is one program per wave: gather (112 VGPRs at its peak), decoder (96 + fragments) and march state (48) add
// up to 256 registers + spills = two waves per SIMD, and the wave spends half its life not issuing.  Here the two
// halves of that program run in DIFFERENT waves of one workgroup, so that neither needs more than 128 registers (four waves per
// SIMD) and the matrix work of one wave runs beside the texture / vector work of another:
//   producer wave p (waves 0 .. NP-1):   depth schedule, tap geometry, quad-cooperative gather, per-plane affines  ->  the two
//       32-point x 32-channel feature tiles of its pair in LDS - the very tile exchange_to_own() used for the quad -> own
//       transposition, so the hand-off costs no LDS traffic the fused kernel did not already have;
//   consumer wave NP + p:                reads its lane's 16 channels in MFMA-operand order, both decoder heads (split-bf16 MFMA),
//       sigmoid, mid-point compositing, outputs.
// Synchronisation is two monotone counters per pair in LDS (`full`: tiles written by the producer, `taken`: tiles read by the
// consumer; geometry and appearance tiles count separately, so each counter moves twice per sample).  Tiles are single-buffered:
// the producer works on sample k+1 while the consumer decodes sample k and only blocks at its write if the consumer has not yet
// read the previous tile - the consumer is the slower role, so that wait is the back-pressure that keeps the pair in step.
// s_barrier cannot do this (every wave of the block would wait for the slowest), and gfx950 has no named barriers.
// Waves w and w + 4 of a workgroup share a SIMD (MI355X_MICROARCH.md, "waves go to SIMDs in the cyclic order"), so with NP = 4
// every SIMD holds a producer and the consumer of the same pair, twice over with two workgroups per CU.
// Outputs are bit-identical to render_kernel's: same tap order in the bilinear sums, same MFMA order in each head (mlp_bf16), same
// march arithmetic (tests/test_render_gpu.py::test_wave_specialised_kernel_is_bit_identical).
// ------------------------------------------------------------------------------------------
#define WS_BUFS 1                                           // tile sets per pair: 2 lets the producer run one sample further ahead (measured: no gain, 6.13 vs 6.12 ms)
constexpr int WS_SET_FLOATS = 2 * XCHG_FLOATS + 32;         // one set: geometry tile, appearance tile, 32 depths
constexpr int WS_T_OFF = 2 * XCHG_FLOATS;
constexpr int WS_FLAG_OFF = WS_BUFS * WS_SET_FLOATS;        // behind the sets: {full, taken, abort, -}
constexpr int WS_PAIR_FLOATS = AFF_FLOATS + WS_FLAG_OFF + 4;
static_assert(WS_PAIR_FLOATS % 4 == 0 && AFF_FLOATS % 4 == 0, "16-byte aligned tiles");
template <int NP> constexpr int ws_lds_bytes() { return (DEC_FLOATS + NP * WS_PAIR_FLOATS) * 4; }
#define WS_SLEEP 2                                          // s_sleep argument of the hand-off poll (measured 0 / 2 / 8: 6.50 / 6.47 / 6.44 ms - polling is not the cost)
constexpr int WS_SPIN_LIMIT = 1 << 18;                      // x (s_sleep 1 + an LDS read) ~ 50 ms: a lost partner ends the wait, never the box

// Wait until *flag has reached `need` (wrap-safe).  Returns false when the wait was abandoned: the flag's pair is then marked
// aborted and every later wait of the pair returns at once - the launch finishes with garbage in that pair's rays and a count in
// depth_minmax[2] instead of hanging the GPU.  The count is not silent: depth_clamp_kernel, which closes every nfe_render
// call, turns ALL outputs of a call with a non-zero count into NaN, leaves the count in the call's workspace (nfe_render_call_status:
// NFE_EHANDOFF for THAT call) and adds it to the process's sticky status word (nfe_render.h, "lost hand-offs").
__device__ int g_ws_spin_limit = WS_SPIN_LIMIT;            // device global, read only on the slow path of a wait (NFE_WS_SPIN_LIMIT: the abort test shortens it)
__device__ __forceinline__ bool ws_wait(unsigned* flags, int which, unsigned need) {
    int spins = 0;
    while (true) {
        const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flags + which, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(v - need) >= 0) return true;
        const unsigned ab = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (ab != 0u) return false;
        if (++spins > __builtin_nontemporal_load(&g_ws_spin_limit)) {
            __hip_atomic_store(flags + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return false;
        }
        __builtin_amdgcn_s_sleep(WS_SLEEP);
    }
}
// Publish: every LDS access this wave issued before is complete (release), then the counter moves.
__device__ __forceinline__ void ws_signal(unsigned* flags, int which, unsigned value, int lane) {
    if (lane == 0) __hip_atomic_store(flags + which, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The producer's gather: as gather_pipelined, but a batch is ONE tap of the quad's four points (4 loads) and three batches are in
// flight (48 data registers instead of 64): the tap order of the bilinear sums is unchanged.
#define NFE_WS_ISSUE_I(S, I)                                                                               \
    vg[S][I] = texel_piece(base_, (unsigned)quad_bcast<I>((int)off_) + qoff_bytes); wq[S][I] = quad_swizzle<I>(wk_);
#define NFE_WS_ISSUE(S, PL, K)                                                                             \
    {                                                                                                      \
        const float* base_ = pg + (PL) * plane_elems;                                                      \
        const unsigned off_ = offs[PL][K];                                                                 \
        const float wk_ = tp[PL].w[K];                                                                     \
        NFE_WS_ISSUE_I(S, 0) NFE_WS_ISSUE_I(S, 1) NFE_WS_ISSUE_I(S, 2) NFE_WS_ISSUE_I(S, 3)                \
    }
#define NFE_WS_CONSUME(S)                                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                     \
        const f32x2 w2 = splat(wq[S][i_]);                                                                 \
        sg[2 * i_ + 0] = pk_fma(w2, f32x2{vg[S][i_].x, vg[S][i_].y}, sg[2 * i_ + 0]);                      \
        sg[2 * i_ + 1] = pk_fma(w2, f32x2{vg[S][i_].z, vg[S][i_].w}, sg[2 * i_ + 1]);                      \
    }
#define NFE_WS_FLIGHT 2            // tap batches (4 loads each) in flight in the producer's gather: 2 or 3
// the producer's gather on the in-bounds fast path (see gather_pipelined_inb): one byte offset per (plane, point)
#define NFE_WSI_ISSUE(S, PL, K)                                                                            \
    {                                                                                                      \
        const char* b_ = reinterpret_cast<const char*>(pg + (PL) * plane_elems) + (((K) >> 1) ? row_bytes : 0); \
        const float wk_ = wt[PL][K];                                                                       \
        vg[S][0] = texel_piece_imm<((K) & 1) * 128>(b_, vo[PL][0]); wq[S][0] = quad_swizzle<0>(wk_);       \
        vg[S][1] = texel_piece_imm<((K) & 1) * 128>(b_, vo[PL][1]); wq[S][1] = quad_swizzle<1>(wk_);       \
        vg[S][2] = texel_piece_imm<((K) & 1) * 128>(b_, vo[PL][2]); wq[S][2] = quad_swizzle<2>(wk_);       \
        vg[S][3] = texel_piece_imm<((K) & 1) * 128>(b_, vo[PL][3]); wq[S][3] = quad_swizzle<3>(wk_);       \
    }
#define NFE_WSI_VOFF(PL)                                                                                   \
    vo[PL][0] = (unsigned)quad_bcast<0>((int)off0[PL]) + qoff_bytes; vo[PL][1] = (unsigned)quad_bcast<1>((int)off0[PL]) + qoff_bytes; \
    vo[PL][2] = (unsigned)quad_bcast<2>((int)off0[PL]) + qoff_bytes; vo[PL][3] = (unsigned)quad_bcast<3>((int)off0[PL]) + qoff_bytes;
template <bool SIGMA_ONLY>
__device__ __forceinline__ void gather_pipelined_ws_inb(const float* __restrict__ pg, int W, long long plane_elems,
                                                        const float* __restrict__ aff, int lane, const InbAxes& a,
                                                        f32x2 (&qn)[8], f32x2 (&qd)[8]) {
    unsigned off0[3];
    float wt[3][4];
    unsigned vo[3][4];
    const long long row_bytes = (long long)W * 128;
    const int ll = launder(lane);
    const int qoff = (ll >> 5) * 16 + (ll & 3) * 4;
    const unsigned qoff_bytes = (unsigned)qoff * 4u;
    float4 vg[NFE_WS_FLIGHT][4];
    float wq[NFE_WS_FLIGHT][4];
    f32x2 sg[8];
    const Taps none{};
#pragma unroll
    for (int c = 0; c < 8; ++c) sg[c] = splat(0.0f);
#define NFE_WS_SLOT(N) ((N) % NFE_WS_FLIGHT)
#define NFE_WSI_STEP(N)                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    NFE_WS_CONSUME(NFE_WS_SLOT(N))                                                                         \
    if ((N) + NFE_WS_FLIGHT < 12) NFE_WSI_ISSUE(NFE_WS_SLOT(N), ((N) + NFE_WS_FLIGHT) / 4, ((N) + NFE_WS_FLIGHT) % 4)
    NFE_INB_PLANE(0, a.x0, a.y0, a.fx, a.fy)
    NFE_WSI_VOFF(0)
    NFE_WSI_ISSUE(0, 0, 0) NFE_WSI_ISSUE(1, 0, 1)
    if (NFE_WS_FLIGHT == 3) NFE_WSI_ISSUE(2 % NFE_WS_FLIGHT, 0, 2)
    NFE_INB_PLANE(1, a.x0, a.z0, a.fx, a.fz)
    NFE_WSI_VOFF(1)
    NFE_WSI_STEP(0) NFE_WSI_STEP(1) NFE_WSI_STEP(2) NFE_WSI_STEP(3)
    plane_affine_acc<SIGMA_ONLY, 0, false, true>(aff, qoff, none, sg, qn, qd);
    NFE_INB_PLANE(2, a.z0, a.x0, a.fz, a.fx)
    NFE_WSI_VOFF(2)
    NFE_WSI_STEP(4) NFE_WSI_STEP(5) NFE_WSI_STEP(6) NFE_WSI_STEP(7)
    plane_affine_acc<SIGMA_ONLY, 1, false, true>(aff, qoff, none, sg, qn, qd);
    NFE_WSI_STEP(8) NFE_WSI_STEP(9) NFE_WSI_STEP(10) NFE_WSI_STEP(11)
    plane_affine_acc<SIGMA_ONLY, 2, false, true>(aff, qoff, none, sg, qn, qd);
#undef NFE_WSI_STEP
#undef NFE_WS_SLOT
}

template <bool SQUARE, bool SIGMA_ONLY>
__device__ __forceinline__ void gather_pipelined_ws(const float* __restrict__ pg, int H, int W, long long plane_elems,
                                                    const float* __restrict__ aff, int lane, float gx, float gy, float gz,
                                                    f32x2 (&qn)[8], f32x2 (&qd)[8]) {
    if (SQUARE) {
        InbAxes ia;
        if (inb_axes(W, gx, gy, gz, ia)) { gather_pipelined_ws_inb<SIGMA_ONLY>(pg, W, plane_elems, aff, lane, ia, qn, qd); return; }
    }
    Taps tp[3];
    unsigned offs[3][4];
    Axis ax_xw, ax_zh;
    if (SQUARE) { ax_xw = axis_geometry(W, gx); NFE_PIPE_GEOM_AX(0, ax_xw, axis_geometry(H, gy)) } else { NFE_PIPE_GEOM(0, gx, gy) }
    const int ll = launder(lane);
    const int qoff = (ll >> 5) * 16 + (ll & 3) * 4;
    const unsigned qoff_bytes = (unsigned)qoff * 4u;
    float4 vg[NFE_WS_FLIGHT][4];
    float wq[NFE_WS_FLIGHT][4];
    f32x2 sg[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) sg[c] = splat(0.0f);
    // tap n = 4 * plane + k goes through slot n % NFE_WS_FLIGHT; consume(n) is followed by issue(n + NFE_WS_FLIGHT)
#define NFE_WS_SLOT(N) ((N) % NFE_WS_FLIGHT)
#define NFE_WS_STEP(N)                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    NFE_WS_CONSUME(NFE_WS_SLOT(N))                                                                         \
    if ((N) + NFE_WS_FLIGHT < 12) NFE_WS_ISSUE(NFE_WS_SLOT(N), ((N) + NFE_WS_FLIGHT) / 4, ((N) + NFE_WS_FLIGHT) % 4)
    NFE_WS_ISSUE(0, 0, 0) NFE_WS_ISSUE(1, 0, 1)
    if (NFE_WS_FLIGHT == 3) NFE_WS_ISSUE(2 % NFE_WS_FLIGHT, 0, 2)
    if (SQUARE) { ax_zh = axis_geometry(H, gz); NFE_PIPE_GEOM_AX(1, ax_xw, ax_zh) } else { NFE_PIPE_GEOM(1, gx, gz) }
    NFE_WS_STEP(0) NFE_WS_STEP(1) NFE_WS_STEP(2) NFE_WS_STEP(3)
    plane_affine_acc<SIGMA_ONLY, 0>(aff, qoff, tp[0], sg, qn, qd);
    if (SQUARE) { NFE_PIPE_GEOM_AX(2, ax_zh, ax_xw) } else { NFE_PIPE_GEOM(2, gz, gx) }
    NFE_WS_STEP(4) NFE_WS_STEP(5) NFE_WS_STEP(6) NFE_WS_STEP(7)
    plane_affine_acc<SIGMA_ONLY, 1>(aff, qoff, tp[1], sg, qn, qd);
    NFE_WS_STEP(8) NFE_WS_STEP(9) NFE_WS_STEP(10) NFE_WS_STEP(11)
    plane_affine_acc<SIGMA_ONLY, 2>(aff, qoff, tp[2], sg, qn, qd);
#undef NFE_WS_STEP
#undef NFE_WS_SLOT
}

// quad layout (channels 16h+4c.. of the quad's four points) -> the pair's tile, exchange_to_own()'s write half
__device__ __forceinline__ void ws_write_tile(float* __restrict__ xp, int lane, const f32x2 (&fq)[8]) {
    lane = launder(lane);
    const int Q = lane >> 2, c = lane & 3, h = lane >> 5;
    const int row0 = 4 * (Q & 7);
    float* wr = xp + row0 * 32 + 4 * ((4 * h + c) ^ xchg_swz(row0));
    float* wr2 = xp + row0 * 32 + 4 * ((4 * h + c) ^ xchg_swz(row0 + 2));
    *reinterpret_cast<float4*>(wr) = make_float4(fq[0][0], fq[0][1], fq[1][0], fq[1][1]);
    *reinterpret_cast<float4*>(wr + 32) = make_float4(fq[2][0], fq[2][1], fq[3][0], fq[3][1]);
    *reinterpret_cast<float4*>(wr2 + 64) = make_float4(fq[4][0], fq[4][1], fq[5][0], fq[5][1]);
    *reinterpret_cast<float4*>(wr2 + 96) = make_float4(fq[6][0], fq[6][1], fq[7][0], fq[7][1]);
}
// ... and its read half: channels 16h..16h+15 of this lane's own point
__device__ __forceinline__ void ws_read_tile(const float* __restrict__ xp, int lane, f32x2 (&fo)[8]) {
    const int h = lane >> 5, j = lane & 31;
    const int rd = j * 32 + 4 * ((4 * h) ^ xchg_swz(j));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(xp + (rd ^ (4 * q)));
        fo[2 * q + 0] = f32x2{v.x, v.y};
        fo[2 * q + 1] = f32x2{v.z, v.w};
    }
}

// DUAL: two plane sets (the producer runs the fused kernel's dual gather); SIGMA_ONLY: the coarse pass of a two-pass render (geometry
// set and geometry head only; the consumer writes the compositing weights, the producer the depths).
template <int NP, int WPS, bool SQUARE, bool GENERIC, bool DUAL = false, bool SIGMA_ONLY = false>
__global__ __launch_bounds__(NP * 128, WPS) void render_ws_kernel(RenderK P) {
    static_assert(!(DUAL && SIGMA_ONLY), "a sigma-only pass reads one plane set");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < DEC_B_G0 / 4; i += NP * 128)
        reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(P.dec + DEC_BF16)[i];
    for (int i = threadIdx.x; i < (DEC_FLOATS - DEC_B_G0) / 4; i += NP * 128)
        reinterpret_cast<float4*>(lds + DEC_B_G0)[i] = reinterpret_cast<const float4*>(P.dec + DEC_B_G0)[i];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave < NP;
    const int pair = producer ? wave : wave - NP;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    float* aff = lds + DEC_FLOATS + pair * WS_PAIR_FLOATS;
    float* tile_g0 = aff + AFF_FLOATS;
    unsigned* flags = reinterpret_cast<unsigned*>(tile_g0 + WS_FLAG_OFF);
    if (producer && lane < 4) flags[lane] = 0u;
    __syncthreads();
    const bool probe = P.clock_probe != nullptr && blockIdx.x == 0 && wave == 0;
    if (probe && lane == 0) { P.clock_probe[0] = __builtin_amdgcn_s_memtime(); P.clock_probe[1] = __builtin_amdgcn_s_memrealtime(); }

    const int S = P.S;
    const int blocks_per_view = (P.M + 31) >> 5;
    const long long total_rb = (long long)P.N * blocks_per_view;
    const long long n_pairs = (long long)gridDim.x * NP;
    unsigned step = 0;                       // samples handed over so far by this pair (both roles count alike)
    bool alive = true;

    // (round 6: s_setprio 1..3 on the producer wave of a pair: 14.52 M cycles per launch against 14.48 M, inside the noise; on the consumer
    // wave: 15.25 M, 5 % slower - profiles/experiments/r06_render_prio.md.  No priorities are set.)
    // per-role register census (round 4, compile-only builds of one role each): consumer 152 VGPRs, producer 169-203
    if (producer) {
        const unsigned long long seed = P.seed_dev ? *P.seed_dev : P.seed;
        int cur_view = -1;
        float tmin = INFINITY, tmax = -INFINITY;
#pragma unroll 1
        for (long long rb = (long long)blockIdx.x * NP + pair; rb < total_rb; rb += n_pairs) {
            const int n = (int)(rb / blocks_per_view), b = (int)(rb % blocks_per_view);
            if (n != cur_view) {
                // the consumer never reads the affines, but the previous block's tiles must have left before anything of this
                // pair's LDS region is rewritten by other lanes than their owners: the affines are private to this wave - no wait
                stage_affine(P.aff, n, aff, lane); cur_view = n;
            }
            int m, px = 0, py = 0;
            if (P.tiled) {
                const int tiles_x = P.R >> 3;
                px = (b % tiles_x) * 8 + (j & 7);
                py = (b / tiles_x) * 4 + (j >> 3);
                m = py * P.R + px;
            } else {
                m = b * 32 + j;
                if (P.R > 0) { py = min(m, P.M - 1) / P.R; px = min(m, P.M - 1) % P.R; }
            }
            const bool valid = m < P.M;
            m = min(m, P.M - 1);
            const long long ray = (long long)n * P.M + m;
            float ox, oy, oz, dx, dy, dz;
            if (P.origins) {
                const float* o = P.origins + ray * 3; const float* d = P.dirs + ray * 3;
                ox = o[0]; oy = o[1]; oz = o[2]; dx = d[0]; dy = d[1]; dz = d[2];
            } else {        // RaySampler.forward, ray_sampler.py:35-61 (as render_kernel)
                const float* c = P.cam2world + n * 16; const float* K = P.intrinsics + n * 9;
                const float fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
                const float inv = 1.0f / (float)P.R;
                const float xc = (float)px * inv + 0.5f * inv, yc = (float)py * inv + 0.5f * inv;
                const float xl = (xc - cx + cy * sk / fy - sk * yc / fy) / fx;
                const float yl = (yc - cy) / fy;
                ox = c[3]; oy = c[7]; oz = c[11];
                float wx = c[0] * xl + c[1] * yl + c[2] + c[3];
                float wy = c[4] * xl + c[5] * yl + c[6] + c[7];
                float wz = c[8] * xl + c[9] * yl + c[10] + c[11];
                dx = wx - ox; dy = wy - oy; dz = wz - oz;
                float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
                dx /= nrm; dy /= nrm; dz /= nrm;
            }
            const float* pg = P.planes_g + (long long)n * P.plane_view_stride;
            const float* pa = P.planes_a + (long long)n * P.plane_view_stride;
            // GENERIC = false (the common launch): stratified depths between scalar limits, jitter from Philox or a buffer - the
            // schedule's constants are wave-uniform and stay in SGPRs; GENERIC = true adds per-ray limits, disparity, depth buffers
            float rs = P.ray_start, re = P.ray_end;
            if (GENERIC && P.depth_mode == DEPTH_PER_RAY) { rs = P.rs_ray[ray]; re = P.re_ray[ray]; }
            const float inv_dm1 = 1.0f / (float)(S - 1);
            const float delta = (re - rs) / (float)(S - 1);
            u32x4 rnd = {0, 0, 0, 0};
            const long long plane_elems = (long long)P.H * P.W * 32;
            const unsigned roff = (unsigned)(ray * S);        // the launch guarantees N * M * S < 2^31: 32-bit element offsets
#pragma unroll 1
            for (int k = 0; k < S; ++k, ++step) {
                float t;
                if (GENERIC && P.depth_mode == DEPTH_BUFFER) {
                    t = P.depth_buf[roff + (unsigned)k];
                } else {
                    float u;
                    if (P.u) {
                        u = P.u[roff + (unsigned)k];
                    } else {
                        if ((k & 3) == 0)
                            rnd = philox4x32_10((unsigned)ray, (unsigned)(k >> 2), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32));
                        unsigned bits = (k & 3) == 0 ? rnd.x : (k & 3) == 1 ? rnd.y : (k & 3) == 2 ? rnd.z : rnd.w;
                        u = u01(bits);
                    }
                    if (GENERIC && P.depth_mode == DEPTH_DISPARITY) {
                        float sd = (float)k * inv_dm1 + u * inv_dm1;
                        t = 1.0f / (1.0f / rs * (1.0f - sd) + 1.0f / re * sd);
                    } else if (GENERIC && P.depth_mode == DEPTH_PER_RAY) {
                        t = rs + ((float)k / (float)(S - 1)) * (re - rs) + u * delta;
                    } else {
                        t = fmaf((float)k, delta, rs) + u * delta;
                    }
                }
                if (P.out_depths && valid && h == 0) P.out_depths[roff + (unsigned)k] = t;
                tmin = fminf(tmin, t); tmax = fmaxf(tmax, t);
                const float gx = P.coord_scale * fmaf(t, dx, ox);
                const float gy = P.coord_scale * fmaf(t, dy, oy);
                const float gz = P.coord_scale * fmaf(t, dz, oz);
                f32x2 qn[8], qd[8];
                int opq;                                     // opaque zero: the (loop-invariant) affine reads stay inside the loop
                asm volatile("s_mov_b32 %0, 0" : "=s"(opq));
                const float* affk = aff + opq;
                {
                    const int l0 = launder(lane);
                    const int qoff0 = (l0 >> 5) * 16 + (l0 & 3) * 4;
                    const float4 bn = *reinterpret_cast<const float4*>(affk + AFF_BSUM + qoff0);
                    const float4 bd = *reinterpret_cast<const float4*>(affk + AFF_BSUM + 32 + qoff0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        qn[2 * i] = f32x2{bn.x, bn.y}; qn[2 * i + 1] = f32x2{bn.z, bn.w};
                        qd[2 * i] = f32x2{bd.x, bd.y}; qd[2 * i + 1] = f32x2{bd.z, bd.w};
                    }
                }
                if (DUAL) gather_pipelined_dual<SQUARE>(pg, pa, P.H, P.W, plane_elems, affk, lane, gx, gy, gz, qn, qd);
                else gather_pipelined_ws<SQUARE, SIGMA_ONLY>(pg, P.H, P.W, plane_elems, affk, lane, gx, gy, gz, qn, qd);
                // geometry tile + depths of sample `step` go to set step % WS_BUFS: the consumer must have read that set's previous
                // content, the geometry tile of sample step - WS_BUFS
                float* tile_g = tile_g0 + (int)(step % WS_BUFS) * WS_SET_FLOATS + opq;
                if (alive) alive = ws_wait(flags, 1, 2u * (step - (unsigned)WS_BUFS) + 1u);
                ws_write_tile(tile_g, lane, qn);
                if (h == 0) tile_g[WS_T_OFF + j] = t;
                if (SIGMA_ONLY) {                     // no appearance tile: both halves of the sample's count move at once
                    ws_signal(flags, 0, 2u * step + 2u, lane);
                } else {
                    ws_signal(flags, 0, 2u * step + 1u, lane);
                    if (alive) alive = ws_wait(flags, 1, 2u * (step - (unsigned)WS_BUFS) + 2u);
                    ws_write_tile(tile_g + XCHG_FLOATS, lane, qd);
                    ws_signal(flags, 0, 2u * step + 2u, lane);
                }
            }
        }
        if (P.depth_minmax) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                tmin = fminf(tmin, __shfl_xor(tmin, off));
                tmax = fmaxf(tmax, __shfl_xor(tmax, off));
            }
            if (lane == 0 && tmin <= tmax) {
                atomicMin(P.depth_minmax + 0, f2ord(tmin));
                atomicMax(P.depth_minmax + 1, f2ord(tmax));
            }
            if (lane == 0 && !alive) atomicAdd(P.depth_minmax + 2, 1u);
        }
    } else {
        float wsig[32];                          // sigma-only pass: row 0 of the geometry head's second layer (mlp_bf16_sigma)
        if (SIGMA_ONLY) sigma_row_weights(lds, lane, wsig);
#pragma unroll 1
        for (long long rb = (long long)blockIdx.x * NP + pair; rb < total_rb; rb += n_pairs) {
            const int n = (int)(rb / blocks_per_view), b = (int)(rb % blocks_per_view);
            int m;
            if (P.tiled) {
                const int tiles_x = P.R >> 3;
                m = ((b / tiles_x) * 4 + (j >> 3)) * P.R + (b % tiles_x) * 8 + (j & 7);
            } else {
                m = b * 32 + j;
            }
            const bool valid = m < P.M;
            m = min(m, P.M - 1);
            const long long ray = (long long)n * P.M + m;
            f32x2 acc_rgb[8], acc_seg[4], prev_rgb[8], prev_seg[4];
            float acc_d = 0.0f, acc_w = 0.0f, T = 1.0f, prev_t = 0.0f, prev_sig = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) { acc_rgb[c] = splat(0.0f); prev_rgb[c] = splat(0.0f); }
#pragma unroll
            for (int c = 0; c < 4; ++c) { acc_seg[c] = splat(0.0f); prev_seg[c] = splat(0.0f); }
#pragma unroll 1
            for (int k = 0; k < S; ++k, ++step) {
                int opq;                                     // opaque zero: keeps the LDS weight reads inside the loop (render_kernel)
                asm volatile("s_mov_b32 %0, 0" : "=s"(opq));
                const float* ldsw = lds + opq;
                const float* tile_g = tile_g0 + (int)(step % WS_BUFS) * WS_SET_FLOATS + opq;
                f32x16 og, oa;
                float t;
                {
                    if (alive) alive = ws_wait(flags, 0, 2u * step + 1u);
                    f32x2 fn[8];
                    ws_read_tile(tile_g, lane, fn);
                    t = tile_g[WS_T_OFF + j];
                    ws_signal(flags, 1, 2u * step + (SIGMA_ONLY ? 2u : 1u), lane);
                    if (SIGMA_ONLY) mlp_bf16_sigma(ldsw, fn, lane, wsig, og);
                    else mlp_bf16(ldsw, fn, 0, lane, og);
                }
                if (!SIGMA_ONLY) {
                    if (alive) alive = ws_wait(flags, 0, 2u * step + 2u);
                    f32x2 fd[8];
                    ws_read_tile(tile_g + XCHG_FLOATS, lane, fd);
                    ws_signal(flags, 1, 2u * step + 2u, lane);
                    mlp_bf16(ldsw, fd, 1, lane, oa);
                }
#pragma unroll
                for (int r = 0; r < (SIGMA_ONLY ? 0 : 16); r += 2) {    // sigmoid(x)*(1+2*0.001) - 0.001, triplane.py:269 (decode_features)
                    const f32x2 d = f32x2{exp2_fast(-oa[r]), exp2_fast(-oa[r + 1])} + splat(1.0f);
                    const f32x2 sg = pk_fma(f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])}, splat(1.002f), splat(-0.001f));
                    oa[r] = sg[0]; oa[r + 1] = sg[1];
                }
                {   // SegMipRayMarcher2.run_forward (ray_marcher.py:68-101), as render_kernel
                    const bool first = k == 0;
                    const float dlt = t - (first ? t : prev_t);
                    const float dens = softplus_f((prev_sig + og[0]) * 0.5f - 1.0f);
                    const float alpha = first ? 0.0f : 1.0f - exp2_fast(-(dens * dlt) * LOG2E);
                    const float w = alpha * T;
                    T = T * (1.0f - alpha + 1e-10f);
                    if (P.out_weights && valid && h == 0 && !first) P.out_weights[ray * (S - 1) + (k - 1)] = w;
                    if (!SIGMA_ONLY) {
                        const f32x2 wh = splat(w * 0.5f);
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            acc_rgb[c] = pk_fma(wh, prev_rgb[c] + f32x2{oa[2 * c], oa[2 * c + 1]}, acc_rgb[c]);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc_seg[c] = pk_fma(wh, prev_seg[c] + f32x2{og[2 + 2 * c], og[3 + 2 * c]}, acc_seg[c]);
                        acc_d = fmaf(w, (prev_t + t) * 0.5f, acc_d);
                        acc_w += w;
                    }
                }
                prev_t = t; prev_sig = og[0];
                if (!SIGMA_ONLY) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) prev_rgb[c] = f32x2{oa[2 * c], oa[2 * c + 1]};
#pragma unroll
                    for (int c = 0; c < 4; ++c) prev_seg[c] = f32x2{og[2 + 2 * c], og[3 + 2 * c]};
                }
            }
            if (valid && !SIGMA_ONLY) {
                const float wb = P.white_back ? (1.0f - acc_w) : 0.0f;
                float rgbv[16], segv[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    rgbv[2 * c] = (acc_rgb[c][0] + wb) * 2.0f - 1.0f;
                    rgbv[2 * c + 1] = (acc_rgb[c][1] + wb) * 2.0f - 1.0f;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) { segv[2 * c] = acc_seg[c][0]; segv[2 * c + 1] = acc_seg[c][1]; }
                const int nsg = h ? 7 : 8;
                if (P.channels_first) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) P.rgb[((long long)n * 32 + 16 * h + c) * P.M + m] = rgbv[c];
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c < nsg) P.seg[((long long)n * 15 + 8 * h + c) * P.M + m] = segv[c];
                } else {
                    float4* o = reinterpret_cast<float4*>(P.rgb + ray * 32 + 16 * h);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[q] = make_float4(rgbv[4 * q], rgbv[4 * q + 1], rgbv[4 * q + 2], rgbv[4 * q + 3]);
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c < nsg) P.seg[ray * 15 + 8 * h + c] = segv[c];
                }
                if (h == 0) {
                    P.depth[ray] = acc_d / acc_w;
                    P.wsum[ray] = acc_w;
                }
            }
        }
        if (P.depth_minmax && lane == 0 && !alive) atomicAdd(P.depth_minmax + 2, 1u);
    }
    if (probe && lane == 0) { P.clock_probe[2] = __builtin_amdgcn_s_memtime(); P.clock_probe[3] = __builtin_amdgcn_s_memrealtime(); }
}


// Composite the depth segments of a split launch in order (see PARTIAL_FLOATS) and finish exactly as the kernel's own
// epilogue does; the coarse pass of a two-pass render (sigma_only) only needs its weights rescaled by the transmittance
// in front of their segment.  One lane per (ray, output channel group): lanes 0..31 of a 64-lane group take rgb channel,
// 32..46 seg channel, 47 depth + weight sum + the weights.
__global__ __launch_bounds__(256) void render_combine_kernel(RenderK P, int sigma_only) {
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = threadIdx.x & 63;
    if (ray >= (long long)P.N * P.M || c > 47) return;
    const int nseg = P.seg_count, S = P.S;
    const float* pp = P.partials + ray * nseg * PARTIAL_FLOATS;
    const int n = (int)(ray / P.M), m = (int)(ray % P.M);
    float T = 1.0f, acc = 0.0f, acc_d = 0.0f, acc_w = 0.0f;
    for (int s = 0; s < nseg; ++s) {
        const float* q = pp + s * PARTIAL_FLOATS;
        if (P.out_weights && T != 1.0f) {            // weights of this segment: all 48 lanes of the ray share the rescale
            const int k0 = (int)((long long)s * S / nseg), k1 = (int)((long long)(s + 1) * S / nseg);
            for (int k = max(k0, 1) + c; k < k1; k += 48) P.out_weights[ray * (S - 1) + (k - 1)] *= T;
        }
        if (c == 47) acc_d = fmaf(T, q[47], acc_d);
        else if (!sigma_only) acc = fmaf(T, q[c], acc);
        acc_w = fmaf(T, q[48], acc_w);
        T *= q[49];
    }
    if (sigma_only) return;
    if (c < 32) {
        const float wb = P.white_back ? (1.0f - acc_w) : 0.0f;
        const float v = (acc + wb) * 2.0f - 1.0f;
        if (P.channels_first) P.rgb[((long long)n * 32 + c) * P.M + m] = v; else P.rgb[ray * 32 + c] = v;
    } else if (c < 47) {
        if (P.channels_first) P.seg[((long long)n * 15 + (c - 32)) * P.M + m] = acc; else P.seg[ray * 15 + (c - 32)] = acc;
    } else {
        P.depth[ray] = acc_d / acc_w;
        P.wsum[ray] = acc_w;
    }
}

// nan_to_num(depth, inf) then clamp to the whole-tensor [min,max] of sampled depths (ray_marcher.py:93-94).
// This kernel closes every nfe_render call, so it is also where a lost wave hand-off of render_ws_kernel becomes an error instead
// of silent garbage: with minmax[2] + minmax[6] != 0 EVERY output the call was given - the four images and the optional taps - is
// overwritten with NaN (an aborted coarse pass feeds garbage weights to importance_kernel, whose merged depths would otherwise go
// on to nfe_render_backward as plausible constants) and (count, 1 call) goes to the process's sticky status word with one
// system-scope 64-bit atomic.  The count itself stays in the workspace for nfe_render_call_status.
struct ClampTaps { float* p[4]; unsigned long long n[4]; };      // tap_depths_all, tap_weights_coarse, tap_depths_fine, tap_sample_colors
__global__ void depth_clamp_kernel(float* depth, float* rgb, float* seg, float* wsum, long long n, const unsigned* minmax, ClampTaps taps,
                                   unsigned long long* host_status) {
    const float lo = ord2f(minmax[0]), hi = ord2f(minmax[1]);
    const unsigned lost = minmax[2] + minmax[6];       // final pass + coarse pass (whose min / max words 4, 5 are scratch)
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    if (lost != 0u) {
        const float bad = __uint_as_float(0x7fc00000u);
        for (long long i = i0; i < n; i += stride) { depth[i] = bad; wsum[i] = bad; }
        for (long long i = i0; i < n * NFE_RGB_CHANNELS; i += stride) rgb[i] = bad;
        for (long long i = i0; i < n * NFE_SEG_CHANNELS; i += stride) seg[i] = bad;
        for (int t = 0; t < 4; ++t)
            if (taps.p[t])
                for (unsigned long long i = (unsigned long long)i0; i < taps.n[t]; i += (unsigned long long)stride) taps.p[t][i] = bad;
        if (i0 == 0 && host_status)
            __hip_atomic_fetch_add(host_status, (1ull << 32) | (unsigned long long)lost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    for (long long i = i0; i < n; i += stride) {
        float d = depth[i];
        if (d != d) d = INFINITY;
        depth[i] = fminf(fmaxf(d, lo), hi);
    }
}

__global__ void minmax_init_kernel(unsigned* minmax) {
    minmax[0] = 0xFFFFFFFFu; minmax[1] = 0u; minmax[2] = 0u;      // [2]: hand-off waits abandoned by render_ws_kernel (must stay 0)
    minmax[4] = 0xFFFFFFFFu; minmax[5] = 0u; minmax[6] = 0u;      // the same three words for the coarse pass of a two-pass call: its min / max are not used, its count is
}

// ------------------------------------------------------------------------------------------
// Importance sampling + merge: one wave per ray, lanes over samples.
// sample_importance / sample_pdf (renderer.py:194-253) and unify_samples (:288-300).
// ------------------------------------------------------------------------------------------
struct ImportanceK {
    const float* t_coarse;   // [NR, D]
    const float* w_coarse;   // [NR, D-1]
    const float* u_fine;     // [NR, Di] or null
    unsigned long long seed; const unsigned long long* seed_dev;
    long long n_rays_total;
    int D, Di;
    float* t_all;            // [NR, D+Di] sorted ascending
    int* src_all;            // optional [NR, D+Di]: coarse index k, or D + ascending rank of the fine sample
    float* tap_fine;         // optional [NR, Di] in draw order
};

// Orders this wave's LDS accesses for the compiler (other lanes of the wave read what a lane wrote).  The hardware executes a
// wave's LDS operations in order; a workgroup-scope fence would also drain the outstanding global stores of the previous ray.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}

// Ascending bitonic sort of 64 * R order-preserving depth keys held R per lane (element i = lane + 64 r), round 4.  Only the sorted
// depths leave the kernel, so the keys carry no index and ties need no order.  Stage (k, j): element i keeps the smaller of (itself,
// element i ^ j) iff bits k and j of i are equal.  j >= 64 pairs two registers of one lane; j < 64 is a lane exchange (ds_bpermute
// through __shfl_xor; the LDS crossbar, no VALU issue slot for the data movement) followed by ONE v_med3_u32 (round 6): the median of
// (own, partner, 0) is the smaller of two unsigned keys, the median of (own, partner, ~0) the larger, so the stage's choice is a third
// operand - the sign-extended bit j of lane ^ (lane >> (log2 k - log2 j)), one v_bfe_i32 per stage on a loop-invariant register -
// instead of a min, a max and a select per key (943 -> ~640 vector instructions per ray at 96 + 96 together with count_leading()).
__device__ __forceinline__ unsigned med3_u32(unsigned a, unsigned b, unsigned c) { return max(min(a, b), min(max(a, b), c)); }
template <int R>
__device__ __forceinline__ void bitonic_sort_keys(unsigned (&v)[R], int lane) {
#pragma unroll
    for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j >= 64) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int q = r ^ (j >> 6);
                    if (q > r) {                                   // registers r < q: element index of r is the smaller one
                        const bool asc = ((64 * r) & k) == 0;      // compile-time: (i & k) for k >= 128 depends on r only
                        const unsigned lo = min(v[r], v[q]), hi = max(v[r], v[q]);
                        v[r] = asc ? lo : hi; v[q] = asc ? hi : lo;
                    }
                }
            } else {
                const int lj = __builtin_ctz((unsigned)j);
                // 0 where the element keeps the minimum, ~0 where it keeps the maximum; for k >= 64 bit k of i is a bit of r
                const int x = k < 64 ? lane ^ (lane >> (__builtin_ctz((unsigned)k) - lj)) : lane;
                const unsigned up = (unsigned)((x << (31 - lj)) >> 31);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const unsigned o = (unsigned)__shfl_xor((int)v[r], j);
                    const bool flip = k >= 64 && ((64 * r) & k) != 0;              // compile-time
                    v[r] = med3_u32(v[r], o, flip ? ~up : up);
                }
            }
        }
    }
}

// Number of leading entries of the ascending array a[0 .. 64 R) that are <= v (UPPER: searchsorted right = True) or < v.  The arrays'
// tails beyond their live entries hold +inf (written once per wave), so the search needs no bounds: log2(64 R) fixed steps - a read, a
// compare and a conditional add each - and one more for an array that is live to its last entry; the divergent `while (lo < hi)` loops
// of rounds 1-5 spent nine instructions per step on the same result.
template <int R, bool UPPER>
__device__ __forceinline__ int count_leading(const float* __restrict__ a, float v) {
    int pos = 0;
#pragma unroll
    for (int step = 32 * R; step >= 1; step >>= 1) {
        const float x = a[pos + step - 1];
        pos += (UPPER ? x <= v : x < v) ? step : 0;
    }
    const float x = a[pos];
    return pos + ((UPPER ? x <= v : x < v) ? 1 : 0);
}

// R = number of 64-sample chunks that hold D and Di (1, 2 or 4: <= 64, <= 128, <= 256 samples); WITH_SRC: the draw index of every
// merged sample is wanted too (density noise only): that path keeps the (depth, index) rank counting of rounds 1-3.
template <int R, bool WITH_SRC>
__global__ __launch_bounds__(256) void importance_kernel(ImportanceK P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int D = P.D, Di = P.Di;
    const int DiP = (Di + 2) & ~1;                       // key pairs are read two at a time; at least one +inf key pads the list
    constexpr int CAP = 64 * R;                          // capacity of the searched arrays (count_leading)
    const int stride = 4 * CAP + 2 * DiP;                // per-wave LDS floats (same formula at the launch); a multiple of 4
    float* tc = lds + wave * stride;                     // [CAP] coarse depths, +inf beyond D
    float* wq = tc + CAP;                                // [CAP] weights, then smoothed weights
    float* cdf = wq + CAP;                               // [CAP] cdf knots (D-2 used), +inf beyond
    uint2* keys = reinterpret_cast<uint2*>(cdf + CAP);   // [DiP] (draw index, order-preserving depth bits), 16-byte aligned
    float* tf = reinterpret_cast<float*>(keys + DiP);    // [CAP] fine depths, ascending, +inf beyond Di
    const int B = D - 3;                                 // number of pdf bins (weights[:,1:-1])
    for (int i = D + lane; i < CAP; i += 64) tc[i] = INFINITY;             // the tails stay: every ray rewrites the live entries only
    for (int i = D - 2 + lane; i < CAP; i += 64) cdf[i] = INFINITY;
    for (int i = Di + lane; i < CAP; i += 64) tf[i] = INFINITY;

    for (long long ray = (long long)blockIdx.x * 4 + wave; ray < P.n_rays_total; ray += (long long)gridDim.x * 4) {
        {   // D <= 64 R: R guarded loads per array (a loop over a run-time D was unrolled eight-fold by the compiler, each copy with its own 64-bit address)
            const float* __restrict__ tsrc = P.t_coarse + ray * D;
            const float* __restrict__ wsrc = P.w_coarse + ray * (D - 1);
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const int i = c * 64 + lane;
                if (i < D) tc[i] = tsrc[i];
                if (i < D - 1) wq[i] = wsrc[i];
            }
        }
        wave_lds_fence();
        // smoothed weights a_i, i=0..D-2 (max_pool1d(k2,s1,p1) then avg_pool1d(k2,s1), +0.01): :205-207
        // only a[1..D-3] are used; q_i = a_{i+1} + 1e-5, i = 0..B-1 (:210, :228)
        float qv[R];
        float part = 0.0f;
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int i = c * 64 + lane;
            float q = 0.0f;
            if (i < B) {
                const int a = i + 1;                     // a in [1, D-3]
                const float w_m1 = wq[a - 1], w_0 = wq[a], w_p1 = wq[a + 1];
                const float m0 = fmaxf(w_m1, w_0), m1 = fmaxf(w_0, w_p1);
                q = ((m0 + m1) * 0.5f + 0.01f) + 1e-5f;
            }
            qv[c] = q;
            part += q;
        }
        float total = part;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
        // cdf knots: cdf[0] = 0, cdf[i+1] = cumsum(pdf)[i]  (:229-232)
        float carry = 0.0f;
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int i = c * 64 + lane;
            float pdf = qv[c] / total;
            float sc = wave_incl_scan(pdf, lane) + carry;
            if (i < B) cdf[i + 1] = sc;
            carry = __shfl(sc, 63);
        }
        if (lane == 0) cdf[0] = 0.0f;
        wave_lds_fence();
        // inverse-CDF sampling (:236-252)
        unsigned fkey[R];                                  // this lane's fine depths as ordered keys (element lane + 64 c), +inf padding
#pragma unroll
        for (int c = 0; c < R; ++c) fkey[c] = 0xFFFFFFFFu;
        u32x4 rb = {0, 0, 0, 0};
        if (!P.u_fine) {
            const unsigned long long seed = P.seed_dev ? *P.seed_dev : P.seed;
            rb = philox4x32_10((unsigned)ray, (unsigned)lane, 1u, 0u, (unsigned)seed, (unsigned)(seed >> 32));
        }
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int e = c * 64 + lane;
            const bool live = e < Di;            // no early `continue`: the lane exchanges below involve every lane
            float u;
            if (P.u_fine) {
                u = live ? P.u_fine[ray * Di + e] : 0.0f;
            } else {
                // Philox block b yields the draws of samples 4b .. 4b + 3: lane b computed it once (rb, below the loop header), sample e
                // fetches word e & 3 of lane e >> 2 (same stream as one Philox call per sample, a quarter of the work)
                const int src = 16 * c + (lane >> 2);
                const unsigned bx = (unsigned)__shfl((int)rb.x, src), by = (unsigned)__shfl((int)rb.y, src);
                const unsigned bz = (unsigned)__shfl((int)rb.z, src), bw = (unsigned)__shfl((int)rb.w, src);
                const unsigned bits = (e & 3) == 0 ? bx : (e & 3) == 1 ? by : (e & 3) == 2 ? bz : bw;
                u = u01(bits);
            }
            // searchsorted(cdf[0..B], u, right=True): number of knots <= u
            const int lo = min(count_leading<R, true>(cdf, u), B + 1);
            const int below = max(lo - 1, 0), above = min(lo, B);
            const float cb = cdf[below], ca = cdf[above];
            const float bb = 0.5f * (tc[below] + tc[below + 1]);     // z_vals_mid (:209)
            const float ba = 0.5f * (tc[above] + tc[above + 1]);
            float den = ca - cb;
            if (den < 1e-5f) den = 1.0f;
            const float t = bb + (u - cb) / den * (ba - bb);
            if (live) fkey[c] = f2ord(t);
            if (WITH_SRC && live) keys[e] = make_uint2((unsigned)e, f2ord(t));
            if (P.tap_fine && live) P.tap_fine[ray * Di + e] = t;
        }
        if (WITH_SRC) {
            for (int e = Di + lane; e < DiP; e += 64) keys[e] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            wave_lds_fence();
        }
        // merge (unify_samples, renderer.py:288-300: only the sorted depths leave this kernel, so ties need no order).
        // 1. rank of each fine depth among the fine depths by counting: the number of (depth, draw index) keys below its own - every
        //    lane walks the whole key list (two keys per LDS read, broadcast) with one 64-bit compare and one add-with-carry per key;
        //    no sorting network, no intermediate barriers.  The ranked depths go to tf[], which is therefore ascending.
        // 2. position of each fine depth = its rank + #coarse <= it (binary search: coarse depths are ascending),
        //    position of each coarse depth = its index + #fine < it (binary search in tf[]).
        if (!WITH_SRC) {            // round 4: sort the keys in registers (28 compare-exchange stages for 128 keys against 98 x 2 compares per lane)
            bitonic_sort_keys<R>(fkey, lane);
#pragma unroll
            for (int c = 0; c < R; ++c)
                if (c * 64 + lane < Di) tf[c * 64 + lane] = ord2f(fkey[c]);
        } else {
            unsigned long long mine[R];
            unsigned rank[R];
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const int e = c * 64 + lane;
                const uint2 k = keys[min(e, DiP - 1)];
                mine[c] = (unsigned long long)k.y << 32 | k.x;
                rank[c] = 0;
            }
            const uint4* kp = reinterpret_cast<const uint4*>(keys);
            for (int j = 0; j < DiP / 2; ++j) {
                const uint4 q = kp[j];
                const unsigned long long k0 = (unsigned long long)q.y << 32 | q.x, k1 = (unsigned long long)q.w << 32 | q.z;
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    if (c * 64 >= Di) break;
                    rank[c] += (k0 < mine[c]) ? 1u : 0u;
                    rank[c] += (k1 < mine[c]) ? 1u : 0u;
                }
            }
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const int e = c * 64 + lane;
                if (e < Di) tf[rank[c]] = ord2f((unsigned)(mine[c] >> 32));
            }
        }
        wave_lds_fence();
        float* out = P.t_all + ray * (D + Di);
        // fine: #coarse <= v (upper bound); coarse: #fine < v (lower bound).  Every lane searches (entries beyond D / Di are the +inf
        // tails: their counts are clamped and never stored), so the 2 R searches are straight-line code the scheduler can interleave
        float vf[R], vc[R];
        int pf[R], pc[R];
#pragma unroll
        for (int c = 0; c < R; ++c) { vf[c] = tf[c * 64 + lane]; vc[c] = tc[c * 64 + lane]; }
#pragma unroll
        for (int c = 0; c < R; ++c) {
            pf[c] = min(count_leading<R, true>(tc, vf[c]), D);
            pc[c] = min(count_leading<R, false>(tf, vc[c]), Di);
        }
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int e = c * 64 + lane;
            if (e < Di) {
                out[e + pf[c]] = vf[c];
                if (P.src_all) P.src_all[ray * (D + Di) + e + pf[c]] = D + e;
            }
            if (e < D) {
                out[e + pc[c]] = vc[c];
                if (P.src_all) P.src_all[ray * (D + Di) + e + pc[c]] = e;
            }
        }
        wave_lds_fence();
    }
}

// ------------------------------------------------------------------------------------------
// a15: point query (renderer.run_model), 32 points per wave step
// ------------------------------------------------------------------------------------------
struct PointK {
    const float* planes_g; const float* planes_a; long long plane_view_stride; int H, W;
    const float* aff[4]; const float* dec;
    const float* coords; int N, Pn; float coord_scale;
    float* rgb; float* sigma; float* seg;
    float density_noise; unsigned long long seed;      // renderer.py:285-286 on points: Philox key (seed; point n*P+m, draw 0)
    const float* dec_cross;
};

template <bool DUAL, int MATH, bool CROSS = false>
__global__ __launch_bounds__(256, 2) void point_kernel(PointK P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_decoder<MATH>(P.dec, lds);
    if (CROSS) stage_cross(P.dec_cross, lds);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    float* aff = lds + LDS_AFF + wave * WAVE_LDS_FLOATS;
    float* xp = aff + AFF_FLOATS;
    __syncthreads();
    const int blocks_per_view = (P.Pn + 31) >> 5;
    const long long total = (long long)P.N * blocks_per_view;
    int cur_view = -1;
    for (long long rb = (long long)blockIdx.x * 4 + wave; rb < total; rb += (long long)gridDim.x * 4) {
        const int n = (int)(rb / blocks_per_view), b = (int)(rb % blocks_per_view);
        if (n != cur_view) { stage_affine(P.aff, n, aff, lane); cur_view = n; }
        int m = b * 32 + j;
        const bool valid = m < P.Pn;
        m = min(m, P.Pn - 1);
        const long long pt = (long long)n * P.Pn + m;
        const float* c = P.coords + pt * 3;
        f32x16 og, oa;
        const float no_wsig[32] = {};                 // (sigma-row weights: only the sigma-only render pass has them)
        eval_point<DUAL, false, MATH, CROSS>(P.planes_g + (long long)n * P.plane_view_stride,
                                P.planes_a + (long long)n * P.plane_view_stride, P.H, P.W, lds, aff, xp,
                                P.coord_scale * c[0], P.coord_scale * c[1], P.coord_scale * c[2], lane, og, oa, no_wsig);
        if (valid) {
            float4* o = reinterpret_cast<float4*>(P.rgb + pt * 32 + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = make_float4(oa[4 * q], oa[4 * q + 1], oa[4 * q + 2], oa[4 * q + 3]);
            const int nseg = h ? 7 : 8;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc)
                if (cc < nseg) P.seg[pt * 15 + 8 * h + cc] = og[2 + cc];
            if (h == 0) P.sigma[pt] = P.density_noise > 0.0f ? fmaf(P.density_noise, sample_gaussian(P.seed, (unsigned)pt, 0u), og[0]) : og[0];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Decoder modules on caller-supplied sampled features (DisentangledOSGDecoder.forward, triplane.py:249-270;
// OSGDecoder.forward :178-190; SegmentationOSGDecoder.forward :209-230): mean over the plane axis, then the two heads.
// 32 points per wave step; lane (j, h) reads channels 16h..16h+15 of point j from every plane (64 contiguous bytes).
// ------------------------------------------------------------------------------------------
struct DecoderK {
    const float* feat_g; const float* feat_a;       // [N, n_planes, Pn, 32]
    int N, n_planes; long long Pn;
    const float* dec; const float* dec_cross;
    float* rgb; float* sigma; float* seg;
};

template <int MATH, bool CROSS = false>
__global__ __launch_bounds__(256, 2) void decoder_kernel(DecoderK P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stage_decoder<MATH>(P.dec, lds);
    if (CROSS) stage_cross(P.dec_cross, lds);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    __syncthreads();
    const long long blocks_per_view = (P.Pn + 31) >> 5;
    const long long total = (long long)P.N * blocks_per_view;
    const float inv_planes = 1.0f / (float)P.n_planes;
    for (long long rb = (long long)blockIdx.x * 4 + wave; rb < total; rb += (long long)gridDim.x * 4) {
        const int n = (int)(rb / blocks_per_view);
        long long m = (rb % blocks_per_view) * 32 + j;
        const bool valid = m < P.Pn;
        m = m < P.Pn ? m : P.Pn - 1;
        f32x2 fn[8], fd[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { fn[c] = splat(0.0f); fd[c] = splat(0.0f); }
        for (int p = 0; p < P.n_planes; ++p) {
            const long long off = (((long long)n * P.n_planes + p) * P.Pn + m) * 32 + 16 * h;
            const float4* g = reinterpret_cast<const float4*>(P.feat_g + off);
            const float4* a = reinterpret_cast<const float4*>(P.feat_a + off);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 x = g[q], y = a[q];
                fn[2 * q] += f32x2{x.x, x.y}; fn[2 * q + 1] += f32x2{x.z, x.w};
                fd[2 * q] += f32x2{y.x, y.y}; fd[2 * q + 1] += f32x2{y.z, y.w};
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) { fn[c] = fn[c] * splat(inv_planes); fd[c] = fd[c] * splat(inv_planes); }
        f32x16 og, oa;
        const float no_wsig[32] = {};
        decode_features<false, MATH, CROSS>(lds, fn, fd, lane, og, oa, no_wsig);
        if (valid) {
            const long long pt = (long long)n * P.Pn + m;
            float4* o = reinterpret_cast<float4*>(P.rgb + pt * 32 + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = make_float4(oa[4 * q], oa[4 * q + 1], oa[4 * q + 2], oa[4 * q + 3]);
            const int nseg = h ? 7 : 8;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc)
                if (cc < nseg) P.seg[pt * 15 + 8 * h + cc] = og[2 + cc];
            if (h == 0) P.sigma[pt] = og[0];
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <bool DUAL, bool SIGMA_ONLY>
static void launch_render_math(const RenderK& P, int math, dim3 grid, hipStream_t st) {
    const bool noise = P.density_noise > 0.0f;
    if (math == NFE_MATH_FP32) {
        if (noise) hipLaunchKernelGGL((render_kernel<DUAL, SIGMA_ONLY, NFE_MATH_FP32, true>), grid, dim3(256), RENDER_LDS_BYTES, st, P);
        else hipLaunchKernelGGL((render_kernel<DUAL, SIGMA_ONLY, NFE_MATH_FP32>), grid, dim3(256), RENDER_LDS_BYTES, st, P);
    } else {
        if (noise && P.tap_colors && !SIGMA_ONLY)          // density_noise with kept colours (round 6): the stored sigma carries its noise, the backward needs no draw
            hipLaunchKernelGGL((render_kernel<DUAL, false, NFE_MATH_BF16X3, true, false, false, false, false, true>), grid, dim3(256), RENDER_LDS_BYTES, st, P);
        else if (noise) hipLaunchKernelGGL((render_kernel<DUAL, SIGMA_ONLY, NFE_MATH_BF16X3, true>), grid, dim3(256), RENDER_LDS_BYTES, st, P);
        else if (P.tap_colors && !SIGMA_ONLY)
            hipLaunchKernelGGL((render_kernel<DUAL, false, NFE_MATH_BF16X3, false, false, false, false, false, true>), grid, dim3(256), RENDER_LDS_BYTES, st, P);
        else if (P.H == P.W)               // shared axis geometry + the in-bounds gather path (round 4: pays with two plane sets too)
            hipLaunchKernelGGL((render_kernel<DUAL, SIGMA_ONLY, NFE_MATH_BF16X3, false, false, false, true>), grid, dim3(256), RENDER_LDS_BYTES, st, P);
        else hipLaunchKernelGGL((render_kernel<DUAL, SIGMA_ONLY, NFE_MATH_BF16X3>), grid, dim3(256), RENDER_LDS_BYTES, st, P);
    }
}

// Which render kernels the last nfe_render call of this thread launched, in order (nfe_render_last_kernels: tests assert that a
// fixture reaches the kernel variant it is meant to pin).
static thread_local char g_kernels[256] = "";
static void note_kernel(const char* name) {
    const size_t have = strlen(g_kernels), add = strlen(name);
    if (have + add + 2 >= sizeof(g_kernels)) return;
    if (have) { g_kernels[have] = ' '; memcpy(g_kernels + have + 1, name, add + 1); } else memcpy(g_kernels, name, add + 1);
}

// Polls of a hand-off wait before it is abandoned: the device global g_ws_spin_limit.  NFE_WS_SPIN_LIMIT (read once) overrides it for
// tests/test_handoff_abort_gpu.py; the symbol is written once per device, and only when the variable is set.
static int apply_ws_spin_limit() {
    static const int v = [] { const char* e = getenv("NFE_WS_SPIN_LIMIT"); const long long x = e ? atoll(e) : 0; return x > 0 && x < (1ll << 30) ? (int)x : 0; }();
    if (!v) return NFE_OK;
    static std::atomic<unsigned long long> done{0};
    const int dev = current_device();
    if (dev < MAX_DEVICES && (done.load(std::memory_order_acquire) >> dev & 1ull)) return NFE_OK;
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_ws_spin_limit), &v, sizeof(int));
    if (e != hipSuccess) return fail(NFE_ELAUNCH, "NFE_WS_SPIN_LIMIT: hipMemcpyToSymbol: %s", hipGetErrorString(e));
    if (dev < MAX_DEVICES) done.fetch_or(1ull << dev, std::memory_order_release);
    return NFE_OK;
}

// Wave-specialised launch (render_ws_kernel): one plane set, full split-bf16 decoder, no noise / cross / kept colours, and enough
// ray blocks to fill the chip without the depth split.  NFE_RENDER_WS=0 keeps the fused kernel (A/B, tests).
static int ws_mode() {
    static const int v = [] { const char* e = getenv("NFE_RENDER_WS"); return e ? atoi(e) : NFE_RENDER_WS_DEFAULT; }();
    return v;
}
// Fewest 32-ray blocks for which the wave-specialised launch is used (NFE_RENDER_WS_MIN_RB overrides for A/B).  4 096: a batch of four
// 128^2 views (2 048 blocks, the FFHQ configuration) stays on the fused kernel - alone the wave-specialised launch is 3-5 % faster there,
// but its 512-thread workgroups with ~150 KB of LDS own a CU, and inside the three-stream pipeline the dense kernels of the neighbouring
// batches then cannot share it: FFHQ 821-836 -> 838-856 views/s with the fused kernel (fp16 convs 1 380-1 394 -> 1 400-1 405).
static long long ws_min_ray_blocks() {
    static const long long v = [] { const char* e = getenv("NFE_RENDER_WS_MIN_RB"); const long long x = e ? atoll(e) : 0; return x > 0 ? x : 4096ll; }();
    return v;
}
template <int NP, int WPS, bool DUAL = false, bool SIGMA_ONLY = false>
static hipError_t launch_render_ws(const RenderK& P, long long total_rb, hipStream_t st) {
    constexpr int bytes = ws_lds_bytes<NP>();
    constexpr int per_cu = (WPS * 4) / (2 * NP);          // workgroups per CU that make WPS waves per SIMD
    static_assert(per_cu >= 1 && per_cu * bytes <= 160 * 1024, "LDS of the resident workgroups");
    long long blocks = (total_rb + NP - 1) / NP;
    const long long cap = (long long)num_cus() * per_cu;
    if (blocks > cap) blocks = cap;
    const bool generic = P.depth_mode != DEPTH_STRATIFIED;
#define NFE_WS_LAUNCH(SQ, GE)                                                                                                    \
    {                                                                                                                            \
        static LdsOptIn opt_;                                                                                                    \
        const hipError_t e_ = opt_.apply(render_ws_kernel<NP, WPS, SQ, GE, DUAL, SIGMA_ONLY>, bytes);                            \
        if (e_ != hipSuccess) return e_;                                                                                         \
        hipLaunchKernelGGL((render_ws_kernel<NP, WPS, SQ, GE, DUAL, SIGMA_ONLY>), dim3((unsigned)blocks), dim3(NP * 128), bytes, st, P); \
    }
    if (P.H == P.W) { if (generic) NFE_WS_LAUNCH(true, true) else NFE_WS_LAUNCH(true, false) }
    else { if (generic) NFE_WS_LAUNCH(false, true) else NFE_WS_LAUNCH(false, false) }
#undef NFE_WS_LAUNCH
    return hipSuccess;
}

static int launch_render(const RenderK& P, bool dual, bool sigma_only, int math, hipStream_t st) {
    if (sigma_only && !P.dec_cross) dual = false;      // a sigma-only pass reads the geometry set only: the one-set variants (SQUARE, in-bounds path) serve it
    const long long total_rb = (long long)P.N * ((P.M + 31) / 32);
    long long blocks = (total_rb + 3) / 4;
    static const int blocks_per_cu = [] {                // tuning/diagnostic knob; default 2 blocks (8 waves) per CU
        const char* e = getenv("NFE_RENDER_BLOCKS_PER_CU");
        const int v = e ? atoi(e) : 0;
        return (v >= 1 && v <= 8) ? v : 2;
    }();
    const long long cap = (long long)num_cus() * blocks_per_cu;      // grid-stride beyond
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks);
    if (P.dec_cross) {       // validated by nfe_render: one plane set, split-bf16 decoder, no density_noise; the coarse pass of a
                             // two-pass render runs the full decoder too (sigma needs the appearance head's hidden layer)
        static LdsOptIn opt, opt_store;
        const hipError_t e = P.tap_colors ? opt_store.apply(render_kernel<false, false, NFE_MATH_BF16X3, false, true, false, false, false, true>, RENDER_LDS_BYTES_CROSS)
                                          : opt.apply(render_kernel<false, false, NFE_MATH_BF16X3, false, true>, RENDER_LDS_BYTES_CROSS);
        if (e != hipSuccess) return fail(NFE_ELAUNCH, "render_kernel<CROSS>: LDS opt-in: %s", hipGetErrorString(e));
        if (P.tap_colors)     // kept colours for the plane-gradient backward of SegmentationOSGDecoder (round 6)
            hipLaunchKernelGGL((render_kernel<false, false, NFE_MATH_BF16X3, false, true, false, false, false, true>), grid, dim3(256), RENDER_LDS_BYTES_CROSS, st, P);
        else hipLaunchKernelGGL((render_kernel<false, false, NFE_MATH_BF16X3, false, true>), grid, dim3(256), RENDER_LDS_BYTES_CROSS, st, P);
        note_kernel(P.tap_colors ? "render_kernel<CROSS,STORE>" : "render_kernel<CROSS>");
        NFE_CHECK_LAUNCH("render_kernel");
        return NFE_OK;
    }
    // Few ray blocks (e.g. one 128^2 view = 512 blocks on 1024 SIMDs): cut the march of each block into depth segments so
    // that every SIMD gets about two waves; render_combine_kernel composites the segments.
    static const bool allow_split = [] { const char* e = getenv("NFE_RENDER_SPLIT"); return !(e && e[0] == '0'); }();
    int nseg = 1;
    if (allow_split && P.partials && math == NFE_MATH_BF16X3 && P.density_noise == 0.0f && total_rb * 2 <= SPLIT_MAX_ITEMS && P.S >= 16) {
        nseg = (int)(SPLIT_MAX_ITEMS / total_rb);
        if (nseg > 8) nseg = 8;
        if (nseg > P.S / 8) nseg = P.S / 8;
    }
    if (nseg >= 2) {
        RenderK Q = P;
        Q.seg_count = nseg;
        long long sblocks = (total_rb * nseg + 3) / 4;
        if (sblocks > cap) sblocks = cap;
        dim3 sgrid((unsigned)sblocks);
#define NFE_LAUNCH_SPLIT(DU, SG)                                                                                                   \
        if (P.H == P.W && !(DU)) hipLaunchKernelGGL((render_kernel<false, SG, NFE_MATH_BF16X3, false, false, true, true>), sgrid, dim3(256), RENDER_LDS_BYTES, st, Q); \
        else hipLaunchKernelGGL((render_kernel<DU, SG, NFE_MATH_BF16X3, false, false, true>), sgrid, dim3(256), RENDER_LDS_BYTES, st, Q);
        if (sigma_only) {
            if (dual) { NFE_LAUNCH_SPLIT(true, true) } else { NFE_LAUNCH_SPLIT(false, true) }
        } else if (P.tap_colors) {
            if (dual) hipLaunchKernelGGL((render_kernel<true, false, NFE_MATH_BF16X3, false, false, true, false, false, true>), sgrid, dim3(256), RENDER_LDS_BYTES, st, Q);
            else hipLaunchKernelGGL((render_kernel<false, false, NFE_MATH_BF16X3, false, false, true, false, false, true>), sgrid, dim3(256), RENDER_LDS_BYTES, st, Q);
        } else {
            if (dual) { NFE_LAUNCH_SPLIT(true, false) } else { NFE_LAUNCH_SPLIT(false, false) }
        }
#undef NFE_LAUNCH_SPLIT
        note_kernel(sigma_only ? "render_kernel<SPLIT,SIGMA_ONLY>+render_combine_kernel" : dual ? "render_kernel<SPLIT,DUAL>+render_combine_kernel" : "render_kernel<SPLIT>+render_combine_kernel");
        NFE_CHECK_LAUNCH("render_kernel (split)");
        const long long rays = (long long)P.N * P.M;
        hipLaunchKernelGGL(render_combine_kernel, dim3((unsigned)((rays + 3) / 4)), dim3(256), 0, st, Q, sigma_only ? 1 : 0);
        NFE_CHECK_LAUNCH("render_combine_kernel");
        return NFE_OK;
    }
    if (ws_mode() && math == NFE_MATH_BF16X3 && P.density_noise == 0.0f && !P.tap_colors && (sigma_only || !P.out_weights) &&
        total_rb >= ws_min_ray_blocks() && (long long)P.N * P.M * P.S < (1ll << 31)) {
        // four pairs per workgroup at two waves per SIMD; the 3- and 4-waves-per-SIMD geometries of round 4 measured slower
        // (profiles/experiments/r04_render_ws.md) and are no longer in this file
        hipError_t e;
        if (sigma_only) { e = launch_render_ws<4, 2, false, true>(P, total_rb, st); note_kernel("render_ws_kernel<4,2,SIGMA_ONLY>"); }
        else if (dual) { e = launch_render_ws<4, 2, true, false>(P, total_rb, st); note_kernel("render_ws_kernel<4,2,DUAL>"); }
        else { e = launch_render_ws<4, 2>(P, total_rb, st); note_kernel("render_ws_kernel<4,2>"); }
        if (e != hipSuccess) return fail(NFE_ELAUNCH, "render_ws_kernel: LDS opt-in (%d bytes): %s", ws_lds_bytes<4>(), hipGetErrorString(e));
        NFE_CHECK_LAUNCH("render_ws_kernel");
        return NFE_OK;
    }
    if (sigma_only) {
        if (dual) launch_render_math<true, true>(P, math, grid, st); else launch_render_math<false, true>(P, math, grid, st);
    } else {
        if (dual) launch_render_math<true, false>(P, math, grid, st); else launch_render_math<false, false>(P, math, grid, st);
    }
    note_kernel(sigma_only ? "render_kernel<SIGMA_ONLY>" : dual ? "render_kernel<DUAL>" : "render_kernel");
    NFE_CHECK_LAUNCH("render_kernel");
    return NFE_OK;
}

static uint64_t align256(uint64_t x) { return (x + 255) & ~uint64_t(255); }

// First pass of nfe_render_backward on the forward kernel's machinery (quad gather, split-bf16 MFMA decoder): sigma_i and
// a_i = <2 g_rgb, rgb_i> + <g_seg, seg_i> of every sample of the sorted depth buffer.  Samples are independent, so the depth
// segments of a ray block simply go to different waves when the launch is small.
int render_eval_pass(const nfe_render_backward_args* a, const float* decoder_packed, float* rec_sig, float* rec_a, hipStream_t st) {
    RenderK P{};
    P.planes_g = a->planes_geo; P.planes_a = a->planes_app; P.plane_view_stride = a->plane_view_stride;
    P.H = a->plane_h; P.W = a->plane_w;
    P.aff[0] = a->geo_scale; P.aff[1] = a->geo_shift; P.aff[2] = a->app_scale; P.aff[3] = a->app_shift;
    P.dec = decoder_packed;
    P.N = a->n_views; P.M = a->n_rays;
    P.R = (a->resolution > 0 && (long long)a->resolution * a->resolution == a->n_rays) ? a->resolution : 0;
    P.tiled = (P.R > 0 && (P.R % 8) == 0) ? 1 : 0;
    P.origins = a->origins; P.dirs = a->dirs; P.cam2world = a->cam2world; P.intrinsics = a->intrinsics;
    P.S = a->n_samples; P.depth_mode = DEPTH_BUFFER; P.depth_buf = a->depths;
    P.coord_scale = 2.0f / a->box_warp;
    P.ev_g_rgb = a->grad_rgb; P.ev_g_seg = a->grad_seg; P.ev_channels_first = a->channels_first; P.ev_sig = rec_sig; P.ev_a = rec_a;
    const long long total_rb = (long long)P.N * ((P.M + 31) / 32);
    int nseg = (int)(SPLIT_MAX_ITEMS / total_rb);
    if (nseg > 8) nseg = 8;
    if (nseg > P.S / 8) nseg = P.S / 8;
    if (nseg < 1) nseg = 1;
    P.seg_count = nseg;
    long long blocks = (total_rb * nseg + 3) / 4;
    const long long cap = (long long)num_cus() * 2;
    if (blocks > cap) blocks = cap;
    const bool dual = a->planes_geo != a->planes_app;
    if (dual) hipLaunchKernelGGL((render_kernel<true, false, NFE_MATH_BF16X3, false, false, true, false, true>), dim3((unsigned)blocks), dim3(256), RENDER_LDS_BYTES, st, P);
    else hipLaunchKernelGGL((render_kernel<false, false, NFE_MATH_BF16X3, false, false, true, false, true>), dim3((unsigned)blocks), dim3(256), RENDER_LDS_BYTES, st, P);
    NFE_CHECK_LAUNCH("render_kernel (eval)");
    return NFE_OK;
}

// The same two quantities from the colours the forward kept (nfe_render_args.tap_sample_colors): one pass over 192 bytes per sample
// instead of the gathers and both decoder heads.  Thread = (ray block, sample, lane of the block); the sums run in the evaluation
// pass's order (16 colour features and 8 logits per channel half, halves added last).
#define NFE_DOT_K 8
#define NFE_DOT_UNROLL 2
constexpr int DOT_K = NFE_DOT_K;      // samples per thread: the ray's 47 cotangents are loaded once per DOT_K samples
__global__ __launch_bounds__(256) void color_dot_kernel(RenderK P, const float* __restrict__ colors) {
    const int S = P.S, blocks_per_view = (P.M + 31) >> 5, ksegs = (S + DOT_K - 1) / DOT_K;
    const long long total = (long long)P.N * blocks_per_view * ksegs * 32;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int j = (int)(i & 31);
        const long long rs = i >> 5;
        const int ks = (int)(rs % ksegs);
        const long long rb = rs / ksegs;
        const int n = (int)(rb / blocks_per_view), b = (int)(rb % blocks_per_view);
        int m;
        if (P.tiled) { const int tiles_x = P.R >> 3; m = ((b / tiles_x) * 4 + (j >> 3)) * P.R + (b % tiles_x) * 8 + (j & 7); }
        else m = b * 32 + j;
        if (m >= P.M) continue;
        const long long ray = (long long)n * P.M + m;
        const long long ev0 = bwd_slot_base(P.R, P.M, S, n, m);            // the records' tile order (nfe_common.h)
        float gr[32], gs[16];
#pragma unroll
        for (int c = 0; c < 32; ++c)
            gr[c] = P.ev_g_rgb ? 2.0f * (P.ev_channels_first ? P.ev_g_rgb[((long long)n * 32 + c) * P.M + m] : P.ev_g_rgb[ray * 32 + c]) : 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            gs[c] = (P.ev_g_seg && c < 15) ? (P.ev_channels_first ? P.ev_g_seg[((long long)n * 15 + c) * P.M + m] : P.ev_g_seg[ray * 15 + c]) : 0.0f;
        const int k1 = min(S, (ks + 1) * DOT_K);
#pragma unroll NFE_DOT_UNROLL
        for (int k = ks * DOT_K; k < k1; ++k) {
            const float* cb = colors + ((rb * S + k) * 48) * 32 + j;
            float a2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float a = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) a = fmaf(gr[16 * h + r], cb[(16 * h + r) * 32], a);
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) a = fmaf(gs[8 * h + cc], (8 * h + cc < 15) ? cb[(32 + 8 * h + cc) * 32] : 0.0f, a);
                a2[h] = a;
            }
            P.ev_sig[ev0 + 64ll * k] = cb[47 * 32];
            P.ev_a[ev0 + 64ll * k] = a2[0] + a2[1];
        }
    }
}

int render_color_dot_pass(const nfe_render_backward_args* a, float* rec_sig, float* rec_a, hipStream_t st) {
    RenderK P{};
    P.N = a->n_views; P.M = a->n_rays; P.S = a->n_samples;
    P.R = (a->resolution > 0 && (long long)a->resolution * a->resolution == a->n_rays) ? a->resolution : 0;
    P.tiled = (P.R > 0 && (P.R % 8) == 0) ? 1 : 0;
    P.ev_g_rgb = a->grad_rgb; P.ev_g_seg = a->grad_seg; P.ev_channels_first = a->channels_first; P.ev_sig = rec_sig; P.ev_a = rec_a;
    const long long total = (long long)P.N * ((P.M + 31) >> 5) * ((P.S + DOT_K - 1) / DOT_K) * 32;
    long long blocks = (total + 255) / 256;
    if (blocks > (1 << 16)) blocks = 1 << 16;
    hipLaunchKernelGGL(color_dot_kernel, dim3((unsigned)blocks), dim3(256), 0, st, P, a->sample_colors);
    NFE_CHECK_LAUNCH("color_dot_kernel");
    return NFE_OK;
}

}  // namespace nfe

using namespace nfe;

extern "C" uint64_t nfe_render_sample_colors_floats(int n_views, int n_rays, int n_samples) {
    if (n_views <= 0 || n_rays <= 0 || n_samples <= 0) return 0;
    return (uint64_t)n_views * (uint64_t)((n_rays + 31) / 32) * (uint64_t)n_samples * 48ull * 32ull;
}

extern "C" uint64_t nfe_render_workspace_bytes(int n_views, int n_rays, int D, int Di) {
    uint64_t nr = (uint64_t)(n_views > 0 ? n_views : 0) * (uint64_t)(n_rays > 0 ? n_rays : 0);
    uint64_t b = 256 + SPLIT_PARTIAL_BYTES;              // depth min/max words, segment composites of depth-split launches
    if (Di > 0) {
        b += align256(nr * (uint64_t)D * 4);             // coarse depths
        b += align256(nr * (uint64_t)(D - 1) * 4);       // coarse weights
        b += align256(nr * (uint64_t)(D + Di) * 4);      // merged depths
    }
    return b;
}

extern "C" const char* nfe_render_last_kernels(void) { return g_kernels; }

extern "C" int nfe_render(const nfe_render_args* a, nfe_stream_t stream) {
    NFE_REQUIRE(a != nullptr, "nfe_render: args is null");
    NFE_REQUIRE(a->struct_size == sizeof(nfe_render_args), "nfe_render: struct_size %u != %zu (ABI mismatch)",
                a->struct_size, sizeof(nfe_render_args));
    NFE_REQUIRE(a->planes_geo && a->planes_app && a->decoder_packed, "nfe_render: planes/decoder pointers are null");
    NFE_REQUIRE(a->plane_h > 0 && a->plane_w > 0 && (long long)a->plane_h * a->plane_w <= (1ll << 25), "nfe_render: bad plane size %dx%d (texel byte offsets are 32-bit: H*W <= 2^25)", a->plane_h, a->plane_w);
    NFE_REQUIRE(a->n_views > 0 && a->n_rays > 0, "nfe_render: n_views=%d n_rays=%d must be positive", a->n_views, a->n_rays);
    const int D = a->depth_resolution, Di = a->depth_resolution_importance;
    NFE_REQUIRE(D >= 2 && D <= NFE_MAX_SAMPLES, "nfe_render: depth_resolution=%d out of [2,%d]", D, NFE_MAX_SAMPLES);
    NFE_REQUIRE(Di >= 0 && Di <= NFE_MAX_SAMPLES, "nfe_render: depth_resolution_importance=%d out of [0,%d]", Di, NFE_MAX_SAMPLES);
    NFE_REQUIRE(Di == 0 || D >= 4, "nfe_render: importance sampling needs depth_resolution >= 4 (got %d)", D);
    NFE_REQUIRE(Di == 0 || (long long)a->n_views * a->n_rays >= 2,
                "nfe_render: importance sampling with a single ray is undefined in the reference (renderer.py:206 squeeze)");
    NFE_REQUIRE((a->origins != nullptr) == (a->dirs != nullptr), "nfe_render: origins and dirs must both be given or both null");
    if (!a->origins) {
        NFE_REQUIRE(a->cam2world && a->intrinsics, "nfe_render: need origins/dirs or cam2world/intrinsics");
        NFE_REQUIRE(a->resolution > 0 && (long long)a->resolution * a->resolution == a->n_rays,
                    "nfe_render: resolution^2 (%d^2) != n_rays (%d)", a->resolution, a->n_rays);
    }
    NFE_REQUIRE((a->ray_start_per_ray != nullptr) == (a->ray_end_per_ray != nullptr), "nfe_render: per-ray limits must come in pairs");
    NFE_REQUIRE(!(a->ray_start_per_ray && a->disparity_space_sampling), "nfe_render: disparity sampling with per-ray (auto) limits: the reference fails here too ([N,M,1] limits do not broadcast against [N,M,D,1] depths, renderer.py:174-181)");
    NFE_REQUIRE(a->box_warp > 0.0f, "nfe_render: box_warp must be positive");
    NFE_REQUIRE(a->decoder_math == NFE_MATH_BF16X3 || a->decoder_math == NFE_MATH_FP32, "nfe_render: unknown decoder_math %d", a->decoder_math);
    NFE_REQUIRE(a->rgb && a->seg && a->depth && a->wsum, "nfe_render: output pointers are null");
    NFE_REQUIRE(a->density_noise >= 0.0f, "nfe_render: density_noise must be >= 0");
    NFE_REQUIRE(a->workspace != nullptr, "nfe_render: workspace is null");
    uint64_t need = nfe_render_workspace_bytes(a->n_views, a->n_rays, D, Di);
    const uint64_t nr0 = (uint64_t)a->n_views * a->n_rays;
    if (Di > 0 && a->density_noise > 0.0f) need += align256(nr0 * (uint64_t)(D + Di) * 4);   // draw index of every merged sample
    if (a->workspace_bytes < need) return fail(NFE_EWORKSPACE, "nfe_render: workspace %llu < %llu bytes",
                                               (unsigned long long)a->workspace_bytes, (unsigned long long)need);
    hipStream_t st = (hipStream_t)stream;
    g_kernels[0] = 0;
    const uint64_t nr = (uint64_t)a->n_views * a->n_rays;
    char* ws = (char*)a->workspace;
    unsigned* minmax = (unsigned*)ws; ws += 256;
    float* partials = (float*)ws; ws += SPLIT_PARTIAL_BYTES;

    RenderK P{};
    P.planes_g = a->planes_geo; P.planes_a = a->planes_app; P.plane_view_stride = a->plane_view_stride;
    P.H = a->plane_h; P.W = a->plane_w;
    P.aff[0] = a->geo_scale; P.aff[1] = a->geo_shift; P.aff[2] = a->app_scale; P.aff[3] = a->app_shift;
    P.dec = a->decoder_packed;
    P.N = a->n_views; P.M = a->n_rays;
    P.R = (a->resolution > 0 && (long long)a->resolution * a->resolution == a->n_rays) ? a->resolution : 0;
    P.tiled = (P.R > 0 && (P.R % 8) == 0) ? 1 : 0;
    P.origins = a->origins; P.dirs = a->dirs; P.cam2world = a->cam2world; P.intrinsics = a->intrinsics;
    P.ray_start = a->ray_start; P.ray_end = a->ray_end; P.rs_ray = a->ray_start_per_ray; P.re_ray = a->ray_end_per_ray;
    P.coord_scale = 2.0f / a->box_warp;
    P.white_back = a->white_back;
    P.rgb = a->rgb; P.seg = a->seg; P.depth = a->depth; P.wsum = a->wsum; P.channels_first = a->channels_first;
    P.seed = a->seed; P.seed_dev = reinterpret_cast<const unsigned long long*>(a->seed_device);
    P.density_noise = a->density_noise;
    P.noise_buf = a->density_noise > 0.0f ? a->density_noise_values : nullptr; P.noise_stride = D + Di;
    P.dec_cross = a->decoder_cross;
    P.clock_probe = reinterpret_cast<unsigned long long*>(a->clock_probe);
    P.partials = a->decoder_cross ? nullptr : partials;
    if (int rc = apply_ws_spin_limit()) return rc;
    if (a->decoder_cross)
        NFE_REQUIRE(a->planes_geo == a->planes_app && a->decoder_math == NFE_MATH_BF16X3 && a->density_noise == 0.0f,
                    "nfe_render: decoder_cross (SegmentationOSGDecoder) needs one plane set, NFE_MATH_BF16X3 and density_noise == 0");
    const int mode = a->ray_start_per_ray ? DEPTH_PER_RAY : (a->disparity_space_sampling ? DEPTH_DISPARITY : DEPTH_STRATIFIED);
    const bool dual = a->planes_geo != a->planes_app;
    const int math = a->decoder_math;

    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, st, minmax);
    NFE_CHECK_LAUNCH("minmax_init_kernel");

    if (a->tap_sample_colors)
        NFE_REQUIRE(math == NFE_MATH_BF16X3, "nfe_render: tap_sample_colors needs NFE_MATH_BF16X3");
    if (Di == 0) {
        P.S = D; P.depth_mode = mode; P.u = a->u_coarse; P.depth_minmax = minmax;
        P.out_depths = a->tap_depths_all;
        P.tap_colors = a->tap_sample_colors;
        int rc = launch_render(P, dual, false, math, st);
        if (rc) return rc;
    } else {
        float* t_c = (float*)ws; ws += align256(nr * (uint64_t)D * 4);
        float* w_c = (float*)ws; ws += align256(nr * (uint64_t)(D - 1) * 4);
        float* t_all = (float*)ws; ws += align256(nr * (uint64_t)(D + Di) * 4);
        int* src_all = a->density_noise > 0.0f ? (int*)ws : nullptr;      // present only with density_noise (checked above)
        // pass 1: coarse densities -> weights (only sigma is needed: geometry net, geometry planes)
        RenderK C = P;
        C.S = D; C.depth_mode = mode; C.u = a->u_coarse; C.depth_minmax = minmax + 4;      // scratch min / max, and the coarse pass's abort count
        C.out_depths = t_c; C.out_weights = w_c;
        int rc = launch_render(C, dual, true, math, st);
        if (rc) return rc;
        if (a->tap_weights_coarse) {
            hipError_t e = hipMemcpyAsync(a->tap_weights_coarse, w_c, nr * (uint64_t)(D - 1) * 4, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return fail(NFE_ELAUNCH, "tap copy: %s", hipGetErrorString(e));
        }
        // pass 2: importance sampling + merge
        ImportanceK I{};
        I.t_coarse = t_c; I.w_coarse = w_c; I.u_fine = a->u_fine; I.seed = a->seed; I.seed_dev = reinterpret_cast<const unsigned long long*>(a->seed_device); I.n_rays_total = (long long)nr;
        I.D = D; I.Di = Di; I.t_all = t_all; I.src_all = src_all; I.tap_fine = a->tap_depths_fine;
        const int cap = (D > Di ? D : Di) <= 64 ? 64 : ((D > Di ? D : Di) <= 128 ? 128 : 256);
        const int lds_bytes = 4 * (4 * cap + 2 * ((Di + 2) & ~1)) * 4;      // four waves, importance_kernel's layout (CAP = 64 R)
        long long blocks = ((long long)nr + 3) / 4;
        if (blocks > (long long)num_cus() * 8) blocks = (long long)num_cus() * 8;
        {
            const int mx = D > Di ? D : Di;
#define NFE_IMP(RR) { if (src_all) hipLaunchKernelGGL((importance_kernel<RR, true>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, I); \
                      else hipLaunchKernelGGL((importance_kernel<RR, false>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, I); }
            if (mx <= 64) NFE_IMP(1) else if (mx <= 128) NFE_IMP(2) else NFE_IMP(4)
#undef NFE_IMP
        }
        note_kernel("importance_kernel");
        NFE_CHECK_LAUNCH("importance_kernel");
        // pass 3: march the merged samples
        RenderK F = P;
        F.S = D + Di; F.depth_mode = DEPTH_BUFFER; F.depth_buf = t_all; F.src_buf = src_all; F.depth_minmax = minmax;
        F.tap_colors = a->tap_sample_colors;
        rc = launch_render(F, dual, false, math, st);
        if (rc) return rc;
        if (a->tap_depths_all) {
            hipError_t e = hipMemcpyAsync(a->tap_depths_all, t_all, nr * (uint64_t)(D + Di) * 4, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return fail(NFE_ELAUNCH, "tap copy: %s", hipGetErrorString(e));
        }
    }
    long long cb = ((long long)nr + 255) / 256;
    if (cb > 4096) cb = 4096;
    ClampTaps taps{};                   // NaN for these too when the call lost a hand-off (see depth_clamp_kernel)
    taps.p[0] = a->tap_depths_all; taps.n[0] = nr * (uint64_t)(D + Di);
    taps.p[1] = Di > 0 ? a->tap_weights_coarse : nullptr; taps.n[1] = nr * (uint64_t)(D - 1);
    taps.p[2] = Di > 0 ? a->tap_depths_fine : nullptr; taps.n[2] = nr * (uint64_t)Di;
    taps.p[3] = a->tap_sample_colors; taps.n[3] = a->tap_sample_colors ? nfe_render_sample_colors_floats(a->n_views, a->n_rays, D + Di) : 0;
    hipLaunchKernelGGL(depth_clamp_kernel, dim3((unsigned)cb), dim3(256), 0, st, a->depth, a->rgb, a->seg, a->wsum, (long long)nr, minmax, taps,
                       handoff_status_word());
    NFE_CHECK_LAUNCH("depth_clamp_kernel");
    return NFE_OK;
}

extern "C" int nfe_point_query(const float* planes_geo, const float* planes_app, int plane_h, int plane_w,
                               int64_t plane_view_stride, const float* geo_scale, const float* geo_shift,
                               const float* app_scale, const float* app_shift, const float* decoder_packed,
                               int decoder_math, const float* coords, int n_views, int n_points, float box_warp,
                               float* rgb, float* sigma, float* seg, float density_noise, uint64_t seed, const float* decoder_cross,
                               nfe_stream_t stream) {
    NFE_REQUIRE(planes_geo && planes_app && decoder_packed && coords, "nfe_point_query: null input pointer");
    NFE_REQUIRE(rgb && sigma && seg, "nfe_point_query: null output pointer");
    NFE_REQUIRE(plane_h > 0 && plane_w > 0 && (long long)plane_h * plane_w <= (1ll << 25), "nfe_point_query: bad plane size %dx%d", plane_h, plane_w);
    NFE_REQUIRE(n_views > 0 && n_points >= 0, "nfe_point_query: bad sizes N=%d P=%d", n_views, n_points);
    NFE_REQUIRE(box_warp > 0.0f, "nfe_point_query: box_warp must be positive");
    NFE_REQUIRE(decoder_math == NFE_MATH_BF16X3 || decoder_math == NFE_MATH_FP32, "nfe_point_query: unknown decoder_math %d", decoder_math);
    NFE_REQUIRE(density_noise >= 0.0f, "nfe_point_query: density_noise must be >= 0");
    NFE_REQUIRE((long long)n_views * n_points < (1ll << 32) || density_noise == 0.0f, "nfe_point_query: density_noise needs N*P < 2^32");
    if (n_points == 0) return NFE_OK;
    PointK P{};
    P.planes_g = planes_geo; P.planes_a = planes_app; P.plane_view_stride = plane_view_stride; P.H = plane_h; P.W = plane_w;
    P.aff[0] = geo_scale; P.aff[1] = geo_shift; P.aff[2] = app_scale; P.aff[3] = app_shift;
    P.dec = decoder_packed; P.coords = coords; P.N = n_views; P.Pn = n_points; P.coord_scale = 2.0f / box_warp;
    P.rgb = rgb; P.sigma = sigma; P.seg = seg; P.density_noise = density_noise; P.seed = seed; P.dec_cross = decoder_cross;
    if (decoder_cross)
        NFE_REQUIRE(planes_geo == planes_app && decoder_math == NFE_MATH_BF16X3, "nfe_point_query: decoder_cross needs one plane set and NFE_MATH_BF16X3");
    const long long total = (long long)n_views * ((n_points + 31) / 32);
    long long blocks = (total + 3) / 4;
    if (blocks > (long long)num_cus() * 8) blocks = (long long)num_cus() * 8;
    hipStream_t st = (hipStream_t)stream;
    const bool dual = planes_geo != planes_app;
    dim3 grid((unsigned)blocks), block(256);
    if (decoder_cross) {
        static LdsOptIn opt;
        const hipError_t e = opt.apply(point_kernel<false, NFE_MATH_BF16X3, true>, RENDER_LDS_BYTES_CROSS);
        if (e != hipSuccess) return fail(NFE_ELAUNCH, "point_kernel<CROSS>: LDS opt-in: %s", hipGetErrorString(e));
        hipLaunchKernelGGL((point_kernel<false, NFE_MATH_BF16X3, true>), grid, block, RENDER_LDS_BYTES_CROSS, st, P);
    } else if (decoder_math == NFE_MATH_FP32) {
        if (dual) hipLaunchKernelGGL((point_kernel<true, NFE_MATH_FP32>), grid, block, RENDER_LDS_BYTES, st, P);
        else hipLaunchKernelGGL((point_kernel<false, NFE_MATH_FP32>), grid, block, RENDER_LDS_BYTES, st, P);
    } else {
        if (dual) hipLaunchKernelGGL((point_kernel<true, NFE_MATH_BF16X3>), grid, block, RENDER_LDS_BYTES, st, P);
        else hipLaunchKernelGGL((point_kernel<false, NFE_MATH_BF16X3>), grid, block, RENDER_LDS_BYTES, st, P);
    }
    NFE_CHECK_LAUNCH("point_kernel");
    return NFE_OK;
}

extern "C" int nfe_decoder_forward(const float* features_geo, const float* features_app, int n_views, int n_planes, int64_t n_points,
                                   const float* decoder_packed, int decoder_math, const float* decoder_cross,
                                   float* rgb, float* sigma, float* seg, nfe_stream_t stream) {
    NFE_REQUIRE(features_geo && features_app && decoder_packed, "nfe_decoder_forward: null input pointer");
    NFE_REQUIRE(rgb && sigma && seg, "nfe_decoder_forward: null output pointer");
    NFE_REQUIRE(n_views > 0 && n_planes > 0 && n_points >= 0, "nfe_decoder_forward: bad sizes N=%d planes=%d P=%lld", n_views, n_planes, (long long)n_points);
    NFE_REQUIRE(decoder_math == NFE_MATH_BF16X3 || decoder_math == NFE_MATH_FP32, "nfe_decoder_forward: unknown decoder_math %d", decoder_math);
    if (decoder_cross)
        NFE_REQUIRE(features_geo == features_app && decoder_math == NFE_MATH_BF16X3, "nfe_decoder_forward: decoder_cross needs one feature set and NFE_MATH_BF16X3");
    if (n_points == 0) return NFE_OK;
    DecoderK P{};
    P.feat_g = features_geo; P.feat_a = features_app; P.N = n_views; P.n_planes = n_planes; P.Pn = n_points;
    P.dec = decoder_packed; P.dec_cross = decoder_cross; P.rgb = rgb; P.sigma = sigma; P.seg = seg;
    const long long total = (long long)n_views * ((n_points + 31) / 32);
    long long blocks = (total + 3) / 4;
    if (blocks > (long long)num_cus() * 8) blocks = (long long)num_cus() * 8;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)blocks), block(256);
    if (decoder_cross) {
        static LdsOptIn opt;
        const hipError_t e = opt.apply(decoder_kernel<NFE_MATH_BF16X3, true>, RENDER_LDS_BYTES_CROSS);
        if (e != hipSuccess) return fail(NFE_ELAUNCH, "decoder_kernel<CROSS>: LDS opt-in: %s", hipGetErrorString(e));
        hipLaunchKernelGGL((decoder_kernel<NFE_MATH_BF16X3, true>), grid, block, RENDER_LDS_BYTES_CROSS, st, P);
    } else if (decoder_math == NFE_MATH_FP32) {
        hipLaunchKernelGGL((decoder_kernel<NFE_MATH_FP32>), grid, block, RENDER_LDS_BYTES, st, P);
    } else {
        hipLaunchKernelGGL((decoder_kernel<NFE_MATH_BF16X3>), grid, block, RENDER_LDS_BYTES, st, P);
    }
    NFE_CHECK_LAUNCH("decoder_kernel");
    return NFE_OK;
}
