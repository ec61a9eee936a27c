/*
 * nfe_render.h — C ABI of libnfe_render.so: the MI355X (gfx950) volumetric-rendering hot path of
 * NeRFFaceEditing (TriPlaneGenerator.synthesis()/sample(): ray sampling, plane statistics,
 * tri-plane gather, dual MLP decoder, importance sampling, alpha compositing).
 *
 * Conventions (they mirror the reference's plugin convention in
 * torch_utils/ops/bias_act.cpp:36-94 and upfirdn2d.cpp:20-96: validate -> launch on the caller's
 * stream -> return, no host sync):
 *   - extern "C", plain pointers and sizes; no torch / pybind types.
 *   - every entry point returns 0 on success or a negative NFE_E* code; nfe_last_error() gives a
 *     thread-local message (the Python shim raises RuntimeError, as TORCH_CHECK does in the
 *     reference: bias_act.cpp:39-55).
 *   - all data pointers are DEVICE pointers to fp32 unless stated; the caller allocates every
 *     output and the workspace, the library never frees or retains caller memory.
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream.
 *   - re-entrant, and stateless EXCEPT for two diagnostics that no result depends on (ABI v15): one pinned host word per process,
 *     the sticky count of lost wave hand-offs (nfe_render_status: written by the device with system-scope atomics from any
 *     stream of any thread, so a count cannot be attributed to a thread or stream - the per-call answer is nfe_render_call_status /
 *     nfe_render_backward_call_status on the CALL'S OWN workspace), and a thread-local list of kernel names
 *     (nfe_render_last_kernels).  Neither makes any call refuse to run or changes what it computes.
 *
 * The reference has no native entry point for the renderer (its renderer is ~25 ATen ops,
 * training/volumetric_rendering/renderer.py:301-363); each function below names the reference
 * Python function(s) it replaces.
 */
#ifndef NFE_RENDER_H
#define NFE_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFE_ABI_VERSION 15

#define NFE_OK 0
#define NFE_EINVAL (-1)      /* bad argument (null pointer, size out of range, unsupported option) */
#define NFE_ELAUNCH (-2)     /* HIP launch / runtime error */
#define NFE_EWORKSPACE (-3)  /* workspace too small */
#define NFE_EHANDOFF (-4)    /* the call this status query is about lost wave hand-offs; its outputs are NaN (see "lost hand-offs") */

#define NFE_PLANE_CHANNELS 32   /* channels per plane (triplane.py:113-115) */
#define NFE_NUM_PLANES 3
#define NFE_RGB_CHANNELS 32     /* decoder_output_dim (triplane.py:49) */
#define NFE_SEG_CHANNELS 15     /* decoder_seg_dim   (triplane.py:49) */
#define NFE_MAX_SAMPLES 256     /* max depth_resolution and max depth_resolution_importance */
/* 4-byte words in a packed decoder blob (see nfe_decoder_pack): fp32 MFMA fragments + biases,
 * followed by the split-bf16 (hi,lo) MFMA fragments */
#define NFE_DECODER_PACKED_FLOATS (4 * 2048 + 64 + 64 + 32 + 32 + 8192)
/* nfe_render_args.decoder_math */
#define NFE_MATH_BF16X3 0   /* default: operands split into bf16 hi+lo, 3 bf16 MFMAs per product
                               (~2^-16 relative per product, fp32 accumulate) */
#define NFE_MATH_FP32 1     /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32), ~5x more matrix-pipe time */

typedef void* nfe_stream_t;

int nfe_abi_version(void);
const char* nfe_last_error(void);

/* ---- a2: RaySampler.forward (training/volumetric_rendering/ray_sampler.py:24-62) -------------
 * cam2world [N,16] row-major 4x4, intrinsics [N,9] row-major 3x3 (normalised by image size),
 * -> origins [N,R*R,3], dirs [N,R*R,3]; pixel m = row*R + col. */
int nfe_ray_sampler(const float* cam2world, const float* intrinsics, int n_views, int resolution,
                    float* origins, float* dirs, nfe_stream_t stream);

/* ---- a12 ('auto' ray limits): math_utils.get_ray_limits_box (math_utils.py:46-98) followed by the
 * invalid-ray fix-up of renderer.py:312-318: rays that miss the [-L/2,L/2]^3 box get
 * start = min(valid starts), end = max(valid STARTS) (sic, as the reference does); if no ray hits the box
 * the (-1,-2) sentinels are left in place.  origins/dirs [n_rays,3] -> ray_start/ray_end [n_rays].
 * scratch: 8 bytes of device memory. */
int nfe_ray_limits_box(const float* origins, const float* dirs, int64_t n_rays, float box_side_length,
                       float* ray_start, float* ray_end, void* scratch, nfe_stream_t stream);

/* ---- a4: compute_mean_var (training/triplane.py:56-60) ---------------------------------------
 * planes [N,C,H*W] (NCHW) -> mean [N,C], std [N,C] = sqrt(unbiased variance). */
int nfe_plane_stats(const float* planes, int n, int c, int hw, float* mean, float* std,
                    nfe_stream_t stream);

/* ---- a4: normalize_plane / denormalize_plane as a per-channel affine (triplane.py:61-68) ------
 * out[n,c,:] = in[n,c,:] * scale[n_s,c] + shift[n_s,c]; scale/shift are [N,C] or [1,C]
 * (n_affine = N or 1).  normalize: scale=1/(std+1e-8), shift=-mean*scale (see nfe_make_affine). */
int nfe_plane_affine(const float* in, const float* scale, const float* shift, int n, int c, int hw,
                     int n_affine, float* out, nfe_stream_t stream);

/* ---- a4 -> a5: the single-gather identity --------------------------------------
 * From the statistics of the raw planes (mean,std [N,C]) and an optional override (new_mean,
 * new_std [N_o,C] with N_o in {1,N}; NULL = no override, triplane.py:98-103) produce the four
 * per-channel affines [N,C] that turn a bilinear sample s of the RAW plane (with in-bounds tap
 * weight sum w) into a sample of the normalised plane (geo) and of the denormalised plane (app):
 *     geo = s*geo_scale + w*geo_shift         app = s*app_scale + w*app_shift               */
int nfe_make_affine(const float* mean, const float* std, const float* new_mean, const float* new_std,
                    int n, int c, int n_override, float* geo_scale, float* geo_shift,
                    float* app_scale, float* app_shift, nfe_stream_t stream);

/* ---- layout: NCHW planes [N,96,H,W] -> gather layout [N,3,H,W,32] (one 128-byte texel per tap) */
int nfe_plane_pack(const float* planes_nchw, int n, int h, int w, float* packed, nfe_stream_t stream);

/* ---- a6: decoder weights -> MFMA operand layout ----------------------------------------------
 * DisentangledOSGDecoder parameters (triplane.py:232-247), FullyConnectedLayer gains applied
 * (networks_stylegan2.py:111-123: weight*lr_mul/sqrt(in), bias*lr_mul).  All device pointers:
 * geo_w0[64,32] geo_b0[64] geo_w1[16,64] geo_b1[16] app_w0[64,32] app_b0[64] app_w1[32,64]
 * app_b1[32] -> packed[NFE_DECODER_PACKED_FLOATS]. */
int nfe_decoder_pack(const float* geo_w0, const float* geo_b0, const float* geo_w1, const float* geo_b1,
                     const float* app_w0, const float* app_b0, const float* app_w1, const float* app_b1,
                     float lr_mul, float* packed, nfe_stream_t stream);

/* ---- SegmentationOSGDecoder (triplane.py:192-230, the `disable_alignment` ablation): sigma and rgb come from `net`,
 * seg from `seg_net`, both reading the same features.  For the fused kernels `seg_net` is the geometry head (its sigma
 * row zero, sigma's bias in geo_b1[0]), `net` rows 1..32 the appearance head, and the sigma row of `net` is this CROSS
 * matrix: cross_w1 [16,64], row r = weights of geometry output r (0 = sigma, 1..15 = seg) on the APPEARANCE head's
 * hidden units -> packed_cross [NFE_DECODER_CROSS_FLOATS].  Pass it as nfe_render_args.decoder_cross /
 * nfe_point_query(decoder_cross); needs planes_geo == planes_app, NFE_MATH_BF16X3, density_noise == 0. */
#define NFE_DECODER_CROSS_FLOATS 2048
int nfe_decoder_pack_cross(const float* cross_w1, float lr_mul, float* packed_cross, nfe_stream_t stream);

/* ---- a5..a12: DisentangledImportanceRenderer.forward (renderer.py:301-363) --------------------*/
typedef struct nfe_render_args {
    uint32_t struct_size;              /* = sizeof(nfe_render_args) */
    /* planes, gather layout [Np,3,H,W,32]; planes_app may equal planes_geo (single gather) */
    const float* planes_geo;           /* source of the geometry ("norm") feature set */
    const float* planes_app;           /* source of the appearance ("denorm") feature set */
    int32_t plane_h, plane_w;
    int64_t plane_view_stride;         /* floats between views; 0 broadcasts one plane set */
    /* optional per-(view,channel) affines [N,96] applied to sampled values (NULL = identity) */
    const float* geo_scale; const float* geo_shift;
    const float* app_scale; const float* app_shift;
    const float* decoder_packed;       /* from nfe_decoder_pack */
    int32_t decoder_math;              /* NFE_MATH_* */
    /* rays: explicit origins/dirs [N,M,3], or (both NULL) generated from cam2world/intrinsics */
    int32_t n_views, n_rays;           /* N, M */
    const float* origins; const float* dirs;
    const float* cam2world; const float* intrinsics;   /* [N,16], [N,9] */
    int32_t resolution;                /* R with M == R*R (0 if unknown: no 2-D tiling) */
    /* rendering_kwargs (train.py:288-313) */
    int32_t depth_resolution;          /* D  >= 2 */
    int32_t depth_resolution_importance; /* Di >= 0 */
    float ray_start, ray_end;          /* scalar limits (ignored when per-ray limits given) */
    const float* ray_start_per_ray;    /* [N,M] or NULL ('auto' branch, renderer.py:312-318) */
    const float* ray_end_per_ray;
    int32_t disparity_space_sampling;  /* renderer.py:174-181 */
    float box_warp;
    int32_t white_back;                /* ray_marcher.py:96-97 */
    /* jitter: external buffers (parity mode) or Philox4x32-10 keyed by seed (NULL pointers) */
    const float* u_coarse;             /* [N,M,D]   or NULL */
    const float* u_fine;               /* [N*M,Di]  or NULL */
    uint64_t seed;
    const uint64_t* seed_device;       /* optional: read the Philox key from device memory instead (lets a
                                          captured hipGraph draw new jitter per replay); NULL = use `seed` */
    /* outputs */
    float* rgb;                        /* [N,M,32]  (or [N,32,M] if channels_first) */
    float* seg;                        /* [N,M,15]  (or [N,15,M]) */
    float* depth;                      /* [N,M] */
    float* wsum;                       /* [N,M] */
    int32_t channels_first;
    float* tap_weights_coarse;         /* optional [N,M,D-1] (two-pass only) */
    float* tap_depths_fine;            /* optional [N,M,Di] */
    float* tap_depths_all;             /* optional [N,M,D+Di] sorted */
    void* workspace; uint64_t workspace_bytes;
    float density_noise;               /* renderer.py:285-286: sigma += N(0,1) * density_noise; the normals are Philox
                                          draws keyed by (seed, ray, draw index), 0 = off.  With importance sampling the
                                          workspace must hold N*M*(D+Di)*4 more bytes (rounded up to 256). */
    const float* decoder_cross;        /* optional, from nfe_decoder_pack_cross (SegmentationOSGDecoder); NULL = none */
    uint64_t* clock_probe;             /* ABI v10, optional, device [4]: the final render launch stamps {shader-cycle counter,
                                          100 MHz reference counter} at its start and its end (workgroup 0): effective shader clock
                                          of that launch = (p[2]-p[0]) / (p[3]-p[1]) x 100 MHz.  Measurement only; NULL = off */
    float* tap_sample_colors;          /* ABI v11, optional, nfe_render_sample_colors_floats() floats, opaque: what the decoders
                                          returned for every sample of the final march (32 colour features, 15 segmentation
                                          logits, sigma - with its density noise, if any).  Handed to nfe_render_backward as
                                          `sample_colors` it replaces that call's re-evaluation pass (gather + both decoder heads for
                                          every sample) by one pass over these values.  Only with the split-bf16 decoder; since ABI v15
                                          also with density_noise (the backward of a noisy render NEEDS them: it has no draws of its
                                          own) and with decoder_cross. */
    const float* density_noise_values; /* ABI v15, optional [N,M,D+Di]: the standard normals themselves instead of the Philox draws, indexed by
                                          draw - coarse sample k at k, the fine sample of ascending rank r at D + r (the reference draws
                                          randn_like per run_model call, renderer.py:285-286: this is the parity hook, like u_coarse /
                                          u_fine for the jitter).  Used only when density_noise > 0. */
} nfe_render_args;

/* floats of nfe_render_args.tap_sample_colors / nfe_render_backward_args.sample_colors for these sizes (S = D + Di) */
uint64_t nfe_render_sample_colors_floats(int n_views, int n_rays, int n_samples);

/* bytes of workspace nfe_render needs for these sizes: the depth min/max words, 13.6 MB for the segment composites of
 * depth-split launches (few rays: every ray block's march is cut into segments marched by different waves), and with
 * importance sampling the coarse depths / weights and the merged depths of every ray (density_noise: see above) */
uint64_t nfe_render_workspace_bytes(int n_views, int n_rays, int depth_resolution,
                                    int depth_resolution_importance);
/* Lost hand-offs (ABI v15; v14 made the NEXT nfe_render of the process fail instead, which blamed a healthy call on another
 * stream or thread, launched nothing for it, and never reported the last call of a process).  Large launches run the
 * wave-specialised kernel: gather waves hand feature tiles to decoder waves of the same workgroup through LDS counters.  A wave
 * that waits longer than ~50 ms for its partner gives up instead of hanging the GPU (never observed; a bounded wait is the safety
 * net).  Such a call is NOT silent garbage, and the failure belongs to THAT call:
 *   - the kernel that closes every nfe_render call overwrites ALL outputs the call was given with NaN: rgb, seg, depth, wsum and
 *     the optional taps (tap_depths_all, tap_weights_coarse, tap_depths_fine, tap_sample_colors) - a backward fed with them
 *     produces NaN gradients, not plausible ones;
 *   - the number of abandoned waits stays in the call's workspace: nfe_render_call_status(workspace, stream, &lost) waits for
 *     `stream`, reads it and returns NFE_EHANDOFF when it is not 0 (NFE_OK otherwise) - the caller's own synchronisation point is
 *     where an asynchronous failure can be reported against the right call (the reference's TORCH_CHECK convention,
 *     bias_act.cpp:39-55, has no asynchronous failures; this is the closest a stream-ordered library gets);
 *   - the same kernel adds (count, 1 call) to the process's sticky status word with ONE 64-bit system-scope atomic, so that a
 *     caller that never synchronises per call can still find out, without any synchronisation, that SOME call was poisoned:
 *     nfe_render_status(&lost, &calls, clear) reads (clear != 0: and resets, one atomic exchange) it at any time.  No later call
 *     is refused because of it.  *lost_handoffs = abandoned waits, *poisoned_calls = render / backward calls whose outputs were set
 *     to NaN since the last clear.  Either may be NULL.  Callable without a GPU (returns zeros). */
int nfe_render(const nfe_render_args* args, nfe_stream_t stream);
int nfe_render_status(uint32_t* lost_handoffs, uint32_t* poisoned_calls, int clear);
/* Per-call status: synchronises `stream`, then reads the abandoned-wait count of the LAST nfe_render call that used `workspace`
 * (both passes of a two-pass call).  Returns NFE_OK (count 0: the outputs are valid), NFE_EHANDOFF (count > 0: every output of that
 * call is NaN; repeat it - NFE_RENDER_WS=0 selects the fused kernel, which has no hand-off), or NFE_ELAUNCH.  *lost_handoffs optional. */
int nfe_render_call_status(const void* workspace, nfe_stream_t stream, uint32_t* lost_handoffs);
/* Names of the render kernels the last nfe_render call of THIS thread launched, space separated, in launch order (e.g.
 * "render_ws_kernel<4,2,SIGMA_ONLY> importance_kernel render_ws_kernel<4,2,DUAL>"); diagnostic, valid until the thread's next call. */
const char* nfe_render_last_kernels(void);

/* ---- a15: renderer.run_model on caller-supplied points (triplane.py:140-157, renderer.py:259-287)
 * coords [N,P,3] -> rgb [N,P,32], sigma [N,P], seg [N,P,15]. Plane/affine/decoder arguments as in
 * nfe_render_args.  density_noise > 0 (renderer.py:285-286): sigma += N(0,1) * density_noise, the normal of point
 * (n, m) being the Philox draw keyed by (seed; n*P + m, draw 0), as in nfe_render. */
int nfe_point_query(const float* planes_geo, const float* planes_app, int plane_h, int plane_w,
                    int64_t plane_view_stride, const float* geo_scale, const float* geo_shift,
                    const float* app_scale, const float* app_shift, const float* decoder_packed,
                    int decoder_math, const float* coords, int n_views, int n_points, float box_warp,
                    float* rgb, float* sigma, float* seg, float density_noise, uint64_t seed,
                    const float* decoder_cross /* optional, see nfe_decoder_pack_cross */, nfe_stream_t stream);

/* ---- a6 as a stand-alone module: DisentangledOSGDecoder.forward (triplane.py:249-270), OSGDecoder.forward (:178-190),
 * SegmentationOSGDecoder.forward (:209-230) on caller-supplied sampled features.
 * features_geo / features_app [N, n_planes, P, 32] (what sample_from_planes returns, renderer.py:55-65; the mean over the
 * plane axis is taken here) -> rgb [N,P,32] (sigmoid clamp applied), sigma [N,P], seg [N,P,15].  features_app may equal
 * features_geo (OSGDecoder; SegmentationOSGDecoder with decoder_cross).  decoder_packed / decoder_cross as in nfe_render_args. */
int nfe_decoder_forward(const float* features_geo, const float* features_app, int n_views, int n_planes, int64_t n_points,
                        const float* decoder_packed, int decoder_math, const float* decoder_cross,
                        float* rgb, float* sigma, float* seg, nfe_stream_t stream);

/* ---- backward of a5..a12 with respect to the plane sets ------------------------------------------
 * The vector-Jacobian product torch autograd computes for DisentangledImportanceRenderer.forward
 * (renderer.py:301-363) w.r.t. `norm_planes` / `denorm_planes`: through SegMipRayMarcher2.run_forward
 * (ray_marcher.py:68-101), DisentangledOSGDecoder.forward (triplane.py:249-270) and F.grid_sample (renderer.py:64).
 * This is what plane optimisation (geometry / appearance editing through utils.decode, utils.py:165-199) needs;
 * decoder parameters are constants here.  The sample depths are constants too: stratified depths do not depend on
 * the planes and importance depths are detached in the reference (renderer.py:198,211), so the caller passes the
 * sorted depths the forward marched (nfe_render_args.tap_depths_all) and the gradient flows through that one march.
 * Passes: per-sample sigma and cotangent-weighted colour (from the forward's kept per-sample outputs, or a re-evaluation), a
 * per-ray reverse recurrence (no divisions by 1 - alpha), the per-sample decoder backward (wave-specialised: gather waves hand
 * feature tiles to decoder waves through LDS counters), and a binned accumulate pass.  density_noise is not supported (absent
 * from every shipped config).
 * Lost hand-offs: the decoder-backward kernel bounds its waits like the render kernel (~100 ms; never observed).  A call that
 * abandoned one does not pass for a result (ABI v15): the kernel that closes the call overwrites BOTH gradient buffers entirely
 * with NaN (every view, whichever chunk of the call lost the hand-off), adds (count, 1 call) to the sticky status word
 * nfe_render_status reads, and leaves the count in the workspace for nfe_render_backward_call_status (same contract as
 * nfe_render_call_status).  The direct and sorted forms (NFE_BWD_SCATTER=direct / sorted, planes beyond the bin table) and the
 * single-wave decoder kernel have no hand-off and never report one. */
typedef struct nfe_render_backward_args {
    uint32_t struct_size;              /* = sizeof(nfe_render_backward_args) */
    const float* planes_geo;           /* as in nfe_render_args (gather layout [Np,3,H,W,32]) */
    const float* planes_app;
    int32_t plane_h, plane_w;
    int64_t plane_view_stride;
    const float* geo_scale; const float* geo_shift;     /* optional affines [N,96], as in nfe_render_args */
    const float* app_scale; const float* app_shift;
    /* raw decoder parameters, as nfe_decoder_pack takes them */
    const float* geo_w0; const float* geo_b0; const float* geo_w1; const float* geo_b1;
    const float* app_w0; const float* app_b0; const float* app_w1; const float* app_b1;
    float lr_mul;
    int32_t n_views, n_rays;
    const float* origins; const float* dirs;            /* [N,M,3] or both NULL: rays from cam2world/intrinsics */
    const float* cam2world; const float* intrinsics;
    int32_t resolution;                /* R with R*R == n_rays: required for camera rays; with origins/dirs a hint that the
                                          rays are an R x R image in row-major order (8x8 ray tiles per wave), 0 = unknown */
    int32_t n_samples;                 /* S = depth_resolution + depth_resolution_importance */
    const float* depths;               /* [N,M,S] ascending per ray: tap_depths_all of the forward call */
    float box_warp;
    int32_t white_back;
    /* cotangents of the four outputs (NULL = zero) */
    const float* grad_rgb;             /* [N,M,32] ([N,32,M] if channels_first) */
    const float* grad_seg;             /* [N,M,15] ([N,15,M]) */
    const float* grad_depth;           /* [N,M] */
    const float* grad_wsum;            /* [N,M] */
    int32_t channels_first;
    /* outputs, ACCUMULATED INTO (the caller zeroes them): gradients w.r.t. the SAMPLED plane sets in gather layout.
     * With affines the chain rule through `scale` is applied, i.e. these are gradients w.r.t. the stored planes.
     * Either may be NULL (that decoder branch is skipped); both may point to the same buffer (single-gather mode). */
    float* grad_planes_geo;
    float* grad_planes_app;
    int64_t grad_view_stride;          /* floats between views of the gradient buffers (0: all views add into one set) */
    void* workspace; uint64_t workspace_bytes;
    const float* sample_colors;        /* ABI v11, optional: tap_sample_colors of the forward call that produced `depths` (same rays,
                                          planes and decoder); NULL = the samples are re-evaluated here */
} nfe_render_backward_args;

uint64_t nfe_render_backward_workspace_bytes(int n_views, int n_rays, int n_samples);
int nfe_render_backward(const nfe_render_backward_args* args, nfe_stream_t stream);
int nfe_render_backward_call_status(const void* workspace, nfe_stream_t stream, uint32_t* lost_handoffs);

#ifdef __cplusplus
}
#endif
#endif /* NFE_RENDER_H */
