/*
 * render_oracle.c — plain-C restatement of the NeRFFaceEditing render core (CPU, fp32, OpenMP over rays).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ as a fast checker and by bench.py's cpu_baseline leg.
 * The product path (nerffaceediting_amd) never links or calls it.
 *
 * Parity pinning: tests/test_oracle_golden.py checks this file against the golden vectors captured
 * from the reference (oracle/gen_golden.py) and against the numpy oracle (oracle/render_oracle.py).
 *
 * It follows DisentangledImportanceRenderer.forward (training/volumetric_rendering/renderer.py:301-363):
 *   sample_stratified           renderer.py:169-192
 *   sample_from_planes          renderer.py:55-65   (grid_sample bilinear, zeros, align_corners=False)
 *   DisentangledOSGDecoder      training/triplane.py:249-270, FullyConnectedLayer networks_stylegan2.py:114-123
 *   SegMipRayMarcher2           ray_marcher.py:68-101
 *   sample_importance/_pdf      renderer.py:194-253
 *   unify_samples               renderer.py:288-300
 * Planes come in the reference layout [N,3,32,H,W]; they are re-laid channels-last internally (a CPU
 * implementation would do the same), which does not change any arithmetic.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define C 32
#define HID 64
#define NSEG 15
#define MAXS 512

typedef struct {
    float w0g[HID * C], b0g[HID], w1g[16 * HID], b1g[16];
    float w0a[HID * C], b0a[HID], w1a[C * HID], b1a[C];
} dec_t;

static float softplusf(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

/* bilinear sample of one channels-last plane [H][W][32] at normalised (gx,gy); accumulates into out[32] */
static void sample_plane(const float* pl, int H, int W, float gx, float gy, float* out) {
    float ix = (gx + 1.0f) * (0.5f * (float)W) - 0.5f, iy = (gy + 1.0f) * (0.5f * (float)H) - 0.5f;
    float x0f = floorf(ix), y0f = floorf(iy);
    float dx = ix - x0f, dy = iy - y0f, ex = 1.0f - dx, ey = 1.0f - dy;
    if (!(x0f > -3.0f)) x0f = -3.0f; if (!(x0f < (float)(W + 2))) x0f = (float)(W + 2);
    if (!(y0f > -3.0f)) y0f = -3.0f; if (!(y0f < (float)(H + 2))) y0f = (float)(H + 2);
    int x0 = (int)x0f, y0 = (int)y0f;
    const int xs[4] = {x0, x0 + 1, x0, x0 + 1}, ys[4] = {y0, y0, y0 + 1, y0 + 1};
    const float ws[4] = {ex * ey, dx * ey, ex * dy, dx * dy};
    for (int c = 0; c < C; ++c) out[c] = 0.0f;
    for (int t = 0; t < 4; ++t) {
        if (xs[t] < 0 || xs[t] >= W || ys[t] < 0 || ys[t] >= H) continue;
        const float* tex = pl + ((size_t)ys[t] * W + xs[t]) * C;
        for (int c = 0; c < C; ++c) out[c] += ws[t] * tex[c];
    }
}

/* run_model at one point: rgb[32], sigma, seg[15]  (renderer.py:259-287 + triplane.py:249-270) */
static void eval_point(const float* ng, const float* dg, int H, int W, const dec_t* d, float scale,
                       float px, float py, float pz, float* rgb, float* sigma, float* seg) {
    const float gx = scale * px, gy = scale * py, gz = scale * pz;
    const float us[3] = {gx, gx, gz}, vs[3] = {gy, gz, gx};          /* p0=(x,y) p1=(x,z) p2=(z,x) */
    float fn[C], fd[C], tmp[C], h[HID];
    const size_t pe = (size_t)H * W * C;
    for (int c = 0; c < C; ++c) fn[c] = fd[c] = 0.0f;
    for (int p = 0; p < 3; ++p) {
        sample_plane(ng + p * pe, H, W, us[p], vs[p], tmp);
        for (int c = 0; c < C; ++c) fn[c] += tmp[c];
        sample_plane(dg + p * pe, H, W, us[p], vs[p], tmp);
        for (int c = 0; c < C; ++c) fd[c] += tmp[c];
    }
    for (int c = 0; c < C; ++c) { fn[c] /= 3.0f; fd[c] /= 3.0f; }   /* mean(1), triplane.py:251-252 */
    for (int u = 0; u < HID; ++u) {
        float a = d->b0g[u];
        for (int c = 0; c < C; ++c) a += fn[c] * d->w0g[u * C + c];
        h[u] = softplusf(a);
    }
    float g[16];
    for (int o = 0; o < 16; ++o) {
        float a = d->b1g[o];
        for (int u = 0; u < HID; ++u) a += h[u] * d->w1g[o * HID + u];
        g[o] = a;
    }
    *sigma = g[0];
    for (int s = 0; s < NSEG; ++s) seg[s] = g[1 + s];
    for (int u = 0; u < HID; ++u) {
        float a = d->b0a[u];
        for (int c = 0; c < C; ++c) a += fd[c] * d->w0a[u * C + c];
        h[u] = softplusf(a);
    }
    for (int o = 0; o < C; ++o) {
        float a = d->b1a[o];
        for (int u = 0; u < HID; ++u) a += h[u] * d->w1a[o * HID + u];
        rgb[o] = (1.0f / (1.0f + expf(-a))) * 1.002f - 0.001f;       /* triplane.py:269 */
    }
}

/* SegMipRayMarcher2.run_forward for one ray with S samples (ray_marcher.py:68-101); returns the
 * unclamped depth (NaN when the weights sum to 0); weights_out (S-1) may be NULL. */
static void march(const float* t, const float* rgb, const float* sig, const float* seg, int S, int white_back,
                  float* o_rgb, float* o_seg, float* o_depth, float* o_wsum, float* weights_out) {
    float T = 1.0f, wsum = 0.0f, dsum = 0.0f;
    float ar[C], as[NSEG];
    for (int c = 0; c < C; ++c) ar[c] = 0.0f;
    for (int c = 0; c < NSEG; ++c) as[c] = 0.0f;
    for (int k = 0; k + 1 < S; ++k) {
        const float delta = t[k + 1] - t[k];
        const float dens = softplusf((sig[k] + sig[k + 1]) / 2.0f - 1.0f);
        const float alpha = 1.0f - expf(-(dens * delta));
        const float w = alpha * T;
        T = T * (1.0f - alpha + 1e-10f);
        if (weights_out) weights_out[k] = w;
        for (int c = 0; c < C; ++c) ar[c] += w * ((rgb[k * C + c] + rgb[(k + 1) * C + c]) / 2.0f);
        for (int c = 0; c < NSEG; ++c) as[c] += w * ((seg[k * NSEG + c] + seg[(k + 1) * NSEG + c]) / 2.0f);
        dsum += w * ((t[k] + t[k + 1]) / 2.0f);
        wsum += w;
    }
    if (o_rgb) {
        for (int c = 0; c < C; ++c) o_rgb[c] = (ar[c] + (white_back ? 1.0f - wsum : 0.0f)) * 2.0f - 1.0f;
        for (int c = 0; c < NSEG; ++c) o_seg[c] = as[c];
        *o_depth = dsum / wsum;
        *o_wsum = wsum;
    }
}

static int cmp_idx_depth(const void* a, const void* b, void* ctx) {
    const float* t = (const float*)ctx;
    const int ia = *(const int*)a, ib = *(const int*)b;
    if (t[ia] < t[ib]) return -1;
    if (t[ia] > t[ib]) return 1;
    return ia - ib;                                  /* stable: coarse before fine on ties */
}

/*
 * norm_planes/denorm_planes: [N,3,32,H,W] (reference layout) or with plane_batch==1 broadcast.
 * dec: 8 arrays in the reference's parameter shapes; lr_mul applied as FullyConnectedLayer does.
 * ray_start_per_ray/ray_end_per_ray: NULL or [N,M] ('auto' branch).
 * u_coarse [N,M,D]; u_fine [N*M,Di] (Di>0).  Outputs rgb [N,M,32], seg [N,M,15], depth [N,M], wsum [N,M].
 * taps (optional, may be NULL): weights_coarse [N,M,D-1], depths_fine [N,M,Di].
 * Returns 0, or -1 on bad sizes.
 */
int nfe_oracle_render(const float* norm_planes, const float* denorm_planes, int plane_batch, int H, int W,
                      const float* gw0, const float* gb0, const float* gw1, const float* gb1,
                      const float* aw0, const float* ab0, const float* aw1, const float* ab1, float lr_mul,
                      const float* origins, const float* dirs, int N, int M, int D, int Di,
                      float ray_start, float ray_end, const float* ray_start_per_ray, const float* ray_end_per_ray,
                      int disparity, float box_warp, int white_back,
                      const float* u_coarse, const float* u_fine,
                      float* rgb, float* seg, float* depth, float* wsum,
                      float* tap_weights_coarse, float* tap_depths_fine, int n_threads) {
    if (D < 2 || D + Di > MAXS || (Di > 0 && D < 4) || N < 1 || M < 1) return -1;
    dec_t* d = (dec_t*)malloc(sizeof(dec_t));
    const float g0 = lr_mul / sqrtf((float)C), g1 = lr_mul / sqrtf((float)HID);
    for (int i = 0; i < HID * C; ++i) { d->w0g[i] = gw0[i] * g0; d->w0a[i] = aw0[i] * g0; }
    for (int i = 0; i < HID; ++i) { d->b0g[i] = gb0[i] * lr_mul; d->b0a[i] = ab0[i] * lr_mul; }
    for (int i = 0; i < 16 * HID; ++i) d->w1g[i] = gw1[i] * g1;
    for (int i = 0; i < C * HID; ++i) d->w1a[i] = aw1[i] * g1;
    for (int i = 0; i < 16; ++i) d->b1g[i] = gb1[i] * lr_mul;
    for (int i = 0; i < C; ++i) d->b1a[i] = ab1[i] * lr_mul;
    /* channels-last copies */
    const size_t pe = (size_t)H * W * C, ve = 3 * pe;
    float* ncl = (float*)malloc(sizeof(float) * ve * plane_batch);
    float* dcl = (float*)malloc(sizeof(float) * ve * plane_batch);
    for (int nb = 0; nb < plane_batch * 3; ++nb)
        for (int c = 0; c < C; ++c)
            for (size_t i = 0; i < (size_t)H * W; ++i) {
                ncl[(size_t)nb * pe + i * C + c] = norm_planes[((size_t)nb * C + c) * H * W + i];
                dcl[(size_t)nb * pe + i * C + c] = denorm_planes[((size_t)nb * C + c) * H * W + i];
            }
    const float scale = 2.0f / box_warp;
    const int S = D + Di;
    float gmin = INFINITY, gmax = -INFINITY;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 64) reduction(min : gmin) reduction(max : gmax)
    for (long long ray = 0; ray < (long long)N * M; ++ray) {
        const int n = (int)(ray / M);
        const float* ng = ncl + (plane_batch == 1 ? 0 : (size_t)n * ve);
        const float* dg = dcl + (plane_batch == 1 ? 0 : (size_t)n * ve);
        const float* o = origins + ray * 3; const float* dr = dirs + ray * 3;
        float t[MAXS], sg[MAXS], col[MAXS * C], sgm[MAXS * NSEG], w[MAXS];
        float rs = ray_start, re = ray_end;
        if (ray_start_per_ray) { rs = ray_start_per_ray[ray]; re = ray_end_per_ray[ray]; }
        for (int k = 0; k < D; ++k) {                 /* sample_stratified, renderer.py:169-192 */
            const float u = u_coarse[ray * D + k];
            if (disparity) {
                const float s = (float)k * (1.0f / (float)(D - 1)) + u * (1.0f / (float)(D - 1));
                t[k] = 1.0f / (1.0f / rs * (1.0f - s) + 1.0f / re * s);
            } else if (ray_start_per_ray) {
                t[k] = rs + ((float)k / (float)(D - 1)) * (re - rs) + u * ((re - rs) / (float)(D - 1));
            } else {
                const float delta = (re - rs) / (float)(D - 1);
                t[k] = (rs + (float)k * delta) + u * delta;
            }
        }
        for (int k = 0; k < D; ++k)
            eval_point(ng, dg, H, W, d, scale, o[0] + t[k] * dr[0], o[1] + t[k] * dr[1], o[2] + t[k] * dr[2],
                       col + k * C, sg + k, sgm + k * NSEG);
        int Sr = D;
        if (Di > 0) {
            march(t, col, sg, sgm, D, white_back, NULL, NULL, NULL, NULL, w);
            if (tap_weights_coarse) memcpy(tap_weights_coarse + ray * (D - 1), w, sizeof(float) * (D - 1));
            /* sample_importance, renderer.py:194-212: max_pool1d(2,1,pad 1) -> avg_pool1d(2,1) -> +0.01 */
            float a[MAXS], cdf[MAXS], zmid[MAXS];
            for (int i = 0; i + 1 < D; ++i) {
                const float wl = i > 0 ? w[i - 1] : -INFINITY, wc = w[i], wr = i + 1 < D - 1 ? w[i + 1] : -INFINITY;
                const float m0 = fmaxf(wl, wc), m1 = fmaxf(wc, wr);
                a[i] = (m0 + m1) * 0.5f + 0.01f;
                zmid[i] = 0.5f * (t[i] + t[i + 1]);
            }
            const int B = D - 3;                      /* sample_pdf on weights[:,1:-1], renderer.py:214-253 */
            float tot = 0.0f;
            for (int i = 0; i < B; ++i) tot += a[i + 1] + 1e-5f;
            cdf[0] = 0.0f;
            float run = 0.0f;
            for (int i = 0; i < B; ++i) { run += (a[i + 1] + 1e-5f) / tot; cdf[i + 1] = run; }
            for (int e = 0; e < Di; ++e) {
                const float u = u_fine[ray * Di + e];
                int ind = 0;
                while (ind <= B && cdf[ind] <= u) ++ind;          /* searchsorted(right=True) */
                const int below = ind - 1 > 0 ? ind - 1 : 0, above = ind < B ? ind : B;
                float den = cdf[above] - cdf[below];
                if (den < 1e-5f) den = 1.0f;
                t[D + e] = zmid[below] + (u - cdf[below]) / den * (zmid[above] - zmid[below]);
            }
            if (tap_depths_fine) memcpy(tap_depths_fine + ray * Di, t + D, sizeof(float) * Di);
            for (int e = 0; e < Di; ++e) {
                const int k = D + e;
                eval_point(ng, dg, H, W, d, scale, o[0] + t[k] * dr[0], o[1] + t[k] * dr[1], o[2] + t[k] * dr[2],
                           col + k * C, sg + k, sgm + k * NSEG);
            }
            /* unify_samples: sort by depth, gather (renderer.py:288-300) */
            int idx[MAXS];
            for (int i = 0; i < S; ++i) idx[i] = i;
            qsort_r(idx, S, sizeof(int), cmp_idx_depth, t);
            float t2[MAXS], sg2[MAXS];
            float* col2 = (float*)malloc(sizeof(float) * S * (C + NSEG));
            float* sgm2 = col2 + S * C;
            for (int i = 0; i < S; ++i) {
                t2[i] = t[idx[i]]; sg2[i] = sg[idx[i]];
                memcpy(col2 + i * C, col + idx[i] * C, sizeof(float) * C);
                memcpy(sgm2 + i * NSEG, sgm + idx[i] * NSEG, sizeof(float) * NSEG);
            }
            memcpy(t, t2, sizeof(float) * S); memcpy(sg, sg2, sizeof(float) * S);
            memcpy(col, col2, sizeof(float) * S * C); memcpy(sgm, sgm2, sizeof(float) * S * NSEG);
            free(col2);
            Sr = S;
        }
        march(t, col, sg, sgm, Sr, white_back, rgb + ray * C, seg + ray * NSEG, depth + ray, wsum + ray, NULL);
        for (int k = 0; k < Sr; ++k) { gmin = fminf(gmin, t[k]); gmax = fmaxf(gmax, t[k]); }
    }
    /* nan_to_num(depth, inf) then clamp to the whole-tensor [min,max] of depths (ray_marcher.py:93-94) */
    for (long long ray = 0; ray < (long long)N * M; ++ray) {
        float dd = depth[ray];
        if (dd != dd) dd = INFINITY;
        depth[ray] = fminf(fmaxf(dd, gmin), gmax);
    }
    free(ncl); free(dcl); free(d);
    return 0;
}

int nfe_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
