"""TEST INFRASTRUCTURE — analytic CPU restatement of the BACKWARD of the renderer's final compositing pass.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; it is the checker of
the HIP path (nfe_render_backward), never the thing measured or shipped.

What it restates: the vector-Jacobian product torch autograd computes for
DisentangledImportanceRenderer.forward (training/volumetric_rendering/renderer.py:301-363) with respect to
`norm_planes` and `denorm_planes`, i.e. through
    SegMipRayMarcher2.run_forward          ray_marcher.py:68-101
    DisentangledOSGDecoder.forward         triplane.py:249-270   (FullyConnectedLayer networks_stylegan2.py:114-127)
    sample_from_planes / F.grid_sample     renderer.py:55-65
The sample depths are constants of this computation: the stratified depths do not depend on the planes and the
importance depths are produced under torch.no_grad() and .detach()ed (renderer.py:198,211), so the gradient flows
only through the march over the final (sorted) depths — which is why the function takes `depths_all`.

Pinned by tests/golden/backward_*.npz: gradients produced by the reference itself under torch autograd
(oracle/gen_golden_backward.py).  The arithmetic here is float64 (it is a derivative checker, not a bit model).
"""
from __future__ import annotations

import numpy as np

F64 = np.float64


def _taps(H, W, gx, gy):
    """Bilinear taps of F.grid_sample(bilinear, zeros, align_corners=False): flat indices [4,P] and weights [4,P]
    (weight 0 for taps outside the plane)."""
    ix = (gx + 1.0) * (W / 2.0) - 0.5
    iy = (gy + 1.0) * (H / 2.0) - 0.5
    x0, y0 = np.floor(ix), np.floor(iy)
    dx, dy = ix - x0, iy - y0
    x0 = np.clip(x0, -2, W + 1).astype(np.int64)
    y0 = np.clip(y0, -2, H + 1).astype(np.int64)
    idx, wgt = [], []
    for (xx, yy, ww) in ((x0, y0, (1 - dx) * (1 - dy)), (x0 + 1, y0, dx * (1 - dy)),
                         (x0, y0 + 1, (1 - dx) * dy), (x0 + 1, y0 + 1, dx * dy)):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        idx.append(np.where(ok, yy * W + xx, 0))
        wgt.append(np.where(ok, ww, 0.0))
    return np.stack(idx), np.stack(wgt)


def _project(coords):
    """renderer.py:39-53 with the axes of generate_planes (:23-37): p0=(x,y), p1=(x,z), p2=(z,x)."""
    x, y, z = coords[:, 0], coords[:, 1], coords[:, 2]
    return ((x, y), (x, z), (z, x))


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _softplus(x):
    return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0))))


def _softplus_grad(x):
    return np.where(x > 20.0, 1.0, _sigmoid(x))


def _fc(dec, name, lr_mul):
    w = dec[name + ".weight"].astype(F64)
    return w * (lr_mul / np.sqrt(w.shape[1])), dec[name + ".bias"].astype(F64) * lr_mul


def render_backward(norm_planes, denorm_planes, dec, origins, dirs, depths_all, options, g_rgb, g_seg, g_depth, g_wsum, sigma_offset=None):
    """norm_planes / denorm_planes [N,3,32,H,W]; origins, dirs [N,M,3]; depths_all [N,M,S] (the sorted depths the
    forward marched); cotangents g_rgb [N,M,32], g_seg [N,M,15], g_depth [N,M,1], g_wsum [N,M,1].
    sigma_offset [N,M,S] (round 6): what renderer.py:285-286 added to sigma (randn_like * density_noise), per MERGED sample - a constant
    of the backward, it only moves the point where the march's softplus is differentiated.
    `dec` with the keys of SegmentationOSGDecoder (net.*, seg_net.*; triplane.py:192-230): sigma and rgb from `net`, seg from `seg_net`,
    both on the DENORM features - the norm planes get a zero gradient.
    Returns (grad_norm_planes, grad_denorm_planes), both [N,3,32,H,W] float64."""
    N, _, C, H, W = norm_planes.shape
    M, S = depths_all.shape[1], depths_all.shape[2]
    lr = float(options.get("decoder_lr_mul", 1))
    wb = bool(options.get("white_back", False))
    segosg = "net.0.weight" in dec
    if segosg:
        nw0, nb0 = _fc(dec, "net.0", lr); nw1, nb1 = _fc(dec, "net.2", lr)
        sw0, sb0 = _fc(dec, "seg_net.0", lr); sw1, sb1 = _fc(dec, "seg_net.2", lr)
    else:
        gw0, gb0 = _fc(dec, "geo_net.0", lr); gw1, gb1 = _fc(dec, "geo_net.2", lr)
        aw0, ab0 = _fc(dec, "app_net.0", lr); aw1, ab1 = _fc(dec, "app_net.2", lr)
    scale = 2.0 / float(options["box_warp"])
    grads = [np.zeros((N, 3, H * W, C), F64), np.zeros((N, 3, H * W, C), F64)]
    for n in range(N):
        t = depths_all[n].astype(F64)                                              # [M,S]
        pts = (origins[n].astype(F64)[:, None, :] + t[:, :, None] * dirs[n].astype(F64)[:, None, :]).reshape(-1, 3) * scale
        taps = [_taps(H, W, u, v) for (u, v) in _project(pts)]
        feats = []
        for planes in (norm_planes, denorm_planes):
            f = np.zeros((M * S, C), F64)
            for p in range(3):
                flat = planes[n, p].astype(F64).reshape(C, H * W).T
                idx, wgt = taps[p]
                for k in range(4):
                    f += flat[idx[k]] * wgt[k][:, None]
            feats.append(f / 3.0)                                                  # mean over planes, triplane.py:251-252
        fn, fd = feats
        if segosg:
            pre_n = fd @ nw0.T + nb0; out_n = _softplus(pre_n) @ nw1.T + nb1          # sigma = ch 0, rgb = ch 1..32
            pre_s = fd @ sw0.T + sb0; out_s = _softplus(pre_s) @ sw1.T + sb1          # seg
            sg = _sigmoid(out_n[:, 1:])
            sigma = out_n[:, 0].reshape(M, S); seg = out_s.reshape(M, S, 15)
        else:
            pre_g = fn @ gw0.T + gb0; out_g = _softplus(pre_g) @ gw1.T + gb1          # sigma = ch 0, seg = ch 1..15
            pre_a = fd @ aw0.T + ab0; y = _softplus(pre_a) @ aw1.T + ab1
            sg = _sigmoid(y)
            sigma = out_g[:, 0].reshape(M, S); seg = out_g[:, 1:].reshape(M, S, 15)
        rgb = (sg * 1.002 - 0.001).reshape(M, S, 32)
        if sigma_offset is not None:
            sigma = sigma + sigma_offset[n].astype(F64).reshape(M, S)
        # ---- forward march (ray_marcher.py:68-101) ----------------------------------------------------------------
        delta = t[:, 1:] - t[:, :-1]
        smid = 0.5 * (sigma[:, :-1] + sigma[:, 1:]) - 1.0
        dens = _softplus(smid)
        alpha = 1.0 - np.exp(-dens * delta)
        om = 1.0 - alpha + 1e-10
        T = np.cumprod(np.concatenate([np.ones((M, 1)), om], 1), 1)[:, :-1]
        w = alpha * T
        wtot = w.sum(1)
        tmid = 0.5 * (t[:, :-1] + t[:, 1:])
        with np.errstate(divide="ignore", invalid="ignore"):
            d0 = (w * tmid).sum(1) / wtot
        ok = np.isfinite(d0) & (wtot != 0)          # nan_to_num + clamp pass no gradient where the ratio is not finite
        # ---- cotangent of the weights -----------------------------------------------------------------------------
        Gr = 2.0 * g_rgb[n].astype(F64)                                            # rgb*2-1, :99
        Gs = g_seg[n].astype(F64)
        a = (rgb * Gr[:, None, :]).sum(-1) + (seg * Gs[:, None, :]).sum(-1)       # [M,S]
        gwj = 0.5 * (a[:, :-1] + a[:, 1:]) + g_wsum[n].astype(F64).reshape(M, 1)
        with np.errstate(divide="ignore", invalid="ignore"):
            gd = np.where(ok, g_depth[n].astype(F64).reshape(M) / wtot, 0.0)
        gwj = gwj + gd[:, None] * np.where(ok[:, None], tmid - np.where(ok, d0, 0.0)[:, None], 0.0)
        if wb:
            gwj = gwj - Gr.sum(-1)[:, None]                                        # rgb + 1 - weight_total, :96-97
        # ---- w_j = alpha_j T_j, T_{j+1} = T_j (1 - alpha_j + 1e-10): reverse recurrence without divisions ------------
        Rj = np.zeros(M)
        galpha = np.zeros_like(alpha)
        for j in range(S - 2, -1, -1):
            galpha[:, j] = T[:, j] * (gwj[:, j] - Rj)
            Rj = gwj[:, j] * alpha[:, j] + om[:, j] * Rj
        gsm = galpha * delta * np.exp(-dens * delta) * _softplus_grad(smid)       # d alpha / d sigma_mid
        gsig = np.zeros((M, S)); gsig[:, :-1] += 0.5 * gsm; gsig[:, 1:] += 0.5 * gsm
        omega = np.zeros((M, S)); omega[:, :-1] += 0.5 * w; omega[:, 1:] += 0.5 * w
        # ---- decoder backward ---------------------------------------------------------------------------------------
        dseg = (omega[:, :, None] * Gs[:, None, :]).reshape(-1, 15)
        dy = (omega[:, :, None] * Gr[:, None, :]).reshape(-1, 32) * 1.002 * sg * (1.0 - sg)
        if segosg:
            dout_n = np.concatenate([gsig.reshape(-1, 1), dy], 1)
            dfd = ((dout_n @ nw1) * _softplus_grad(pre_n)) @ nw0 + ((dseg @ sw1) * _softplus_grad(pre_s)) @ sw0
            dfn = np.zeros_like(dfd)
        else:
            dout_g = np.concatenate([gsig.reshape(-1, 1), dseg], 1)
            dfn = ((dout_g @ gw1) * _softplus_grad(pre_g)) @ gw0
            dfd = ((dy @ aw1) * _softplus_grad(pre_a)) @ aw0
        # ---- scatter (grid_sample backward wrt input) ----------------------------------------------------------------
        for gi, df in enumerate((dfn, dfd)):
            for p in range(3):
                idx, wgt = taps[p]
                for k in range(4):
                    np.add.at(grads[gi][n, p], idx[k], df * (wgt[k][:, None] / 3.0))
    to5 = lambda g: g.reshape(N, 3, H, W, C).transpose(0, 1, 4, 2, 3)
    return to5(grads[0]), to5(grads[1])
