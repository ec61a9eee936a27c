"""CPU oracle of TriPlaneGenerator.mapping/synthesis/sample (training/triplane.py:69-157), composed from
the dense oracle (torch-CPU) and the render oracle (numpy).  TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch

from . import dense_oracle as dor
from . import render_oracle as orc

BACKBONE_RES = [4, 8, 16, 32, 64, 128, 256]


def _sub(p, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


def decoder_np(p):
    return {k[len("decoder."):]: v.numpy() for k, v in p.items() if k.startswith("decoder.")}


def mapping(p, z, c, rendering_kwargs, truncation_psi=1.0, truncation_cutoff=None, num_ws=14, num_layers=2):
    """triplane.py:69-72."""
    if rendering_kwargs["c_gen_conditioning_zero"]:
        c = torch.zeros_like(c)
    return dor.mapping(_sub(p, "backbone.mapping."), z, c * rendering_kwargs.get("c_scale", 0), num_ws, num_layers,
                       truncation_psi, truncation_cutoff)


def synthesis(p, ws, c, rendering_kwargs, R, u_coarse, u_fine, planes_mean=None, planes_var=None, noise_mode="const"):
    """triplane.py:74-138 -> dict of numpy arrays with the reference's keys."""
    c = c.numpy()
    N = c.shape[0]
    cam2world, intrinsics = c[:, :16].reshape(N, 4, 4), c[:, 16:25].reshape(N, 3, 3)
    o, d = orc.ray_sampler(cam2world, intrinsics, R)
    planes = dor.synthesis_network(_sub(p, "backbone.synthesis."), ws, BACKBONE_RES, noise_mode=noise_mode).numpy()
    pm = planes_mean if (planes_mean is None or isinstance(planes_mean, int)) else np.asarray(planes_mean)
    pv = planes_var if (planes_var is None or isinstance(planes_var, int)) else np.asarray(planes_var)
    norm, denorm, mean, std = orc.synthesis_planes(planes, pm, pv)
    rgb, seg, depth, _ = orc.render_chunked(norm, denorm, decoder_np(p), o, d, rendering_kwargs, u_coarse, u_fine, chunk=2048)
    feat = torch.from_numpy(rgb.transpose(0, 2, 1).reshape(N, 32, R, R).copy())
    image = dor.superresolution_8xdc(_sub(p, "superresolution."), feat[:, :3].contiguous(), feat, ws,
                                     noise_mode=rendering_kwargs["superresolution_noise_mode"],
                                     sr_antialias=rendering_kwargs["sr_antialias"])
    return {"image": image.numpy(), "image_seg": seg.transpose(0, 2, 1).reshape(N, 15, R, R),
            "image_raw": feat[:, :3].numpy(), "image_depth": depth.transpose(0, 2, 1).reshape(N, 1, R, R),
            "plane_mean": mean, "plane_var": std, "planes": planes}


def sample_mixed(p, coordinates, ws, rendering_kwargs, noise_mode="const"):
    """triplane.py:150-157."""
    planes = dor.synthesis_network(_sub(p, "backbone.synthesis."), ws, BACKBONE_RES, noise_mode=noise_mode).numpy()
    return orc.point_query(planes, decoder_np(p), coordinates, rendering_kwargs)
