"""RaySampler with the reference's interface (training/volumetric_rendering/ray_sampler.py:18-62),
backed by nfe_ray_sampler.  Inside TriPlaneGenerator.synthesis the rays are generated in the render
kernel itself and this module is bypassed; it exists for callers that use it directly
(utils.py:171, projector.py:81)."""
import torch

from ... import ops


class RaySampler(torch.nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, cam2world_matrix, intrinsics, resolution):
        """cam2world_matrix (N,4,4), intrinsics (N,3,3), resolution int -> ray_origins, ray_dirs (N,M,3)."""
        return ops.ray_sampler(cam2world_matrix, intrinsics, resolution)
