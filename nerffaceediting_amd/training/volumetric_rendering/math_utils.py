"""get_ray_limits_box with the reference's signature (training/volumetric_rendering/math_utils.py:46-98).
The fused entry point also applies the invalid-ray fix-up the renderer does right after it
(renderer.py:313-317); `raw=True` is not offered because no caller of the reference uses the raw values."""
from ... import ops


def get_ray_limits_box(rays_o, rays_d, box_side_length):
    shape = rays_o.shape
    rs, re = ops.ray_limits_box(rays_o.reshape(1, -1, 3), rays_d.reshape(1, -1, 3), box_side_length)
    return rs.reshape(*shape[:-1], 1), re.reshape(*shape[:-1], 1)
