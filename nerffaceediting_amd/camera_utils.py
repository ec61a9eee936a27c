"""Camera pose helpers with the reference's names (camera_utils.py:22-149).  These build the 25-float
camera labels that feed the path; a handful of 3-vector operations per view, done with torch on
whatever device the caller asks for."""
import math

import torch


def _normalize_vecs(v):
    return v / torch.norm(v, dim=-1, keepdim=True)                       # math_utils.py:33-37


def create_cam2world_matrix(forward_vector, origin):
    """camera_utils.py:118-137: y-up, no roll."""
    forward_vector = _normalize_vecs(forward_vector)
    up = torch.tensor([0, 1, 0], dtype=torch.float, device=origin.device).expand_as(forward_vector)
    right = -_normalize_vecs(torch.cross(up, forward_vector, dim=-1))
    up = _normalize_vecs(torch.cross(forward_vector, right, dim=-1))
    rot = torch.eye(4, device=origin.device).unsqueeze(0).repeat(forward_vector.shape[0], 1, 1)
    rot[:, :3, :3] = torch.stack((right, up, forward_vector), axis=-1)
    trans = torch.eye(4, device=origin.device).unsqueeze(0).repeat(forward_vector.shape[0], 1, 1)
    trans[:, :3, 3] = origin
    return trans @ rot


def _origins(h, v, radius):
    v = torch.clamp(v, 1e-5, math.pi - 1e-5)
    phi = torch.arccos(1 - 2 * (v / math.pi))
    o = torch.zeros((h.shape[0], 3), device=h.device)
    o[:, 0:1] = radius * torch.sin(phi) * torch.cos(math.pi - h)
    o[:, 2:3] = radius * torch.sin(phi) * torch.sin(math.pi - h)
    o[:, 1:2] = radius * torch.cos(phi)
    return o


class GaussianCameraPoseSampler:
    """camera_utils.py:22-56."""
    @staticmethod
    def sample(horizontal_mean, vertical_mean, horizontal_stddev=0, vertical_stddev=0, radius=1, batch_size=1, device="cpu"):
        h = torch.randn((batch_size, 1), device=device) * horizontal_stddev + horizontal_mean
        v = torch.randn((batch_size, 1), device=device) * vertical_stddev + vertical_mean
        o = _origins(h, v, radius)
        return create_cam2world_matrix(_normalize_vecs(-o), o)


class LookAtPoseSampler:
    """camera_utils.py:59-86."""
    @staticmethod
    def sample(horizontal_mean, vertical_mean, lookat_position, horizontal_stddev=0, vertical_stddev=0, radius=1,
               batch_size=1, device="cpu"):
        h = torch.randn((batch_size, 1), device=device) * horizontal_stddev + horizontal_mean
        v = torch.randn((batch_size, 1), device=device) * vertical_stddev + vertical_mean
        o = _origins(h, v, radius)
        return create_cam2world_matrix(_normalize_vecs(lookat_position.to(o.device) - o), o)


class UniformCameraPoseSampler:
    """camera_utils.py:88-116."""
    @staticmethod
    def sample(horizontal_mean, vertical_mean, horizontal_stddev=0, vertical_stddev=0, radius=1, batch_size=1, device="cpu"):
        h = (torch.rand((batch_size, 1), device=device) * 2 - 1) * horizontal_stddev + horizontal_mean
        v = (torch.rand((batch_size, 1), device=device) * 2 - 1) * vertical_stddev + vertical_mean
        o = _origins(h, v, radius)
        return create_cam2world_matrix(_normalize_vecs(-o), o)


def FOV_to_intrinsics(fov_degrees, device="cpu"):
    """camera_utils.py:140-149 (keeps the reference's truncated constants)."""
    focal_length = float(1 / (math.tan(fov_degrees * 3.14159 / 360) * 1.414))
    return torch.tensor([[focal_length, 0, 0.5], [0, focal_length, 0.5], [0, 0, 1]], device=device)
