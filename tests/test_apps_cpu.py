"""CPU: host-side helpers of the batch drivers (nerffaceediting_amd/apps.py) that need no GPU."""
import numpy as np
import scipy.interpolate
import torch

from nerffaceediting_amd import apps


def test_interpolate_ws_is_the_reference_construction():
    """gen_videos.py:103-113,135-136: interp1d(kind='cubic') over the keyframes tiled 2*wraps+1 times, evaluated at f / w_frames."""
    rng = np.random.RandomState(0)
    K, w_frames, wraps = 3, 5, 2
    key = rng.randn(K, 14, 8).astype(np.float32)
    got = apps.interpolate_ws(torch.from_numpy(key), w_frames=w_frames, wraps=wraps).numpy()
    x = np.arange(-K * wraps, K * (wraps + 1))
    ref = scipy.interpolate.interp1d(x, np.tile(key, [wraps * 2 + 1, 1, 1]), kind="cubic", axis=0)
    want = np.stack([ref(f / w_frames) for f in range(K * w_frames)])
    assert got.shape == (K * w_frames, 14, 8)
    assert np.allclose(got, want, atol=1e-6)
    assert np.allclose(got[::w_frames], key, atol=1e-5)                  # the curve passes through the keyframes
    lin = apps.interpolate_ws(torch.from_numpy(key), w_frames=4, kind="linear").numpy()
    assert np.allclose(lin[2], 0.5 * (key[0] + key[1]), atol=1e-6)


def test_seed_to_z_and_uint8_conversion():
    z = apps.seed_to_z(7, 512, device="cpu")
    assert z.shape == (1, 512) and np.allclose(z.numpy(), np.random.RandomState(7).randn(1, 512).astype(np.float32))
    img = torch.tensor([-1.2, -1.0, 0.0, 1.0, 1.3]).reshape(1, 1, 1, 5).expand(1, 3, 1, 5)
    u8 = apps.to_uint8(img)
    assert u8.dtype == torch.uint8 and u8.shape == (1, 1, 5, 3)
    assert u8[0, 0, :, 0].tolist() == [0, 0, 128, 255, 255]               # gen_samples.py:177


def test_create_samples_and_volume_match_reference():
    """apps.create_samples against the reference's own create_samples (gen_samples.py:79-101, incl. its float-division
    quirk on the x / y columns), and the flip + border trim of gen_samples.py:204-216."""
    from tests._golden import load
    z = load("density_grid")
    R = int(z["shape_res"])
    pts, origin, voxel = apps.create_samples(N=R, voxel_origin=[0, 0, 0], cube_length=1.0)
    assert pts.shape == (1, R ** 3, 3) and np.array_equal(pts.numpy(), z["samples"])
    assert np.allclose(origin, z["voxel_origin"]) and abs(voxel - float(z["voxel_size"])) < 1e-12
    # z is the fastest axis and the only one on the lattice; x / y advance fractionally (the reference's float division)
    assert abs(float(pts[0, 1, 2] - pts[0, 0, 2]) - voxel) < 1e-6 and 0 < float(pts[0, 1, 1] - pts[0, 0, 1]) < voxel
    vol = apps.density_to_volume(torch.from_numpy(z["sigma_grid"]))
    assert np.array_equal(vol.numpy(), z["sigma_volume"])
