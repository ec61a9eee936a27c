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
