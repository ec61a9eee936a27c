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


def test_camera_samples_and_video_schedule_match_reference():
    """utils.get_camera_samples (utils.py:130-144) and the cameras utils.render_video visits (:45-80), against captures of the
    reference's own functions (oracle/gen_golden.py:gen_camera_samples)."""
    import types
    from tests._golden import load
    from nerffaceediting_amd import utils
    from nerffaceediting_amd.camera_utils import FOV_to_intrinsics, LookAtPoseSampler
    z = load("camera_samples")
    G = types.SimpleNamespace(rendering_kwargs={"avg_camera_pivot": [0, 0, 0.2], "avg_camera_radius": 2.7})
    cams = utils.get_camera_samples(G, torch.device("cpu"))
    assert len(cams) == 9 and all(c.shape == (1, 25) for c in cams)
    assert np.allclose(torch.cat(cams, 0).numpy(), z["cams_pivot02"], atol=1e-6)
    G0 = types.SimpleNamespace(rendering_kwargs={})
    assert np.allclose(torch.cat(utils.get_camera_samples(G0, torch.device("cpu")), 0).numpy(), z["cams_default"], atol=1e-6)

    def cams_of(G_, sched):
        intr = FOV_to_intrinsics(18.837)
        pivot = torch.tensor(G_.rendering_kwargs.get("avg_camera_pivot", [0, 0, 0]), dtype=torch.float32)
        radius = G_.rendering_kwargs.get("avg_camera_radius", 2.7)
        return torch.cat([torch.cat([LookAtPoseSampler.sample(p, y, pivot, radius=radius).reshape(-1, 16), intr.reshape(-1, 9)], 1)
                          for p, y in sched], 0).numpy()
    s12 = utils.video_camera_schedule(12, 15.0, 12.0)
    assert len(s12) == 12 and np.allclose(cams_of(G, s12), z["video_cams_12"], atol=1e-6)      # default start: no interpolation leg
    s9 = utils.video_camera_schedule(9, 10.0, 20.0, init_pitch=1.2, init_yaw=1.7)
    assert len(s9) == 9 + 9 // 4 and np.allclose(cams_of(G0, s9), z["video_cams_9_interp"], atol=1e-6)


def test_sr_gradient_support_table():
    """decode()['image'] carries plane gradients for the reference's head classes at any neural rendering resolution (sr_grad.py);
    an unknown head class falls back to the node that raises in backward (utils._NotDifferentiableImage)."""
    from nerffaceediting_amd import sr_grad

    class SuperresolutionHybrid8XDC:          # only the class name and the resolution decide
        input_resolution = 128

    class SuperresolutionHybrid4X:
        input_resolution = 128

    class SomeOtherHead:
        input_resolution = 128
    assert sr_grad.supported(SuperresolutionHybrid8XDC(), 128) and sr_grad.supported(SuperresolutionHybrid8XDC(), 64)
    assert sr_grad.supported(SuperresolutionHybrid8XDC(), 512) and not sr_grad.supported(SuperresolutionHybrid8XDC(), 0)
    assert sr_grad.supported(SuperresolutionHybrid4X(), 128) and not sr_grad.supported(SomeOtherHead(), 128)


def test_render_tensor_follows_the_reference_conventions():
    """utils.render_tensor (utils.py:11-30): [-1,1] -> uint8 by x/2 + .5, * 255, truncating cast; one image -> its own size, a
    batch -> torchvision's make_grid layout (2 px of zero padding around every image, `nrow` images per row), a list of [1,C,H,W]
    tensors is concatenated, one-channel inputs are broadcast to RGB.  torchvision is not installed here: the grid is checked
    against make_grid's documented geometry (known-answer)."""
    from nerffaceediting_amd import utils
    one = torch.linspace(-1, 1, 3 * 4 * 5).view(1, 3, 4, 5)
    im = utils.render_tensor(one)
    assert im.size == (5, 4) and im.mode == "RGB"
    want = ((one[0] / 2 + .5).permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    assert np.array_equal(np.asarray(im), want)
    assert np.asarray(im)[0, 0, 0] == 0 and np.asarray(im)[-1, -1, -1] == 255
    gray = utils.render_tensor(torch.zeros(1, 1, 4, 5))                           # one channel -> RGB, 0 -> 127 (truncation of 127.5)
    assert gray.mode == "RGB" and np.asarray(gray).min() == 127 == np.asarray(gray).max()
    flat = utils.render_tensor(torch.full((4, 5), 0.25), normalize=False)         # a 2-D tensor stays one-channel ('L')
    assert flat.mode == "L" and flat.size == (5, 4) and int(np.asarray(flat)[0, 0]) == 63
    batch = [torch.full((1, 3, 4, 5), v) for v in np.linspace(-1, 1, 10)]
    grid = np.asarray(utils.render_tensor(batch, nrow=8))
    assert grid.shape == ((4 + 2) * 2 + 2, (5 + 2) * 8 + 2, 3)
    assert grid[:2].max() == 0 and grid[:, :2].max() == 0                           # padding is pad_value 0 -> 0
    for k in range(10):
        y, x = divmod(k, 8)
        tile = grid[2 + y * 6:2 + y * 6 + 4, 2 + x * 7:2 + x * 7 + 5]
        assert tile.min() == tile.max() == int((np.float32(np.linspace(-1, 1, 10)[k]) / 2 + .5) * 255), k
    assert grid[2 + 6:2 + 6 + 4, 2 + 2 * 7:].max() == 0                             # the unused cells of the last row stay empty
