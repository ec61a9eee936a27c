"""Checkpoint converter (tools/convert_checkpoint.py) + loader (nerffaceediting_amd/checkpoint.py): a pickle written
by the reference's own persistence machinery round-trips into this package's generator.  Needs the reference
tree, so it only runs in the build container (never on the GPU box)."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")

RK = dict(superresolution_module="training.superresolution.SuperresolutionHybrid8XDC", sr_antialias=True, c_gen_conditioning_zero=False,
          c_scale=1, superresolution_noise_mode="none", depth_resolution=12, depth_resolution_importance=12, ray_start=2.25, ray_end=3.3,
          box_warp=1, disparity_space_sampling=False, clamp_mode="softplus", decoder_lr_mul=1, avg_camera_radius=2.7, avg_camera_pivot=[0, 0, 0.2])


def _ref_generator():
    sys.path.insert(0, REF)
    from training.triplane import TriPlaneGenerator as RefG
    torch.manual_seed(3)
    return RefG(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2),
                rendering_kwargs=dict(RK), sr_kwargs=dict(channel_base=2048, channel_max=16, fused_modconv_default="inference_only"),
                channel_base=2048, channel_max=16, fused_modconv_default="inference_only", num_fp16_res=0, conv_clamp=None).eval().requires_grad_(False)


def test_reference_pickle_roundtrip(tmp_path):
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import convert_checkpoint as cc
    from nerffaceediting_amd.checkpoint import load_generator
    G = _ref_generator()
    G.register_buffer("dataset_label_std", torch.arange(25, dtype=torch.float32))     # training_loop.py:192
    G.neural_rendering_resolution = 128
    pkl = tmp_path / "net.pkl"
    with open(pkl, "wb") as f:
        pickle.dump(dict(G=G, D=torch.nn.Linear(1, 1), G_ema=G, training_set_kwargs=None, augment_pipe=None), f)
    import legacy
    with open(pkl, "rb") as f:
        data = legacy.load_network_pkl(f)
    meta = cc.convert(data["G_ema"], str(tmp_path / "ffhq"))
    assert meta["class"] == "TriPlaneGenerator" and not meta["converted_from_single_decoder"]
    mine = load_generator(str(tmp_path / "ffhq"), device="cpu")
    ref_sd, my_sd = G.state_dict(), mine.state_dict()
    assert set(ref_sd) == set(my_sd)
    for k in ref_sd:
        assert torch.equal(ref_sd[k].float(), my_sd[k].float()), k
    assert mine.neural_rendering_resolution == 128 and mine.rendering_kwargs["ray_end"] == 3.3
    assert mine.init_kwargs["channel_max"] == 16


def test_single_decoder_split():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import convert_checkpoint as cc
    rng = np.random.RandomState(0)
    state = {"decoder.net.0.weight": rng.randn(64, 32).astype(np.float32), "decoder.net.0.bias": rng.randn(64).astype(np.float32),
             "decoder.net.2.weight": rng.randn(33, 64).astype(np.float32), "decoder.net.2.bias": rng.randn(33).astype(np.float32),
             "backbone.x": np.zeros(3, np.float32)}
    out = cc.split_eg3d_decoder(state)
    assert out["decoder.geo_net.2.weight"].shape == (16, 64) and out["decoder.app_net.2.weight"].shape == (32, 64)
    assert np.array_equal(out["decoder.geo_net.2.weight"][0], state["decoder.net.2.weight"][0])      # training_loop.py:205
    assert np.array_equal(out["decoder.app_net.2.bias"], state["decoder.net.2.bias"][1:])              # :211
    assert np.array_equal(out["decoder.app_net.0.weight"], state["decoder.net.0.weight"]) and "decoder.net.0.weight" not in out


def test_vis_parsing_maps_matches_reference_formula():
    from nerffaceediting_amd import utils as U
    torch.manual_seed(0)
    seg = torch.randn(2, 15, 9, 7)
    img = U.vis_parsing_maps(seg)
    lab = torch.argmax(seg, 1, keepdim=True)
    want = torch.zeros(2, 3, 9, 7)
    for i, col in enumerate(U.PART_COLORS):                        # utils.py:113-116
        want = torch.where(lab == i, torch.tensor(col, dtype=torch.float32).view(1, 3, 1, 1).expand_as(want), want)
    assert torch.equal(img, want / 255.0 * 2 - 1)
    assert torch.equal(U.vis_parsing_maps(img, inverse=True), lab)


def test_checkpoint_golden_record_is_current(tmp_path):
    """tests/golden/checkpoint_e2e.json (what the GPU test trusts) == a fresh pickle -> legacy.load_network_pkl -> convert()
    run on the reference class here."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import gen_golden_checkpoint as gg
    meta, digests = gg.converted_record(str(tmp_path))
    with open(os.path.join(os.path.dirname(__file__), "golden", "checkpoint_e2e.json")) as f:
        rec = json.load(f)
    assert rec["converter_json"] == meta
    assert rec["sha256"] == digests and len(digests) == 180
