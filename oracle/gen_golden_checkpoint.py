"""Golden record of a checkpoint conversion (TEST INFRASTRUCTURE; build container only: needs /root/reference).

A reference TriPlaneGenerator (the reduced-width generator of tests/golden/dense_e2e.npz, parameters from
oracle.dense_params.generator_params(51)) is pickled by the reference's own persistence machinery, read back with the
reference's legacy.load_network_pkl, and converted by tools/convert_checkpoint.convert().  The converter's output is
9 MB of seeded random numbers, so the fixture keeps its .json verbatim plus the SHA-256 of every tensor it wrote
(tests/golden/checkpoint_e2e.json): the GPU test regenerates the tensors from the seed, proves them identical to what the
converter wrote by those hashes, writes them in the converter's format, and goes load_generator() -> synthesis() ->
reference outputs of dense_e2e.npz.  tests/test_checkpoint_cpu.py re-runs this whole flow where the reference is present.

    python oracle/gen_golden_checkpoint.py
"""
import hashlib
import json
import os
import pickle
import sys
import tempfile

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

SEED, CHANNEL_BASE, CHANNEL_MAX = 51, 4096, 32
RENDERING_KWARGS = dict(superresolution_module="training.superresolution.SuperresolutionHybrid8XDC", sr_antialias=True,
                        c_gen_conditioning_zero=False, c_scale=1, superresolution_noise_mode="none", depth_resolution=12,
                        depth_resolution_importance=12, ray_start=2.25, ray_end=3.3, box_warp=1,
                        disparity_space_sampling=False, clamp_mode="softplus", decoder_lr_mul=1,
                        avg_camera_radius=2.7, avg_camera_pivot=[0, 0, 0.2])


def tensor_digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes()).hexdigest()


def converted_record(workdir):
    """reference generator -> pickle -> legacy.load_network_pkl -> convert(): returns (meta json dict, {name: sha256})."""
    import convert_checkpoint as cc
    import legacy
    from training.triplane import TriPlaneGenerator as RefG
    from oracle.dense_params import generator_params
    G = RefG(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2),
             rendering_kwargs=dict(RENDERING_KWARGS), sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default="inference_only"),
             channel_base=CHANNEL_BASE, channel_max=CHANNEL_MAX, fused_modconv_default="inference_only", num_fp16_res=0,
             conv_clamp=None).eval().requires_grad_(False)
    sd = G.state_dict()
    for k, v in generator_params(SEED, CHANNEL_BASE, CHANNEL_MAX).items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        sd[k] = v.clone()
    G.load_state_dict(sd)
    G.neural_rendering_resolution = 32
    pkl = os.path.join(workdir, "network-snapshot.pkl")
    with open(pkl, "wb") as f:                       # what training_loop.py:384-395 writes
        pickle.dump(dict(G=G, D=torch.nn.Linear(1, 1), G_ema=G, training_set_kwargs=None, augment_pipe=None), f)
    with open(pkl, "rb") as f:
        data = legacy.load_network_pkl(f)
    prefix = os.path.join(workdir, "converted")
    meta = cc.convert(data["G_ema"], prefix)
    with np.load(prefix + ".npz") as z:
        digests = {k: tensor_digest(z[k]) for k in z.files}
    with open(prefix + ".json") as f:
        meta_json = json.load(f)
    assert meta_json == json.loads(json.dumps(meta))
    return meta_json, digests


def main():
    with tempfile.TemporaryDirectory() as td:
        meta, digests = converted_record(td)
    out = os.path.join(ROOT, "tests", "golden", "checkpoint_e2e.json")
    with open(out, "w") as f:
        json.dump({"converter_json": meta, "sha256": digests, "seed": SEED, "channel_base": CHANNEL_BASE, "channel_max": CHANNEL_MAX,
                   "torch_version": torch.__version__}, f, indent=1, sort_keys=True)
    print(f"wrote {out}: {len(digests)} tensors")


if __name__ == "__main__":
    main()
