#!/usr/bin/env python3
"""Offline converter: a reference network pickle -> flat weights + constructor arguments for this package.

The reference's pickles embed module source and execute it on load (torch_utils/persistence.py:181-229,
legacy.py:24-60), so this runs where the reference tree is importable, once per checkpoint:

    python tools/convert_checkpoint.py --reference /path/to/NeRFFaceEditing --pkl network-snapshot.pkl --out ckpt/ffhq

writes  ckpt/ffhq.npz   state_dict of the chosen network (SURVEY.md App. B names, fp32 numpy arrays)
        ckpt/ffhq.json  init_args / init_kwargs (incl. rendering_kwargs) / neural_rendering_resolution
which nerffaceediting_amd.checkpoint.load_generator() reads.  EG3D pickles (single OSGDecoder with 1+32 outputs)
get the reference's own decoder split (training/training_loop.py:202-214): geometry head = layer 0 + output row 0,
appearance head = layer 0 + output rows 1..32; the seg rows of the geometry head stay at their init value (zeros here).
"""
import argparse
import json
import os
import sys

import numpy as np


def to_plain(obj):
    if isinstance(obj, dict):
        return {str(k): to_plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_plain(v) for v in obj]
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    return obj


def split_eg3d_decoder(state):
    """decoder.net.{0,2}.* (OSGDecoder, triplane.py:167) -> decoder.geo_net / decoder.app_net (training_loop.py:202-214)."""
    out = {k: v for k, v in state.items() if not k.startswith("decoder.net.")}
    w0, b0 = state["decoder.net.0.weight"], state["decoder.net.0.bias"]
    w2, b2 = state["decoder.net.2.weight"], state["decoder.net.2.bias"]
    for head in ("geo_net", "app_net"):
        out[f"decoder.{head}.0.weight"], out[f"decoder.{head}.0.bias"] = w0.copy(), b0.copy()
    gw = np.zeros((16, w2.shape[1]), np.float32); gb = np.zeros((16,), np.float32)
    gw[:1], gb[:1] = w2[:1], b2[:1]
    out["decoder.geo_net.2.weight"], out["decoder.geo_net.2.bias"] = gw, gb
    out["decoder.app_net.2.weight"], out["decoder.app_net.2.bias"] = w2[1:].copy(), b2[1:].copy()
    return out


def convert(G, out_prefix):
    state = {k: v.detach().cpu().numpy().astype(np.float32) if v.dtype.is_floating_point else v.detach().cpu().numpy()
             for k, v in G.state_dict().items()}
    eg3d = "decoder.net.0.weight" in state and "decoder.geo_net.0.weight" not in state
    if eg3d:
        state = split_eg3d_decoder(state)
    meta = {"class": type(G).__name__, "init_args": to_plain(list(getattr(G, "init_args", []))),
            "init_kwargs": to_plain(dict(getattr(G, "init_kwargs", {}))),
            "rendering_kwargs": to_plain(dict(getattr(G, "rendering_kwargs", {}))),
            "neural_rendering_resolution": int(getattr(G, "neural_rendering_resolution", 64)),
            "converted_from_single_decoder": bool(eg3d), "num_tensors": len(state)}
    os.makedirs(os.path.dirname(os.path.abspath(out_prefix)), exist_ok=True)
    np.savez(out_prefix + ".npz", **state)
    with open(out_prefix + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    return meta


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="root of the reference source tree")
    ap.add_argument("--pkl", required=True)
    ap.add_argument("--out", required=True, help="output prefix (writes <out>.npz and <out>.json)")
    ap.add_argument("--which", default="G_ema", choices=["G", "G_ema"])
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    import legacy                                   # noqa: E402  (reference module)
    with open(args.pkl, "rb") as f:
        data = legacy.load_network_pkl(f)
    meta = convert(data[args.which], args.out)
    print(f"wrote {args.out}.npz / .json: {meta['num_tensors']} tensors, class {meta['class']}")


if __name__ == "__main__":
    main()
