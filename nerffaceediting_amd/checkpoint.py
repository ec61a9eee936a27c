"""Load generators converted by tools/convert_checkpoint.py (flat .npz state_dict + .json constructor arguments).
The reference reloads a pickle by re-instantiating the class from init_args/init_kwargs and copying parameters by
name (gen_samples.py:146-151, torch_utils/misc.py:157-164); this is the same, without unpickling code."""
import json

import numpy as np
import torch

from .training.triplane import TriPlaneGenerator


def load_generator(prefix, device="cuda"):
    with open(prefix + ".json") as f:
        meta = json.load(f)
    kwargs = dict(meta["init_kwargs"])
    G = TriPlaneGenerator(*meta["init_args"], **kwargs).eval().requires_grad_(False)
    with np.load(prefix + ".npz") as z:
        state = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    own = G.state_dict()
    for k, v in state.items():                      # buffers the training loop adds (training_loop.py:192)
        if k not in own and "." not in k:
            G.register_buffer(k, torch.zeros_like(v))
    missing, unexpected = G.load_state_dict(state, strict=False)
    if missing or unexpected:
        raise RuntimeError(f"checkpoint does not match the generator: missing {missing[:5]}, unexpected {unexpected[:5]}")
    G.rendering_kwargs = dict(meta["rendering_kwargs"]) or G.rendering_kwargs
    G.neural_rendering_resolution = int(meta["neural_rendering_resolution"])
    return G.to(device)
