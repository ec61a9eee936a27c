"""Loader for the committed golden fixtures (made by oracle/gen_golden.py from the reference)."""
import ast
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RENDER_CASES = sorted(os.path.basename(p)[len("render_"):-len(".npz")]
                      for p in glob.glob(os.path.join(GOLDEN, "render_*.npz")))


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def load_render_case(tag):
    z = load("render_" + tag)
    case = dict(
        planes=z["planes"], cam2world=z["cam2world"], intrinsics=z["intrinsics"], R=int(z["R"]),
        swap=bool(int(z["swap"])), u_coarse=z["u_coarse"], u_fine=z["u_fine"],
        options=ast.literal_eval(str(z["options"])),
        dec={k[4:]: z[k] for k in z.files if k.startswith("dec.")},
        out={k[4:]: z[k] for k in z.files if k.startswith("out.")},
        tap={k[4:]: z[k] for k in z.files if k.startswith("tap.")},
    )
    case["options"]["white_back"] = bool(case["options"].get("white_back", 0))
    case["options"]["disparity_space_sampling"] = bool(case["options"].get("disparity_space_sampling", 0))
    return case


def max_abs(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) if a.size else 0.0
