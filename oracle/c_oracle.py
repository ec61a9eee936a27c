"""ctypes wrapper of oracle/librender_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librender_oracle.so")
_lib = None
F = ctypes.POINTER(ctypes.c_float)
_NAMES = ["geo_net.0.weight", "geo_net.0.bias", "geo_net.2.weight", "geo_net.2.bias",
          "app_net.0.weight", "app_net.0.bias", "app_net.2.weight", "app_net.2.bias"]


def build(native=False):
    global _lib
    cmd = ["make", "-C", _HERE, "-B"] + (["NATIVE=1"] if native else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    _lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.nfe_oracle_render.restype = ctypes.c_int
        _lib.nfe_oracle_max_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(F)


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def max_threads():
    return load().nfe_oracle_max_threads()


def render(norm_planes, denorm_planes, dec, origins, dirs, options, u_coarse, u_fine=None, ray_limits=None,
           taps=False, threads=None):
    """Same contract as oracle.render_oracle.render (scalar or per-ray limits; not the 'auto' keyword).
    threads: None = min(host cores, 16) (checker use); 0 = every core OpenMP offers (bench cpu_baseline)."""
    lib = load()
    if threads is None:
        threads = min(os.cpu_count() or 1, 16)
    norm_planes, denorm_planes, origins, dirs = map(_f, (norm_planes, denorm_planes, origins, dirs))
    Np, _, _, H, W = norm_planes.shape
    N, M, _ = origins.shape
    D = int(options["depth_resolution"])
    Di = int(options.get("depth_resolution_importance", 0) or 0)
    u_coarse = _f(u_coarse).reshape(N, M, D)
    u_fine = _f(u_fine).reshape(N * M, Di) if Di > 0 else None
    rs = re = None
    if ray_limits is not None:
        rs, re = _f(ray_limits[0]).reshape(N, M), _f(ray_limits[1]).reshape(N, M)
    w = [_f(dec[k]) for k in _NAMES]
    rgb = np.empty((N, M, 32), np.float32); seg = np.empty((N, M, 15), np.float32)
    depth = np.empty((N, M, 1), np.float32); wsum = np.empty((N, M, 1), np.float32)
    tw = np.empty((N, M, D - 1), np.float32) if (taps and Di > 0) else None
    tf = np.empty((N, M, Di), np.float32) if (taps and Di > 0) else None
    rc = lib.nfe_oracle_render(
        _p(norm_planes), _p(denorm_planes), ctypes.c_int(Np), ctypes.c_int(H), ctypes.c_int(W),
        *[_p(x) for x in w], ctypes.c_float(options.get("decoder_lr_mul", 1)),
        _p(origins), _p(dirs), ctypes.c_int(N), ctypes.c_int(M), ctypes.c_int(D), ctypes.c_int(Di),
        ctypes.c_float(0.0 if rs is not None else options["ray_start"]),
        ctypes.c_float(0.0 if rs is not None else options["ray_end"]), _p(rs), _p(re),
        ctypes.c_int(int(bool(options.get("disparity_space_sampling", False)))), ctypes.c_float(options["box_warp"]),
        ctypes.c_int(int(bool(options.get("white_back", False)))), _p(u_coarse), _p(u_fine),
        _p(rgb), _p(seg), _p(depth), _p(wsum), _p(tw), _p(tf), ctypes.c_int(threads))
    if rc != 0:
        raise ValueError("nfe_oracle_render: bad sizes")
    out = (rgb, seg, depth, wsum)
    return out + ({"weights_coarse": tw, "depths_fine": tf},) if taps else out
