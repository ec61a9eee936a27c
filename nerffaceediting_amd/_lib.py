"""ctypes binding of libnfe_render.so (include/nfe_render.h).  Fails loudly when the library is absent."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# NFE_RENDER_LIB lets tools/ load an experimental build of the same ABI (kernel ablations)
LIB_PATH = os.environ.get("NFE_RENDER_LIB") or os.path.join(_HERE, "libnfe_render.so")

NFE_ABI_VERSION = 15
NFE_MAX_SAMPLES = 256
NFE_DECODER_PACKED_FLOATS = 4 * 2048 + 64 + 64 + 32 + 32 + 8192
NFE_DECODER_CROSS_FLOATS = 2048
NFE_MATH_BF16X3, NFE_MATH_FP32 = 0, 1

FP = c_void_p      # device pointers travel as integers


class RenderArgs(ctypes.Structure):
    """Mirror of ``nfe_render_args`` (include/nfe_render.h) — keep field order identical."""
    _fields_ = [
        ("struct_size", c_uint32),
        ("planes_geo", FP), ("planes_app", FP),
        ("plane_h", c_int32), ("plane_w", c_int32),
        ("plane_view_stride", c_int64),
        ("geo_scale", FP), ("geo_shift", FP), ("app_scale", FP), ("app_shift", FP),
        ("decoder_packed", FP), ("decoder_math", c_int32),
        ("n_views", c_int32), ("n_rays", c_int32),
        ("origins", FP), ("dirs", FP), ("cam2world", FP), ("intrinsics", FP),
        ("resolution", c_int32),
        ("depth_resolution", c_int32), ("depth_resolution_importance", c_int32),
        ("ray_start", c_float), ("ray_end", c_float),
        ("ray_start_per_ray", FP), ("ray_end_per_ray", FP),
        ("disparity_space_sampling", c_int32),
        ("box_warp", c_float), ("white_back", c_int32),
        ("u_coarse", FP), ("u_fine", FP), ("seed", c_uint64), ("seed_device", FP),
        ("rgb", FP), ("seg", FP), ("depth", FP), ("wsum", FP),
        ("channels_first", c_int32),
        ("tap_weights_coarse", FP), ("tap_depths_fine", FP), ("tap_depths_all", FP),
        ("workspace", FP), ("workspace_bytes", c_uint64), ("density_noise", c_float), ("decoder_cross", FP),
        ("clock_probe", FP), ("tap_sample_colors", FP), ("density_noise_values", FP),
    ]


class RenderBackwardArgs(ctypes.Structure):
    """Mirror of ``nfe_render_backward_args`` (include/nfe_render.h)."""
    _fields_ = [
        ("struct_size", c_uint32),
        ("planes_geo", FP), ("planes_app", FP),
        ("plane_h", c_int32), ("plane_w", c_int32),
        ("plane_view_stride", c_int64),
        ("geo_scale", FP), ("geo_shift", FP), ("app_scale", FP), ("app_shift", FP),
        ("geo_w0", FP), ("geo_b0", FP), ("geo_w1", FP), ("geo_b1", FP),
        ("app_w0", FP), ("app_b0", FP), ("app_w1", FP), ("app_b1", FP),
        ("lr_mul", c_float),
        ("n_views", c_int32), ("n_rays", c_int32),
        ("origins", FP), ("dirs", FP), ("cam2world", FP), ("intrinsics", FP),
        ("resolution", c_int32), ("n_samples", c_int32),
        ("depths", FP),
        ("box_warp", c_float), ("white_back", c_int32),
        ("grad_rgb", FP), ("grad_seg", FP), ("grad_depth", FP), ("grad_wsum", FP),
        ("channels_first", c_int32),
        ("grad_planes_geo", FP), ("grad_planes_app", FP),
        ("grad_view_stride", c_int64),
        ("workspace", FP), ("workspace_bytes", c_uint64), ("sample_colors", FP),
    ]


class ConvArgs(ctypes.Structure):
    """Mirror of ``nfe_conv_args`` (include/nfe_dense.h)."""
    _fields_ = [
        ("struct_size", c_uint32), ("mode", c_int32), ("math", c_int32),
        ("x", FP), ("styles", FP), ("packed", FP), ("dcoef", FP), ("noise", FP), ("noise_n_stride", c_int64), ("noise_strength", c_float),
        ("bias", FP),
        ("n", c_int32), ("h", c_int32), ("w", c_int32), ("cin", c_int32), ("cout", c_int32),
        ("lrelu", c_int32), ("act_gain", c_float), ("clamp", c_float),
        ("skip", FP), ("out_planes", c_int32), ("out", FP), ("scratch", FP), ("scratch_floats", c_uint64),
        ("next_styles", FP), ("next_split", FP), ("x_split", FP),
        ("rgb_weight", FP), ("rgb_styles", FP), ("rgb_bias", FP), ("rgb_skip", FP), ("rgb_out", FP), ("rgb_channels", c_int32), ("rgb_clamp", c_float),
    ]


class FcGroup(ctypes.Structure):
    """Mirror of ``nfe_fc_group``."""
    _fields_ = [("x", FP), ("x_stride", c_int64), ("w", FP), ("b", FP), ("y", FP), ("in_features", c_int32), ("out_features", c_int32),
                ("weight_gain", c_float), ("bias_gain", c_float)]


class DemodGroup(ctypes.Structure):
    """Mirror of ``nfe_demod_group``."""
    _fields_ = [("styles", FP), ("wsq", FP), ("dcoef", FP), ("cin", c_int32), ("cout", c_int32), ("styles_norm", FP)]


NFE_MAX_GROUPS = 32
NFE_CONV_BF16X3, NFE_CONV_BF16, NFE_CONV_F16 = 0, 1, 2
NFE_CONV_3X3, NFE_CONV_3X3_UP2, NFE_CONV_1X1 = 0, 1, 2

_SIGNATURES = {
    "nfe_abi_version": (c_int, []),
    "nfe_last_error": (c_char_p, []),
    "nfe_ray_sampler": (c_int, [FP, FP, c_int, c_int, FP, FP, c_void_p]),
    "nfe_ray_limits_box": (c_int, [FP, FP, c_int64, c_float, FP, FP, FP, c_void_p]),
    "nfe_plane_stats": (c_int, [FP, c_int, c_int, c_int, FP, FP, c_void_p]),
    "nfe_plane_affine": (c_int, [FP, FP, FP, c_int, c_int, c_int, c_int, FP, c_void_p]),
    "nfe_make_affine": (c_int, [FP, FP, FP, FP, c_int, c_int, c_int, FP, FP, FP, FP, c_void_p]),
    "nfe_plane_pack": (c_int, [FP, c_int, c_int, c_int, FP, c_void_p]),
    "nfe_decoder_pack": (c_int, [FP] * 8 + [c_float, FP, c_void_p]),
    "nfe_decoder_pack_cross": (c_int, [FP, c_float, FP, c_void_p]),
    "nfe_decoder_forward": (c_int, [FP, FP, c_int, c_int, c_int64, FP, c_int, FP, FP, FP, FP, c_void_p]),
    "nfe_render_workspace_bytes": (c_uint64, [c_int, c_int, c_int, c_int]),
    "nfe_render_sample_colors_floats": (c_uint64, [c_int, c_int, c_int]),
    "nfe_render": (c_int, [POINTER(RenderArgs), c_void_p]),
    "nfe_render_status": (c_int, [POINTER(c_uint32), POINTER(c_uint32), c_int]),
    "nfe_render_call_status": (c_int, [FP, c_void_p, POINTER(c_uint32)]),
    "nfe_render_last_kernels": (c_char_p, []),
    "nfe_render_backward_workspace_bytes": (c_uint64, [c_int, c_int, c_int]),
    "nfe_render_backward": (c_int, [POINTER(RenderBackwardArgs), c_void_p]),
    "nfe_render_backward_call_status": (c_int, [FP, c_void_p, POINTER(c_uint32)]),
    # include/nfe_dense.h
    "nfe_nchw_to_nhwc": (c_int, [FP, c_int, c_int, c_int, c_int, FP, c_void_p]),
    "nfe_nhwc_to_nchw": (c_int, [FP, c_int, c_int, c_int, c_int, FP, c_void_p]),
    "nfe_nhwc_to_planes": (c_int, [FP, c_int, c_int, c_int, FP, c_void_p]),
    "nfe_plane_stats_nhwc": (c_int, [FP, c_int, c_int, c_int, FP, FP, FP, c_void_p]),
    "nfe_fully_connected": (c_int, [FP, FP, FP, c_int, c_int, c_int, c_float, c_float, c_int, FP, c_int, c_void_p]),
    "nfe_fully_connected_grouped": (c_int, [POINTER(FcGroup), c_int, c_int, c_void_p]),
    "nfe_conv_demod_grouped": (c_int, [POINTER(DemodGroup), c_int, c_int, c_void_p]),
    "nfe_normalize_2nd_moment": (c_int, [FP, c_int, c_int, FP, c_int, c_void_p]),
    "nfe_broadcast_truncate": (c_int, [FP, FP, c_int, c_int, c_int, c_float, c_int, FP, c_void_p]),
    "nfe_conv_packed_words": (c_uint64, [c_int, c_int, c_int]),
    "nfe_conv_pack": (c_int, [FP, c_int, c_int, c_int, FP, FP, c_void_p]),
    "nfe_conv_pack_f16": (c_int, [FP, c_int, c_int, c_int, c_int, FP, FP, c_void_p]),
    "nfe_conv_demod": (c_int, [FP, FP, c_int, c_int, c_int, FP, FP, c_void_p]),
    "nfe_modulated_conv": (c_int, [POINTER(ConvArgs), c_void_p]),
    "nfe_conv_scratch_floats": (c_uint64, [c_int] * 7),
    "nfe_conv_split_floats": (c_uint64, [c_int] * 5),
    "nfe_conv_accepts_split": (c_int, [c_int] * 5),
    "nfe_conv_fuses_rgb": (c_int, [c_int] * 8),
    "nfe_conv_splits_in_epilogue": (c_int, [c_int] * 6),
    "nfe_conv_describe": (c_int, [c_int] * 8 + [c_char_p, c_int]),
    "nfe_upfirdn2d": (c_int, [FP, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, FP, c_void_p]),
    "nfe_resize_bilinear": (c_int, [FP, c_int, c_int, c_int, c_int, c_int, c_int, c_int, FP, c_void_p]),
    "nfe_resize_bilinear_backward": (c_int, [FP, c_int, c_int, c_int, c_int, c_int, c_int, c_int, FP, c_void_p]),
    "nfe_upfirdn2d_polyphase": (c_int, [FP, c_int, c_int, c_int, c_int, c_int, c_int, c_float, FP, c_void_p]),
    "nfe_bias_act_backward": (c_int, [FP, FP, FP, FP, FP, c_int, FP, c_float, c_float, c_int, c_int64, c_int, FP, c_void_p]),
    "nfe_point_query": (c_int, [FP, FP, c_int, c_int, c_int64, FP, FP, FP, FP, FP, c_int, FP, c_int, c_int, c_float,
                                FP, FP, FP, c_float, c_uint64, FP, c_void_p]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises ImportError when the .so is missing."""
    global _lib
    if _lib is None:
        # torch must be imported first: it carries its own HIP runtime (same SONAME as /opt/rocm's); loading
        # our .so before it would bind the kernels to a second runtime instance that never sees torch's device.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C nerffaceediting_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        if lib.nfe_abi_version() != NFE_ABI_VERSION:
            raise ImportError(f"libnfe_render.so ABI {lib.nfe_abi_version()} != expected {NFE_ABI_VERSION}")
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc, what=""):
    """Turn a non-zero return code into RuntimeError (the reference's TORCH_CHECK behaviour,
    torch_utils/ops/bias_act.cpp:39-55)."""
    if rc != 0:
        msg = load().nfe_last_error()
        raise RuntimeError(f"{what or 'nfe'} failed ({rc}): {msg.decode() if msg else ''}")
