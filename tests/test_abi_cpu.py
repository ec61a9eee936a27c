"""CPU: the C-ABI library loads and exports every symbol include/nfe_render.h declares; the ctypes
mirror of nfe_render_args matches the C struct; host-side argument handling works without a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("nfe_render.h", "nfe_dense.h")]


def _declared_functions():
    names = set()
    for h in HEADERS:
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(nfe_[a-z_0-9]+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from nerffaceediting_amd import _lib
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 11
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    assert sorted(_lib.exported_symbols()) == declared, "ctypes signature table out of sync with the header"
    assert lib.nfe_abi_version() == _lib.NFE_ABI_VERSION


def test_render_args_struct_matches_header(tmp_path):
    """Compile a probe against the real header and compare sizeof/offsetof with the ctypes mirror."""
    from nerffaceediting_amd import _lib
    fields = [f[0] for f in _lib.RenderArgs._fields_]
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"nfe_render.h\"\nint main(){\n"
    prog += 'printf("%zu\\n", sizeof(nfe_render_args));\n'
    for f in fields:
        prog += f'printf("{f} %zu\\n", offsetof(nfe_render_args, {f}));\n'
    prog += 'printf("%d %d %d\\n", NFE_DECODER_PACKED_FLOATS, NFE_MAX_SAMPLES, NFE_ABI_VERSION);return 0;}\n'
    c = tmp_path / "probe.c"
    c.write_text(prog)
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    assert int(out[0]) == ctypes.sizeof(_lib.RenderArgs)
    for line, f in zip(out[1:], fields):
        name, off = line.split()
        assert name == f and int(off) == getattr(_lib.RenderArgs, f).offset, (line, f)
    consts = [int(x) for x in out[1 + len(fields)].split()]
    assert consts == [_lib.NFE_DECODER_PACKED_FLOATS, _lib.NFE_MAX_SAMPLES, _lib.NFE_ABI_VERSION]


def test_conv_args_struct_matches_header(tmp_path):
    from nerffaceediting_amd import _lib
    fields = [f[0] for f in _lib.ConvArgs._fields_]
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"nfe_dense.h\"\nint main(){\n"
    prog += 'printf("%zu\\n", sizeof(nfe_conv_args));\n'
    for f in fields:
        prog += f'printf("{f} %zu\\n", offsetof(nfe_conv_args, {f}));\n'
    prog += "return 0;}\n"
    c = tmp_path / "probe2.c"
    c.write_text(prog)
    exe = tmp_path / "probe2"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    assert int(out[0]) == ctypes.sizeof(_lib.ConvArgs)
    for line, f in zip(out[1:], fields):
        name, off = line.split()
        assert name == f and int(off) == getattr(_lib.ConvArgs, f).offset, (line, f)


def test_render_backward_args_struct_matches_header(tmp_path):
    from nerffaceediting_amd import _lib
    fields = [f[0] for f in _lib.RenderBackwardArgs._fields_]
    prog = "#include <stdio.h>\n#include <stddef.h>\n#include \"nfe_render.h\"\nint main(){\n"
    prog += 'printf("%zu\\n", sizeof(nfe_render_backward_args));\n'
    for f in fields:
        prog += f'printf("{f} %zu\\n", offsetof(nfe_render_backward_args, {f}));\n'
    prog += "return 0;}\n"
    c = tmp_path / "probe3.c"
    c.write_text(prog)
    exe = tmp_path / "probe3"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    assert int(out[0]) == ctypes.sizeof(_lib.RenderBackwardArgs)
    for line, f in zip(out[1:], fields):
        name, off = line.split()
        assert name == f and int(off) == getattr(_lib.RenderBackwardArgs, f).offset, (line, f)
    lib = _lib.load()
    assert lib.nfe_render_backward_workspace_bytes(2, 64, 24) >= 3 * 2 * 64 * 24 * 4
    a = _lib.RenderBackwardArgs()
    a.struct_size = 8
    assert lib.nfe_render_backward(ctypes.byref(a), None) == -1 and b"struct_size" in lib.nfe_last_error()


def test_workspace_query_and_argument_errors_without_gpu():
    from nerffaceediting_amd import _lib
    lib = _lib.load()
    single = lib.nfe_render_workspace_bytes(4, 512 * 512, 64, 0)          # depth min/max words + the segment composites of depth-split launches
    assert 256 <= single <= 32 << 20 and single == lib.nfe_render_workspace_bytes(1, 64, 8, 0)      # independent of the ray count
    two_pass = lib.nfe_render_workspace_bytes(1, 128 * 128, 96, 96)
    assert two_pass >= single + 128 * 128 * (96 + 95 + 192) * 4
    # validation happens before any launch: a null args pointer / bad struct size is refused
    assert lib.nfe_render(None, None) == -1
    assert b"args is null" in lib.nfe_last_error()
    a = _lib.RenderArgs()
    a.struct_size = 8
    assert lib.nfe_render(ctypes.byref(a), None) == -1
    assert b"struct_size" in lib.nfe_last_error()
    assert lib.nfe_ray_sampler(None, None, 1, 8, None, None, None) == -1
    with pytest.raises(RuntimeError):
        _lib.check(-1, "nfe_ray_sampler")


def test_ops_refuse_cpu_tensors():
    import torch
    from nerffaceediting_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.plane_pack(torch.zeros(1, 96, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.ray_sampler(torch.eye(4)[None], torch.eye(3)[None], 8)


def test_backward_workspace_is_bounded():
    """nfe_render_backward_workspace_bytes: four floats per sample slot of whole 64-ray tiles for the march records (round 4: kept in
    the tile order the kernels walk), plus the binned scatter's chunk buffers, which stop growing once a chunk holds 2^23 sample
    slots (256-byte feature-gradient row + 68 bytes of records per slot)."""
    from nerffaceediting_amd import _lib
    lib = _lib.load()
    align = lambda x: (x + 255) & ~255
    fixed = lambda n, m, s: lib.nfe_render_backward_workspace_bytes(n, m, s) - 4 * align(4 * n * ((m + 63) // 64) * 64 * s)
    per_slot = 256 + 3 * (8 + 16 + 8 + 4)
    small = fixed(3, 70, 12)                       # three views x two 64-ray tiles x 12 samples = 4608 slots
    assert 4608 * per_slot <= small <= 4608 * per_slot + (6 << 20)
    cap = fixed(1, 512 * 512, 192)
    assert cap == fixed(8, 512 * 512, 192)         # more views, same chunk
    assert (1 << 23) * per_slot * 0.99 <= cap <= (1 << 23) * per_slot + (8 << 20)
    assert lib.nfe_render_backward_workspace_bytes(0, 64, 4) < (1 << 23)


def test_status_queries_are_callable_without_a_gpu():
    """the two diagnostics that make the library not-quite-stateless (include/nfe_render.h, conventions) answer on a
    box without a GPU: the sticky hand-off word reads (0, 0) - with and without clear - and the thread-local kernel list is empty;
    the per-call queries reject a null workspace with NFE_EINVAL instead of touching a device."""
    from nerffaceediting_amd import _lib
    lib = _lib.load()
    for clear in (0, 1, 0):
        lost, calls = ctypes.c_uint32(7), ctypes.c_uint32(7)
        assert lib.nfe_render_status(ctypes.byref(lost), ctypes.byref(calls), clear) == 0
        assert (lost.value, calls.value) == (0, 0)
    assert lib.nfe_render_status(None, None, 0) == 0                       # both outputs are optional
    k = lib.nfe_render_last_kernels()
    assert k is not None and k.decode() == ""
    for fn in (lib.nfe_render_call_status, lib.nfe_render_backward_call_status):
        assert fn(None, None, None) == -1 and b"workspace is null" in lib.nfe_last_error()
