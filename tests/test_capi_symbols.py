"""CPU: libgm3d_hip.so builds, loads without a GPU and exports exactly what include/gm3d.h declares
(no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    txt = open(os.path.join(ROOT, "include", "gm3d.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return {m.group(1): m.group(2) for m in re.finditer(r"\b(gm3d_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", txt)}


def test_library_exports_every_declared_symbol():
    from gm3d_amd import build
    lib = ctypes.CDLL(build.build())
    decl = declared()
    assert len(decl) >= 11
    for name in decl:
        assert hasattr(lib, name), name
    from gm3d_amd import _capi
    assert set(_capi.SIGNATURES) == set(decl)
    for name, args in decl.items():
        n = 0 if args.strip() == "void" else len(args.split(","))
        assert len(_capi.SIGNATURES[name]) == n, name
    assert _capi.lib.gm3d_abi_version() == 1
    assert b"invalid" in _capi.lib.gm3d_strerror(-1)


def test_argument_errors_are_detected_before_any_launch():
    """NULL pointers / bad sizes return GM3D_EINVAL / EUNSUPPORTED on the host: safe without a GPU."""
    from gm3d_amd._capi import lib, GM3D_EINVAL, GM3D_EUNSUPPORTED
    assert lib.gm3d_fps(None, 1, 1024, 64, None, None, None) == GM3D_EINVAL
    one = ctypes.c_void_p(16)  # never dereferenced: the shape check fails first
    assert lib.gm3d_fps(one, 1, 100000, 64, one, None, None) == GM3D_EUNSUPPORTED
    assert lib.gm3d_knn(one, one, 1, 8, 2, 9, None, one, None) == GM3D_EINVAL          # k > N
    assert lib.gm3d_knn(one, one, 1, 1024, 2, 65, None, one, None) == GM3D_EUNSUPPORTED
    assert lib.gm3d_attention_fwd(one, one, None, 1, 129, 6, 0.125, 1, None) == GM3D_EUNSUPPORTED
    assert lib.gm3d_attention_fwd(one, one, None, 1, 64, 6, 0.125, 7, None) == GM3D_EINVAL
    assert lib.gm3d_chamfer_fwd(one, one, 1, 0, 32, one, one, one, one, None) == GM3D_EINVAL


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from gm3d_amd import ops
    with pytest.raises(RuntimeError):
        ops.fps(torch.zeros(1, 16, 3), 4)
    with pytest.raises(RuntimeError):
        ops.chamfer(torch.zeros(1, 32, 3), torch.zeros(1, 32, 3))
    with pytest.raises(RuntimeError):
        ops.attention(torch.zeros(1, 4, 3 * 6 * 64), 6, 0.125)
