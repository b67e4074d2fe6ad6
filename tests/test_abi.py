"""CPU tests of the drop-in boundary: libccc_amd.so loads, exports every symbol include/ccc_amd.h declares,
and fails loudly (no CPU fallback) when no gfx950 device is visible.  No compute calls here."""
import ctypes
import os
import re

import pytest

from centroidalcontrolcollection_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names.update(re.findall(r"\b(ccc_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_builds_and_loads():
    path = build.build_lib()
    assert os.path.exists(path)
    assert _lib.load() is not None
    assert _lib.load().ccc_abi_version() >= 1


def test_every_declared_symbol_is_exported():
    L = ctypes.CDLL(build.build_lib())
    declared = _declared_symbols()
    assert declared, "no declarations parsed from include/*.h"
    for name in declared:
        assert hasattr(L, name), "include/*.h declares %s but libccc_amd.so does not export it" % name
    assert sorted(_lib.ABI_SYMBOLS) == declared


def test_library_contains_gfx950_code_object():
    data = open(build.build_lib(), "rb").read()
    assert b"gfx950" in data


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="GPU present: the loud-failure path is only reachable without a device")
def test_no_device_fails_loudly_instead_of_falling_back():
    from centroidalcontrolcollection_amd import LinearMpcZmp

    with pytest.raises(_lib.CccError) as e:
        LinearMpcZmp(1.0, 2.0, 0.0625)
    assert e.value.code == _lib.CCC_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_invalid_arguments_are_rejected_before_touching_the_device():
    L = _lib.load()
    h = ctypes.c_void_p()
    assert L.ccc_zmp_create(-1.0, 2.0, 0.1, 0, ctypes.byref(h)) == _lib.CCC_ERR_INVALID_ARGUMENT
    assert L.ccc_zmp_create(1.0, 2.0, 0.0, 0, ctypes.byref(h)) == _lib.CCC_ERR_INVALID_ARGUMENT
    assert b"must be > 0" in L.ccc_last_error_string()
    assert L.ccc_zmp_horizon_steps(None) == -1


def test_shard_bounds_partition():
    """ccc_shard_bounds (no device needed): contiguous, balanced, covering -- the partition of SURVEY.md 8(e) and the
    same one sharding.shard_bounds gives the torch.distributed path."""
    from centroidalcontrolcollection_amd import sharding

    L = ctypes.CDLL(build.build_lib())
    L.ccc_shard_bounds.restype = ctypes.c_int
    L.ccc_shard_bounds.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64),
                                   ctypes.POINTER(ctypes.c_int64)]
    for n, world in ((65536, 8), (10, 3), (5, 8), (0, 4), (4097, 2)):
        got = []
        for r in range(world):
            b, e = ctypes.c_int64(), ctypes.c_int64()
            assert L.ccc_shard_bounds(n, world, r, ctypes.byref(b), ctypes.byref(e)) == 0
            got.append((b.value, e.value))
        assert got == sharding.shard_bounds(n, world)
    b, e = ctypes.c_int64(), ctypes.c_int64()
    assert L.ccc_shard_bounds(10, 0, 0, ctypes.byref(b), ctypes.byref(e)) != 0
