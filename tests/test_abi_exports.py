"""CPU-side checks of the boundary: the library builds for gfx950, loads without a GPU and exports
every symbol include/phdhip.h declares; the ctypes parameter block matches the C struct."""
import ctypes as C
import os
import re

from monorfs_amd import _lib
from monorfs_amd.abi import PhdParams, prm3d_defaults

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "phdhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phd_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    so = _lib.build()
    lib = C.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libphdhip.so does not export %s" % n
    assert set(names) == set(_lib.EXPORTS)


def test_params_struct_matches_c_defaults():
    lib = _lib.load()
    assert lib.phd_api_version() == 4
    c = PhdParams()
    lib.phd_default_params(C.byref(c), 7, 640, 33)
    py = prm3d_defaults(7, 640, 33)
    for name, _ in PhdParams._fields_:
        a, b = getattr(c, name), getattr(py, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, name


def test_create_fails_loudly_without_device_or_bad_params():
    import torch
    lib = _lib.load()
    p = prm3d_defaults(4, 600, 16)
    p.max_components = 10   # < max_quantity
    assert not lib.phd_create(C.byref(p), 0)
    assert b"capacities" in lib.phd_create_error()
    # MaxQuantity beyond what one particle's prune can keep in LDS is refused at creation, not at the first launch
    p = prm3d_defaults(4, 6000, 16)
    p.max_quantity = 6000
    assert not lib.phd_create(C.byref(p), 0)
    assert b"max_quantity too large" in lib.phd_create_error()
    if not torch.cuda.is_available():
        p = prm3d_defaults(4, 600, 16)
        assert not lib.phd_create(C.byref(p), 0)
        assert b"no HIP device" in lib.phd_create_error() or b"device" in lib.phd_create_error()


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "monorfs_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libphd_oracle" not in text and "import orc" not in text, f
