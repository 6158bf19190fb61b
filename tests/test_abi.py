"""The shared library loads on a machine without a GPU and exports every entry point that
include/segk.h declares; the ctypes table binds exactly that set.  No kernel is launched."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "segk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(segk_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = declared_symbols()
    assert "segk_conv_fwd" in names and "segk_confmat" in names and len(names) >= 25


def test_library_exports_every_declared_symbol():
    from segk import _lib
    assert os.path.exists(_lib.LIB_PATH), "libsegk.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_table_matches_header():
    from segk import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.segk_version() >= 100


def test_missing_gpu_is_loud():
    """Ops refuse CPU tensors instead of computing on the host."""
    import torch
    from segk import ops
    with pytest.raises(ValueError):
        ops.argmax(torch.zeros(1, 3, 4, 4))
