"""CPU suite: the product library loads and exports every symbol include/vdb.h
declares (no compute without a GPU), and fails loudly without one."""
import ctypes as C
import os
import re

import pytest

from vectordb_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "vdb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vdb_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _header_symbols() == sorted(_capi.SYMBOLS)


def test_hip_library_exports_every_symbol(hip_lib):
    for sym in _header_symbols():
        assert hasattr(hip_lib, sym), sym


def test_oracle_library_exports_same_abi(oracle_lib):
    for sym in _header_symbols():
        assert hasattr(oracle_lib, sym), sym


def test_no_gpu_means_loud_failure(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = hip_lib.vdb_create(C.byref(h), 768, 0, 0, 0)
    assert rc == -2  # VDB_ERR_DEVICE: no CPU fallback
    assert b"no CPU fallback" in hip_lib.vdb_last_error(None)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vectordb_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libvdb_oracle" not in src, f
