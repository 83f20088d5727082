"""CPU suite: the C-ABI shared object loads and exports every symbol include/fumi_hip.h declares (no compute)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "fumi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fumi_hip_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from fumi_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = ctypes.CDLL(hip.LIB_PATH)
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    for s in declared:
        assert hasattr(L, s), f"{s} declared in include/fumi_hip.h but not exported"
    assert sorted(hip.SYMBOLS) == declared, "fumi_amd/hip.py binds a different symbol set than the header declares"


def test_strerror_and_version_without_gpu():
    from fumi_amd import hip
    L = hip.lib()
    assert L.fumi_hip_version() >= 100
    assert L.fumi_hip_strerror(0) == b"ok"
    assert L.fumi_hip_strerror(-1) == b"invalid argument"


def test_product_path_fails_loudly_without_gpu_tensors():
    import pytest
    import torch
    from fumi_amd import hip
    with pytest.raises(hip.FumiHipError):
        hip.Workspace("cpu")
    with pytest.raises(hip.FumiHipError):
        hip.glove_bag(None, torch.zeros(1, 2, dtype=torch.int64), torch.zeros(3, 4), 0)
