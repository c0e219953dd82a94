"""The C-ABI library: builds, loads, and exports exactly what include/dm_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "dm_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from diffusion_models_amd import _lib

    assert sorted(_lib.EXPORTS) == _declared()


def test_library_loads_and_exports_every_symbol():
    from diffusion_models_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.dm_abi_version.restype = ctypes.c_int
    assert lib.dm_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback_without_gpu():
    import torch

    import diffusion_models_amd as dm

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        dm.Unet(dim=32, dim_mults=(1, 2))  # handle creation needs a HIP device; nothing falls back to the CPU
    with pytest.raises(RuntimeError):
        dm.Unet(dim=32, dim_mults=(1, 2), device="cpu")


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "diffusion-models_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
