"""CPU: the C-ABI library loads and exports every symbol include/smalfit.h declares (no compute calls)."""
import os
import re

from smalify_amd import _lib

HEADER = os.path.join(os.path.dirname(__file__), "..", "include", "smalfit.h")


def test_every_declared_symbol_is_exported_and_bound():
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(smalfit_[a-z_0-9]+)\s*\(", text))
    assert len(declared) >= 20
    if _lib.needs_rebuild():
        _lib.build_library()
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    # every bound signature is declared in the header (no private entry points in the binding)
    assert set(_lib.SIGNATURES) <= declared | {"smalfit_debug_set"}
    assert lib.smalfit_version() >= 1


def test_missing_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    from smalify_amd import engine as eng, synthetic
    import pytest
    with pytest.raises(eng.SmalfitError):
        eng.DeviceModel(synthetic.synthetic_model())
