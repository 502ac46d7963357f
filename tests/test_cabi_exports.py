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
    assert set(_lib.SIGNATURES) <= declared
    # ... and the product exports nothing else: no developer switches, no undeclared entry points
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("smalfit_")}
    assert exported == declared, exported ^ declared
    assert lib.smalfit_version() >= 1


def test_missing_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    from smalify_amd import engine as eng, synthetic
    import pytest
    with pytest.raises(eng.SmalfitError):
        eng.DeviceModel(synthetic.synthetic_model())


def test_struct_layouts_match_the_header(tmp_path):
    """every struct the binding passes by pointer has the layout gcc gives the declaration in include/smalfit.h:
    same field names, offsets and total size (guards against the ctypes mirror drifting from the header)"""
    import ctypes as C
    import subprocess
    pairs = {"smalfit_model_desc": _lib.ModelDesc, "smalfit_fit_args": _lib.FitArgs, "smalfit_fit3d_args": _lib.Fit3dArgs,
             "smalfit_adam_args": _lib.AdamArgs, "smalfit_shard_args": _lib.ShardArgs, "smalfit_rccl_ctx": _lib.RcclCtx}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "smalfit.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for field, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, field, cname, field))
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = {(a, b): int(c) for a, b, c in (ln.split() for ln in out.strip().splitlines())}
    for cname, cls in pairs.items():
        assert got[(cname, "sizeof")] == C.sizeof(cls), cname
        for field, _ in cls._fields_:
            assert got[(cname, field)] == getattr(cls, field).offset, (cname, field)
