"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares;
argument checks answer without touching a device."""
import ctypes
import glob
import os
import re

import pytest

from conftest import ROOT


def _declared():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(fpsg_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    so = os.path.join(ROOT, "fpsg_amd", "libfpsg_hip.so")
    if not os.path.exists(so):
        g.build()
    from fpsg_amd import _hip
    return _hip.load()


def test_exports_every_declared_symbol(lib):
    decl = _declared()
    assert "fpsg_chamfer_fwd" in decl and "fpsg_version" in decl
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"


def test_binding_table_matches_header(lib):
    from fpsg_amd import _hip
    assert sorted(_hip.SIGNATURES) == _declared()


def test_version_and_argument_checks(lib):
    assert lib.fpsg_version() == 1
    # null pointers / bad shapes are refused on the host, before any HIP call
    rc = lib.fpsg_chamfer_fwd(None, None, 1, 8, 8, None, None, None, None, None)
    assert rc == -1 and b"null pointer" in lib.fpsg_last_error()
    rc = lib.fpsg_chamfer_fwd(None, None, 0, 8, 8, None, None, None, None, None)
    assert rc == -2 and b"positive" in lib.fpsg_last_error()
    rc = lib.fpsg_chamfer_bwd(None, None, None, None, None, None, 2, 0, 3, None, None, None)
    assert rc == -2


def test_product_path_has_no_cpu_fallback():
    import torch
    from fpsg_amd._hip import FpsgHipError
    from fpsg_amd.metrics import chamfer_distance
    with pytest.raises(FpsgHipError):
        chamfer_distance(torch.rand(1, 4, 3), torch.rand(1, 4, 3))


def test_package_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for path in glob.glob(os.path.join(ROOT, "fpsg_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
    for path in glob.glob(os.path.join(ROOT, "fpsg_amd", "csrc", "*")):
        if os.path.isfile(path):   # comments may cite the oracle as the specification; code may not use it
            src = re.sub(r"(?m)^\s*#(?!\s*include)[^\n]*", "", open(path, errors="ignore").read())
            assert not re.search(r"#\s*include[^\n]*oracle", src), path
            assert "fpsg_oracle" not in re.sub(r"//[^\n]*", "", src), path
