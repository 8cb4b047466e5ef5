"""CPU: libuwu_hip.so builds, loads, and exports every symbol include/uwu_hip.h declares (no compute)."""
import ctypes
import os
import re

from tests.conftest import ROOT


def header_symbols():
    src = open(os.path.join(ROOT, "include", "uwu_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uwu_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_all_header_symbols():
    from uwudiff_amd import build, lib

    build.build(verbose=False)
    cd = ctypes.CDLL(lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(cd, s)]
    assert not missing, f"not exported: {missing}"
    # the ctypes table binds exactly the header's entry points
    assert sorted(lib.exported_symbols()) == syms
    l = lib.load()
    assert l.uwu_version() >= 1
    assert isinstance(l.uwu_last_error(), bytes)


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    import pytest
    import torch

    from uwudiff_amd import lib

    with pytest.raises(lib.UwuError):
        lib.ptr(torch.zeros(4))


def test_product_does_not_import_oracle():
    bad = []
    for pkg in ("uwudiff_amd", "duwu", "tools", "test_scripts"):
        for dp, _, fs in os.walk(os.path.join(ROOT, pkg)):
            for f in fs:
                if f.endswith(".py"):
                    s = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_ctypes_arity_matches_header():
    """Every ctypes signature has exactly as many parameters as the C prototype in include/uwu_hip.h."""
    from uwudiff_amd import lib

    src = open(os.path.join(ROOT, "include", "uwu_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = dict(re.findall(r"\b(uwu_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S))
    for name, (_, args) in lib._SIGS.items():
        params = protos[name].strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), f"{name}: header has {n} params, ctypes table has {len(args)}"
