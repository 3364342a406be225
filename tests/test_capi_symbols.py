"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/stgcn_hip.h declares (no compute calls here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "stgcn_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(stgcn_[a-z0-9_]+)\s*\(", hdr)))


def test_header_lists_expected_entry_points():
    from stgcn_amd import _lib
    assert declared_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_hip_library_builds_loads_and_exports_all_symbols():
    from stgcn_amd import _lib, build
    try:
        path = build.build()
    except RuntimeError as e:          # no hipcc on this machine
        pytest.skip(str(e))
    L = _lib._Lib(path)
    for sym in declared_symbols():
        assert hasattr(L.dll, sym), sym
    assert L.backend == "hip-gfx950" and L.dll.stgcn_version() >= 1


def test_missing_library_fails_loudly(tmp_path):
    from stgcn_amd import _lib
    with pytest.raises(_lib.StgcnError):
        _lib._Lib(str(tmp_path / "nope.so"))


def test_product_package_never_imports_oracle():
    import ast
    pkg = os.path.join(ROOT, "stgcn_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            tree = ast.parse(open(os.path.join(pkg, fn)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                assert not any(n.split(".")[0] in ("oracle", "tests") for n in names), f"{fn} imports {names}"
