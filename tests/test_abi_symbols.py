"""The C-ABI library loads and exports every symbol include/threecrate_hip.h declares
(no compute calls: this runs without a GPU)."""
import os
import re

from threecrate_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "threecrate_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_python_binding_agree():
    assert declared_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    for sym in declared_symbols():
        assert hasattr(L, sym), sym
    assert L.tc_abi_version() == 2


def test_no_device_means_gpu_error_not_fallback():
    import threecrate_amd as tc
    L = _lib.load()
    if L.tc_device_count() > 0:
        return   # on a GPU box the context must work instead (covered by the -m gpu tests)
    import pytest
    with pytest.raises(tc.GpuError):
        tc.GpuContext(0)


def test_product_never_imports_the_oracle():
    """The product path may not import, link, dlopen or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "threecrate_amd")
    banned = re.compile(r"(^|\s)(import\s+oracle|from\s+oracle)|tc_oracle|libtc_oracle|tco_[a-z]|oracle/|oracle\.oracle")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert not banned.search(src), f"{f} references the oracle"
    out = os.popen(f"readelf -d {os.path.join(pkg, 'libthreecrate_hip.so')} 2>/dev/null").read()
    assert "oracle" not in out
