"""The C-ABI libraries load and export every symbol their headers declare (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

from helpers import ROOT
from timberborn_support_solver_amd import _lib


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:mi355sat|tbs)_[a-z0-9_]+)\s*\(", text)))


def test_solver_library_exports_the_whole_abi():
    L = _lib.solver_lib()
    names = declared_functions("mi355sat.h")
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n


def test_host_library_exports_the_whole_abi():
    L = _lib.host_lib()
    names = declared_functions("tbs_host.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "SOLVER_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_solver", None)
    with pytest.raises(_lib.NativeLibraryMissing, match="no CPU fallback"):
        _lib.solver_lib()


def test_product_sources_never_touch_the_oracle():
    """The oracle is a checker: nothing under the package (or bench's measured leg) may import it."""
    pkg = os.path.join(ROOT, "timberborn_support_solver_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                src = open(os.path.join(d, f), errors="ignore").read()
                assert "oracle" not in src.lower(), os.path.join(d, f)


def test_signature_and_opts_struct_layout():
    from timberborn_support_solver_amd.solver import Mi355SatOpts, Mi355SatStats
    L = _lib.solver_lib()
    L.mi355sat_abi_sizes.restype = ctypes.c_uint64
    st_size = ctypes.c_uint64(0)
    assert L.mi355sat_abi_sizes(ctypes.byref(st_size)) == ctypes.sizeof(Mi355SatOpts)     # the header's structs,
    assert st_size.value == ctypes.sizeof(Mi355SatStats) == 8 * (9 + 4 + 8 + 8 + 2)       # as the compiler laid them out (31 x 8 bytes)
    L.mi355sat_signature.restype = ctypes.c_char_p
    assert b"mi355sat" in L.mi355sat_signature()
    L.mi355sat_release_cached_memory()        # nothing parked, no device touched: must be a no-op
