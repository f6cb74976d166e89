"""The C-ABI library builds for gfx950, loads without a GPU and exports every
symbol include/afqmc_hip.h declares.  No compute calls here."""
import os
import re

import pytest

from pauxy_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "afqmc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(afq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), s
    # and the ctypes table covers the header one to one
    assert sorted(_lib.SIGNATURES) == syms


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.AfqLibraryError):
        _lib.load(str(tmp_path / "libafqmc_hip.so"))


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "pauxy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(dirpath, f)


def test_every_entry_point_refuses_a_null_handle():
    """Every function of the C-ABI that takes the handle first returns AFQ_EINVAL for a null handle (no dereference, no
    HIP call): run in a child process so that a crash would be a failure, not the end of the test session."""
    import subprocess
    import sys
    code = r'''
import ctypes, sys
sys.path.insert(0, %r)
from pauxy_amd import _lib
lib = _lib.load()
n = 0
for name, args in _lib.SIGNATURES.items():
    if not args or args[0] is not _lib._h:
        continue
    call = []
    for a in args:
        if a in (ctypes.c_int, ctypes.c_int64, ctypes.c_uint64):
            call.append(0)
        elif a is ctypes.c_double:
            call.append(0.0)
        elif hasattr(a, '_flags_') and hasattr(a, '_restype_'):
            call.append(a())
        else:
            call.append(None)
    r = getattr(lib, name)(*call)
    if name == "afq_last_error":
        assert r == b"null handle", r
    else:
        assert r == -1, (name, r)
    n += 1
print("checked", n)
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert int(out.stdout.split()[-1]) >= 80
