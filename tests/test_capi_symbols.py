"""The C-ABI library loads and exports every symbol include/hgx.h declares (no compute calls)."""
import ctypes
import os
import re

from hisatgenotype_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hgx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hgx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    lib = ctypes.CDLL(capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libhgx.so does not export %s" % n
    assert sorted(capi.SYMBOLS) == names


def test_version_and_error_string():
    lib = capi.lib()
    assert lib.hgx_version() >= 100
    assert isinstance(lib.hgx_last_error(), bytes)
    assert capi.a_pad(7000) == 7168 and capi.a_pad(512) == 512 and capi.a_pad(1) == 512


def test_invalid_arguments_are_reported_not_crashed():
    lib = capi.lib()
    rc = lib.hgx_locus_create(None, None)
    assert rc == -1 and b"invalid argument" in lib.hgx_last_error()
