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


def test_switches_live_in_the_process_not_in_the_environment(monkeypatch):
    """Path-forcing switches are an in-process test hook (hgx_test_switch_set): set, cleared one by one or all at once; an
    environment variable of the old name does nothing (the library reads no path-selecting environment variable)."""
    import ctypes as C
    from hisatgenotype_amd import capi, engine
    L = capi.lib()
    L.hgx_test_switch.restype = C.c_char_p
    monkeypatch.setenv("HGX_EM_NO_EMX", "1")
    assert L.hgx_test_switch(b"em_skip") is None
    engine.test_switch("em_skip", "emx")
    engine.test_switch("em_mid_nnz", 123)
    assert L.hgx_test_switch(b"em_skip") == b"emx" and L.hgx_test_switch(b"em_mid_nnz") == b"123"
    engine.test_switch("em_skip", None)
    assert L.hgx_test_switch(b"em_skip") is None and L.hgx_test_switch(b"em_mid_nnz") == b"123"
    with engine.test_switches(front="host"):
        assert L.hgx_test_switch(b"front") == b"host"
    assert L.hgx_test_switch(b"front") is None
    engine.test_switch(None)
    assert L.hgx_test_switch(b"em_mid_nnz") is None


def test_em_mode_values():
    """hgx_em_set_fast: 0 the reference's order where the default gate allows, 1 table lookups, -1 the reference's order at every size;
    the call returns the previous setting (thread-local)."""
    from hisatgenotype_amd import engine
    assert engine.em_set_fast(True) is False
    assert engine.em_set_fast(-1) is True
    assert engine.em_set_fast(False) == -1
    assert engine.em_set_fast(False) is False
