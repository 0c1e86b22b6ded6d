"""CPU: the C-ABI library loads and exports every symbol include/dbgk.h declares (no compute)."""
import ctypes
import os
import re

from dbg_assembly_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dbgk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dbgk_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 25
    L = ctypes.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libdbgk.so does not export %s" % n
    assert sorted(s[0] for s in capi.SYMBOLS) == names  # the binding covers the header exactly


def test_abi_version_and_errors():
    L = capi.lib()
    want = int(re.search(r"#define\s+DBGK_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "dbgk.h")).read()).group(1))
    assert L.dbgk_abi_version() == want
    assert L.dbgk_strerror(capi.ERR_TABLE_FULL) == b"k-mer table full"
    # argument validation happens before any device work
    cfg = capi.Config(0, 250, 1009, 0, 0, 0, 0)
    h = ctypes.c_void_p()
    assert L.dbgk_create(ctypes.byref(cfg), ctypes.byref(h)) == capi.ERR_ARG
    cfg = capi.Config(33, 250, 1009, 0, 0, 0, 0)
    assert L.dbgk_create(ctypes.byref(cfg), ctypes.byref(h)) == capi.ERR_ARG


def test_struct_layouts_match_header():
    assert ctypes.sizeof(capi.Config) == 4 + 4 + 8 + 4 + 4 + 8 + 8 + 32
    assert ctypes.sizeof(capi.Stats) == 6 * 8 + 8 + 8  # + other_bytes (ABI 5)
    assert ctypes.sizeof(capi.LinkStats) == 261 * 8
    assert ctypes.sizeof(capi.SynthParams) == 48
    assert capi.NODE_DTYPE.itemsize == 16


def test_prime_helpers_match_reference_kats():
    kat = {}
    for line in open(os.path.join(ROOT, "tests", "golden", "kat.txt")):
        t = line.split("\t")
        if t[0] == "find_next_prime":
            kat[int(t[1])] = int(t[2])
        if t[0] == "is_prime":
            assert capi.is_prime_ref(int(t[1])) == bool(int(t[2]))
    assert len(kat) >= 15
    for n, p in kat.items():
        assert capi.find_next_prime_ref(n) == p


def test_no_fallback_without_gpu():
    """On a box without a GPU the product must refuse to run, not fall back."""
    if capi.lib().dbgk_device_count() > 0:
        return
    try:
        capi.Graph(31, 1009)
    except capi.DbgkError as e:
        assert e.status == capi.ERR_HIP
    else:
        raise AssertionError("dbgk_create succeeded without a GPU")
