"""CPU: libmpcasm.so loads and exports exactly the C ABI that include/mpcasm.h
declares (no compute call is made here)."""
import ctypes
import os
import re

from mpcasm import capi

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                      "include", "mpcasm.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpcasm_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert "mpcasm_fill_su" in names and "mpcasm_assemble" in names and len(names) >= 12
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), "%s declared in mpcasm.h but not exported" % name
    # the ctypes binding covers the whole header, nothing more
    assert sorted(capi.SIGNATURES) == names


def test_library_answers_without_a_device():
    lib = capi.load()
    assert lib.mpcasm_abi_version() >= 1000
    assert lib.mpcasm_device_count() >= 0
    assert lib.mpcasm_status_string(0) == b"ok"
    assert lib.mpcasm_status_string(-2) == b"malformed plan tables"
    assert lib.mpcasm_set_option(99, 0) == -1
    assert lib.mpcasm_plan_destroy(None) == 0
    assert lib.mpcasm_plan_create(None, 0, None, 0, None) == -1
    assert set(capi.STATUS) == {0, -1, -2, -3, -4, -5}
