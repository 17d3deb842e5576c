"""The C-ABI library loads and exports every symbol include/nvbio_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import nvbio_amd
from nvbio_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nvbio_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nvbio_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_symbol():
    assert os.path.exists(nvbio_amd.LIB_PATH), "build with python -m nvbio_amd.build"
    L = ctypes.CDLL(nvbio_amd.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), s
    assert nvbio_amd.lib().nvbio_hip_abi_version() == 2
    assert nvbio_amd.lib().nvbio_hip_arch() == b"gfx950"


def test_struct_layouts_match_header():
    # sizes the C compiler gives the ABI structs (LP64): 8+8+4+4+8+8+4+4 / 16 / 4+4+20+4 +8+8+8 +4+4 +8 +4+4+16+16
    assert ctypes.sizeof(_lib.StringSetStruct) == 48
    assert ctypes.sizeof(_lib.GotohSchemeStruct) == 16
    assert ctypes.sizeof(_lib.FMIndexStruct) == 120
    assert _lib.FMIndexStruct.dimer.offset == 64 and _lib.FMIndexStruct.dimer_S.offset == 80


def test_invalid_arguments_are_rejected_without_a_gpu():
    L = nvbio_amd.lib()
    sc = _lib.GotohSchemeStruct(2, -1, -2, -1)
    ps, ts = _lib.StringSetStruct(), _lib.StringSetStruct()
    ps.bits, ts.bits = 4, 2
    # null outputs -> hipErrorInvalidValue (1); nothing is launched
    assert L.nvbio_hip_banded_gotoh_score(ctypes.byref(sc), 1, 15, ctypes.byref(ps), ctypes.byref(ts), 0, 0, 10, None, None, None) == 1
    # an 8-bit TEXT stream (or a 3-bit pattern stream) is outside the ABI's contract -> hipErrorNotSupported (801); 8-bit patterns are inside it
    ps.bits = 3
    assert L.nvbio_hip_banded_gotoh_score(ctypes.byref(sc), 1, 15, ctypes.byref(ps), ctypes.byref(ts), 0, 0, 10, None, None, None) == 801
    ps.bits, ts.bits = 8, 8
    assert L.nvbio_hip_banded_gotoh_score(ctypes.byref(sc), 1, 15, ctypes.byref(ps), ctypes.byref(ts), 0, 0, 10, None, None, None) == 801
    ts.bits = 2
    assert L.nvbio_hip_banded_gotoh_score(ctypes.byref(sc), 1, 15, ctypes.byref(ps), ctypes.byref(ts), 0, 0, 10, None, None, None) == 1
    # an empty batch is legal
    ps.bits = 4
    assert L.nvbio_hip_banded_gotoh_score(ctypes.byref(sc), 1, 15, ctypes.byref(ps), ctypes.byref(ts), 0, 0, 0, None, None, None) == 0
    f = _lib.FMIndexStruct()
    assert L.nvbio_hip_fm_rank(ctypes.byref(f), None, None, 10, None, None) == 1
    assert L.nvbio_hip_fm_filter_temp_bytes(1000) > 8000


def test_product_does_not_import_the_oracle():
    """The product package must never route through oracle/ (only tests, smoke and the
    cpu_baseline leg of bench.py may)."""
    pkg = os.path.join(ROOT, "nvbio_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"(from|import)\s+oracle|oracle[/.]|pyoracle|nvbio_oracle", src), os.path.join(dirpath, f)
