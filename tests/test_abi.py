"""CPU-only checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol that
include/tortoise_mi355x.h declares, the ctypes mirrors have the library's struct sizes, and a call
without a GPU fails loudly through tt_last_error (never silently falls back)."""
import ctypes as C
import os
import re

import pytest

from tortoise_tts_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(E.LIB_PATH):
        from tortoise_tts_amd.build import build
        build(verbose=False)
    return E.load_library()


def declared_symbols(header="tortoise_mi355x.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ttx?_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    """Both headers under include/: the drop-in boundary and the operator-level test entries."""
    names = declared_symbols()
    assert 30 <= len(names) <= 60, len(names)  # the boundary a maintainer binds stays small: experiments and test hooks live elsewhere
    assert not [n for n in names if n.startswith(("tt_op_", "ttx_"))], "test / experiment entries belong in tortoise_mi355x_test.h"
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert set(names) == set(E._PROTOS), set(names) ^ set(E._PROTOS)
    tnames = declared_symbols("tortoise_mi355x_test.h")
    missing = [n for n in tnames if not hasattr(lib, n)]
    assert not missing, f"declared in the test header but not exported: {missing}"
    assert set(tnames) == set(E._TEST_PROTOS), set(tnames) ^ set(E._TEST_PROTOS)


def test_struct_mirrors_match(lib):
    for i, st in enumerate(E.BOUNDARY_STRUCTS):
        assert C.sizeof(st) == lib.tt_struct_size(i), st.__name__
    assert lib.tt_abi_version() == 6


def test_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc = lib.tt_init()
    assert rc != 0
    assert b"hip" in lib.tt_last_error().lower() or b"device" in lib.tt_last_error().lower()
    with pytest.raises(E.EngineError):
        E.check(rc)


def test_header_constants_match_the_host_mirror():
    """Every integer `#define TT_*` of the header that the ctypes host names too has the header's value (dtype codes, option ids)."""
    src = open(os.path.join(ROOT, "include", "tortoise_mi355x.h")).read()
    defines = {k: int(v) for k, v in re.findall(r"^#define\s+(TT_[A-Z0-9_]+)\s+(-?\d+)\s*$", src, flags=re.M)}
    assert {"TT_BF16", "TT_F16", "TT_F32", "TT_AR_OPT_LOOKAHEAD", "TT_DIFF_OPT_OVERLAP_PREPASS", "TT_DIFF_OPT_FUSED_GN"} <= set(defines)
    mirrored = [k for k in defines if hasattr(E, k)]
    assert len(mirrored) >= 6
    for k in mirrored:
        assert getattr(E, k) == defines[k], k
