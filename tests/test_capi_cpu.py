"""CPU (no GPU needed): the C-ABI library loads, exports every symbol include/wrs.h declares, host-only entry
points work, and every compute entry fails loudly with WRS_ENODEVICE instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import whisper_rs_amd as W
from whisper_rs_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="wrs.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wrs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = W.load_library()
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/wrs.h but not exported by {L.LIB_PATH}"
    assert set(L.SYMBOLS) == set(names), set(L.SYMBOLS) ^ set(names)


def test_product_library_is_the_boundary_and_nothing_else():
    """libwrs_mi355x.so (what a caller links: bench.py, examples/transcribe, the Rust crate) exports every entry of
    include/wrs.h and NOT ONE test hook of include/wrs_debug.h; the hooks live in libwrs_mi355x_debug.so (the same
    objects + csrc/debug_api.cpp), which is what this test session has loaded."""
    import ctypes
    assert L.IS_DEBUG_LIB and L.LIB_PATH == L.DEBUG_LIB_PATH, "tests/conftest.py selects the test build"
    product = ctypes.CDLL(L.PRODUCT_LIB_PATH)
    for n in _declared_symbols():
        assert hasattr(product, n), f"{n} is declared in include/wrs.h but not exported by libwrs_mi355x.so"
    leaked = [n for n in _declared_symbols("wrs_debug.h") if hasattr(product, n)]
    assert not leaked, f"test hooks exported by the product library: {leaked}"
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L.PRODUCT_LIB_PATH], capture_output=True, text=True).stdout
    assert "wrs_debug_" not in out, "a wrs_debug_* symbol is defined in the product library"


def test_library_exports_every_debug_symbol():
    """include/wrs_debug.h (test hooks, not part of the boundary) is held to the same rule."""
    from whisper_rs_amd import debug as DBG
    lib = DBG.lib()
    names = _declared_symbols("wrs_debug.h")
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/wrs_debug.h but not exported by libwrs_mi355x_debug.so"
    assert set(DBG.SYMBOLS) == set(names), set(DBG.SYMBOLS) ^ set(names)
    assert not (set(names) & set(_declared_symbols())), "a debug hook leaked into the boundary header"


def test_version_and_device_count():
    lib = W.load_library()
    assert b"gfx950" in lib.wrs_version()
    assert W.device_count() >= 0


def test_host_only_entry_points():
    f = W.audio.mel_filter()
    gold = np.loadtxt(os.path.join(ROOT, "tests", "golden", "mel-filter-8x8.csv"), delimiter=",", dtype=np.float32)
    np.testing.assert_allclose(f[:8, :8], gold, rtol=1e-5, atol=1e-8)  # reference audio.rs:273-274
    x = np.arange(10, dtype=np.float32).reshape(2, 5)
    np.testing.assert_array_equal(W.audio.pad_or_trim(x, 3), x[:, :3])
    np.testing.assert_array_equal(W.audio.pad_or_trim(x, 7)[:, 5:], 0)


@pytest.mark.skipif(W.device_count() > 0, reason="checks the no-GPU failure mode")
def test_compute_entries_fail_loudly_without_gpu():
    with pytest.raises(W.WrsError) as e:
        W.Whisper(W.MODEL_DIMS["nano"])
    assert e.value.code == L.WRS_ENODEVICE and "no CPU fallback" in str(e.value)
    with pytest.raises(W.WrsError) as e:
        W.audio.log_mel_spectrogram(np.zeros(16000, np.float32))
    assert e.value.code == L.WRS_ENODEVICE


def test_struct_layouts_match_header():
    """field order of the ctypes mirrors == include/wrs.h (a silent mismatch would corrupt options)."""
    txt = open(os.path.join(ROOT, "include", "wrs.h")).read()
    dims = re.search(r"typedef struct wrs_model_dims \{(.*?)\}", txt, re.S).group(1)
    assert re.findall(r"int64_t (\w+);", dims) == [f for f, _ in L.ModelDimsC._fields_]
    opts = re.search(r"typedef struct wrs_decode_options \{(.*?)\} wrs_decode_options", txt, re.S).group(1)
    opts = re.sub(r"/\*.*?\*/", "", opts, flags=re.S)
    assert re.findall(r"(\w+);", opts) == [f for f, _ in L.DecodeOptionsC._fields_]
    assert C.sizeof(L.TokenIdsC) == 40
