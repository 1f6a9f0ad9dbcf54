import os
import sys
import dataclasses

# tests drive the library through the test hooks of include/wrs_debug.h: load the test build (the product's objects + debug_api.cpp)
os.environ["WRS_DEBUG_LIB"] = "1"

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def _has_gpu():
    try:
        import whisper_rs_amd as W
        return W.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """The HIP library on a gfx950 device.  GPU tests FAIL (not skip) if it is missing: a silent
    fallback would void every parity claim."""
    import whisper_rs_amd as W
    W.load_library()
    assert W.device_count() > 0, "no gfx950 device visible: -m gpu tests must run on the MI355X box"
    return W


def oracle_ids(n_vocab=51864):
    from whisper_rs_amd.consts import TokenIds
    from oracle import decode as OD
    return OD.TokenIds(**dataclasses.asdict(TokenIds.for_vocab(n_vocab)))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
