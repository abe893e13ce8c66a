import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _fresh_knobs():
    """The library parses its GEMMUL8_* testing knobs ONCE (csrc/oz2_knobs.hpp).  A test that changed one (gpu_util.setknob) leaves a
    stale snapshot behind when monkeypatch restores the environment: re-parse after every test, if the library is loaded at all."""
    yield
    import gemmul8_amd as g
    if g._lib is not None:
        g._lib.gemmul8_reload_knobs()
