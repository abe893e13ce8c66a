import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# GEMMUL8_TEST_SOAK=1: the long forms of the GPU grids (every size of the reference's debug/test.cu sweep, every op pair for the real types in the sub-matrix
# grid, the extra shapes of the heaviest parity cases).  The default run hits every configuration, kernel path and op pair at least once and stays inside
# half of the driver's step limit (profiles/r06_gpu_suite_durations.json); the soak forms are for one-off runs.
SOAK = os.environ.get("GEMMUL8_TEST_SOAK", "0") == "1"


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


@pytest.fixture(autouse=True)
def _fp8_bound_mode_under_test(request):
    """FP8 accurate mode: the ORACLE's default inflation of the bound GEMM is the reference's (k+1)*2^-24 (src/find_max.hpp:82-96), the
    PRODUCT's default is its engine-safe formula (include/gemmul8_c.h gemmul8_set_fp8_bound_mode: gfx950's FP8 MFMA truncates).  Every
    -m gpu test therefore starts with the product's safe mode selected EXPLICITLY ON BOTH SIDES (gpu_util.select_fp8_bound_mode); tests of
    the reference formula select mode 1 on both sides themselves (parity_case(..., bound_mode=REFERENCE)).  Afterwards each side is back on
    its own default."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import gpu_util as gu
    gu.select_fp8_bound_mode(gu.SAFE)
    yield
    gu.restore_fp8_bound_defaults()


# ---- per-file wall time of the -m gpu suite (the driver's step limit is 1200 s; keep `pytest -m gpu` under 600 s and SEE where it goes):
# written to gpurun_out/gpu_suite_durations.json at the end of any session that ran GPU tests; the round's copy lives in profiles/.
_DUR = {}


def pytest_runtest_logreport(report):
    if report.when in ("setup", "call", "teardown"):
        f = report.nodeid.split("::")[0]
        d = _DUR.setdefault(f, [0.0, 0])
        d[0] += report.duration
        d[1] += report.when == "call"


def pytest_sessionfinish(session, exitstatus):
    gpu = {f: v for f, v in _DUR.items() if os.path.basename(f).startswith("test_gpu_")}
    if not gpu or sum(v[1] for v in gpu.values()) < 20:   # a full (or nearly full) GPU run only
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    rec = {"total_s": round(sum(v[0] for v in gpu.values()), 1), "exitstatus": int(exitstatus),
           "files": {f: {"seconds": round(v[0], 1), "tests": v[1]} for f, v in sorted(gpu.items(), key=lambda kv: -kv[1][0])}}
    with open(os.path.join(out, "gpu_suite_durations.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
