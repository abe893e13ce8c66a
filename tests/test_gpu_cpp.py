"""-m gpu: the C++ drop-in surface -- include/gemmul8.hpp (gemmul8::gemm / gemmLt / workSize) and the
LD_PRELOAD hipBLAS hook -- exercised by compiled C++ programs (tests/cpp, built by build())."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "build")
LIB = os.path.join(ROOT, "gemmul8_amd", "lib", "libgemmul8.so")


def run(cmd, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(p.stdout)
    assert p.returncode == 0 and "ALL OK" in p.stdout, p.stdout[-2000:]
    return p.stdout


def test_cpp_api_dropin():
    assert os.path.exists(os.path.join(BIN, "test_api")), "tests/cpp not built (run __graft_entry__.build())"
    run([os.path.join(BIN, "test_api")], {})


def test_ld_preload_hook():
    assert os.path.exists(os.path.join(BIN, "test_hook")), "tests/cpp not built (run __graft_entry__.build())"
    out = run([os.path.join(BIN, "test_hook")],
              {"LD_PRELOAD": LIB, "GEMMUL8_NUM_MOD_D": "15", "GEMMUL8_NUM_MOD_S": "8", "GEMMUL8_SKIP_SCALE_A": "1",
               "GEMMUL8_SKIP_SCALE_B": "1", "GEMMUL8_MAX_M": "256", "GEMMUL8_MAX_N": "256", "GEMMUL8_MAX_K": "1024",
               "GEMMUL8_MAX_NUM_MOD": "15"})
    assert "bitwise" in out
