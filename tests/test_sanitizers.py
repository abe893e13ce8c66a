"""Host-side sanitizer runs (reference: debug/Makefile:49-53 compute-sanitizer targets): the CPU oracle and the
LD_PRELOAD hook's host logic built with AddressSanitizer + UndefinedBehaviorSanitizer and driven through every
dtype x backend x mode x op combination / every hook entry point against mock HIP + hipBLAS libraries (tests/sanitize)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "sanitize")
FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


def _run(cmd, env=None, cwd=None):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env, cwd=cwd)
    assert p.returncode == 0, p.stdout[-4000:]
    return p.stdout


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_sweep")
    _run(["gcc", "-std=c11", *FLAGS, "-ffp-contract=off", "-frounding-math", "-mavx2", "-mfma", "-Wno-unused-function",
          os.path.join(SAN, "oracle_sweep.c"), os.path.join(ROOT, "oracle", "oz2_oracle.c"), "-lm", "-o", exe])
    out = _run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert "ALL OK" in out, out[-2000:]


def test_hook_under_asan_ubsan(tmp_path):
    """gemmul8_amd/csrc/oz2_hook.cpp (the only product file with hand-rolled lifetime logic) with ASan + UBSan, five host threads."""
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("ROCm clang++ not available")
    d = str(tmp_path)
    common = [clang, "-std=c++20", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-fPIC", "-Wno-unused-value", "-Wno-deprecated-declarations"]
    _run(common + ["-O1", "-shared", os.path.join(SAN, "mock_gpu.cpp"), "-o", os.path.join(d, "libmockgpu.so")])
    _run(common + FLAGS + ["-shared", os.path.join(ROOT, "gemmul8_amd", "csrc", "oz2_hook.cpp"), "-o", os.path.join(d, "libhook_san.so"), "-ldl", "-lpthread"])
    _run(common + FLAGS + [os.path.join(SAN, "hook_driver.cpp"), "-o", os.path.join(d, "hook_driver"), "-L" + d, "-lhook_san", "-lmockgpu",
                           "-Wl,-rpath," + d, "-lpthread"])
    out = _run([os.path.join(d, "hook_driver")], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1",
                                                         LD_LIBRARY_PATH=d, GEMMUL8_MIN_FLOPS="0"))
    assert "ALL OK" in out, out[-3000:]
