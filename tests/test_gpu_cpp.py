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
    env.setdefault("GEMMUL8_MIN_FLOPS", "0")   # the test matrices are small: emulate every call (the reference's behaviour), not the automatic floor
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


def test_ld_preload_hook_hipblaslt_and_range_passthrough():
    """hipblasLtMatmul interception (plain S / D matmul, in place and out of place: bitwise equal to the direct call; BIAS epilogue: direct
    result + bias, bitwise; GEMMUL8_MIN_FLOPS floor: native) and the k > 2^17 passthrough of the hipBLAS hook."""
    assert os.path.exists(os.path.join(BIN, "test_hook_lt")), "tests/cpp not built (run __graft_entry__.build())"
    out = run([os.path.join(BIN, "test_hook_lt")], {"LD_PRELOAD": LIB, "GEMMUL8_NUM_MOD_D": "15", "GEMMUL8_NUM_MOD_S": "8"})
    assert "hipblasLtMatmul<float>" in out and "hipblasLtMatmul<double>" in out and "passed to the native routine" in out
    assert "hipblasLtMatmul<float> with a BIAS epilogue == direct gemmul8_gemm + bias (bitwise)" in out


@pytest.mark.parametrize("kind", ["blocks", "moduli", "fp64sum"])
def test_ld_preload_hook_dist_opt_in_one_rank(kind):
    """GEMMUL8_DIST: the hook's sharded path (RCCL communicator bootstrapped from RANK / WORLD_SIZE / MASTER_* inside libgemmul8.so,
    plan cache, all-gather of C) with the one rank this box allows: the same C++ program, the same bits as the direct call."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = run([os.path.join(BIN, "test_hook")],
              {"LD_PRELOAD": LIB, "GEMMUL8_NUM_MOD_D": "15", "GEMMUL8_NUM_MOD_S": "8", "GEMMUL8_DIST": kind, "RANK": "0", "WORLD_SIZE": "1",
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert "bitwise" in out


def test_ld_preload_hook_rocblas_entry_points():
    """GEMMUL8_HOOK_ROCBLAS=1: rocblas_dgemm / rocblas_gemm_ex / rocblas_dgemm_strided_batched called directly by an application are emulated
    (error ~1e-16 of sum |a||b|, below anything the native FP64 routine leaves at k = 1500; counted by GEMMUL8_HOOK_STATS); without the
    variable the same program runs on rocBLAS untouched.  No counterpart in the reference (src/hook.cu:846-1055 hooks the BLAS-level names only)."""
    exe = os.path.join(BIN, "test_hook_rocblas")
    assert os.path.exists(exe), "tests/cpp not built (run __graft_entry__.build())"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    env.update({"LD_PRELOAD": LIB, "GEMMUL8_NUM_MOD_D": "15", "GEMMUL8_NUM_MOD_Z": "16", "GEMMUL8_HOOK_STATS": "1"})
    env.pop("GEMMUL8_MIN_FLOPS", None)
    on = subprocess.run([exe, "on"], env=dict(env, GEMMUL8_HOOK_ROCBLAS="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(on.stdout)
    assert on.returncode == 0 and "rocblas hook test passed (emulated)" in on.stdout, on.stdout[-2000:]
    assert "stats: emulated 4 GEMM calls" in on.stdout, on.stdout[-2000:]
    off = subprocess.run([exe, "off"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(off.stdout)
    assert off.returncode == 0 and "rocblas hook test passed (native)" in off.stdout, off.stdout[-2000:]
    assert "stats: emulated" not in off.stdout or "stats: emulated 0 GEMM calls" in off.stdout, off.stdout[-2000:]   # no call ever reached try_emulate
