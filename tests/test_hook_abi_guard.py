"""The opt-in interposition of rocBLAS's INTERNAL rocblas_internal_gemm_template (GEMMUL8_HOOK_ROCBLAS=1; what rocSOLVER's trailing updates
call) is bound to the rocBLAS releases it was tested with (oz2_hook.cpp rocblas_internal_abi_ok): the mangled name pins the parameter types,
not their meaning.  Checked here without a GPU and without rocBLAS: tests/cpp/fake_rocblas.cpp stands in for librocblas.so (version string from
the environment), tests/cpp/test_rocblas_abi_guard.cpp calls the template by its mangled name as rocSOLVER does.  No counterpart in the
reference (src/hook.cu hooks the cuBLAS / hipBLAS names only)."""
import os
import subprocess

import gemmul8_amd as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "build")


def run_guard(version, extra=None):
    exe = os.path.join(BUILD, "test_rocblas_abi_guard")
    fake = os.path.join(BUILD, "libfake_rocblas.so")
    assert os.path.exists(exe) and os.path.exists(fake), "tests/cpp not built (python -c 'import __graft_entry__ as g; g.build()')"
    env = dict(os.environ)
    env.update({"LD_PRELOAD": f"{g.LIB_PATH} {fake} /opt/rocm/lib/libamdhip64.so", "GEMMUL8_HOOK_ROCBLAS": "1", "GEMMUL8_NUM_MOD_D": "14",
                "FAKE_ROCBLAS_VERSION": version})
    env.update(extra or {})
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    fields = dict(kv.split("=") for kv in p.stdout.split())
    return {k: int(v) for k, v in fields.items()}, p.stderr


def test_version_allow_list():
    L = g.lib()
    L.gemmul8_hook_rocblas_version_tested.restype = int
    for v, want in ((b"5.0.2.20250912-42-1199-g2584e35062", 1), (b"5.2.0.5b515cf1bc", 1), (b"5.1.0", 0), (b"5.3.0", 0), (b"6.0.0", 0),
                    (b"4.4.1", 0), (b"", 0), (b"15.2.0", 0)):
        assert L.gemmul8_hook_rocblas_version_tested(v) == want, v
    assert L.gemmul8_hook_rocblas_version_tested(None) == 0


def test_untested_rocblas_release_takes_the_passthrough():
    out, err = run_guard("9.9.9.deadbeef")
    # handed to the next definition untouched: one native call, rc = the fake's 0, and the hook never asked for the handle's stream
    assert out == {"rc": 0, "native": 1, "stream_queries": 0}, (out, err)
    assert "is not one this build was tested with" in err and "9.9.9.deadbeef" in err


def test_tested_release_is_intercepted():
    """With a tested version string the same call enters the emulation (which then fails for want of a GPU in this container, or is served on
    a GPU box): either way the fake's native routine is NOT what ran first -- the hook asked for the stream and took the call."""
    out, err = run_guard("5.2.0.5b515cf1bc")
    assert out["stream_queries"] == 1 and out["native"] == 0, (out, err)
    assert "is not one this build was tested with" not in err


def test_override_switch():
    out, err = run_guard("9.9.9.deadbeef", {"GEMMUL8_ROCBLAS_ABI_UNCHECKED": "1"})
    assert out["stream_queries"] == 1 and out["native"] == 0, (out, err)
