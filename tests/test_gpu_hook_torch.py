"""-m gpu: the LD_PRELOAD drop-in under an unmodified PyTorch process.  Python loads its HIP runtime and hipBLAS late and
privately, so the preload library is the device-code-free shim libgemmul8_preload.so, which binds to libgemmul8.so on the
first intercepted call (gemmul8_amd/csrc/oz2_hook.cpp, OZ2_HOOK_SHIM)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "gemmul8_amd", "lib", "libgemmul8_preload.so")


def _run(extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hook_torch_demo.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"TFLOPS, normwise err ([0-9.e+-]+)", out.stdout)
    mb = re.search(r"bmm normwise err ([0-9.e+-]+)", out.stdout)
    assert m and mb, out.stdout
    assert "graph replay equals eager: True" in out.stdout, out.stdout   # the hooked call captured in a HIP graph and replayed
    return float(m.group(1)), float(mb.group(1))


def test_torch_matmul_is_emulated_under_ld_preload():
    assert os.path.exists(SHIM), "libgemmul8_preload.so not built"
    native, native_b = _run({})
    passthrough, passthrough_b = _run({"LD_PRELOAD": SHIM})                    # hook loaded, no GEMMUL8_NUM_MOD_D: native routine
    emulated, emulated_b = _run({"LD_PRELOAD": SHIM, "GEMMUL8_NUM_MOD_D": "18"})   # 18 moduli: more accurate than FP64 DGEMM
    assert (passthrough, passthrough_b) == (native, native_b)
    assert native > 1e-15 and emulated < 1e-15, (native, emulated)
    assert native_b > 4e-16 and emulated_b < 4e-16, (native_b, emulated_b)     # torch.bmm -> strided-batched hook
