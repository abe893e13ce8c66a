"""-m gpu: what the LD_PRELOAD hook reaches inside an unmodified PyTorch process (tools/hook_reach.py, INTEGRATION.md "What the
hook reaches").  (1) A blocked right-looking LU written with torch ops -- the HPL shape: its trailing updates are DGEMMs with k = the panel
width -- has them EMULATED under the hook (GEMMUL8_HOOK_STATS counts >= 90 % of the trailing-update flops), and its factorization residual is
no worse than the native run's.  (2) torch.linalg.lu_factor itself (hipSOLVER / rocSOLVER getrf): with the hipBLAS-level hooks alone (what
the reference's hook offers, src/hook.cu:846-1055) NONE of its flops are intercepted -- rocSOLVER's trailing updates call rocBLAS's exported
C++ template rocblas_internal_gemm_template<T>, not a BLAS entry point; with GEMMUL8_HOOK_ROCBLAS=1, which interposes that template as
well, at least half of the factorization's 2/3 n^3 flops run emulated and the residual is no worse than native.  Numbers:
gpurun_out/hook_reach.jsonl -> profiles/archive/r04_hook_reach.jsonl, INTEGRATION.md "What the hook reaches"."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "gemmul8_amd", "lib", "libgemmul8_preload.so")


def run(what, n, hooked, extra=None, nb=1024, rocblas=True):
    env = dict(os.environ)
    env.pop("GEMMUL8_MIN_FLOPS", None)
    env.pop("GEMMUL8_HOOK_ROCBLAS", None)
    if hooked:
        env.update({"LD_PRELOAD": SHIM, "GEMMUL8_NUM_MOD_D": "15", "GEMMUL8_HOOK_STATS": "1"})
        if rocblas:
            env["GEMMUL8_HOOK_ROCBLAS"] = "1"
    env.update(extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hook_reach.py"), "--n", str(n), "--nb", str(nb), "--what", what], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    recs = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    m = re.search(r"stats: emulated (\d+) GEMM calls \(([0-9.]+) TFLOP\), native (\d+) GEMM calls through the hooked entry points \(([0-9.]+) TFLOP\)", p.stderr)
    stats = {"emulated_calls": int(m.group(1)), "emulated_tflop": float(m.group(2)), "native_calls": int(m.group(3)), "native_tflop": float(m.group(4))} if m else None
    return recs, stats


def record(rec):
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "hook_reach.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    print("hook reach:", json.dumps(rec))


def test_blocked_lu_trailing_updates_are_emulated():
    n, nb = 8192, 1024
    (nat,), _ = run("blocked_lu", n, False, nb=nb)
    (hk,), stats = run("blocked_lu", n, True, nb=nb)
    record({"experiment": "blocked_lu", "native": nat, "hooked": hk, "hook_stats": stats})
    assert stats is not None and stats["emulated_calls"] >= n // nb - 1
    # every trailing update ran twice (one untimed warm-up call of the factorization + one timed): >= 90 % of 2 x the update flops were emulated
    assert stats["emulated_tflop"] >= 0.9 * 2 * hk["trailing_update_tflop"], (stats, hk)
    assert hk["residual"] <= max(nat["residual"] * 1.05, 0.5), (hk, nat)     # ||PA - LU|| / (||A|| n eps): no worse than native


def test_torch_lu_factor_trailing_updates_reached_through_rocblas_internal_template():
    n = 8192
    getrf = 2 / 3 * n ** 3 * 1e-12          # TFLOP of one factorization; tools/hook_reach.py runs it 3 times (1 warm-up + 2 timed)
    nat, _ = run("lu_factor", n, False)
    hk0, st0 = run("lu_factor", n, True, rocblas=False)     # hipBLAS / hipBLASLt names only: the reference's reach
    hk1, st1 = run("lu_factor", n, True, rocblas=True)
    record({"experiment": "torch.linalg.lu_factor", "n": n, "native": nat, "hooked_hipblas_names_only": hk0, "stats_hipblas_names_only": st0,
            "hooked_with_rocblas": hk1, "stats_with_rocblas": st1, "getrf_tflop_per_call": getrf})
    assert st0 is not None and st1 is not None
    inside = st1["emulated_tflop"] - st0["emulated_tflop"]   # what only the rocBLAS-level interposition reaches = rocSOLVER's own GEMMs
    assert inside >= 0.5 * 3 * getrf, (st0, st1, getrf)
    assert hk1[0]["residual"] <= max(nat[0]["residual"] * 1.05, 0.5), (hk1, nat)
    assert hk0[0]["residual"] <= max(nat[0]["residual"] * 1.05, 0.5), (hk0, nat)


def test_torch_solve_under_the_hook():
    n = 8192
    nat, _ = run("solve", n, False)
    hk, stats = run("solve", n, True)
    record({"experiment": "torch.linalg.solve", "n": n, "native": nat, "hooked": hk, "hook_stats": stats})
    assert hk[0]["residual"] <= max(nat[0]["residual"] * 1.05, 0.5), (hk, nat)
