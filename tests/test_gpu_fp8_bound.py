"""FP8 backend, accurate mode: the CONSEQUENCE of a bound that comes out low (VERDICT r2 weak #2 / ADVICE r2).

gfx950's v_mfma_scale_f32_16x16x128_f8f6f4 truncates products more than 13 binades below the largest of their group of 8
(profiles/r02_f8_mfma_accumulation.txt), so the bound GEMM's non-negative sums come out low by up to ~8e-4 -- more than the
reference's (k+1)*2^-24 inflation (GEMMul8/src/find_max.hpp:82-96) covers.  A low bound matters only through
floor(log2P - 0.5*log2(max)) (scaling_accu_real.hpp:6-18): if the exact maximum leaves the pre-floor value just BELOW an integer, the
low maximum pushes it across, the shift of that row AND column grows by one, |A'B'| reaches ~P and the CRT wraps: the element comes
back with the wrong sign, not a wrong last bit.

The matrices below are built to sit in that window: every value is exactly representable in e4m3 after the first scaling (no
round-up slack), every product is non-negative (no cancellation slack), every group of 8 consecutive k holds one big product and
seven products just under 2^-13 of it (maximal engine loss), and the number / size of the groups is chosen so that the pre-floor value
lands ~2e-4 below an integer with the exact maximum.  The final C is compared with the EXACT product (the values are small dyadic
rationals: float64 arithmetic on them is exact).
  * default inflation (gemmul8_set_fp8_bound_mode(0)): every element must be correct -- asserted;
  * the reference's formula (mode 1): recorded (gpurun_out/fp8_bound_adversarial.json) and printed, not asserted either way --
    it documents whether the hazard is real on this engine (DESIGN.md 4 quotes the numbers).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = 4096
BIG_A, SMALL_A = 128.0, 0.9375       # bound-plane units (value * 2^7); 0.9375 = 1.875 * 2^-1 is an e4m3 number
# per group of 8: products BIG_A * b and 7 * SMALL_A * (b / 64): each small product = 0.9155 * 2^-13 of the big one


def log2P_fp8(N):
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "gemmul8_amd", "csrc", "tables.inc")).read()
    body = re.search(r"GEMMUL8_LOG2P_FP8\[19\]\s*=\s*\{([^}]*)\}", txt).group(1)
    return [float.fromhex(x.strip().rstrip("f")) for x in body.split(",")][N - 2]


def build_case(N, delta=2.0e-4, k=K):
    """Column pattern whose exact bound sum S_x puts L - 0.5*c*log2(S_x*(1+ku_ref)) about `delta` below an integer while the engine's
    (all small products dropped) sum S_l puts it above.  Returns (A, B, exact product value, S_x, S_l)."""
    L = log2P_fp8(N)
    c = 1.0 + 6.0 * 2.0 ** -24
    ku = (k + 1) * 2.0 ** -24
    loss = 7 * SMALL_A * 2.0 / (BIG_A * 128.0)          # relative engine loss of a non-negative sum of such groups
    ngroups = k // 8
    # S_l = sum over groups of BIG_A * b_g with b_g = 128 * 2^-j, j = 0..6  (units 2^14 * 2^-j); at most ngroups groups
    best = None
    for b_int in range(int(L) - 14, int(L) - 8):
        # v_l = b_int + delta  ->  log2(S_l*(1+ku)) = 2 (L - b_int - delta) / c
        S_l = 2.0 ** (2.0 * (L - b_int - delta) / c) / (1 + ku)
        if not (2.0 ** 20 <= S_l <= ngroups * 2.0 ** 14 * 0.98):
            continue
        units = int(round(S_l / 2.0 ** 8))               # multiples of BIG_A * 2 (j = 6)
        counts = []
        for j in range(7):
            u = 2 ** (6 - j)
            counts.append(units // u if j == 0 else (units // u) % 2 if j < 6 else units % 2)
            if j == 0:
                units -= counts[0] * u
            else:
                units -= counts[j] * u
        if sum(counts) <= ngroups:
            best = counts
            break
    assert best is not None, "no group decomposition found"
    a_row = np.zeros(k)
    b_col = np.zeros(k)
    g = 0
    for j, cnt in enumerate(best):
        for _ in range(cnt):
            a_row[8 * g] = BIG_A
            a_row[8 * g + 1:8 * g + 8] = SMALL_A
            b_col[8 * g] = 128.0 * 2.0 ** -j
            b_col[8 * g + 1:8 * g + 8] = 2.0 * 2.0 ** -j
            g += 1
    S_x = float(a_row @ b_col)
    S_l = float((a_row * (a_row == BIG_A)) @ b_col)
    assert abs((S_x - S_l) / S_x - loss / (1 + loss)) < 1e-9
    v_x = L - 0.5 * c * np.log2(S_x * (1 + ku))
    v_l = L - 0.5 * c * np.log2(S_l * (1 + ku))
    assert np.floor(v_l) == np.floor(v_x) + 1, (v_x, v_l)   # the window: exact below the integer, engine value above
    m = n = 48
    A = np.tile(a_row / 128.0, (m, 1))                       # amax of every row = 1.0 -> sft0 = 7 -> bound plane = a_row exactly
    B = np.tile((b_col / 128.0)[:, None], (1, n))
    return A, B, S_x / 128.0 ** 2, v_x, v_l


@pytest.mark.parametrize("N", [6, 8, 12])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fp8_bound_adversarial_no_wrap(N, dtype):
    import gemmul8_amd as g
    import gpu_util as gu
    if dtype == np.float32 and N > 8:
        pytest.skip("float32 quantised integers would exceed 2^24 * ... keep the float case at N <= 8")
    A, B, exact, v_x, v_l = build_case(N)
    A, B = A.astype(dtype), B.astype(dtype)
    ref = np.full((A.shape[0], B.shape[1]), exact)
    lib = g.lib()
    out = {}
    try:
        for mode in (0, 1):
            assert lib.gemmul8_set_fp8_bound_mode(mode) >= 0
            C = gu.hip_gemm(A, B, N, fastmode=False, backend=g.FP8)
            rel = np.abs(C.astype(np.float64) - ref) / ref
            out[mode] = {"max_rel_err": float(rel.max()), "wrong_elements": int((rel > 1e-3).sum()), "elements": int(rel.size),
                         "sign_flips": int((C < 0).sum())}
    finally:
        lib.gemmul8_set_fp8_bound_mode(0)
    rec = {"N": N, "dtype": np.dtype(dtype).name, "k": K, "prefloor_exact_max": float(v_x), "prefloor_engine_max": float(v_l),
           "default_inflation": out[0], "reference_inflation": out[1]}
    print("fp8 bound adversarial:", json.dumps(rec))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "fp8_bound_adversarial.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    tol = 2.0 ** -20 if dtype == np.float32 else 2.0 ** -40
    assert out[0]["wrong_elements"] == 0 and out[0]["max_rel_err"] < tol, f"default inflation: {out[0]}"


def test_fp8_bound_never_below_exact_on_wide_rows():
    """Rows spanning > 20 binades with the big products scattered over the groups: the device's inflated maxima must not fall below
    the exactly accumulated ones (bounds_case asserts exactly that for real types)."""
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(7)
    m, n, k = 96, 80, 2048
    A = np.abs(rng.standard_normal((m, k))) * 2.0 ** rng.integers(-22, 1, (m, k))
    B = np.abs(rng.standard_normal((k, n))) * 2.0 ** rng.integers(-22, 1, (k, n))
    gu.bounds_case(A, B, 10, backend=g.FP8)
    gu.bounds_case(A.astype(np.float32), B.astype(np.float32), 6, backend=g.FP8)
