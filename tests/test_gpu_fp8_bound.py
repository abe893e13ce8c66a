"""FP8 backend, accurate mode: the CONSEQUENCE of a bound that comes out low (VERDICT r2 weak #2 / ADVICE r2).

gfx950's v_mfma_scale_f32_16x16x128_f8f6f4 truncates products more than 13 binades below the largest of their group of 8
(profiles/archive/r02_f8_mfma_accumulation.txt), so the bound GEMM's non-negative sums come out low by up to ~8e-4 -- more than the
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


@pytest.mark.parametrize("N,dtype", [(6, np.float64), (8, np.float64), (12, np.float64), (6, np.float32), (8, np.float32)])
def test_fp8_bound_adversarial_no_wrap(N, dtype):
    import gemmul8_amd as g
    import gpu_util as gu
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


def build_case_cplx(N, k=1024, delta=1.2e-4):
    """Complex counterpart.  The bound of |Re C| is T = C0 + C1, C0 = sum (|Ar|-|Ai|)(|Br|-|Bi|) (both signs),
    C1 = sum |Ar||Bi| + |Ai||Br| (find_max.hpp:117-140).  Two kinds of groups of 8 consecutive k, every value an e4m3 number:
      X  k0: A = BIG, B = b            k1-7: A = SMALL, B = b/64          -> T += BIG b + 7 SMALL b/64, the small products of C0 vanish
      Y  k0: A = BIG, B = -i b         k1-7: A = SMALL (1 + i), B = -i b/64
         -> C0 -= BIG b (big NEGATIVE term), C1 += BIG b + 7 SMALL b/64 (the small products of the |Ar||Bi| GEMM vanish beside BIG b),
            T += 7 SMALL b/64 (= |Ai||Bi|; in C0 these positions are exact zeros: RU(|Ar| - |Ai|) = 0, nothing to lose)
    Engine: C0 = sumX_big - sumY_big, C1 = sumY_big; exact T = sumX_big (1 + e) + e sumY_big with e = 7 SMALL / (64 BIG): with sumY ~ sumX the
    engine's T is low by ~2 e = 12.8 * 2^-13, more than ku = 7 * 2^-13 + 4 (k+1) 2^-24 = 9 * 2^-13 at k = 1024 covers when it multiplies
    C0 + C1 only.  The sums are chosen so that the pre-floor value of the shift sits `delta` ABOVE an integer with the engine's value
    inflated the round-3 way and below it with the exact T.  Re C = T exactly (every product of Re C is non-negative), so a shift that is
    one too large wraps the CRT."""
    L = log2P_fp8(N)
    c = 1.0 + 6.0 * 2.0 ** -24
    ku_eng = 1.75 * 2.0 ** -11 + 4 * (k + 1) * 2.0 ** -24
    e = 7 * SMALL_A * 2.0 / (BIG_A * 128.0)
    ngroups = k // 8
    JMAX = 8                                              # b = 128 * 2^-j bound-plane units, j = 0..8 (small = 2 * 2^-j >= 2^-7: an e4m3 number)
    best = None
    for b_int in range(int(L) - 14, int(L) - 6):
        S_l = 2.0 ** (2.0 * (L - b_int - delta) / c) / (1 + ku_eng)     # engine-side T = sumX_big (round-3 combination)
        unit = BIG_A * 128.0 * 2.0 ** -JMAX
        units = int(round(S_l / unit))
        counts, rest = [], units
        for j in range(JMAX + 1):
            u = 2 ** (JMAX - j)
            cnt = rest // u
            counts.append(cnt)
            rest -= cnt * u
        nX = sum(counts)
        nY = int(0.97 * units * unit / (BIG_A * 128.0))   # Y groups all with b = 128: sumY_big = nY * 2^14 ~ 0.97 sumX_big
        if nX + nY <= ngroups and nY >= 8:
            best = (counts, nY, b_int)
            break
    assert best is not None, "no group decomposition found"
    counts, nY, b_int = best
    a = np.zeros(k, np.complex128)
    b = np.zeros(k, np.complex128)
    g = 0
    for j, cnt in enumerate(counts):
        for _ in range(cnt):
            a[8 * g] = BIG_A
            a[8 * g + 1:8 * g + 8] = SMALL_A
            b[8 * g] = 128.0 * 2.0 ** -j
            b[8 * g + 1:8 * g + 8] = 2.0 * 2.0 ** -j
            g += 1
    for _ in range(nY):
        a[8 * g] = BIG_A
        a[8 * g + 1:8 * g + 8] = SMALL_A * (1 + 1j)
        b[8 * g] = -128.0j
        b[8 * g + 1:8 * g + 8] = -2.0j
        g += 1
    ar, ai, br, bi = np.abs(a.real), np.abs(a.imag), np.abs(b.real), np.abs(b.imag)
    T = float(ar @ br + ai @ bi)
    C1 = float(ar @ bi + ai @ br)
    big = ar == BIG_A
    X_big = float((ar * big) @ br)
    Y_big = float((ar * big) @ bi)
    assert T > C1 and abs(T - (X_big * (1 + e) + e * Y_big)) < 1e-6 * T
    v_exact = L - 0.5 * c * np.log2(T)                                   # the shift any valid bound must not exceed
    v_r3 = L - 0.5 * c * np.log2(X_big * (1 + ku_eng))                   # engine value, round-3 combination
    v_new = L - 0.5 * c * np.log2((X_big - Y_big) + Y_big * (1 + ku_eng) + ku_eng * (abs(X_big - Y_big) + 2 * Y_big * (1 + ku_eng)))
    assert np.floor(v_r3) == np.floor(v_exact) + 1, (v_exact, v_r3)       # the window
    assert np.floor(v_new) <= np.floor(v_exact), (v_exact, v_new)
    m = n = 48
    A = np.tile(a / 128.0, (m, 1))                     # amax of every row / column = 1.0 -> sft0 = 7 -> bound planes = the patterns exactly
    B = np.tile((b / 128.0)[:, None], (1, n))
    exact = complex((a / 128.0) @ (b / 128.0))         # small dyadic rationals: exact in float64
    assert abs(exact.real - T / 128.0 ** 2) < 1e-9 * T / 128.0 ** 2
    return A, B, exact, v_exact, v_r3, v_new


@pytest.mark.parametrize("N,dtype", [(8, np.complex128), (12, np.complex128), (8, np.complex64)])
def test_fp8_bound_adversarial_complex(N, dtype):
    """Complex accurate mode on matrices built so that the engine's loss on the mixed-sign product decides the shift: with the default
    combination (mode 0) every element of C must equal the exact product; the round-3 default (mode 2: same ku, reference's combination)
    and the reference's formula (mode 1) are recorded (gpurun_out/fp8_bound_adversarial_complex.jsonl), not asserted."""
    import gemmul8_amd as g
    import gpu_util as gu
    A, B, exact, v_exact, v_r3, v_new = build_case_cplx(N)
    A, B = A.astype(dtype), B.astype(dtype)
    gu.bounds_case(A, B, N, backend=g.FP8)             # the guarantee itself: device maxima >= exact un-inflated max(T, C1)
    lib = g.lib()
    out = {}
    try:
        for mode in (0, 2, 1):
            assert lib.gemmul8_set_fp8_bound_mode(mode) >= 0
            Cd = gu.hip_gemm(A, B, N, fastmode=False, backend=g.FP8).astype(np.complex128)
            rel = np.abs(Cd - exact) / abs(exact)
            out[mode] = {"max_rel_err": float(rel.max()), "wrong_elements": int((rel > 1e-3).sum()), "elements": int(rel.size),
                         "re_sign_flips": int((Cd.real < 0).sum())}
    finally:
        lib.gemmul8_set_fp8_bound_mode(0)
    rec = {"N": N, "dtype": np.dtype(dtype).name, "k": A.shape[1], "prefloor_exact_T": float(v_exact), "prefloor_engine_round3": float(v_r3),
           "prefloor_engine_default": float(v_new), "default": out[0], "round3_default": out[2], "reference": out[1]}
    print("fp8 bound adversarial complex:", json.dumps(rec))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "fp8_bound_adversarial_complex.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    tol = 2.0 ** -20 if dtype == np.complex64 else 2.0 ** -40
    assert out[0]["wrong_elements"] == 0 and out[0]["max_rel_err"] < tol, f"default combination: {out[0]}"


def _fuzz_case(seed):
    """The operands tests/test_gpu_fuzz.py draws for `seed` (kept in step with it by test_fuzz_regressions_are_the_fuzz_cases)."""
    from test_gpu_fuzz import DIMS_K, DIMS_MN, _rand
    rng = np.random.default_rng(9000 + seed)
    dtype = [np.float64, np.float32, np.complex128, np.complex64][seed % 4]
    fp8 = (seed // 4) % 3 == 2
    is_f32 = dtype in (np.float32, np.complex64)
    N = int(rng.integers(2, 14 if is_f32 else 21))
    rng.integers(0, 2)
    m, n = (int(rng.choice(DIMS_MN)) for _ in range(2))
    k = int(rng.choice(DIMS_K))
    cplx = np.dtype(dtype).kind == "c"
    if fp8 or cplx:
        m, n = min(m, 257), min(n, 256)
    for opts in (["", "dma", "reg"], ["", "128", "256"], ["", "0", "1"], ["", "1", "2", "3"]):
        rng.choice(opts)
    opA = str(rng.choice(["N", "T", "C"] if cplx else ["N", "T"]))
    opB = str(rng.choice(["N", "T", "C"] if cplx else ["N", "T"]))
    phi = float(rng.choice([0.0, 1.0, 3.0]))
    A = _rand((m, k) if opA == "N" else (k, m), dtype, rng, phi)
    B = _rand((k, n) if opB == "N" else (n, k), dtype, rng, phi)
    if rng.integers(0, 3) == 0 and m > 2:
        (A if opA == "N" else A.T)[m // 2, :] = 0
    return A, B, N, opA, opB, fp8


@pytest.mark.parametrize("seed", [7388, 10113])
def test_fp8_bound_fuzz_regressions(seed):
    """Round 4: a 12000-seed run of tests/test_gpu_fuzz.py found these two real-type FP8 cases (one row / one column, exponent range
    phi = 3, k = 400) whose bound maxima came out 4.7e-5 / 2.2e-6 BELOW the exact sums with the relative inflation alone: the engine aligns a
    group of 8 products to the largest sum of the operands' exponent FIELDS, and an e4m3 subnormal carries the field of 2^-6
    (tools/ubench/f8_accum2.hip, profiles/archive/r04_f8_accum2.txt).  The default inflation now has an absolute part (oz2_gemm_f8.hip bound_kabs);
    bounds_case asserts exact un-inflated maximum <= device value."""
    import gemmul8_amd as g
    import gpu_util as gu
    A, B, N, opA, opB, fp8 = _fuzz_case(seed)
    assert fp8 and A.dtype.kind == "f"
    gu.bounds_case(A, B, N, opA=opA, opB=opB, backend=g.FP8)
    gu.parity_case(A, B, N, False, opA=opA, opB=opB, backend=g.FP8)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fp8_bound_subnormal_reference_product(dtype):
    """The mechanism of the cases above, built to its worst: in every group of 8 the product with the largest exponent-field sum is
    (subnormal 2^-9) x 2^7 -- field sum 1, true value 2^-2 -- and the seven others, 2^-6 x 1.75 * 2^-7 = 1.75 * 2^-13, lie just under the
    alignment grid 2^-12 of that reference and vanish: the engine's sum is 0.6 % low, six times what the relative inflation covers.  With the
    absolute part of the default inflation the device's maxima are >= the exact sums (bounds_case); with the round-3 inflation (mode 2)
    they are not (recorded)."""
    import gemmul8_amd as g
    import gpu_util as gu
    k, groups = 512, 63
    a = np.zeros(k)
    b = np.zeros(k)
    for gi in range(groups):
        a[8 * gi] = 2.0 ** -9
        b[8 * gi] = 128.0
        a[8 * gi + 1:8 * gi + 8] = 2.0 ** -6
        b[8 * gi + 1:8 * gi + 8] = 1.75 * 2.0 ** -7
    a[8 * groups] = 128.0                                  # the row's maximum (amax = 1.0 -> sft0 = 7 -> bound plane = a exactly); its partner is 0
    m, n = 48, 40
    A = np.tile(a / 128.0, (m, 1)).astype(dtype)
    B = np.tile((b / 128.0)[:, None], (1, n)).astype(dtype)
    exact = float(a @ b)
    try:
        gu.bounds_case(A, B, 8, backend=g.FP8, bound_mode=2)   # the round-3 inflation, selected on both sides
        below_with_round3 = False
    except AssertionError as e:
        below_with_round3 = "BELOW the exact sum" in str(e)
    gu.bounds_case(A, B, 8, backend=g.FP8)               # default (bound_mode = SAFE): the guarantee holds
    gu.parity_case(A, B, 8, False, backend=g.FP8)
    rec = {"case": "subnormal reference product", "dtype": np.dtype(dtype).name, "k": k, "exact_bound_sum": exact,
           "engine_loss_expected": 7 * 1.75 * 2.0 ** -13 * groups / exact, "round3_inflation_below_exact": below_with_round3}
    print("fp8 bound subnormal reference:", json.dumps(rec))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "fp8_bound_subnormal_reference.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    assert below_with_round3, "the round-3 inflation was expected to fall below the exact sum on this input"


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
    # complex: wide rows with random signs -- the mixed-sign product C0 is far smaller than the magnitudes of its terms
    Ac = A * np.exp(2j * np.pi * rng.random((m, k)))
    Bc = B * np.exp(2j * np.pi * rng.random((k, n)))
    gu.bounds_case(Ac, Bc, 10, backend=g.FP8)
    gu.bounds_case(Ac.astype(np.complex64), Bc.astype(np.complex64), 6, backend=g.FP8)


N_STRESS = int(os.environ.get("GEMMUL8_FP8_BOUND_STRESS", "24"))   # one-off sweeps: GEMMUL8_FP8_BOUND_STRESS=3000


@pytest.mark.parametrize("seed", range(N_STRESS))
def test_fp8_bound_guarantee_stress(seed):
    """Random operands aimed at the regime where the fuzz sweep found the bound below the exact sum: few rows / columns (the maximum cannot
    hide behind a well-aligned partner), exponent ranges of up to ~40 binades, many zeros, so that most entries of the e4m3 bound planes
    are subnormal or zero and the few large ones of A rarely meet the large ones of B.  Real and complex; bounds_case asserts
    exact un-inflated maximum <= device value."""
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(424242 + seed)
    cplx = seed % 3 == 2
    f32 = (seed // 3) % 2 == 1
    m = int(rng.choice([1, 2, 5, 17, 64]))
    n = int(rng.choice([1, 2, 3, 9, 48]))
    k = int(rng.choice([8, 64, 127, 400, 1024, 2300]))
    phi = float(rng.choice([2.0, 4.0, 6.0]))
    dens = float(rng.choice([0.1, 0.5, 1.0]))

    def mat(shape):
        x = (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape)) * (rng.random(shape) < dens)
        if cplx:
            x = x + 1j * (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape)) * (rng.random(shape) < dens)
        return x
    A, B = mat((m, k)), mat((k, n))
    A[0, 0] = A[0, 0] or 1.0          # no all-zero operand
    B[0, 0] = B[0, 0] or 1.0
    dt = (np.complex64 if f32 else np.complex128) if cplx else (np.float32 if f32 else np.float64)
    gu.bounds_case(A.astype(dt), B.astype(dt), 6 if f32 else 10, backend=g.FP8)
