"""The reference's own correctness sweep (GEMMul8/debug/test.cu:14-18,106-141,247-299,327-345), restated as a parity grid: square problems
m = n = k = 32 ... 47 (nmin / nmax there), every moduli count the type accepts (the reference sweeps 6..14 / 7..19; here 2..13 / 2..20),
every op pair (real: N/T; complex: N/T/C), fast and accurate mode, its five (alpha, beta) pairs, its data (uniform in [0, 2): mt19937 / RAND_MAX).
The reference checks a relative-error threshold against the vendor GEMM (err > 1 prints a line); here every case is held BIT-EXACT against the
oracle: shifts within the App. C policy, residue planes, C_mid and C identical given the device's shifts.  The (alpha, beta) pair cycles with
the case index instead of multiplying the grid by five: every pair meets every size, op pair and moduli count several times over."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import SOAK  # noqa: E402

SIZES = range(32, 48) if SOAK else (32, 37, 41, 47)   # debug/test.cu sweeps every size 32..47: GEMMUL8_TEST_SOAK=1 does too (4x the time)
AB_REAL = [(1.0, 0.0), (1.0, 1.0), (-1.0, 0.0), (-1.0, 1.0), (-1.5, 1.5)]
AB_CPLX = [(1.0, 0.0), (1.0, 1.0), (-1.0, 0.0), (-1.0, 1.0), (-1.5 + 1.2j, 1.5 + 1.2j)]
REAL_OPS = [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")]
CPLX_OPS = [(a, b) for a in "NTC" for b in "NTC"]


def fill(shape, dtype, rng):
    x = rng.random(shape) * 2.0
    if np.dtype(dtype).kind == "c":
        x = x + 1j * rng.random(shape) * 2.0
    return x.astype(dtype)


def light_case(gu, ol, g, A, B, N, fast, opA, opB, alpha, beta, C0, backend):
    """gpu_util.parity_case without the separate bounds pass (one device call, two oracle calls)."""
    Cd, it = gu.hip_gemm(A, B, N, fastmode=fast, backend=backend, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0, want_intermediates=True)
    _, ito = ol.gemm(A, B, N, fastmode=fast, backend=backend, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0, want_intermediates=True)
    gu.shifts_close(it["sftA"], ito["sftA"], "sftA")
    gu.shifts_close(it["sftB"], ito["sftB"], "sftB")
    Co, ito = ol.gemm(A, B, N, fastmode=fast, backend=backend, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0, sftA_in=it["sftA"],
                      sftB_in=it["sftB"], want_intermediates=True)
    assert np.array_equal(it["A_lo"], ito["A_lo"]) and np.array_equal(it["B_lo"], ito["B_lo"]), "operand planes differ"
    assert np.array_equal(it["C_mid"], ito["C_mid"]), "C_mid planes differ"
    assert gu.bits_equal(Cd, Co), f"final C differs in {np.sum(Cd != Co)} elements"


def run_grid(dtype, opA, opB, backend_name, n_list):
    import gemmul8_amd as g
    import gpu_util as gu
    import oracle_lib as ol
    be = getattr(g, backend_name)
    cplx = np.dtype(dtype).kind == "c"
    pairs = AB_CPLX if cplx else AB_REAL
    rng = np.random.default_rng(9999)                   # SEED of debug/test.cu:19-21
    Afull, Bfull, Cfull = (fill((47, 47), dtype, rng) for _ in range(3))   # the reference carves every size out of one buffer
    idx = 0
    for s in SIZES:
        A, B, C0 = (np.ascontiguousarray(X.reshape(-1)[:s * s].reshape(s, s)) for X in (Afull, Bfull, Cfull))
        for N in n_list:
            for fast in (True, False):
                alpha, beta = pairs[idx % 5]
                idx += 1
                try:
                    if idx % 7 == 0:   # every seventh case also runs the accurate mode's bounds pass bit-exactly (gpu_util.parity_case)
                        gu.parity_case(A, B, N, fast, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0, backend=be)
                    else:
                        light_case(gu, ol, g, A, B, N, fast, opA, opB, alpha, beta, C0, be)
                except AssertionError as e:
                    raise AssertionError(f"size {s}, N={N}, {'fast' if fast else 'accurate'}, op {opA}{opB}, alpha={alpha}, beta={beta}: {e}") from e
    return idx


@pytest.mark.parametrize("opA,opB", REAL_OPS)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_reference_grid_real_int8(dtype, opA, opB):
    n_list = range(2, 21) if dtype == np.float64 else range(2, 14)
    assert run_grid(dtype, opA, opB, "INT8", n_list) == len(SIZES) * len(n_list) * 2


@pytest.mark.parametrize("opA,opB", CPLX_OPS)
@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_reference_grid_complex_int8(dtype, opA, opB):
    n_list = range(2, 21) if dtype == np.complex128 else range(2, 14)
    assert run_grid(dtype, opA, opB, "INT8", n_list) == len(SIZES) * len(n_list) * 2


@pytest.mark.parametrize("dtype,opA,opB", [(np.float64, "N", "N"), (np.float32, "T", "T"), (np.complex128, "C", "N"), (np.complex64, "N", "C")])
def test_reference_grid_fp8(dtype, opA, opB):
    """the same grid with -DUseFP8 (debug/test.cu:29-33): one op pair per type"""
    n_list = range(2, 21) if dtype in (np.float64, np.complex128) else range(2, 14)
    assert run_grid(dtype, opA, opB, "FP8", n_list) == len(SIZES) * len(n_list) * 2
