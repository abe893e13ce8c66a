"""-m gpu parity on SUB-MATRIX VIEWS: lda > rows(A), ldb > rows(B), ldc > m, base pointers that are only element-aligned, C updated in
place inside a larger matrix -- the calling pattern of the hook's LU / QR trailing updates (src/hook.cu:609-730 forwards the caller's
lda / ldb / ldc unchanged; include/gemmul8.hpp:107-112; debug/test.cu:247-299 is the reference's own op x alpha/beta matrix).

Every case is compared BIT FOR BIT with the CPU oracle running on the same bytes (oracle_lib.gemm_embedded): operand planes, C_mid,
the m x n window of C, and every byte of the enclosing C buffer outside the window (must be untouched); A's and B's buffers must
come back unchanged.  All four types, both backends and both FP8 plane formats, op in {N, T, C}^2, (alpha, beta) in
{(1, 0), (-1, 1), (0.75, -0.5)}, ld = rows + {1, 7, 64}, base offset {1, 3} elements, both scaling modes.

Also here: the FP8 K-concatenation gate (csrc/oz2_driver.hip f8_concat_ok: k <= 32768) from both sides in both plane formats."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LD_EXTRA = (1, 7, 64)
BASE_OFF = (1, 3)
AXPBY = ((1, 0), (-1, 1), (0.75, -0.5))
OPS9 = [a + b for a in "NTC" for b in "NTC"]
# moduli counts: float types take 2..13 (tests/test_cabi.py); FP8 backend: config 3's N = 6 for single, 12 for double
NMOD = {("int8", "float32"): 8, ("int8", "float64"): 14, ("int8", "complex64"): 7, ("int8", "complex128"): 15,
        ("fp8", "float32"): 6, ("fp8", "float64"): 12, ("fp8", "complex64"): 6, ("fp8", "complex128"): 12}


def rand(shape, dtype, rng, phi=1.0):
    x = (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))
    if np.dtype(dtype).kind == "c":
        x = x + 1j * (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))
    return x.astype(dtype)


def stored(rows, cols, op):
    return (rows, cols) if op == "N" else (cols, rows)


@pytest.mark.parametrize("ops", OPS9)
@pytest.mark.parametrize("dtype", ["float32", "float64", "complex64", "complex128"])
@pytest.mark.parametrize("variant", ["int8", "fp6", "e4m3"])
def test_submatrix_views_bit_exact(variant, dtype, ops, monkeypatch):
    import gemmul8_amd as g
    import gpu_util as gu
    from conftest import SOAK
    opA, opB = ops
    if not SOAK and dtype.startswith("float") and "C" in ops and ops != "CC":
        pytest.skip("real types: op C is op T (one C/C case kept per type and backend); GEMMUL8_TEST_SOAK=1 runs all nine")
    dt = np.dtype(dtype)
    backend = g.INT8 if variant == "int8" else g.FP8
    if variant != "int8":
        gu.setknob(monkeypatch, "GEMMUL8_FP8_PLANES", variant)
    N = NMOD[("int8" if variant == "int8" else "fp8", dtype)]
    seed = OPS9.index(ops) * 131 + ["float32", "float64", "complex64", "complex128"].index(dtype) * 17 + ["int8", "fp6", "e4m3"].index(variant)
    rng = np.random.default_rng(seed)
    m, n, k = 70 + seed % 5, 66 + seed % 3, 45 + seed % 7      # n >= 64: the FP6 panel images are in use for variant "fp6"
    A = rand(stored(m, k, opA), dt, rng)
    B = rand(stored(k, n, opB), dt, rng)
    C0 = rand((m, n), dt, rng)
    # every (alpha, beta) appears in each test, every (ld_extra, base_off) pair in every pair of neighbouring tests; the three buffers get DIFFERENT ld / offsets
    combos = list(itertools.product(LD_EXTRA, BASE_OFF))
    for i, (alpha, beta) in enumerate(AXPBY):
        ca, cb, cc = (combos[(seed + 2 * i + s) % 6] for s in (0, 1, 3))
        fast = (i + seed) % 2 == 1
        if dt.kind == "c" and i == 2:
            alpha, beta = 0.75 - 0.25j, -0.5 + 1.5j
        _, fmt = gu.parity_case_embedded(A, B, C0, N, fast, opA, opB, alpha, beta, backend, (ca[0], cb[0], cc[0]), (ca[1], cb[1], cc[1]), rng)
        assert fmt == (1 if variant == "fp6" else 0)


@pytest.mark.parametrize("variant,dtype", [("int8", "float64"), ("int8", "complex128"), ("fp6", "float32"), ("e4m3", "float32"), ("fp6", "complex64")])
def test_submatrix_views_bit_exact_on_the_large_tile_kernels(variant, dtype, monkeypatch):
    """The same through the 256 x 256 persistent kernels and the LDS-staged row-strided readers: several tiles, ragged edges, k not a
    multiple of anything, in-place update with beta != 0."""
    import gemmul8_amd as g
    import gpu_util as gu
    dt = np.dtype(dtype)
    backend = g.INT8 if variant == "int8" else g.FP8
    if variant != "int8":
        gu.setknob(monkeypatch, "GEMMUL8_FP8_PLANES", variant)
    N = NMOD[("int8" if variant == "int8" else "fp8", dtype)]
    from conftest import SOAK
    rng = np.random.default_rng(4242)
    m, n, k = (521, 389, 777) if SOAK else (300, 261, 520)   # (several 256-tiles with ragged edges either way; the scalar oracle is what takes the time)
    cases = (("N", "N", False, (1, 7, 64), (1, 3, 1), (-1, 1)), ("T", "N", True, (7, 64, 1), (3, 1, 3), (0.75, -0.5)),
             ("N", "T", False, (64, 1, 7), (1, 1, 3), (1, 0)), ("C", "C", True, (7, 7, 1), (3, 3, 1), (-1, 1)))
    for (opA, opB, fast, ex, off, ab) in (cases if SOAK else (cases[0], cases[3])):
        A = rand(stored(m, k, opA), dt, rng)
        B = rand(stored(k, n, opB), dt, rng)
        C0 = rand((m, n), dt, rng)
        gu.parity_case_embedded(A, B, C0, N, fast, opA, opB, ab[0], ab[1], backend, ex, off, rng)


@pytest.mark.parametrize("k", [32768, 32769])
@pytest.mark.parametrize("planes", ["fp6", "e4m3"])
def test_fp8_k_concatenation_gate_from_both_sides(planes, k, monkeypatch):
    """csrc/oz2_driver.hip f8_concat_ok: the square moduli's C0 + C1 products are K-concatenated for k <= 32768 and run as separate
    launches above; the FP6 fused three-segment loop has its own bound (kp <= 65024).  k = 32768 and 32769 in both plane formats, m = n = 9
    (e4m3 planes; n < 64 keeps format 0 whatever the knob says) and n = 64 (the FP6 images), bit-exact against the oracle."""
    import gemmul8_amd as g
    import gpu_util as gu
    gu.setknob(monkeypatch, "GEMMUL8_FP8_PLANES", planes)
    rng = np.random.default_rng(k + len(planes))
    m, n = (9, 9) if planes == "e4m3" else (9, 64)
    A, B = rand((m, k), np.float32, rng), rand((k, n), np.float32, rng)
    _, it = gu.hip_gemm(A, B, 6, backend=g.FP8, want_intermediates=True)
    assert it["lo_format"] == (1 if planes == "fp6" else 0)
    gu.parity_case(A, B, 6, False, backend=g.FP8)
    gu.parity_case(A, B, 7, True, backend=g.FP8)
