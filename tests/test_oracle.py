"""CPU tests (no GPU): pin the oracle against (i) the reference's known-answer sample, (ii) exact
big-integer identities, (iii) OpenBLAS (BASELINE config 1: SGEMM 256^3, moduli=2), (iv) the
reference's workSize formula values quoted in SURVEY.md section 8."""
import json
import os
from fractions import Fraction

import numpy as np
import pytest

import bigint_ref as bi
import oracle_lib as ol

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def kat():
    d = json.load(open(os.path.join(GOLD, "kat_dgemm_4x5x3.json")))
    A = np.array([float.fromhex(x) for x in d["A"]]).reshape((d["m"], d["k"]), order="F")
    B = np.array([float.fromhex(x) for x in d["B"]]).reshape((d["k"], d["n"]), order="F")
    Cx = np.array([float.fromhex(x) for x in d["C_exact"]]).reshape((d["m"], d["n"]), order="F")
    return A, B, Cx


@pytest.mark.parametrize("backend,N", [(ol.INT8, 15), (ol.FP8, 13)])
@pytest.mark.parametrize("fast", [False, True])
def test_kat_sample(backend, N, fast):
    """sample/dgemm_cuBLAS_int8.cu (N=15) and dgemm_cuBLASLt_fp8.cu (N=13): Frobenius error vs hC_exact."""
    A, B, Cx = kat()
    C = ol.gemm(A, B, N, fastmode=fast, backend=backend)
    err = np.sqrt(((C - Cx) ** 2).sum())
    assert err < 4e-15, err  # native DGEMM gives ~1e-15 here; the emulation is at least as good
    # the exact product from rationals agrees with the reference's hC_exact to the last bit
    for i in range(4):
        for j in range(3):
            ex = sum(Fraction(A[i, l]) * Fraction(B[l, j]) for l in range(5))
            assert float(ex) == Cx[i, j]


@pytest.mark.parametrize("dtype,N,fast", [(np.float64, 2, True), (np.float64, 5, False), (np.float64, 8, True), (np.float64, 14, False),
                                          (np.float64, 16, False), (np.float64, 20, True), (np.float32, 2, True), (np.float32, 5, False),
                                          (np.float32, 8, True), (np.float32, 13, False)])  # float types: 2..13 moduli (gemmul8.hpp:30)
def test_bigint_identities_real(dtype, N, fast):
    rng = np.random.default_rng(1000 + N)
    m, n, k = 7, 5, 19
    A = ((rng.random((m, k)) - 0.5) * np.exp(2 * rng.standard_normal((m, k)))).astype(dtype)
    B = ((rng.random((k, n)) - 0.5) * np.exp(2 * rng.standard_normal((k, n)))).astype(dtype)
    A[2, 3] = 0.0
    C, it = ol.gemm(A, B, N, fastmode=fast, want_intermediates=True)
    mods = bi.MODULI_INT8[:N]
    Ai = bi.int_matrix(A, it["sftA"])
    Bi = bi.int_matrix(B.T, it["sftB"])
    # residues of the truncated integers, as int8
    for t, p in enumerate(mods):
        for i in range(m):
            for kk in range(k):
                assert np.int8(it["A_lo"][0, t, i, kk].view(np.int8)) == bi.as_int8(bi.sym(Ai[i][kk], p))
        for j in range(n):
            for kk in range(k):
                assert np.int8(it["B_lo"][0, t, j, kk].view(np.int8)) == bi.as_int8(bi.sym(Bi[j][kk], p))
    # exact integer product, residues, CRT uniqueness, final value
    P = 1
    for p in mods:
        P *= p
    for i in range(m):
        for j in range(n):
            ex = sum(Ai[i][kk] * Bi[j][kk] for kk in range(k))
            assert 2 * abs(ex) < P, "shift selection must keep |A'B'| < P/2"
            for t, p in enumerate(mods):
                assert int(it["C_mid"][t, j, i]) == bi.as_int8(bi.sym(ex, p))
            val = Fraction(ex) * Fraction(2) ** (int(it["sftA"][i]) + int(it["sftB"][j]))
            got = Fraction(float(C[i, j]))
            if val == 0:
                assert got == 0
            else:
                tol = Fraction(1, 2 ** 51) if dtype == np.float64 else Fraction(1, 2 ** 23)
                assert abs(got - val) <= tol * abs(val)


@pytest.mark.parametrize("N,fast", [(3, False), (9, True), (20, False)])
def test_bigint_identities_complex(N, fast):
    rng = np.random.default_rng(7 + N)
    m, n, k = 4, 3, 11
    A = (rng.standard_normal((m, k)) + 1j * rng.standard_normal((m, k))).astype(np.complex128)
    B = (rng.standard_normal((k, n)) + 1j * rng.standard_normal((k, n))).astype(np.complex128)
    C, it = ol.gemm(A, B, N, fastmode=fast, want_intermediates=True)
    mods = bi.MODULI_INT8[:N]
    Ar, Ai_ = bi.int_matrix(A.real, it["sftA"]), bi.int_matrix(A.imag, it["sftA"])
    Br, Bi_ = bi.int_matrix(B.real.T, it["sftB"]), bi.int_matrix(B.imag.T, it["sftB"])
    P = 1
    for p in mods:
        P *= p
    for i in range(m):
        for j in range(n):
            cr = sum(Ar[i][l] * Br[j][l] - Ai_[i][l] * Bi_[j][l] for l in range(k))
            ci = sum(Ar[i][l] * Bi_[j][l] + Ai_[i][l] * Br[j][l] for l in range(k))
            assert 2 * max(abs(cr), abs(ci)) < P
            for t, p in enumerate(mods):
                assert int(it["C_mid"][t, j, i, 0]) == bi.as_int8(bi.sym(cr, p))
                assert int(it["C_mid"][t, j, i, 1]) == bi.as_int8(bi.sym(ci, p))
            sc = Fraction(2) ** (int(it["sftA"][i]) + int(it["sftB"][j]))
            for val, got in ((cr * sc, C[i, j].real), (ci * sc, C[i, j].imag)):
                assert abs(Fraction(float(got)) - val) <= Fraction(1, 2 ** 50) * max(abs(val), abs(Fraction(float(abs(C[i, j])))) / 2 ** 3)
    # third plane = wrap(re + im) of the int8-cast residues
    for t, p in enumerate(mods):
        re = it["A_lo"][0, t].view(np.int8).astype(int)
        im = it["A_lo"][1, t].view(np.int8).astype(int)
        s = re + im
        h = p // 2
        w = np.where(s > h, s - p, np.where(s < -h, s + p, s))
        assert np.array_equal(it["A_lo"][2, t].view(np.int8), w.astype(np.int8))


def test_ops_and_axpby():
    """op(N/T/C) x alpha/beta variants (debug/test.cu:106-141) against numpy in high precision."""
    rng = np.random.default_rng(3)
    m, n, k = 33, 34, 47
    for dtype, N, tol in ((np.float64, 14, 1e-13), (np.complex128, 15, 1e-13), (np.float32, 8, 3e-6)):
        for opA in "NTC":
            for opB in "NTC":
                if dtype != np.complex128 and "C" in (opA, opB):
                    continue
                sa = (m, k) if opA == "N" else (k, m)
                sb = (k, n) if opB == "N" else (n, k)
                A = rng.standard_normal(sa).astype(dtype)
                B = rng.standard_normal(sb).astype(dtype)
                if dtype == np.complex128:
                    A = A + 1j * rng.standard_normal(sa)
                    B = B + 1j * rng.standard_normal(sb)
                C0 = rng.standard_normal((m, n)).astype(dtype)
                f = {"N": lambda x: x, "T": lambda x: x.T, "C": lambda x: x.conj().T}
                for alpha, beta in ((1, 0), (1, 1), (-1, 0), (-1, 1), (-1.5, 1.5)):
                    if dtype == np.complex128 and alpha == -1.5:
                        alpha, beta = -1.5 + 1.2j, 1.5 + 1.2j
                    C = ol.gemm(A, B, N, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0)
                    hp = np.complex256 if dtype == np.complex128 else np.longdouble
                    ref = alpha * (f[opA](A).astype(hp) @ f[opB](B).astype(hp)) + beta * C0.astype(hp)
                    scale = np.abs(f[opA](A)).astype(hp) @ np.abs(f[opB](B)).astype(hp) + np.abs(C0)
                    assert np.max(np.abs(C - ref) / scale) < tol


def test_config1_sgemm_256_vs_openblas():
    """BASELINE config 1: SGEMM 256x256x256, moduli=2, INT8 -- host plumbing check vs OpenBLAS (numpy).
    With N=2, P = 65280: operands are quantised to ~4 bits, so the check is normwise."""
    rng = np.random.default_rng(12345)
    A = (rng.random((256, 256)) - 0.5).astype(np.float32)
    B = (np.random.default_rng(54321).random((256, 256)) - 0.5).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    for fast in (True, False):
        C, it = ol.gemm(A, B, 2, fastmode=fast, want_intermediates=True)
        assert np.max(np.abs(C - ref) / bound) < 2.0 ** -3
        # exact identity: CRT of the two residue planes reproduces A'^T B' exactly
        Ai = np.array(bi.int_matrix(A, it["sftA"]), dtype=np.int64)
        Bi = np.array(bi.int_matrix(B.T, it["sftB"]), dtype=np.int64)
        ex = Ai @ Bi.T
        assert np.all(2 * np.abs(ex) < 65280)
        want = ex.astype(np.float64) * np.exp2(it["sftA"].astype(np.float64))[:, None] * np.exp2(it["sftB"].astype(np.float64))[None, :]
        assert np.array_equal(C.astype(np.float64), want.astype(np.float32).astype(np.float64))


def test_worksize_matches_survey_table():
    """SURVEY.md section 8 table (from gemmul8_real.hpp:14-46, gemmul8_complex.hpp:14-46)."""
    GiB = 2.0 ** 30
    tot, _, _ = ol.work_size(False, ol.INT8, 8192, 8192, 8192, 14)
    assert abs(tot / GiB - 2.875) < 0.001
    tot, _, _ = ol.work_size(False, ol.FP8, 16384, 16384, 16384, 6)
    assert abs(tot / GiB - 12.0) < 0.001
    tot, _, _ = ol.work_size(False, ol.INT8, 16384, 16384, 16384, 16)
    assert abs(tot / GiB - 13.0) < 0.001
    tot, _, _ = ol.work_size(True, ol.INT8, 8192, 8192, 8192, 20)
    assert abs(tot / GiB - 10.75) < 0.001


@pytest.mark.parametrize("backend,name,N", [(ol.INT8, "INT8", 15), (ol.FP8, "FP8", 13)])
@pytest.mark.parametrize("fast", [False, True])
def test_kat_result_bit_patterns_regression(backend, name, N, fast):
    """Regression pin of the restatement itself: result bits, shifts and residue planes of the reference's known-answer sample
    as committed in tests/golden/kat_result_bits.json (tools/make_kat_result_bits.py).  The GPU twin of this test
    (tests/test_gpu_parity.py::test_kat_result_bit_patterns) holds the HIP path to the same bits."""
    A, B, _ = kat()
    gold = json.load(open(os.path.join(GOLD, "kat_result_bits.json")))[f"{name}_N{N}_{'fast' if fast else 'accurate'}"]
    C, it = ol.gemm(A, B, N, fastmode=fast, backend=backend, want_intermediates=True)
    assert [float(x).hex() for x in C.flatten(order="F")] == gold["C"]
    assert it["sftA"].tolist() == gold["sftA"] and it["sftB"].tolist() == gold["sftB"]
    assert it["C_mid"].flatten().tolist() == gold["C_mid"]


@pytest.mark.parametrize("mode", [1, 0])
def test_fp8_bound_maxima_against_numpy(mode):
    """oz2_bound_maxima_f8 (exported for the GPU bound-maxima parity test; the function oz2_bound_shifts runs) against an
    independent numpy statement of find_max.hpp:82-96: decode e4m3, exact products and sums, one rounding to float,
    inflate by ku rounding up -- mode 1: the reference's ku = (k+1)*2^-24; mode 0: the product's engine-safe
    ku = 7*2^-13 + 4(k+1)*2^-24 plus, round 4, the absolute term kabs = 7*pad256(k)*2^-14 added rounding up
    (include/gemmul8_c.h, gemmul8_set_fp8_bound_mode)."""
    def e4m3(b):
        b = b.astype(np.int64)
        e, mnt = (b >> 3) & 15, b & 7
        v = np.where(e == 0, mnt * 2.0 ** -9, (8 + mnt) * 2.0 ** (e - 10))
        return np.where(b & 128, -v, v)
    rng = np.random.default_rng(4)
    m, n, k = 9, 6, 33
    A = ((rng.random((m, k)) - 0.5) * np.exp(rng.standard_normal((m, k)))).astype(np.float32)
    B = ((rng.random((k, n)) - 0.5) * np.exp(rng.standard_normal((k, n)))).astype(np.float32)
    oA, _ = ol.extract_bounds(A, "N", True, ol.FP8)
    oB, _ = ol.extract_bounds(B, "N", False, ol.FP8)
    ol.set_fp8_bound_mode(mode)
    try:
        rmax, cmax = ol.bound_maxima(oA, oB, ol.FP8)
    finally:
        ol.set_fp8_bound_mode(ol.FP8_BOUND_REFERENCE)   # the oracle's default
    ku = ol.fp8_bound_ku(k, mode)
    assert ku == ((k + 1) * 2.0 ** -24 if mode == 1 else 7 * 2.0 ** -13 + 4 * (k + 1) * 2.0 ** -24)   # exact in float32 at this k
    prod = (e4m3(oA[0]) @ e4m3(oB[0]).T).astype(np.float32).astype(np.float64)   # [m][n], exact in double
    def ru32(x):
        up = x.astype(np.float32)
        return np.where(up.astype(np.float64) < x, np.nextafter(up, np.float32(np.inf)), up)
    up = ru32(prod + prod * ku)                                                  # exact in double (48-bit product), one rounding up
    if mode == 0:
        kabs = 7.0 * ((k + 255) // 256 * 256) * 2.0 ** -14
        up = ru32(up.astype(np.float64) + kabs)                                  # second rounding up, like the device's __fadd_ru
    assert np.array_equal(rmax, up.max(axis=1)) and np.array_equal(cmax, up.max(axis=0))


@pytest.mark.parametrize("dtype,backend,N", [(np.float64, 0, 14), (np.complex64, 0, 7), (np.float32, 1, 6), (np.complex128, 1, 12)])
def test_oracle_on_submatrix_views_equals_the_contiguous_call(dtype, backend, N):
    """oracle_lib.gemm_embedded (explicit lda / ldb / ldc > rows, element-offset base pointers, C updated in place) against the same
    operands stored contiguously: identical planes, C_mid and window of C; nothing outside the window written (the checker of tests/test_gpu_ld.py)."""
    import oracle_lib as ol
    rng = np.random.default_rng(77)
    m, n, k = 13, 9, 21
    dt = np.dtype(dtype)

    def rnd(shape):
        x = rng.standard_normal(shape)
        if dt.kind == "c":
            x = x + 1j * rng.standard_normal(shape)
        return np.asfortranarray(x.astype(dt))
    for opA, opB, fast in (("N", "N", False), ("T", "C", True), ("C", "N", False)):
        A = rnd((m, k) if opA == "N" else (k, m))
        B = rnd((k, n) if opB == "N" else (n, k))
        C0 = rnd((m, n))
        alpha, beta = (0.75, -0.5)
        Cref, itr = ol.gemm(A, B, N, fastmode=fast, backend=backend, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0, want_intermediates=True)
        bufs = []
        for M, ex, off in ((A, 7, 1), (B, 1, 3), (C0, 64, 1)):
            ld = M.shape[0] + ex
            buf = np.full(off + ld * M.shape[1] + 5, 123.25, dt)
            buf[off:off + ld * M.shape[1]].reshape(M.shape[1], ld)[:, :M.shape[0]] = M.T
            bufs.append((buf, off, ld))
        (bA, oA, lda), (bB, oB, ldb), (bC, oC, ldc) = bufs
        before = bC.copy()
        it = ol.gemm_embedded(bA, oA, lda, bB, oB, ldb, bC, oC, ldc, m, n, k, N, fast, backend, opA, opB, alpha, beta)
        for key in ("sftA", "sftB", "A_lo", "B_lo", "C_mid"):
            assert np.array_equal(it[key], itr[key]), key
        win = bC[oC:oC + ldc * n].reshape(n, ldc)
        assert np.array_equal(np.ascontiguousarray(win[:, :m].T).view(np.uint8), np.ascontiguousarray(Cref).view(np.uint8))
        win[:, :m] = before[oC:oC + ldc * n].reshape(n, ldc)[:, :m]
        assert np.array_equal(bC.view(np.uint8), before.view(np.uint8)), "the oracle wrote outside the m x n window"


def test_oracle_fp8_bound_default_is_the_reference_formula():
    """the oracle's DEFAULT FP8 accurate-mode inflation is the reference's (k+1)*2^-24 (find_max.hpp:82-96), not the
    product's engine-safe one; out of the box it reproduces the numpy statement of the reference formula bit for bit."""
    assert ol.get_fp8_bound_mode() == ol.FP8_BOUND_REFERENCE
    rng = np.random.default_rng(40)
    m, n, k = 7, 5, 300
    A = (rng.random((m, k)) - 0.5).astype(np.float32)
    B = (rng.random((k, n)) - 0.5).astype(np.float32)
    oA, _ = ol.extract_bounds(A, "N", True, ol.FP8)
    oB, _ = ol.extract_bounds(B, "N", False, ol.FP8)
    rmax, _ = ol.bound_maxima(oA, oB, ol.FP8)
    prod = (ol.e4m3_decode(oA[0]) @ ol.e4m3_decode(oB[0]).T).astype(np.float32)
    ku = np.float32((k + 1) * 2.0 ** -24)
    exact = prod.astype(np.float64) * (1.0 + float(ku))          # ku * t + t in double, then rounded UP to float
    up = exact.astype(np.float32)
    up = np.where(up.astype(np.float64) < exact, np.nextafter(up, np.float32(np.inf)), up)
    assert np.array_equal(rmax, up.max(axis=1))
