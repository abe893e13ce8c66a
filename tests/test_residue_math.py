"""CPU model of the floating-point residue reductions used by the GEMM epilogues and the quantise kernels (oz2_device.hpp:
mod_i32_sym_odd_f64, mod_small_sym_odd, residue_sym_bytes*) checked against exact integer arithmetic.  numpy float32 / float64
multiply, rint and fma-free differences are the same IEEE operations the kernels are compiled to (-ffp-contract=off)."""
import numpy as np
import pytest

INT8_MODULI = [256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193, 191, 181, 179, 173]


def sym_exact(a, p):
    r = np.mod(a, p)
    return np.where(r > (p - 1) // 2, r - p, r)


def model_odd(a, p):
    """mod_i32_sym_odd_f64: ONE FP64 quotient step, exact for every int32 (v_cvt_f64_i32, v_mul_f64, v_rndne_f64, v_fma_f64)."""
    invp = np.float64(1.0) / np.float64(p)
    x = a.astype(np.float64)
    q = np.rint(x * invp)
    r = x - q * np.float64(p)          # = fma(-q, p, x): both are exact here (integers below 2^53)
    assert np.all(r == np.rint(r))
    return r.astype(np.int64)


def test_moduli_table_matches():
    import ctypes, os
    # the list above is the generated table (first 20 INT8 moduli)
    inc = open(os.path.join(os.path.dirname(__file__), "..", "gemmul8_amd", "csrc", "tables.inc")).read()
    body = inc[inc.index("GEMMUL8_MODULI_INT8"):]
    body = body[body.index("{") + 1:body.index("}")]
    assert [int(x) for x in body.replace("\n", " ").split(",") if x.strip()] == INT8_MODULI


@pytest.mark.parametrize("p", [m for m in INT8_MODULI if m & 1])
def test_one_step_fp64_reduction_is_exact(p):
    rng = np.random.default_rng(p)
    lim = 2 ** 31 - 1
    parts = [rng.integers(-lim, lim + 1, size=2_000_000, dtype=np.int64),
             np.arange(-lim, -lim + 70_000, dtype=np.int64), np.arange(lim - 70_000, lim + 1, dtype=np.int64),
             np.arange(-70_000, 70_000, dtype=np.int64)]
    # every residue class at many quotients, including the rounding ties q*p +- (p-1)/2, (p+1)/2
    qs = np.concatenate([rng.integers(-(lim // p), lim // p, size=4000), [-(lim // p), lim // p - 1, 0, 1, -1]])
    rs = np.arange(-(p // 2) - 1, p // 2 + 2)
    parts.append(np.clip((qs[:, None] * p + rs[None, :]).ravel(), -lim, lim))
    a = np.concatenate(parts)
    got = model_odd(a, p)
    assert np.array_equal(got, sym_exact(a, p))


@pytest.mark.parametrize("p", [m for m in INT8_MODULI if m & 1])
def test_biased_accumulator_byte_dot_reduction_is_exact(p):
    """INT8 GEMM epilogue, odd moduli (oz2_gemm_i8_epi.hpp red(), RED_ODD; oz2_device.hpp mod_small_sym_u): the accumulator
    starts at -2^31, so its register read as unsigned is u = x + 2^31 for any int32 sum x; s = sum_j byte_j(u) (256^j mod p) +
    ((-2^31) mod p) (v_dot4_u32_u8), q from the low 24 bits of fma(float(s), RN(1/p), 2^23), r = s - q p (v_mad_i32_i24)."""
    rng = np.random.default_rng(1000 + p)
    lim = 2 ** 31 - 1
    parts = [rng.integers(-lim - 1, lim + 1, size=2_000_000, dtype=np.int64),
             np.arange(-lim - 1, -lim + 70_000, dtype=np.int64), np.arange(lim - 70_000, lim + 1, dtype=np.int64),
             np.arange(-70_000, 70_000, dtype=np.int64)]
    qs = np.concatenate([rng.integers(-(lim // p), lim // p, size=4000), [-(lim // p), lim // p - 1, 0, 1, -1]])
    rs = np.arange(-(p // 2) - 1, p // 2 + 2)
    parts.append(np.clip((qs[:, None] * p + rs[None, :]).ravel(), -lim - 1, lim))
    x = np.concatenate(parts)
    u = (x + 2 ** 31).astype(np.uint64)                       # the register contents
    w = [pow(256, j, p) for j in range(4)]                    # GemmArgs.dotw bytes (fill_common)
    c = (p - (2 ** 31) % p) % p                               # GemmArgs.dotc
    assert all(0 <= v < 256 for v in w)
    s = sum(((u >> np.uint64(8 * j)) & np.uint64(0xFF)).astype(np.int64) * w[j] for j in range(4)) + c
    assert s.max() < 2 ** 18
    invp = np.float32(1.0) / np.float32(p)
    prod = s.astype(np.float32).astype(np.float64) * np.float64(invp)        # exact in float64 (24 x 24 bits)
    qf = (prod + np.float64(8388608.0)).astype(np.float32)                   # the rounding of the fp32 fma (the float64 sum is off by < 2^-29, far inside the 1/(2p) margin to a tie)
    bits = qf.view(np.uint32).astype(np.int64) & 0xFFFFFF                    # v_mad_i32_i24 reads the low 24 bits (positive here)
    assert np.array_equal(bits, qf.astype(np.int64) - 8388608)
    r = s - bits * p
    assert np.array_equal(r, sym_exact(x, p))


@pytest.mark.parametrize("p", INT8_MODULI)
def test_round4_pair_form_and_p256_low_byte(p):
    """Round 4 (oz2_gemm_i8_epi.hpp red_odd_pair): the dot product's constant carries the bit pattern of 2^23 (dotc + 0x4B000000), so the
    register READ AS A FLOAT is 2^23 + s; s = that - 2^23 (exact), q from fma(s, RN(1/p), 2^23), v_mad_i32_i24(q bits, -p, dot-product bits):
    the LOW BYTE is the canonical residue's.  p = 256 runs through the same form since round 4 (one reduction form per kernel): its dot
    product is the accumulator's low byte (weights 1, 0, 0, 0; constant 0) and every quotient leaves the low byte in place."""
    rng = np.random.default_rng(2000 + p)
    lim = 2 ** 31 - 1
    x = np.concatenate([rng.integers(-lim - 1, lim + 1, size=1_000_000, dtype=np.int64), np.arange(-lim - 1, -lim + 70_000, dtype=np.int64),
                        np.arange(lim - 70_000, lim + 1, dtype=np.int64), np.arange(-70_000, 70_000, dtype=np.int64)])
    u = (x + 2 ** 31).astype(np.uint64)
    w = [pow(256, j, p) for j in range(4)]                    # fill_common: 1 | (256 % p) << 8 | ...
    c = (p - (2 ** 31) % p) % p
    s = sum(((u >> np.uint64(8 * j)) & np.uint64(0xFF)).astype(np.int64) * w[j] for j in range(4)) + c
    assert 0 <= s.min() and s.max() < 2 ** 18
    reg = (s + 0x4B000000).astype(np.uint32)                  # v_dot4_u32_u8 with the constant dotc + 0x4B000000: no carry into the exponent
    sm = reg.view(np.float32)
    assert np.array_equal(sm.astype(np.float64), 8388608.0 + s)
    sf = (sm.astype(np.float64) - 8388608.0).astype(np.float32)   # v_pk_add_f32: exact
    assert np.array_equal(sf.astype(np.int64), s)
    invp = np.float32(1.0) / np.float32(p)
    qm = (sf.astype(np.float64) * np.float64(invp) + np.float64(8388608.0)).astype(np.float32)   # v_pk_fma_f32: one rounding at unit spacing
    q24 = qm.view(np.uint32).astype(np.int64) & 0xFFFFFF      # v_mad_i32_i24 reads the low 24 bits of the pattern
    r = q24 * (-p) + reg.astype(np.int64)                     # ... and adds the dot product's PATTERN (2^23's exponent bits included)
    want = sym_exact(x, p) if p & 1 else x                    # p = 256: the residue's byte is the accumulator's low byte
    assert np.array_equal(r & 0xFF, want & 0xFF)


def test_short_k_form_p256_low_byte():
    """RED_ODD_SMALL with p = 256 (round 4: no separate form): x - (2^22 + q) 256 keeps the low byte of x for ANY q."""
    x = np.arange(-(1 << 23), (1 << 23) + 1, dtype=np.int64)
    invp = np.float32(1.0) / np.float32(256)
    qf = (x.astype(np.float32).astype(np.float64) * np.float64(invp) + np.float64(12582912.0)).astype(np.float32)
    low24 = qf.view(np.uint32).astype(np.int64) & 0xFFFFFF
    assert np.array_equal((x - low24 * 256) & 0xFF, x & 0xFF)


@pytest.mark.parametrize("p", [m for m in INT8_MODULI if m & 1])
def test_mod_small_sym_u_exhaustive(p):
    """oz2_device.hpp mod_small_sym_u over its whole documented domain 0 <= s < 2^22 (the epilogue's byte-dot sums stay below 2^18)."""
    s_ = np.arange(0, 1 << 22, dtype=np.int64)
    invp = np.float32(1.0) / np.float32(p)
    qf = (s_.astype(np.float32).astype(np.float64) * np.float64(invp) + np.float64(8388608.0)).astype(np.float32)
    q = qf.view(np.uint32).astype(np.int64) & 0xFFFFFF
    assert np.array_equal(s_ - q * p, sym_exact(s_, p))


@pytest.mark.parametrize("p", [m for m in INT8_MODULI if m & 1])
def test_short_k_accumulator_reduction_low_byte(p):
    """INT8 GEMM epilogue for K <= 512 (oz2_gemm_i8.hip RED_ODD_SMALL): |x| <= 512 * 127^2 < 2^23; the quotient is read from the low 24
    bits of fma(float(x), RN(1/p), 1.5 * 2^23) (= 2^22 + q for either sign of q) and v_mad_i32_i24 returns x - (2^22 + q) p, whose LOW
    BYTE is the canonical residue's -- the only byte the epilogue stores.  EXHAUSTIVE over |x| <= 2^23; beyond that bound the single
    quotient is no longer safe (|x| |RN(1/p) - 1/p| reaches 1/(2p)): the test also pins where the first wrong byte appears."""
    def low_bytes(x):
        invp = np.float32(1.0) / np.float32(p)
        prod = x.astype(np.float32).astype(np.float64) * np.float64(invp)
        qf = (prod + np.float64(12582912.0)).astype(np.float32)      # one rounding at unit spacing (the float64 sum is off by < 2^-28)
        low24 = qf.view(np.uint32).astype(np.int64) & 0xFFFFFF
        assert np.array_equal(low24, (1 << 22) + (qf.astype(np.int64) - 12582912))
        return (x - low24 * p) & 0xFF
    lim = 1 << 23
    x = np.arange(-lim, lim + 1, dtype=np.int64)
    assert np.array_equal(low_bytes(x), sym_exact(x, p) & 0xFF)
    assert 512 * 127 * 127 < lim
    if p == 255:   # the kernel's K bound is not slack: the form fails a little above 2^23
        y = np.arange(lim, 2 * lim, dtype=np.int64)
        bad = np.nonzero(low_bytes(y) != (sym_exact(y, p) & 0xFF))[0]
        assert len(bad) and y[bad[0]] == 8454907


@pytest.mark.parametrize("p", [m for m in INT8_MODULI if m & 1])
def test_one_step_small(p):
    a = np.arange(-65535, 65536, dtype=np.int64)
    invp = np.float32(1.0) / np.float32(p)
    got = a - np.rint(a.astype(np.float32) * invp).astype(np.int64) * p
    assert np.array_equal(got, sym_exact(a, p))


@pytest.mark.parametrize("p", INT8_MODULI)
def test_byte_dot_residue(p):
    """oz2_device.hpp residue_sym_bytes: sum_i byte_i(M) * (256^i mod p), one fp32 quotient step, sign applied last."""
    rng = np.random.default_rng(1000 + p)
    M = np.concatenate([rng.integers(0, 2 ** 53, size=400_000, dtype=np.int64), np.arange(0, 70_000, dtype=np.int64),
                        (2 ** 53 - 1 - np.arange(0, 1000)).astype(np.int64),
                        # rounding ties / range ends of every residue class
                        (rng.integers(0, 2 ** 53 // p, size=3000)[:, None] * p + np.arange(-(p // 2) - 1, p // 2 + 2)[None, :]).ravel().clip(0)])
    c = [pow(256, i, p) for i in range(7)]
    s = np.zeros_like(M)
    for i in range(7):
        s += ((M >> (8 * i)) & 0xFF) * c[i]
    assert s.max() < 2 ** 19
    invp = np.float32(1.0) / np.float32(p)
    r = s - np.rint(s.astype(np.float32) * invp).astype(np.int64) * p
    want = sym_exact(M, p)
    if p & 1:
        assert np.array_equal(r, want)
        assert np.array_equal(-r, sym_exact(-M, p))
    else:  # p = 256: compare as int8 bytes (+128 and -128 are the same byte)
        assert np.array_equal(r.astype(np.int8), want.astype(np.int8))
        assert np.array_equal((-r).astype(np.int8), sym_exact(-M, p).astype(np.int8))


@pytest.mark.parametrize("p", INT8_MODULI)
def test_byte_dot_residue_signed_fma(p):
    """oz2_device.hpp residue_sym_bytes_e0: bytes of M or of the 56-bit two's complement 2^56 - M (negative values) plus
    k56 = (-2^56 mod p); quotient from one fma(float(s), 1/p, 2^23) whose low 24 bits are rint(s/p)."""
    rng = np.random.default_rng(2000 + p)
    M = np.concatenate([rng.integers(1, 2 ** 53, size=300_000, dtype=np.int64), np.arange(1, 70_000, dtype=np.int64),
                        (2 ** 53 - 1 - np.arange(0, 1000)).astype(np.int64),
                        (rng.integers(1, 2 ** 53 // p, size=3000)[:, None] * p + np.arange(-(p // 2) - 1, p // 2 + 2)[None, :]).ravel()])
    c = [pow(256, i, p) for i in range(7)]
    k56 = (p - pow(2, 56, p)) % p
    invp = np.float64(np.float32(1.0) / np.float32(p))
    for neg in (False, True):
        Mt = (2 ** 56 - M) if neg else M
        s = np.full_like(M, k56 if neg else 0)
        for i in range(7):
            s += ((Mt >> (8 * i)) & 0xFF) * c[i]
        assert s.max() < 2 ** 20
        # fma in fp32 == exact double sum rounded once to fp32 (all operands fit)
        qf = (s.astype(np.float64) * invp + 8388608.0).astype(np.float32)
        q = qf.view(np.int32).astype(np.int64) & 0xFFFFFF
        r = s - q * p
        want = sym_exact(-M if neg else M, p)
        if p & 1:
            assert np.array_equal(r, want)
        else:
            assert np.array_equal(r.astype(np.int8), want.astype(np.int8))


@pytest.mark.parametrize("p", INT8_MODULI)
def test_byte_dot_residue_shifted_120bit(p):
    """oz2_device.hpp shifted_bytes + residue_sym_bytes128: bytes of M*2^E (E < 64) or of 2^120 - M*2^E, four dot4, one fma."""
    rng = np.random.default_rng(3000 + p)
    c = [pow(256, i, p) for i in range(15)]
    k120 = (p - pow(2, 120, p)) % p
    invp = np.float64(np.float32(1.0) / np.float32(p))
    Ms = [int(x) for x in rng.integers(1, 2 ** 53, size=4000)] + [1, 2 ** 53 - 1, 2 ** 52, p, p - 1, (p + 1) // 2, 255, 256]
    Es = [int(x) for x in rng.integers(0, 64, size=len(Ms))]
    Es[-8:] = [63, 63, 0, 1, 40, 26, 7, 8]
    for neg in (False, True):
        s_list, want = [], []
        for M, E in zip(Ms, Es):
            X = M << E
            Xt = (2 ** 120 - X) if neg else X
            s = (k120 if neg else 0) + sum(((Xt >> (8 * i)) & 0xFF) * c[i] for i in range(15))
            s_list.append(s)
            v = -X if neg else X
            r = v % p
            want.append(r - p if r > (p - 1) // 2 else r)
        s = np.array(s_list, dtype=np.int64)
        assert s.max() < 2 ** 20
        qf = (s.astype(np.float64) * invp + 8388608.0).astype(np.float32)
        q = qf.view(np.int32).astype(np.int64) & 0xFFFFFF
        r = s - q * p
        want = np.array(want, dtype=np.int64)
        if p & 1:
            assert np.array_equal(r, want)
        else:
            assert np.array_equal(r.astype(np.int8), want.astype(np.int8))


FP8_MODULI = [1089, 1024, 961, 841, 625, 529, 511, 509, 503, 499, 491, 487, 481, 479, 467, 463, 461, 457, 449, 443]


@pytest.mark.parametrize("p", FP8_MODULI)
def test_wide_byte_dot_residue_fp8_moduli(p):
    """FP8 moduli (up to 1089): constants split c = 32 c_hi + c_lo, two byte sums, one fma; 56-bit (E = 0) and 120-bit forms.
    p = 1024: the tie -512 becomes +512 (residues live in (-p/2, p/2])."""
    rng = np.random.default_rng(4000 + p)
    c = [pow(256, i, p) for i in range(15)]
    invp = np.float64(np.float32(1.0) / np.float32(p))

    def sym(v):
        r = v % p
        return r - p if r > p // 2 else r

    def finish(s):
        s = np.array(s, dtype=np.int64)
        assert s.max() < 2 ** 23
        qf = (s.astype(np.float64) * invp + 8388608.0).astype(np.float32)
        q = qf.view(np.int32).astype(np.int64) & 0xFFFFFF
        r = s - q * p
        if p % 2 == 0:
            r = np.where(r == -(p // 2), p // 2, r)
        return r

    Ms = [int(x) for x in rng.integers(1, 2 ** 53, size=3000)] + [1, 2 ** 53 - 1, p, p // 2, p // 2 + 1, 512, 1536, 3 * 512 * 7]
    Ms += [int(q) * p + r for q in rng.integers(0, 2 ** 53 // p - 1, size=300) for r in (p // 2 - 1, p // 2, p // 2 + 1, p - 1, 0, 1)]
    Ms = [m for m in Ms if 0 < m < 2 ** 53]
    for neg in (False, True):
        k56, k120 = (p - pow(2, 56, p)) % p, (p - pow(2, 120, p)) % p
        # 56-bit form
        s_list = []
        for M in Ms:
            Mt = (2 ** 56 - M) if neg else M
            lo = (k56 if neg else 0) + sum(((Mt >> (8 * i)) & 0xFF) * (c[i] & 31) for i in range(7))
            hi = sum(((Mt >> (8 * i)) & 0xFF) * (c[i] >> 5) for i in range(7))
            s_list.append(lo + 32 * hi)
        assert max(s_list) < 2 ** 21 + 2048
        assert np.array_equal(finish(s_list), np.array([sym(-M if neg else M) for M in Ms]))
        # 120-bit form
        Es = [int(x) for x in rng.integers(0, 64, size=len(Ms))]
        s_list = []
        for M, E in zip(Ms, Es):
            X = M << E
            Xt = (2 ** 120 - X) if neg else X
            lo = (k120 if neg else 0) + sum(((Xt >> (8 * i)) & 0xFF) * (c[i] & 31) for i in range(15))
            hi = sum(((Xt >> (8 * i)) & 0xFF) * (c[i] >> 5) for i in range(15))
            s_list.append(lo + 32 * hi)
        assert np.array_equal(finish(s_list), np.array([sym(-(M << E) if neg else (M << E)) for M, E in zip(Ms, Es)]))



# ---------------------------------------------------------------------------------------------------------------------
# Round 3: the quantise kernels' two-level FLOATING-POINT reduction (oz2_scale.hip: emit4_mod_float, residue_from_small).
#   level 1 (FP64, per pair P = p_a * p_b):  q = rint(xs * RN(1/P)),  R = fma(-q, P, xs)   -- exact: R = xs - q P, |R| small
#   level 2 (FP32, per modulus):            qf = fma(float(R), RN_f32(1/p), 1.5 * 2^23)   -- rounds to 1.5 * 2^23 + rint(R/p)
#                                           r  = R - q p                                   -- v_mad_i32_i24 on the low 24 bits of qf
# Modelled with exact integer / rational arithmetic and numpy's IEEE operations, compared with the exact symmetric residue.
FP8_MODULI = [1089, 1024, 961, 841, 625, 529, 511, 509, 503, 499, 491, 487, 481, 479, 467, 463, 461, 457, 449, 443]


def _level1(xs_int, P):
    """xs_int: python ints (|xs| < 2^53 -> one step; larger -> two steps as the kernel's BIG path).  Returns exact R (python ints)."""
    invP = np.float64(1.0) / np.float64(P)
    out = []
    for x in xs_int:
        xf = np.float64(x)
        assert int(xf) == x                      # the kernel's xs is an exactly representable double
        q = int(np.rint(xf * invP))
        R = x - q * P                            # = fma(-q, P, xs) exactly when representable: checked next
        assert abs(R) < 2 ** 53
        if abs(x) >= 2 ** 53:
            assert abs(R) <= P // 2 + abs(x) * 2.0 ** -51 + 2
            q2 = int(np.rint(np.float64(R) * invP))
            R = R - q2 * P
        assert abs(R) <= P // 2 + 2, (x, P, R)
        out.append(R)
    return out


def _level2(R, p):
    """Single-fma quotient on a signed |R| < 2^21 and the i24 multiply-add; returns the int the kernel derives (full value)."""
    from fractions import Fraction
    invp = np.float32(1.0) / np.float32(p)
    exact = Fraction(int(R)) * Fraction(float(invp)) + Fraction(12582912)      # R * RN(1/p) + 1.5 * 2^23, exactly
    fl = exact.numerator // exact.denominator
    rem = exact - fl
    qf = fl + (1 if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl % 2 == 1) else 0)   # RN-even at unit spacing
    assert 2 ** 23 <= qf < 2 ** 24
    low24 = qf - 2 ** 23                          # low 24 bits of the float's BIT PATTERN: the mantissa field (bit 23 = exponent LSB = 0) = 2^22 + q
    assert 0 <= low24 < 2 ** 23
    raw = (low24 * (-p) + int(R))                 # v_mad_i32_i24
    return raw + (p << 22)                        # the kernel adds p << 22 (mod 2^32) where it needs the value; low byte unchanged


def _values(rng, P, big):
    lim = 2 ** 53 - 1
    vals = [int(v) for v in rng.integers(-lim, lim, size=300)]
    vals += [int(q) * P + d for q in rng.integers(-(lim // P), lim // P, size=60) for d in (-(P // 2) - 1, -(P // 2), -1, 0, 1, P // 2, P // 2 + 1)]
    vals += [0, 1, -1, lim, -lim, P, -P, P // 2, -(P // 2)]
    vals = [v for v in vals if abs(v) <= lim]
    if big:   # |xs| >= 2^53: 53-bit mantissa times a power of two (what trunc(ldexp(x, s)) produces for num_moduli > 15)
        for _ in range(200):
            m = int(rng.integers(2 ** 52, 2 ** 53)) * (1 if rng.integers(0, 2) else -1)
            vals.append(m * 2 ** int(rng.integers(1, 38)))
    return vals


@pytest.mark.parametrize("moduli,big", [(INT8_MODULI, True), (FP8_MODULI, True)])
def test_float_domain_two_level_residue(moduli, big):
    rng = np.random.default_rng(len(moduli) + moduli[0])
    pairs = [(moduli[t], moduli[t + 1]) for t in range(0, 19)] + [(p, 1) for p in moduli]   # every consecutive pair (any t_begin) and the odd tail
    for pa, pb in pairs:
        P = pa * pb
        xs = _values(rng, P, big)
        Rs = _level1(xs, P)
        for p in (pa, pb):
            if p == 1:
                continue
            for x, R in zip(xs, Rs):
                r = _level2(R, p)
                assert (r - x) % p == 0, (x, p, r)
                if p & 1:
                    assert abs(r) <= (p - 1) // 2, (x, p, r)          # the canonical symmetric representative
                else:
                    assert abs(r) <= p // 2, (x, p, r)                # p = 256: +-128 is the same int8 byte; p = 1024: -512 is mapped to +512


@pytest.mark.parametrize("moduli", [INT8_MODULI, FP8_MODULI])
def test_level2_quotient_exhaustive(moduli):
    """Level 2 of the float-domain residue (oz2_scale.hip residue_from_small) EXHAUSTIVELY over |R| <= 2^20, which contains every
    level-1 remainder (|R| <= P/2 + 2, P = product of two moduli <= 1089 * 1024): vectorised form of _level2."""
    R = np.arange(-(1 << 20), (1 << 20) + 1, dtype=np.int64)
    Rf = R.astype(np.float32).astype(np.float64)                     # |R| < 2^24: exact
    for p in moduli:
        invp = np.float32(1.0) / np.float32(p)
        qf = (Rf * np.float64(invp) + np.float64(12582912.0)).astype(np.float32)   # float64 sum off by < 2^-29: far inside the margin to a tie
        low24 = qf.view(np.uint32).astype(np.int64) & 0xFFFFFF
        r = R - low24 * p + (p << 22)
        assert np.all((r - R) % p == 0), p
        if p & 1:
            assert np.array_equal(r, sym_exact(R, p)), p
        else:
            assert np.all(np.abs(r) <= p // 2), p



# ---------------------------------------------------------------------------------------------------------------------------------
# FP8 GEMM epilogue (oz2_gemm_f8.hip f8_epilogue_mod, round 4): ONE reduction form for every modulus, q = ceil(x / p - 1/2),
# r = x - q p in (-p/2, p/2]: the symmetric residue for odd p (x / p - 1/2 is never an integer there), the reference's representative
# +512 of the tie for p = 1024 (conv_hi2mid / mod.hpp keep (-512, 512]).
def _sym_half_open(x, p):
    r = np.mod(x, p)
    return np.where(r > p // 2, r - p, r)          # (-p/2, p/2]: odd p -> [-(p-1)/2, (p-1)/2]; p = 1024 -> [-511, 512]


@pytest.mark.parametrize("p", FP8_MODULI)
def test_fp8_epilogue_small_reduction_exhaustive(p):
    """red_small: q = ceilf(fmaf(float(v), RN32(1/p), -0.5f)), r = v - q p; exhaustive over |v| < 2^18 (the combined value k0 R0 + k1 R1 +
    k2 R2 and the complex differences stay below that)."""
    v = np.arange(-(1 << 18), (1 << 18) + 1, dtype=np.int64)
    invp = np.float32(1.0) / np.float32(p)
    y = (v.astype(np.float64) * np.float64(invp) - 0.5).astype(np.float32)      # the product is exact in float64 (19 x 24 bits), the sum too: ONE rounding, as in the fma
    q = np.ceil(y.astype(np.float64)).astype(np.int64)
    r = v - q * p
    assert np.array_equal(r, _sym_half_open(v, p))
    if p & 1:
        assert np.array_equal(r, sym_exact(v, p))


@pytest.mark.parametrize("p", FP8_MODULI)
def test_fp8_epilogue_accumulator_reduction(p):
    """red_acc (round 5): the LOOSE fp32 residue of an exact integer accumulator |c| <= 2^24 (k <= 65536 with |a|, |b| <= 16):
    t = RN32(c * RN32(1/p)), q = rint(t) (ties to even), r = fma(-q, p, c).  Modelled in exact integer arithmetic with numpy's float32 product (one
    rounding): r == c (mod p), |r| <= (1/2 + 2^-7) p (quotients reach 2^15.2 for the smallest moduli: half an ulp of the product plus the error of RN32(1/p)), q p and r are exactly representable where the fma forms them, and r fits the int16 scratch
    planes; then the combination k0 R0 + k1 R1 + k2 R2 of three loose residues stays inside the exhaustively tested range of red_small (2^18),
    whose result is the canonical residue of the exact value."""
    rng = np.random.default_rng(3000 + p)
    lim = 1 << 24
    qs = np.concatenate([rng.integers(-(lim // p), lim // p, size=400), [-(lim // p), lim // p - 1, 0, 1, -1]])
    rs = np.arange(-(p // 2) - 2, p // 2 + 3)
    xs = np.concatenate([np.clip((qs[:, None] * p + rs[None, :]).ravel(), -lim, lim), rng.integers(-lim, lim + 1, size=200000),
                         np.arange(-lim, -lim + 4000), np.arange(lim - 4000, lim + 1)])
    xs = np.unique(xs)
    invp = np.float32(1.0) / np.float32(p)
    t = xs.astype(np.float32) * invp                       # |c| <= 2^24 is exact in float32; one rounding in the product
    q = np.rint(t.astype(np.float64)).astype(np.int64)     # rint of a float32 value: ties to even, as v_rndne_f32
    r = xs - q * p
    assert np.all(np.mod(r - xs, p) == 0)
    assert np.max(np.abs(r)) <= (0.5 + 2.0 ** -7) * p + 1e-9, (p, np.max(np.abs(r)))
    assert np.max(np.abs(q * p)) < (1 << 26) and np.max(np.abs(r)) < 32768
    # the combination of three loose residues (worst coefficients: Karatsuba 240, -15, 16 for p <= 511; squares s, s, 1) and its canonical reduction
    t_idx = FP8_MODULI.index(p)
    k0, k1, k2 = ((int(round(p ** 0.5)),) * 2 + (1,)) if t_idx < 6 else (240, -15, 16)
    bound = (abs(k0) + abs(k1) + abs(k2)) * np.max(np.abs(r))
    assert bound < (1 << 18), (p, bound)
    a = rng.choice(r, size=(3, 50000))
    v = k0 * a[0] + k1 * a[1] + k2 * a[2]
    y = (v.astype(np.float64) * np.float64(invp) - 0.5).astype(np.float32)
    got = v - np.ceil(y.astype(np.float64)).astype(np.int64) * p
    assert np.array_equal(got, _sym_half_open(v, p))


@pytest.mark.parametrize("p", FP8_MODULI)
def test_fp8_fused_three_segment_tile_loop(p):
    """Round 5, FP6 kernel (csrc/oz2_gemm_f8_epi.hpp f8_fill_planes which = 7 / 8): the three products S0, S1, S2 of a modulus run as ONE tile loop; behind
    segment 1 the accumulators become m1 x loose(acc), behind segment 2 m2 x loose(acc), and the canonical residue of gam x loose(final acc) must equal
    the canonical residue of the reference's value -- s (S0 + S1) + S2 for the square moduli (mod.hpp:176-189), 256 S0 + 16 (S2 - S0 - S1) + S1 for the
    Karatsuba ones (mod.hpp:117-129) -- while every accumulator stays an exact integer below 2^24 at the largest padded k the fused form is used for
    (65024: |S| <= 65024 * 256)."""
    rng = np.random.default_rng(5000 + p)
    t = FP8_MODULI.index(p)
    lim = 65024 * 256
    n = 200000
    S = rng.integers(-lim, lim + 1, size=(3, n))
    S[:, :8] = np.array([[lim, lim, lim], [-lim, -lim, -lim], [lim, -lim, lim], [-lim, lim, -lim], [0, 0, 0], [1, -1, 1], [lim, 0, -lim], [p, p, p]]).T
    invp = np.float32(1.0) / np.float32(p)

    def loose(c):
        assert np.max(np.abs(c)) < (1 << 24), (p, np.max(np.abs(c)))          # exact in float32, as an MFMA accumulator
        q = np.rint((c.astype(np.float32) * invp).astype(np.float64)).astype(np.int64)
        return c - q * p
    if t < 6:
        s = int(round(p ** 0.5))
        m1, m2, gam = 1, s, 1
        value = s * (S[0] + S[1]) + S[2]
    else:
        inv16 = next(i for i in range(1, p) if (16 * i) % p == 1)
        m2 = ((p - 15) * inv16) % p
        m2 = m2 - p if m2 > p // 2 else m2
        m1, gam = -16, 16
        value = 256 * S[0] + 16 * (S[2] - S[0] - S[1]) + S[1]
    acc = m1 * loose(S[0]) + S[1]
    acc = m2 * loose(acc) + S[2]
    v = gam * loose(acc)
    assert np.max(np.abs(v)) < (1 << 18)
    y = (v.astype(np.float64) * np.float64(invp) - 0.5).astype(np.float32)
    got = v - np.ceil(y.astype(np.float64)).astype(np.int64) * p
    assert np.array_equal(got, _sym_half_open(value, p))


# ---------------------------------------------------------------------------------------------------------------------------------
# FP6 plane writers (oz2_scale.hip, round 5): the residue, the split and the e2m3 codes in fp32 (put_f6_planes_f, quantise_f6_pair_kernel) against the
# integer chain of the e4m3 writer (residue_from_small + fp8_split_sq / fp8_split_kara + f6_code), exhaustively over every level-1 remainder.
FP8_SQRT = [33, 32, 31, 29, 25, 23]


@pytest.mark.parametrize("t", range(20))
def test_fp6_writer_float_chain_equals_integer_chain(t):
    p = FP8_MODULI[t]
    R = np.arange(-(1 << 20), (1 << 20) + 1, dtype=np.int64)
    Rf = R.astype(np.float32)
    invp = np.float32(1.0) / np.float32(p)
    magic = np.float32(12582912.0)
    # integer chain: q from the low 24 bits of fma(R, 1/p, 1.5 * 2^23), r = R - q p, the even-modulus tie to +p/2
    qf = (Rf.astype(np.float64) * np.float64(invp) + np.float64(magic)).astype(np.float32)
    q_int = (qf.view(np.uint32).astype(np.int64) & 0xFFFFFF) - (1 << 22)
    r_int = R - q_int * p
    if p % 2 == 0:
        r_int = np.where(r_int == -(p // 2), p // 2, r_int)
    # float chain: q = fma(..) - magic (exact: both are integers below 2^24), r = fma(-q, p, R) (exact), the same tie rule
    q_f = qf - magic
    r_f = (-q_f.astype(np.float64) * np.float64(p) + Rf.astype(np.float64)).astype(np.float32)
    if p % 2 == 0:
        r_f = np.where(r_f == np.float32(-0.5 * p), np.float32(0.5 * p), r_f)
    assert np.array_equal(r_f.astype(np.int64), r_int) and np.all(np.abs(r_int) <= p // 2)
    # split: squares hi = rint(r / s) (float multiply by RN(1/s), as fp8_split_sq), lo = r - s hi; Karatsuba: hi = sign(r) ceil(|r| / 16), lo = r - 16 hi
    if t < 6:
        s = FP8_SQRT[t]
        inv = np.float32(1.0) / np.float32(s)
        hi_f = np.rint(r_f * inv)
        lo_f = (r_f.astype(np.float64) - np.float64(s) * hi_f.astype(np.float64)).astype(np.float32)
        a = r_int.astype(np.float32)
        hi_i = np.rint(a * inv).astype(np.int64)
        lo_i = r_int - s * hi_i
        pieces_f, pieces_i = [hi_f, lo_f], [hi_i, lo_i]
    else:
        hi_f = np.copysign(np.ceil(np.abs(r_f) * np.float32(0.0625)), r_f)
        lo_f = (r_f.astype(np.float64) - 16.0 * hi_f.astype(np.float64)).astype(np.float32)
        qa = (np.abs(r_int) + 15) >> 4
        hi_i = np.where(r_int < 0, -qa, qa)
        lo_i = r_int - 16 * hi_i
        pieces_f, pieces_i = [hi_f, lo_f, hi_f + lo_f], [hi_i, lo_i, hi_i + lo_i]
    for pf_, pi_ in zip(pieces_f, pieces_i):
        assert np.array_equal(pf_.astype(np.int64), pi_), t
        assert np.abs(pi_).max() <= 16, (t, np.abs(pi_).max())     # every piece is an exact e2m3 code: sign << 5 | |v| (value v / 8)
        # the code the four-k writer builds in fp32: v < 0 ? 32 - v : v;  the hardware pack of the lane-per-fragment writer returns sign << 5 | |v| (tools/ubench/cvt_fp6.hip)
        code_f = np.where(pf_ < 0, np.float32(32.0) - pf_, pf_).astype(np.int64)
        code_i = np.where(pi_ < 0, 32 - pi_, pi_)
        assert np.array_equal(code_f, code_i) and code_i.max() <= 48


def test_fp6_panel_image_codec_round_trip():
    """tests/gpu_util.py f6_plane_image / f6_plane_values (the test-side encoder / decoder of the FP6 panel images, csrc/oz2_gemm_f6.hip layout): inverse of
    each other on ragged row counts, several K-steps and a last row block that is not a multiple of 16 rows."""
    import gpu_util as gu
    rng = np.random.default_rng(7)
    for rows, rows_img, kp in ((300, 512, 256), (333, 333, 384), (70, 80, 128), (16, 16, 128)):
        v = rng.integers(-16, 17, size=(rows, kp)).astype(np.int8)
        nb = (rows_img + 255) // 256
        img = gu.f6_plane_image(v, rows_img, kp, nb * 256 * (kp // 4 * 3))
        assert np.array_equal(gu.f6_plane_values(img, rows, rows_img, kp), v)


@pytest.mark.parametrize("p", INT8_MODULI)
def test_magic_bias_reduction(p):
    """INT8 GEMM epilogue for K <= 256 (round 6, oz2_gemm_i8_epi.hpp RED_MAGIC): the accumulators start at 0x4B400000, the bit pattern of the float
    1.5 * 2^23, and |x| <= 256 * 128^2 = 2^22: the INTEGER sum pattern 0x4B400000 + x IS the float 1.5 * 2^23 + x (ulp 1 on [2^23, 2^24], ends included), so
    float(x) = as_float(acc) - 1.5 * 2^23 exactly -- the quotient is RED_ODD_SMALL's, bit for bit -- and v_mad_i32_i24 on the biased register returns the
    canonical residue's LOW BYTE (the bias has no bit below 2^22).  EXHAUSTIVE over |x| <= 2^22."""
    lim = 1 << 22
    x = np.arange(-lim, lim + 1, dtype=np.int64)
    acc = (np.int64(0x4B400000) + x).astype(np.uint32)
    f = acc.view(np.float32)
    assert np.array_equal(f.astype(np.float64), 12582912.0 + x)                          # the pattern is the float, at both ends too
    xf = (f.astype(np.float64) - 12582912.0).astype(np.float32)                          # v_pk_add_f32: exact
    assert np.array_equal(xf.astype(np.int64), x)
    invp = np.float32(1.0) / np.float32(p)
    qf = (xf.astype(np.float64) * np.float64(invp) + np.float64(12582912.0)).astype(np.float32)     # v_pk_fma_f32: one rounding at unit spacing
    qf_small = (x.astype(np.float32).astype(np.float64) * np.float64(invp) + np.float64(12582912.0)).astype(np.float32)
    assert np.array_equal(qf.view(np.uint32), qf_small.view(np.uint32))                  # RED_ODD_SMALL's quotient
    low24 = qf.view(np.uint32).astype(np.int64) & 0xFFFFFF
    got = (low24 * (-p) + acc.astype(np.int64)) & 0xFF                                   # v_mad_i32_i24 (low 24 bits of src0, 32-bit add), low byte
    want = (sym_exact(x, p) & 0xFF) if p != 256 else (x & 0xFF)
    assert np.array_equal(got, want)
