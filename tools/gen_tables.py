#!/usr/bin/env python3
"""Generate the CRT constant tables of the Ozaki-II emulation from big-integer rules.

Nothing here is copied from the reference: every table is derived from the moduli lists
with Python big integers / mpmath and written as C99 hex-float literals.  The rules are the
ones stated in SURVEY.md App. B; tools/verify_tables_vs_reference.py re-parses the
reference's literals (GEMMul8/src/table.hpp:80-151, 161-203, 209-258, 277-838) in the build
container and checks that every generated value is bit-identical.

Outputs (identical content, two consumers):
    gemmul8_amd/csrc/tables.inc   -- product (HIP library)
    oracle/tables.inc             -- CPU oracle (test infrastructure)

Rules
-----
moduli           INT8: 256,255,253,...   FP8: 1089,1024,961,...        (table.hpp:12-53)
P(N)             = prod_{t<N} p_t
Pneg_hi, Pneg_lo = RN(-P), RN(-P - Pneg_hi)                              (table.hpp:80-126)
invP             = RN(1/P)                                               (table.hpp:138-151)
w_t(N)           = q_t * P/p_t,  q_t = (P/p_t)^-1 mod p_t in [0,p_t)
qPi1[N][t]       = RN(w_t)                                               (table.hpp:277-327)
qPi2[N][t]       = {h, l}: h = w_t with everything below bit
                   2^(emax - (53 - ceil(log2 rho))) chopped (emax = max_t bitlen(w_t),
                   rho = sum_t floor(p_t/2)), l = RN(w_t - h)            (table.hpp:332-550)
log2P[N]         = RD_f32(log2(P-1)/2 - 0.5), except the two N=2 entries which the
                   reference defines by literal (+1 float ulp-ish off the formula); they are
                   carried as data (table.hpp:166,187).
pow2mod[t][e]    = symmetric residue of 2^e mod p_t, e in [0,64)  (own layout; the
                   reference's mod_pow2, table.hpp:209-258, is the same function of (p,e)).
"""
import math
import os
import struct
import sys
from fractions import Fraction

import mpmath

MODULI = {
    "INT8": [256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193, 191, 181, 179, 173],
    "FP8": [1089, 1024, 961, 841, 625, 529, 511, 509, 503, 499, 491, 487, 481, 479, 467, 463, 461, 457, 449, 443],
}
# thresholds (common.hpp:15-27): CRT sum kept in one double up to P_IS_DOUBLE moduli
P_IS_DOUBLE = {"INT8": 6, "FP8": 5}
# FP8: first six moduli are squares (or 2^10): residues split as a = s*hi + lo
SQRT_MODULI = [33, 32, 31, 29, 25, 23]
# log2P for N=2 is defined by literal in the reference (does not follow the formula)
LOG2P_N2 = {"INT8": float.fromhex("0x1.dfd1ecp+2"), "FP8": float.fromhex("0x1.316baep+3")}


def rn_double(x: Fraction) -> float:
    """Round-to-nearest-even of an exact rational to binary64."""
    if x == 0:
        return 0.0
    sign = -1 if x < 0 else 1
    x = abs(x)
    # exponent e with 2^e <= x < 2^(e+1)
    e = x.numerator.bit_length() - x.denominator.bit_length()
    if Fraction(2) ** e > x:
        e -= 1
    if Fraction(2) ** (e + 1) <= x:
        e += 1
    scale = Fraction(2) ** (52 - e)
    y = x * scale  # in [2^52, 2^53)
    fl = y.numerator // y.denominator
    rem = y - fl
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)):
        fl += 1
    return sign * math.ldexp(float(fl), e - 52)


def crt_weights(mods):
    P = 1
    for p in mods:
        P *= p
    w = []
    for p in mods:
        Pi = P // p
        q = pow(Pi % p, -1, p)
        w.append(q * Pi)
    return P, w


def qpi2_split(mods, w):
    rho = sum(p // 2 for p in mods)
    keep = 53 - math.ceil(math.log2(rho))
    emax = max(x.bit_length() for x in w)
    cut = emax - keep  # bits below 2^cut are chopped
    out = []
    for x in w:
        h = (x >> cut) << cut if cut > 0 else x
        out.append((float(h), rn_double(Fraction(x - h))))
        assert int(float(h)) == h
    return out


def rd_float32(x: mpmath.mpf) -> float:
    """Round an mpmath value DOWN to binary32."""
    f = struct.unpack("f", struct.pack("f", float(x)))[0]
    if mpmath.mpf(f) > x:
        # step one float32 ulp down
        bits = struct.unpack("I", struct.pack("f", f))[0]
        bits = bits - 1 if f > 0 else bits + 1
        f = struct.unpack("f", struct.pack("I", bits))[0]
    assert mpmath.mpf(f) <= x
    return f


def sym(r, p):
    r %= p
    return r - p if r > p // 2 else r


def build(backend):
    mods = MODULI[backend]
    t = {"P_hi": [], "P_lo": [], "invP": [], "qpi1": [], "qpi2h": [], "qpi2l": [], "log2P": []}
    mpmath.mp.prec = 400
    for N in range(2, 21):
        P, w = crt_weights(mods[:N])
        hi = rn_double(Fraction(-P))
        lo = rn_double(Fraction(-P) - Fraction(hi))
        t["P_hi"].append(hi)
        t["P_lo"].append(lo)
        t["invP"].append(rn_double(Fraction(1, P)))
        t["qpi1"].append([rn_double(Fraction(x)) for x in w] + [0.0] * (20 - N))
        if N > P_IS_DOUBLE[backend]:
            s = qpi2_split(mods[:N], w)
        else:
            s = [(0.0, 0.0)] * N
        t["qpi2h"].append([a for a, _ in s] + [0.0] * (20 - N))
        t["qpi2l"].append([b for _, b in s] + [0.0] * (20 - N))
        if N == 2:
            t["log2P"].append(LOG2P_N2[backend])
        else:
            t["log2P"].append(rd_float32(mpmath.log(mpmath.mpf(P - 1), 2) / 2 - mpmath.mpf("0.5")))
    t["pow2"] = [[sym(pow(2, e, p), p) for e in range(64)] for p in mods]
    return t


def fmt_d(x):
    return float(x).hex()


def emit(path):
    lines = []
    A = lines.append
    A("// GENERATED by tools/gen_tables.py -- do not edit.")
    A("// CRT constants of the Ozaki-II emulation, derived from the moduli lists with big integers.")
    A("// Index convention: [N-2] for per-num_moduli rows (N = 2..20), [t] for per-modulus entries.")
    A("#pragma once")
    for be in ("INT8", "FP8"):
        t = build(be)
        mods = MODULI[be]
        A(f"static const int GEMMUL8_MODULI_{be}[20] = {{{', '.join(map(str, mods))}}};")
        A(f"static const double GEMMUL8_PNEG_HI_{be}[19] = {{{', '.join(map(fmt_d, t['P_hi']))}}};")
        A(f"static const double GEMMUL8_PNEG_LO_{be}[19] = {{{', '.join(map(fmt_d, t['P_lo']))}}};")
        A(f"static const double GEMMUL8_INVP_{be}[19] = {{{', '.join(map(fmt_d, t['invP']))}}};")
        A(f"static const float GEMMUL8_LOG2P_{be}[19] = {{{', '.join(float(x).hex() + 'f' for x in t['log2P'])}}};")
        for name, key in (("QPI1", "qpi1"), ("QPI2_HI", "qpi2h"), ("QPI2_LO", "qpi2l")):
            A(f"static const double GEMMUL8_{name}_{be}[19][20] = {{")
            for row in t[key]:
                A("    {" + ", ".join(map(fmt_d, row)) + "},")
            A("};")
        A(f"static const short GEMMUL8_POW2MOD_{be}[20][64] = {{")
        for row in t["pow2"]:
            A("    {" + ", ".join(map(str, row)) + "},")
        A("};")
    A(f"static const int GEMMUL8_SQRT_MODULI_FP8[6] = {{{', '.join(map(str, SQRT_MODULI))}}};")
    text = "\n".join(lines) + "\n"
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = sys.argv[1:] or [os.path.join(root, "gemmul8_amd/csrc/tables.inc"), os.path.join(root, "oracle/tables.inc")]
    for o in outs:
        emit(o)
        print("wrote", o)
