"""Independent big-integer model of the exact phases (Python ints / Fractions).  Used to pin the
oracle (and through it the HIP path) to the mathematical specification of SURVEY.md App. A:
A' = trunc(A*2^s), residues symmetric mod p, exact integer products, CRT reconstruction."""
from fractions import Fraction

import numpy as np

MODULI_INT8 = [256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193, 191, 181, 179, 173]
MODULI_FP8 = [1089, 1024, 961, 841, 625, 529, 511, 509, 503, 499, 491, 487, 481, 479, 467, 463, 461, 457, 449, 443]


def trunc_scaled(x, s):
    """trunc(x * 2^s) exactly, for a float/np.float32/np.float64 x."""
    f = Fraction(float(x)) * (Fraction(2) ** int(s))
    n = f.numerator // f.denominator if f >= 0 else -((-f.numerator) // f.denominator)
    return n


def sym(a, p):
    r = a % p
    return r - p if r > p // 2 else r


def as_int8(r):
    return ((r + 128) % 256) - 128


def int_matrix(X, shifts, rows_are_axis0=True):
    """X: logical (rows x k) array; shifts: NEGATED int16 per row -> list of lists of Python ints."""
    out = []
    for r in range(X.shape[0]):
        s = -int(shifts[r])
        out.append([trunc_scaled(X[r, kk], s) for kk in range(X.shape[1])])
    return out
