"""-m gpu: seeded random sweep over dtype x backend x mode x op x shape x num_moduli x (alpha, beta) through the C ABI,
each case bit-exact against the oracle (gpu_util.parity_case).  Shapes straddle the 256 x 256 tile, the 128-byte K-step
and the 256-padding of k; sizes are kept small enough for the scalar oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIMS_MN = [1, 2, 31, 255, 256, 257, 300]
DIMS_K = [1, 5, 127, 128, 129, 255, 256, 257, 400]


def _rand(shape, dtype, rng, phi):
    x = (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))
    if np.dtype(dtype).kind == "c":
        x = x + 1j * (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))
    return x.astype(dtype)


@pytest.mark.parametrize("seed", range(96))
def test_random_case_bit_exact(seed):
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(9000 + seed)
    dtype = [np.float64, np.float32, np.complex128, np.complex64][seed % 4]
    backend = g.FP8 if (seed // 4) % 3 == 2 else g.INT8
    is_f32 = dtype in (np.float32, np.complex64)
    N = int(rng.integers(2, 14 if is_f32 else 21))
    fast = bool(rng.integers(0, 2))
    m, n = (int(rng.choice(DIMS_MN)) for _ in range(2))
    k = int(rng.choice(DIMS_K))
    if backend == g.FP8 or np.dtype(dtype).kind == "c":   # 3-9x the oracle work
        m, n = min(m, 257), min(n, 256)
    cplx = np.dtype(dtype).kind == "c"
    opA = str(rng.choice(["N", "T", "C"] if cplx else ["N", "T"]))
    opB = str(rng.choice(["N", "T", "C"] if cplx else ["N", "T"]))
    phi = float(rng.choice([0.0, 1.0, 3.0]))
    A = _rand((m, k) if opA == "N" else (k, m), dtype, rng, phi)
    B = _rand((k, n) if opB == "N" else (n, k), dtype, rng, phi)
    if rng.integers(0, 3) == 0 and m > 2:
        (A if opA == "N" else A.T)[m // 2, :] = 0        # an all-zero row of op(A)
    alpha, beta = [(1.0, 0.0), (1.0, 1.0), (-1.0, 0.0), (-1.0, 1.0), (0.75, -0.5), (2.0, 0.0)][int(rng.integers(0, 6))]
    if cplx and rng.integers(0, 2):
        alpha, beta = alpha + 0.5j, beta - 0.25j
    C0 = _rand((m, n), dtype, rng, 0.0) if beta != 0 else None
    gu.parity_case(A, B, N, fast, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0, backend=backend)
