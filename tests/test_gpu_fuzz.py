"""-m gpu: seeded random sweep over dtype x backend x mode x op x shape x num_moduli x (alpha, beta) through the C ABI,
each case bit-exact against the oracle (gpu_util.parity_case).  Shapes straddle the 256 x 256 tile, the 128-byte K-step
and the 256-padding of k; sizes are kept small enough for the scalar oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("GEMMUL8_FUZZ_SEEDS", "96"))  # the suite runs 96; longer one-off sweeps: GEMMUL8_FUZZ_SEEDS=1000

DIMS_MN = [1, 2, 31, 255, 256, 257, 300, 513]
DIMS_K = [1, 5, 127, 128, 129, 255, 256, 257, 400]


def _rand(shape, dtype, rng, phi):
    x = (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))
    if np.dtype(dtype).kind == "c":
        x = x + 1j * (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))
    return x.astype(dtype)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_case_bit_exact(seed, monkeypatch):
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
    # kernel-selection switches (round 3): the CRT's register / LDS-DMA forms (the latter needs whole 1024-byte units per column:
    # m = 1024 real, 512 complex) and the bound GEMM's 128 / 256 tiles, drawn per seed; "" = the library's own choice
    crt_force = str(rng.choice(["", "dma", "reg"]))
    tile_force = str(rng.choice(["", "128", "256"]))
    nt_force = str(rng.choice(["", "0", "1"]))       # residue-store policy of the INT8 GEMM (oz2_gemm_i8.hip nt_residue_stores)
    if crt_force == "dma" and backend == g.INT8:
        m, n = (512 if np.dtype(dtype).kind == "c" else 1024), min(n, 64)
    if crt_force:
        gu.setknob(monkeypatch, "GEMMUL8_CRT_KERNEL", crt_force)
    if tile_force:
        gu.setknob(monkeypatch, "GEMMUL8_BOUND_TILE", tile_force)
    if nt_force:
        gu.setknob(monkeypatch, "GEMMUL8_EPI_NT", nt_force)
    cb_force = str(rng.choice(["", "1", "2", "3"]))   # column-block width of the tile walk (oz2_gemm_common.hpp map_colblock)
    if cb_force:
        gu.setknob(monkeypatch, "GEMMUL8_MAP_COLBLOCK", cb_force)
    if backend == g.FP8 and seed % 5 == 0:  # FP8 backend: the e4m3 byte planes + e4m3 kernel instead of the FP6 panel images (round 5; no rng draw: the shapes of a seed stay)
        gu.setknob(monkeypatch, "GEMMUL8_FP8_PLANES", "e4m3")
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
    # FP8 accurate mode: once with the product's default (engine-safe) bound inflation and once with the reference's (k+1)*2^-24
    # (src/find_max.hpp:82-96), each selected on BOTH sides
    for bound_mode in ((gu.SAFE, gu.REFERENCE) if backend == g.FP8 and not fast else (gu.SAFE,)):
        gu.parity_case(A, B, N, fast, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0, backend=backend, bound_mode=bound_mode)


N_SEEDS_LARGE_K = int(os.environ.get("GEMMUL8_FUZZ_LARGEK_SEEDS", "12"))
DIMS_K_LARGE = [513, 640, 1023, 1024, 1300, 2049, 4097, 5120, 5121, 5400, 8193]
DIMS_MN_LARGE_K = [1, 31, 129, 255, 256, 257, 300]


@pytest.mark.parametrize("seed", range(N_SEEDS_LARGE_K))
def test_random_case_large_k_bit_exact(seed, monkeypatch):
    """The sweep above keeps k <= 400 for the scalar oracle's sake, i.e. padded k <= 512: every INT8 case of it runs the short-K
    instantiation of the GEMM (accumulators from 0, three-instruction residue).  This one draws k from 513 ... 8193 -- the
    K-step-barrier kernels with the byte-dot-product reduction on biased accumulators (padded k <= 5120), the ping-pong kernels
    beyond, the FP8 K-concatenation gate, the complex three-segment bound GEMMs -- on shapes cut down so that the oracle still
    finishes in seconds (m n k N x GEMMs-per-modulus <= 2e9)."""
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(77000 + seed)
    dtype = [np.float64, np.float32, np.complex128, np.complex64][seed % 4]
    backend = g.FP8 if (seed // 4) % 3 == 2 else g.INT8
    cplx = np.dtype(dtype).kind == "c"
    is_f32 = dtype in (np.float32, np.complex64)
    N = int(rng.integers(2, 14 if is_f32 else 21))
    fast = bool(rng.integers(0, 2))
    m, n = (int(rng.choice(DIMS_MN_LARGE_K)) for _ in range(2))
    k = int(rng.choice(DIMS_K_LARGE))
    if backend == g.FP8:
        k = min(k, 5400)
    weight = (3 if cplx else 1) * (3 if backend == g.FP8 else 1)
    while m * n * k * N * weight > 2e9:       # shrink the larger side, keeping it off the tile boundary
        if m >= n:
            m = max(1, m // 2 + 1)
        else:
            n = max(1, n // 2 + 1)
    nt_force = str(rng.choice(["", "0", "1"]))
    if backend == g.FP8 and seed % 5 == 0:
        gu.setknob(monkeypatch, "GEMMUL8_FP8_PLANES", "e4m3")
    if nt_force:
        gu.setknob(monkeypatch, "GEMMUL8_EPI_NT", nt_force)
    tile_force = str(rng.choice(["", "128", "256"]))
    if tile_force:
        gu.setknob(monkeypatch, "GEMMUL8_BOUND_TILE", tile_force)
    opA = str(rng.choice(["N", "T", "C"] if cplx else ["N", "T"]))
    opB = str(rng.choice(["N", "T", "C"] if cplx else ["N", "T"]))
    phi = float(rng.choice([0.0, 1.0, 3.0]))
    A = _rand((m, k) if opA == "N" else (k, m), dtype, rng, phi)
    B = _rand((k, n) if opB == "N" else (n, k), dtype, rng, phi)
    alpha, beta = [(1.0, 0.0), (1.0, 1.0), (-1.0, 0.0), (0.75, -0.5)][int(rng.integers(0, 4))]
    C0 = _rand((m, n), dtype, rng, 0.0) if beta != 0 else None
    # FP8 accurate mode: once with the product's default (engine-safe) bound inflation and once with the reference's (k+1)*2^-24
    # (src/find_max.hpp:82-96), each selected on BOTH sides
    for bound_mode in ((gu.SAFE, gu.REFERENCE) if backend == g.FP8 and not fast else (gu.SAFE,)):
        gu.parity_case(A, B, N, fast, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0, backend=backend, bound_mode=bound_mode)


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.complex128])
@pytest.mark.parametrize("fast", [False, True])
def test_extreme_exponents_bit_exact(dtype, fast):
    """Rows / columns scaled over (almost) the whole exponent range, subnormals, zeros and exact powers of two: the exact
    trunc(x * 2^s) representation and the shift logic against the oracle."""
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(31337)
    m, n, k = 70, 45, 200
    is32 = np.dtype(dtype) in (np.dtype(np.float32),)
    span = 60 if is32 else 450
    A = _rand((m, k), dtype, rng, 1.0)
    B = _rand((k, n), dtype, rng, 1.0)
    A = (A * np.exp2(rng.integers(-span, span, size=(m, 1)).astype(np.float64))).astype(dtype)
    B = (B * np.exp2(rng.integers(-span, span, size=(1, n)).astype(np.float64))).astype(dtype)
    A[3, :] = 0
    A[7, ::3] = 0
    B[:, 5] = 0
    A[11, :] = np.exp2(rng.integers(-20, 20, size=k)).astype(dtype)           # exact powers of two
    tiny = np.finfo(np.float32 if is32 else np.float64).tiny
    A[13, :] = (rng.random(k) * tiny).astype(dtype)                           # subnormal row
    N = 12 if is32 else 16
    gu.parity_case(A, B, N, fast)
    if not is32:
        gu.parity_case(A, B, 9, fast, backend=g.FP8)
