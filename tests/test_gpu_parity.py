"""-m gpu parity tests: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Policy: shifts may differ from the host oracle only at rare floor boundaries of
the hardware log2 (|diff| <= 1, flagged and counted); with the DEVICE's shifts fed to the oracle,
A_lo / B_lo / C_mid planes and the final C must be bit-identical."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rand(shape, dtype, rng, phi=1.0):
    x = (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))
    if np.dtype(dtype).kind == "c":
        x = x + 1j * (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))
    return x.astype(dtype)


def test_library_loads_and_reports():
    import gemmul8_amd as g
    assert b"gfx950" in g.lib().gemmul8_version()


def test_kat_sample_on_gpu():
    """Reference sample vectors (sample/dgemm_cuBLAS_int8.cu:24-38), N=15 accurate, through the HIP path."""
    import gpu_util as gu
    d = json.load(open(os.path.join(GOLD, "kat_dgemm_4x5x3.json")))
    A = np.array([float.fromhex(x) for x in d["A"]]).reshape((4, 5), order="F")
    B = np.array([float.fromhex(x) for x in d["B"]]).reshape((5, 3), order="F")
    Cx = np.array([float.fromhex(x) for x in d["C_exact"]]).reshape((4, 3), order="F")
    for fast in (False, True):
        C = gu.hip_gemm(A, B, 15, fastmode=fast)
        assert np.sqrt(((C - Cx) ** 2).sum()) < 4e-15
        gu.parity_case(A, B, 15, fast)


@pytest.mark.parametrize("dtype,N", [(np.float64, 2), (np.float64, 6), (np.float64, 7), (np.float64, 8), (np.float64, 14), (np.float64, 15), (np.float64, 16), (np.float64, 20), (np.float32, 2), (np.float32, 6), (np.float32, 7), (np.float32, 8)])  # float types: 2..13 moduli (more is rejected: tests/test_cabi.py)
@pytest.mark.parametrize("fast", [False, True])
def test_parity_small_real(dtype, fast, N):
    import gpu_util as gu
    rng = np.random.default_rng(100 * N + fast)
    m, n, k = 37, 41, 300
    A, B = rand((m, k), dtype, rng), rand((k, n), dtype, rng)
    A[5, :] = 0  # all-zero row
    B[:, 7] = 0  # all-zero column
    A[3, 4] = np.finfo(dtype).tiny / 4  # subnormal
    gu.parity_case(A, B, N, fast)


@pytest.mark.parametrize("opA,opB", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
@pytest.mark.parametrize("alpha,beta", [(1, 0), (1, 1), (-1, 0), (-1, 1), (-1.5, 1.5)])
def test_parity_ops_axpby(opA, opB, alpha, beta):
    """op x (alpha,beta) matrix of debug/test.cu:106-141 on odd sizes."""
    import gpu_util as gu
    rng = np.random.default_rng(5)
    m, n, k = 45, 33, 47
    A = rand((m, k) if opA == "N" else (k, m), np.float64, rng)
    B = rand((k, n) if opB == "N" else (n, k), np.float64, rng)
    C0 = rand((m, n), np.float64, rng)
    for fast in (False, True):
        gu.parity_case(A, B, 14, fast, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0)


@pytest.mark.parametrize("m,n,k", [(256, 256, 256), (300, 520, 700), (1, 1, 1), (513, 255, 1025), (128, 128, 4096)])
def test_parity_shapes(m, n, k):
    import gpu_util as gu
    rng = np.random.default_rng(m + n + k)
    A, B = rand((m, k), np.float64, rng), rand((k, n), np.float64, rng)
    gu.parity_case(A, B, 14, False)
    gu.parity_case(A, B, 9, True)


def test_config1_sgemm_256_moduli2():
    """BASELINE config 1 through the GPU path as well."""
    import gpu_util as gu
    A = (np.random.default_rng(12345).random((256, 256)) - 0.5).astype(np.float32)
    B = (np.random.default_rng(54321).random((256, 256)) - 0.5).astype(np.float32)
    for fast in (True, False):
        gu.parity_case(A, B, 2, fast)


@pytest.mark.parametrize("dtype,N", [(np.complex128, 2), (np.complex128, 7), (np.complex128, 13), (np.complex128, 16), (np.complex128, 20), (np.complex64, 2), (np.complex64, 7), (np.complex64, 13)])  # float types: 2..13 moduli (more is rejected: tests/test_cabi.py)
@pytest.mark.parametrize("fast", [False, True])
def test_parity_small_complex(dtype, fast, N):
    import gpu_util as gu
    rng = np.random.default_rng(7000 + 10 * N + fast)
    m, n, k = 37, 41, 300
    A, B = rand((m, k), dtype, rng), rand((k, n), dtype, rng)
    A[5, :] = 0
    B[:, 7] = 0
    gu.parity_case(A, B, N, fast)


@pytest.mark.parametrize("opA,opB", [("N", "N"), ("T", "N"), ("N", "C"), ("C", "T"), ("C", "C")])
def test_parity_complex_ops_axpby(opA, opB):
    """complex op(N/T/C) x (alpha,beta) pairs of debug/test.cu:106-141 (incl. -1.5+1.2i, 1.5+1.2i)."""
    import gpu_util as gu
    rng = np.random.default_rng(11)
    m, n, k = 45, 33, 47
    A = rand((m, k) if opA == "N" else (k, m), np.complex128, rng)
    B = rand((k, n) if opB == "N" else (n, k), np.complex128, rng)
    C0 = rand((m, n), np.complex128, rng)
    for alpha, beta in ((1, 0), (1, 1), (-1, 0), (-1, 1), (-1.5 + 1.2j, 1.5 + 1.2j)):
        gu.parity_case(A, B, 15, False, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0)
    gu.parity_case(A, B, 12, True, opA=opA, opB=opB, alpha=-1.5 + 1.2j, beta=1.5 + 1.2j, C0=C0)


def test_parity_complex_shapes():
    import gpu_util as gu
    rng = np.random.default_rng(99)
    from conftest import SOAK
    for (m, n, k) in [(256, 256, 256), (300, 520, 700), (1, 1, 1)] + ([(513, 255, 1025)] if SOAK else []):   # (the largest: 18 s of scalar oracle; real types run it always)
        A, B = rand((m, k), np.complex128, rng), rand((k, n), np.complex128, rng)
        gu.parity_case(A, B, 20, False)
        gu.parity_case(A, B, 9, True)


@pytest.mark.parametrize("dtype,N", [(np.float64, 14), (np.complex128, 9)])
def test_parity_long_k_int8(dtype, N):
    """k > 65536: accumulators may exceed 2^30, so the GEMM epilogue must leave the fp32 residue path for the integer
    multiply-high path (oz2_gemm_i8.hip RED_GENERIC) -- bit-exact against the oracle."""
    import gpu_util as gu
    rng = np.random.default_rng(66)
    m, n, k = 40, 24, 66048 + 5
    A, B = rand((m, k), dtype, rng), rand((k, n), dtype, rng)
    gu.parity_case(A, B, N, False)
    gu.parity_case(A, B, N, True)


@pytest.mark.parametrize("backend,dtype,N,k", [("INT8", np.float64, 15, 1 << 17), ("INT8", np.complex64, 7, 1 << 17),
                                               ("FP8", np.float32, 7, 1 << 16), ("FP8", np.complex128, 13, 1 << 16)])
def test_parity_at_the_k_limits(backend, dtype, N, k):
    """The largest inner dimension each backend accepts (INT8: 2^17, the reference's int32 accumulation limit; FP8: 2^16, exact
    FP32 accumulation of e4m3 products): one past it is E_ARG, at it the result is bit-exact against the oracle."""
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(k % 1000 + N)
    m, n = 9, 6
    A, B = rand((m, k), dtype, rng, phi=0.5), rand((k, n), dtype, rng, phi=0.5)
    be = getattr(g, backend)
    gu.parity_case(A, B, N, False, backend=be)
    gu.parity_case(A, B, N, True, backend=be)
    tdt = {np.float64: torch.float64, np.float32: torch.float32, np.complex64: torch.complex64, np.complex128: torch.complex128}[dtype]
    A1 = torch.zeros((k + 1, m), dtype=tdt, device="cuda")
    B1 = torch.zeros((n, k + 1), dtype=tdt, device="cuda")
    with pytest.raises(Exception):
        g.gemm(A1, B1, N, backend=be)


@pytest.mark.parametrize("m,n,k", [(0, 5, 7), (5, 0, 7), (5, 7, 0), (0, 0, 0)])
@pytest.mark.parametrize("backend", ["INT8", "FP8"])
def test_empty_dimensions_leave_c_untouched(m, n, k, backend):
    """m, n or k == 0: success and C untouched -- the behaviour the reference defines at its hook (hook.cu:616-617; NOT the
    BLAS beta*C semantics); the C ABI applies it to direct calls too."""
    import ctypes as C
    import gemmul8_amd as g
    lib = g.lib()
    be = getattr(g, backend)
    Cm = torch.full((max(n, 1), max(m, 1)), 3.25, dtype=torch.float64, device="cuda")
    A = torch.ones((max(k, 1), max(m, 1)), dtype=torch.float64, device="cuda")
    B = torch.ones((max(n, 1), max(k, 1)), dtype=torch.float64, device="cuda")
    al, bt = np.array([2.0]), np.array([0.0])
    st = torch.cuda.current_stream().cuda_stream
    work = torch.empty(1024, dtype=torch.uint8, device="cuda")
    rc = lib.gemmul8_gemm(st, g.D, be, 0, 0, m, n, k, al.ctypes.data, A.data_ptr(), max(m, 1), B.data_ptr(), max(k, 1),
                          bt.ctypes.data, Cm.data_ptr(), max(m, 1), 14 if backend == "INT8" else 12, 0, work.data_ptr(), None, None, 0, 0, 0, 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert bool((Cm == 3.25).all())


def test_fp8_rejects_k_beyond_exactness_bound():
    """FP8 products are only exact in FP32 for k <= 65536 (src/gemmul8_real.hpp k limit): the C ABI returns E_ARG."""
    import ctypes as C
    import gemmul8_amd as g
    lib = g.lib()
    m = n = 8
    k = 65537
    A = torch.zeros((k, m), dtype=torch.float32, device="cuda")
    B = torch.zeros((n, k), dtype=torch.float32, device="cuda")
    with pytest.raises(Exception):
        g.gemm(A, B, 6, backend=g.FP8)


@pytest.mark.parametrize("dtype,N", [(np.float64, 2), (np.float64, 5), (np.float64, 6), (np.float64, 7), (np.float64, 12), (np.float64, 13), (np.float64, 20), (np.float32, 2), (np.float32, 5), (np.float32, 6), (np.float32, 7), (np.float32, 12), (np.float32, 13)])  # float types: 2..13 moduli (more is rejected: tests/test_cabi.py)
@pytest.mark.parametrize("fast", [False, True])
def test_parity_small_real_fp8(dtype, fast, N):
    """FP8-e4m3 backend (gemmLt<T,FP8> in the reference): e4m3 planes, int16 C_mid, final C -- bit-exact."""
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(300 * N + fast)
    m, n, k = 37, 41, 300
    A, B = rand((m, k), dtype, rng), rand((k, n), dtype, rng)
    A[5, :] = 0
    B[:, 7] = 0
    gu.parity_case(A, B, N, fast, backend=g.FP8)


@pytest.mark.parametrize("m,n,k", [(256, 256, 256), (260, 300, 520), (1, 1, 1), (257, 255, 513)])
def test_parity_shapes_fp8(m, n, k):
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(m + n + k + 1)
    A, B = rand((m, k), np.float32, rng), rand((k, n), np.float32, rng)
    gu.parity_case(A, B, 6, False, backend=g.FP8)   # BASELINE config 3 parameters (SGEMM, moduli=6)
    gu.parity_case(A, B, 13, True, backend=g.FP8, opA="T" if m == k else "N")
    Ad, Bd = A.astype(np.float64), B.astype(np.float64)
    gu.parity_case(Ad, Bd, 13, False, backend=g.FP8, alpha=-1.5, beta=1.5, C0=rand((m, n), np.float64, rng))


@pytest.mark.parametrize("dtype,N", [(np.complex128, 2), (np.complex128, 6), (np.complex128, 7), (np.complex128, 13), (np.complex128, 20), (np.complex64, 2), (np.complex64, 6), (np.complex64, 7), (np.complex64, 13)])  # float types: 2..13 moduli (more is rejected: tests/test_cabi.py)
@pytest.mark.parametrize("fast", [False, True])
def test_parity_small_complex_fp8(dtype, fast, N):
    """FP8 backend, complex types: 9 e4m3 GEMMs per modulus (gemmul8_complex.hpp:170-195), three separately inflated bound
    products in accurate mode, interleaved int16 (Cr, Ci) planes -- bit-exact against the oracle."""
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(700 * N + fast)
    m, n, k = 37, 41, 300
    A, B = rand((m, k), dtype, rng), rand((k, n), dtype, rng)
    A[5, :] = 0
    B[:, 7] = 0
    gu.parity_case(A, B, N, fast, backend=g.FP8)


@pytest.mark.parametrize("m,n,k", [(256, 256, 256), (260, 300, 520), (1, 1, 1)])
def test_parity_shapes_complex_fp8(m, n, k):
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(m + n + k + 5)
    from conftest import SOAK
    A, B = rand((m, k), np.complex128, rng), rand((k, n), np.complex128, rng)
    gu.parity_case(A, B, 12, False, backend=g.FP8)
    gu.parity_case(A, B, 8, True, backend=g.FP8, opA="C" if m == k else "N", opB="T" if n == k else "N")
    if SOAK or k < 500:   # (nine scalar-oracle GEMMs per modulus: the largest shape's third case is 8 s)
        gu.parity_case(A, B, 13, False, backend=g.FP8, alpha=-1.5 + 0.5j, beta=0.25 - 2j, C0=rand((m, n), np.complex128, rng))


def test_kat_sample_fp8_on_gpu():
    """sample/dgemm_cuBLASLt_fp8.cu: the 4x5x3 known-answer vectors with N=13 on the FP8 backend."""
    import gemmul8_amd as g
    import gpu_util as gu
    d = json.load(open(os.path.join(GOLD, "kat_dgemm_4x5x3.json")))
    A = np.array([float.fromhex(x) for x in d["A"]]).reshape((4, 5), order="F")
    B = np.array([float.fromhex(x) for x in d["B"]]).reshape((5, 3), order="F")
    Cx = np.array([float.fromhex(x) for x in d["C_exact"]]).reshape((4, 3), order="F")
    C = gu.hip_gemm(A, B, 13, backend=g.FP8)
    assert np.sqrt(((C - Cx) ** 2).sum()) < 4e-15
    gu.parity_case(A, B, 13, False, backend=g.FP8)


@pytest.mark.parametrize("dt", ["float32", "float64", "complex64", "complex128"])
@pytest.mark.parametrize("backend", ["INT8", "FP8"])
@pytest.mark.parametrize("fast", [False, True])
def test_result_independent_of_workspace_contents(dt, backend, fast):
    """The workspace is scratch: whatever it holds on entry (the reference never clears it either), ragged shapes included,
    the result must be the same bits as with a zeroed workspace."""
    import gemmul8_amd as g
    tdt = getattr(torch, dt)
    be = getattr(g, backend)
    N = {"float32": 6, "float64": 13, "complex64": 6, "complex128": 12}[dt]
    gen = torch.Generator(device="cuda").manual_seed(11)
    for (m, n, k) in [(520, 392, 1031), (77, 300, 129), (257, 255, 640)]:
        rdt = torch.float32 if dt in ("float32", "complex64") else torch.float64

        def rnd(shape):
            x = torch.rand(shape, generator=gen, dtype=rdt, device="cuda") - 0.5
            if tdt.is_complex:
                x = torch.complex(x, torch.rand(shape, generator=gen, dtype=rdt, device="cuda") - 0.5)
            return x.contiguous()
        A, B = rnd((k, m)), rnd((n, k))
        tot, _, _ = g.work_size(tdt.is_complex, be, m, n, k, N)
        C0, _, _ = g.gemm(A, B, N, fastmode=fast, backend=be, work=torch.zeros(tot, dtype=torch.uint8, device="cuda"))
        for fill in ("ff", "a5", "random"):
            if fill == "random":
                w = torch.randint(0, 256, (tot,), generator=gen, dtype=torch.uint8, device="cuda")
            else:
                w = torch.full((tot,), int(fill, 16), dtype=torch.uint8, device="cuda")
            C1, _, _ = g.gemm(A, B, N, fastmode=fast, backend=be, work=w)
            assert torch.equal(C0.view(torch.uint8), C1.view(torch.uint8)), (m, n, k, fill)


# ---------------------------------------------------------------------------------------------------------------------
# Accurate-mode scaling phase, first half: bound planes, sft0 and bound maxima BIT-EXACT.
# parity_case() already runs bounds_case() for every accurate-mode case above; these add the layouts and operand shapes that
# exercise the tile masks of the MAX epilogue (rows/cols beyond m/n inside a 256-tile, several tiles, K-major and strided
# operands, conj) and the skip-scaling carving, where the bound planes have their own slot.
@pytest.mark.parametrize("backend", ["INT8", "FP8"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.complex128, np.complex64])
@pytest.mark.parametrize("opA,opB", [("N", "N"), ("T", "N"), ("N", "T"), ("C", "C")])
def test_bounds_bit_exact(backend, dtype, opA, opB):
    import gemmul8_amd as g
    import gpu_util as gu
    if np.dtype(dtype).kind != "c" and "C" in (opA, opB):
        opA, opB = "T", "T"
    rng = np.random.default_rng(hash((backend, np.dtype(dtype).name, opA, opB)) % 2**31)
    be = getattr(g, backend)
    for (m, n, k) in [(37, 41, 300), (300, 520, 700), (513, 255, 1025), (1, 1, 1)]:
        A = rand((m, k) if opA == "N" else (k, m), dtype, rng, phi=2.0)
        B = rand((k, n) if opB == "N" else (n, k), dtype, rng, phi=2.0)
        if m > 5:
            (A if opA == "N" else A.T)[5, :] = 0                       # all-zero row
            (A if opA == "N" else A.T)[3, min(4, k - 1)] = np.finfo(dtype).tiny / 4   # subnormal
        gu.bounds_case(A, B, 13, opA=opA, opB=opB, backend=be)
        gu.bounds_case(A, B, 13, opA=opA, opB=opB, backend=be, skip_layout=True)


@pytest.mark.parametrize("tile", ["128", "256"])
@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_bound_gemm_both_tile_sizes(dtype, tile, monkeypatch):
    """The INT8 bound GEMM has two kernels: the persistent 256 x 256-tile one and the 128 x 128-tile one that small products take
    (oz2_gemm_i8_small.hip).  GEMMUL8_BOUND_TILE forces either; the exact integer maxima must equal the oracle's with both, on shapes
    with ragged tile edges, several tiles per dimension, K-major and strided operands, and the complex K-concatenated products."""
    import gemmul8_amd as g
    import gpu_util as gu
    from conftest import SOAK
    gu.setknob(monkeypatch, "GEMMUL8_BOUND_TILE", tile)
    rng = np.random.default_rng(11)
    for (m, n, k), (opA, opB) in [((37, 41, 300), ("N", "N")), ((300, 520, 700), ("T", "N")), ((513, 255, 1025), ("N", "T")),
                                  ((129, 385, 520), ("T", "T")), ((1, 1, 1), ("N", "N"))]:
        if np.dtype(dtype).kind == "c" and opA == "T":
            opA = "C"
        if np.dtype(dtype).kind == "c" and m == 513 and not SOAK:
            continue
        A = rand((m, k) if opA == "N" else (k, m), dtype, rng, phi=2.0)
        B = rand((k, n) if opB == "N" else (n, k), dtype, rng, phi=2.0)
        gu.bounds_case(A, B, 14, opA=opA, opB=opB, backend=g.INT8)
        gu.parity_case(A, B, 14, False, opA=opA, opB=opB)


@pytest.mark.parametrize("kernel", ["dma", "reg"])
@pytest.mark.parametrize("dtype,m", [(np.float64, 1024), (np.float32, 2048), (np.complex128, 512), (np.complex64, 1024)])
def test_crt_kernels_both_forms(dtype, m, kernel, monkeypatch):
    """The CRT has two kernels for INT8 residues: the register form (oz2_crt.hip: crt_kernel) and the LDS-DMA form (crt_dma_kernel: slices
    through global_load_lds, sign-extending byte reads) that large products with whole 1024-byte units per column take.
    GEMMUL8_CRT_KERNEL forces either; both must reproduce the oracle bit for bit, for every axpby form."""
    import gpu_util as gu
    gu.setknob(monkeypatch, "GEMMUL8_CRT_KERNEL", kernel)
    rng = np.random.default_rng(5)
    n, k = 37, 160
    cplx = np.dtype(dtype).kind == "c"
    N = 14 if np.dtype(dtype).itemsize >= 8 and not cplx or np.dtype(dtype) == np.complex128 else 7
    A = rand((m, k), dtype, rng, phi=1.0)
    B = rand((k, n), dtype, rng, phi=1.0)
    C0 = rand((m, n), dtype, rng, phi=0.0)
    for alpha, beta in [(1.0, 0.0), (1.0, 1.0), (-1.0, 1.0), (-1.5, 0.5), (0.75, 0.0)]:
        if cplx and alpha == -1.5:
            alpha, beta = -1.5 + 0.5j, 0.5 - 0.25j
        gu.parity_case(A, B, N, True, alpha=alpha, beta=beta, C0=C0)
    gu.parity_case(A, B, N, False, C0=C0)


@pytest.mark.parametrize("policy", ["0", "1"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.complex128])
def test_residue_store_policy_both(dtype, policy, monkeypatch):
    """The INT8 GEMM writes its residue planes with non-temporal stores when the operand planes of the launch fit the Infinity Cache
    and the output is large (oz2_gemm_i8.hip nt_residue_stores); GEMMUL8_EPI_NT forces either policy.  Same bits either way."""
    import gpu_util as gu
    gu.setknob(monkeypatch, "GEMMUL8_EPI_NT", policy)
    rng = np.random.default_rng(77)
    m, n, k = 300, 290, 200
    A = rand((m, k), dtype, rng, phi=1.0)
    B = rand((k, n), dtype, rng, phi=1.0)
    gu.parity_case(A, B, 12 if dtype != np.float32 else 7, False)
    gu.parity_case(A, B, 12 if dtype != np.float32 else 7, True, alpha=-1.0, beta=1.0, C0=rand((m, n), dtype, rng, phi=0.0))


@pytest.mark.parametrize("width", ["1", "2", "3"])
@pytest.mark.parametrize("dtype,backend_fp8", [(np.float64, False), (np.complex64, False), (np.float32, True)])
def test_tile_walk_column_blocks(dtype, backend_fp8, width, monkeypatch):
    """Planes wider than ~160 MiB of B panels are walked in column blocks (oz2_gemm_common.hpp map_colblock, map_tile); the width is a
    kernel argument and GEMMUL8_MAP_COLBLOCK forces it, so small products exercise the blocked walk, ragged last block included
    (5 tile-columns in blocks of 1, 2, 3).  Same bits as the oracle."""
    import gemmul8_amd as g
    import gpu_util as gu
    gu.setknob(monkeypatch, "GEMMUL8_MAP_COLBLOCK", width)
    rng = np.random.default_rng(int(width))
    m, n, k = 520, 1100, 96
    A = rand((m, k), dtype, rng, phi=1.0)
    B = rand((k, n), dtype, rng, phi=1.0)
    gu.parity_case(A, B, 8 if dtype == np.float64 else 6, False, backend=g.FP8 if backend_fp8 else g.INT8)


@pytest.mark.parametrize("launches", ["1", "2"])
@pytest.mark.parametrize("tile", ["128", "256"])
@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_complex_bound_one_or_two_launches(dtype, tile, launches, monkeypatch):
    """Complex INT8 bound (find_max.hpp:99-114): maxima of C1 = |Ar||Bi| + |Ai||Br| and of C1 + C0.  Shipped: ONE 3-segment launch whose
    maxima epilogue also runs on the partial sums after the second segment; GEMMUL8_CPLX_BOUND_LAUNCHES=2: the 2-segment + 3-segment
    pair of rounds 1-2.  Both kernels (128 / 256 tiles), several K-steps per segment; parity_case asserts the integer row / column
    maxima, the shifts and everything downstream against the oracle."""
    import gpu_util as gu
    gu.setknob(monkeypatch, "GEMMUL8_CPLX_BOUND_LAUNCHES", launches)
    gu.setknob(monkeypatch, "GEMMUL8_BOUND_TILE", tile)
    rng = np.random.default_rng(11)
    m, n, k = 300, 270, 700
    A = rand((m, k), dtype, rng, phi=2.0)
    B = rand((k, n), dtype, rng, phi=2.0)
    gu.parity_case(A, B, 10 if dtype == np.complex128 else 6, False)


@pytest.mark.parametrize("k", [256, 512, 768, 1024, 4096])
def test_epilogue_reduction_on_extreme_accumulators(k):
    """The requantise epilogue on accumulators AT the bounds the residue planes allow (planes of +-127 with long runs of equal sign, so
    |sum| reaches k * 127^2), fed straight to gemmul8_lowprec_gemm: K <= 512 takes the three-instruction form (exact below 2^23; k = 256 also
    runs a row of -128 x -128 = the 2^22 bound of the float-pattern form that -DOZ2_MOD256=1 builds use there),
    longer K the byte-dot form on biased accumulators (any int32).  Random operands never come near these values; the expected residues
    are plain integer arithmetic."""
    import ctypes as C
    import gemmul8_amd as g
    L = g.lib()
    m, n, N = 256, 192, 14
    tot, _, _ = g.work_size(False, g.INT8, m, n, k, N)
    work = torch.zeros(tot, dtype=torch.uint8, device="cuda")
    Lo = g.Layout()
    g.check(L.gemmul8_get_layout(g.D, g.INT8, m, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(Lo)))
    assert Lo.kp == k
    rng = np.random.default_rng(k)
    base = work.data_ptr()

    def planes(rows, rows_pad):
        x = rng.choice(np.array([-127, 127, 126, -126, 0, 1], dtype=np.int8), size=(N, rows, k), p=[0.3, 0.3, 0.15, 0.15, 0.05, 0.05])
        x[:, 0, :] = 127                      # rows of one sign: the largest sums
        x[:, 1, :] = -127
        x[:, 2, ::2] = 127
        if k == 256:
            x[:, 3, :] = -128                 # 256 * 128^2 = 2^22: the bound of the float-pattern form (RED_MAGIC, -DOZ2_MOD256=1 builds)
        out = np.zeros((N, rows_pad, k), dtype=np.int8)
        out[:, :rows] = x
        return x, out
    A, Ap = planes(m, Lo.sizeA // k)
    B, Bp = planes(n, Lo.sizeB // k)
    offA, offB, offC = Lo.A_lo - base, Lo.B_lo - base, Lo.C_mid - base
    work[offA:offA + Ap.size] = torch.from_numpy(Ap.view(np.uint8).ravel()).cuda()
    work[offB:offB + Bp.size] = torch.from_numpy(Bp.view(np.uint8).ravel()).cuda()
    st = torch.cuda.current_stream().cuda_stream
    g.check(L.gemmul8_lowprec_gemm(st, g.D, g.INT8, m, n, k, N, 0, N, C.byref(Lo)))
    torch.cuda.synchronize()
    got = work[offC:offC + N * Lo.sizeC].cpu().numpy().view(np.int8).reshape(N, Lo.sizeC // Lo.mp, Lo.mp)[:, :n, :m]
    moduli = [256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199]
    for t, p in enumerate(moduli):
        acc = A[t].astype(np.int64) @ B[t].astype(np.int64).T          # (m, n)
        assert np.abs(acc).max() == (k * 128 * 128 if k == 256 else k * 127 * 127)   # the bound is reached
        r = np.mod(acc, p)
        r = np.where(r > (p - 1) // 2, r - p, r) if p != 256 else ((acc + 128) % 256 - 128)
        assert np.array_equal(got[t], r.T.astype(np.int8)), (k, p)


@pytest.mark.parametrize("k", [512, 8192, 65024, 65536])
def test_fp8_residue_gemms_on_extreme_accumulators(k, monkeypatch):
    """FP8 backend, the three residue GEMMs of a modulus on accumulators AT the bound the planes allow: integer pieces of +-16 with long runs of equal
    sign (|sum| reaches 256 k per product; k = 65536 -> 2^24, the last exactly representable value), fed straight to gemmul8_lowprec_gemm in every
    form the driver has -- FP6 panel images through the fused three-segment tile loop (k <= 65024; csrc/oz2_gemm_f6.hip EPI_FUSED: loose fp32 residues
    between the segments), FP6 through three launches (k = 65536, or GEMMUL8_FP8_FUSED=0), e4m3 byte planes (the round-4 kernel).  All three must agree
    bit for bit and equal plain integer arithmetic: squares value = s (hi.lo + lo.hi) + lo.lo, Karatsuba moduli 240 hi.hi - 15 lo.lo + 16 (hi+lo).(hi+lo)
    (src/mod.hpp:159-189 pieces, src/conv_hi2mid_real.hpp combine).  Random operands never come near these sums."""
    import ctypes as C
    import gemmul8_amd as g
    import gpu_util as gu
    from bigint_ref import MODULI_FP8
    L = g.lib()
    m, n, N = 80, 96, 8   # six square moduli + two Karatsuba moduli
    rng = np.random.default_rng(k)
    nplanes = 2 * 6 + 3 * 2

    def step_planes(rows):  # one K-step (128 columns) of every plane; the full plane repeats it along k
        x = rng.choice(np.array([-16, 16, 15, -15, 0, 1], dtype=np.int8), size=(nplanes, rows, 128), p=[0.3, 0.3, 0.15, 0.15, 0.05, 0.05])
        x[:, 0, :] = 16
        x[:, 1, :] = -16
        x[:, 2, ::2] = 16
        return x
    A, B = step_planes(m), step_planes(n)
    reps = k // 128
    st = torch.cuda.current_stream().cuda_stream
    got = {}
    for form in ("fp6", "fp6_three_launches", "e4m3"):
        gu.setknob(monkeypatch, "GEMMUL8_FP8_PLANES", "e4m3" if form == "e4m3" else "fp6")
        gu.setknob(monkeypatch, "GEMMUL8_FP8_FUSED", "0" if form == "fp6_three_launches" else None)
        tot, _, _ = g.work_size(False, g.FP8, m, n, k, N)
        work = torch.zeros(tot, dtype=torch.uint8, device="cuda")
        Lo = g.Layout()
        g.check(L.gemmul8_get_layout(g.S, g.FP8, m, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(Lo)))
        assert Lo.kp == k and Lo.lo_format == (0 if form == "e4m3" else 1) and Lo.num_mat == nplanes
        base = work.data_ptr()
        for X, ptr, size, rows_img in ((A, Lo.A_lo, Lo.sizeA, Lo.mp), (B, Lo.B_lo, Lo.sizeB, n)):
            for q in range(nplanes):
                if form == "e4m3":
                    img = np.zeros((size // k, k), np.uint8)
                    img[:X.shape[1]] = np.tile(gu.e4m3_of_ints(X[q]), (1, reps))
                else:  # the panel of a K-step is the same in every K-step: encode one, repeat it inside each row block
                    nb = (rows_img + 255) // 256
                    one = gu.f6_plane_image(X[q], rows_img, 128, nb * 256 * 96)
                    img = np.zeros(size, np.uint8)
                    for tb in range(nb):
                        rp = 256 if tb < nb - 1 else (min(256, rows_img - 256 * tb) + 15) // 16 * 16
                        img[tb * 256 * (k // 4 * 3):][:reps * rp * 96] = np.tile(one[tb * 256 * 96:][:rp * 96], reps)
                off = ptr - base + q * size
                work[off:off + size] = torch.from_numpy(img.ravel()).cuda()
        g.check(L.gemmul8_lowprec_gemm(st, g.S, g.FP8, m, n, k, N, 0, N, C.byref(Lo)))
        torch.cuda.synchronize()
        offC = Lo.C_mid - base
        got[form] = work[offC:offC + 2 * N * Lo.sizeC].cpu().numpy().view(np.int16).reshape(N, Lo.sizeC // Lo.mp, Lo.mp)[:, :n, :m].copy()
        del work
    assert np.array_equal(got["fp6"], got["e4m3"]) and np.array_equal(got["fp6"], got["fp6_three_launches"])
    A64, B64 = A.astype(np.int64), B.astype(np.int64)
    reached = 0
    for t in range(N):
        pm = MODULI_FP8[t]
        if t < 6:
            q, s = 2 * t, int(round(pm ** 0.5))
            c0, c1, c2 = A64[q] @ B64[q + 1].T, A64[q + 1] @ B64[q].T, A64[q + 1] @ B64[q + 1].T
            val = s * (c0 + c1) + c2
        else:
            q = 12 + 3 * (t - 6)
            c0, c1, c2 = A64[q] @ B64[q].T, A64[q + 1] @ B64[q + 1].T, A64[q + 2] @ B64[q + 2].T
            val = 240 * c0 - 15 * c1 + 16 * c2
        reached = max(reached, reps * int(max(np.abs(c0).max(), np.abs(c1).max(), np.abs(c2).max())))
        r = got["fp6"][t].T.astype(np.int64)                 # C_mid is stored [n][mp]
        assert not ((r - reps * val) % pm).any(), (k, pm)
        assert np.abs(r).max() <= pm // 2, (k, pm)
    assert reached == 256 * k                                # the bound is reached


@pytest.mark.parametrize("kernel", ["dma", "reg"])
@pytest.mark.parametrize("dtype,N", [(np.float64, 14), (np.float64, 20), (np.float32, 7), (np.complex128, 16)])
def test_crt_on_extreme_residues(dtype, N, kernel, monkeypatch):
    """CRT + unscale + axpby on residue planes AT the ends of their ranges (all +-(p-1)/2, alternating by plane, zeros, +-1), fed straight
    to gemmul8_crt and compared bit for bit with the oracle's inverse scaling (inverse_scaling_real.hpp:8-89): the reduction mod P
    next to its rounding boundaries, which random products never produce.  Both kernels (LDS-DMA slices / register form)."""
    import ctypes as C
    import gemmul8_amd as g
    import oracle_lib as ol
    import gpu_util as gu
    gu.setknob(monkeypatch, "GEMMUL8_CRT_KERNEL", kernel)
    cplx = np.dtype(dtype).kind == "c"
    comps = 2 if cplx else 1
    m, n = (512 if cplx else 1024), 24
    moduli = [256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193, 191, 181, 179, 173][:N]
    rng = np.random.default_rng(N)
    Cm = np.zeros((N, n, m, comps), np.int8)
    for t, p in enumerate(moduli):
        hi = 127 if p == 256 else (p - 1) // 2
        lo = -128 if p == 256 else -hi
        pool = np.array([hi, lo, hi - 1, lo + 1, 0, 1, -1], dtype=np.int64)
        Cm[t] = rng.choice(pool, size=(n, m, comps)).astype(np.int8)
        Cm[t, 0] = hi                                   # column 0: every plane at its maximum
        Cm[t, 1] = lo                                   # column 1: every plane at its minimum
        Cm[t, 2] = hi if t % 2 == 0 else lo             # column 2: alternating
        Cm[t, 3] = 0
    Cm[0, 3] = 1                                        # column 3: the value 1 in the first plane only
    sftA = rng.integers(-70, 70, size=m).astype(np.int16)
    sftB = rng.integers(-70, 70, size=n).astype(np.int16)
    code = g._dtype_code(torch.from_numpy(np.zeros(1, dtype)).dtype)
    for alpha, beta in [(1.0, 0.0), (-1.0, 1.0), (0.75, -0.5)]:
        al, be = np.array([alpha], dtype), np.array([beta], dtype)
        C0 = rand((n, m), dtype, rng, phi=0.0)          # column-major m x n as (n, m)
        want = C0.copy()
        ol.lib().oz2_invscal(code, ol.INT8, N, m, n, ol._p(Cm), ol._p(sftA), ol._p(sftB), ol._p(al), ol._p(be), ol._p(want), m, 0)
        dC = torch.from_numpy(C0.copy()).cuda()
        dM = torch.from_numpy(Cm.copy()).cuda()
        dA, dB = torch.from_numpy(sftA).cuda(), torch.from_numpy(sftB).cuda()
        g.check(g.lib().gemmul8_crt(torch.cuda.current_stream().cuda_stream, code, g.INT8, N, m, n, dM.data_ptr(), m, n * m, dA.data_ptr(),
                                    dB.data_ptr(), al.ctypes.data, be.ctypes.data, dC.data_ptr(), m))
        torch.cuda.synchronize()
        got = dC.cpu().numpy()
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (dtype, N, kernel, alpha, beta)


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("k", [512, 1024, 3000])
@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.complex128])
def test_parity_constant_and_sign_pattern_matrices(dtype, k, fast):
    """Operands whose scaled integers repeat along k (constant rows, +-1 checkerboards, one huge entry): every product of a dot product
    has the same residues, so the INT32 sums reach k * r_a * r_b -- the largest accumulators the whole pipeline can produce -- and the
    bound GEMM's sums reach the value the shift formula is designed around.  Whole pipeline against the oracle, bit for bit."""
    import gpu_util as gu
    rng = np.random.default_rng(k)
    m, n = 70, 66
    A = np.ones((m, k), dtype)
    B = np.ones((k, n), dtype)
    A[1] = -1.0
    A[2, ::2] = -1.0
    A[3] = 0.999999                       # scaled integer just below a power of two
    A[4] = 1.000001
    A[5] = rng.choice([-1.0, 1.0], size=k)
    A[6, 0] = 1e6                         # one dominant entry in the row
    B[:, 1] = -1.0
    B[::2, 2] = -1.0
    B[:, 3] = 3.0
    B[:, 4] = 1.0 / 3.0
    if np.dtype(dtype).kind == "c":
        A = A + 1j * np.roll(A, 1, axis=0)
        B = B - 1j * np.roll(B, 1, axis=1)
    N = 7 if dtype == np.float32 else 14
    gu.parity_case(A.astype(dtype), B.astype(dtype), N, fast)


def test_residue_store_policy_auto_at_size(monkeypatch):
    """8192 x 8192 x 512, 14 moduli: 112 MiB of operand planes and 896 MiB of residues -- the shape class where the library picks
    non-temporal residue stores by itself.  The result must equal the forced default-policy run bit for bit."""
    import torch
    import gemmul8_amd as g
    torch.manual_seed(3)
    n, k = 8192, 512
    A = torch.randn((k, n), dtype=torch.float64, device="cuda")
    B = torch.randn((n, k), dtype=torch.float64, device="cuda")
    import gpu_util as gu
    gu.setknob(monkeypatch, "GEMMUL8_EPI_NT", None)
    C_auto, _, work = g.gemm(A, B, 14)
    gu.setknob(monkeypatch, "GEMMUL8_EPI_NT", "0")
    C_plain, _, _ = g.gemm(A, B, 14, work=work)
    torch.cuda.synchronize()
    assert torch.equal(C_auto, C_plain)
    ref = (B.double() @ A.double())  # tensors are transposed views of the column-major matrices: C^T = B^T-view @ A^T-view
    assert float((C_auto - ref).abs().max() / ref.abs().max()) < 1e-13


@pytest.mark.parametrize("dtype", [np.float32, np.complex128])
def test_bounds_fp8_exact_when_fp32_sums_are_exact(dtype):
    """Operands of one binade: every e4m3 bound value is a multiple of 8 in [64, 256], so the FP32 accumulation of the
    products is exact in any order and the device's inflated float maxima must equal the oracle's BITS."""
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(21)
    m, n, k = 130, 70, 200

    def one_binade(shape):
        x = (0.25 + 0.25 * rng.random(shape)) * rng.choice([-1.0, 1.0], shape)
        if np.dtype(dtype).kind == "c":
            x = x + 1j * (0.25 + 0.25 * rng.random(shape)) * rng.choice([-1.0, 1.0], shape)
        return x.astype(dtype)
    nd = gu.bounds_case(one_binade((m, k)), one_binade((k, n)), 12, backend=g.FP8)
    assert nd == 0


@pytest.mark.parametrize("shape", [(130, 70, 200), (257, 96, 640)])
@pytest.mark.parametrize("dtype,N", [(np.float32, 6), (np.float64, 12), (np.complex64, 7), (np.complex128, 12)])
def test_fp8_reference_bound_formula_end_to_end(dtype, N, shape):
    """gemmul8_set_fp8_bound_mode(1) = the reference's own inflation ku = (k+1) 2^-24 (GEMMul8/src/find_max.hpp:82-96) on device AND
    oracle, through the WHOLE accurate-mode pipeline.  Operands of one binade make every e4m3 bound value a multiple of 8 in [64, 256]:
    the FP32 accumulation of the bound products is exact in any order, so how gfx950's FP8 MFMA adds (the reason for the default mode 0)
    cannot show -- the inflated maxima must equal the oracle's BITS, hence the shifts, the residue planes, C_mid and C (VERDICT r4 weak #1:
    the reference-faithful path was selectable but never asserted)."""
    import gemmul8_amd as g
    import gpu_util as gu
    import oracle_lib as ol
    rng = np.random.default_rng(1000 + N + shape[0])
    m, n, k = shape

    def one_binade(sh):
        x = (0.25 + 0.25 * rng.random(sh)) * rng.choice([-1.0, 1.0], sh)
        if np.dtype(dtype).kind == "c":
            x = x + 1j * (0.25 + 0.25 * rng.random(sh)) * rng.choice([-1.0, 1.0], sh)
        return x.astype(dtype)
    A, B = one_binade((m, k)), one_binade((k, n))
    assert gu.bounds_case(A, B, N, backend=g.FP8, bound_mode=gu.REFERENCE) == 0, "mode 1: inflated bound maxima differ from the oracle's bits"
    nd = gu.parity_case(A, B, N, False, backend=g.FP8, bound_mode=gu.REFERENCE)     # bound planes / maxima / planes / C_mid / C bit-exact inside
    assert nd == 0, f"mode 1 with exact bound sums: {nd} shifts differ from the oracle"
    assert ol.get_fp8_bound_mode() == ol.FP8_BOUND_REFERENCE and g.lib().gemmul8_set_fp8_bound_mode(1) == 1   # both sides really ran the reference formula


@pytest.mark.parametrize("backend,dtype,N", [("INT8", np.float64, 14), ("INT8", np.complex64, 9), ("FP8", np.float32, 6), ("FP8", np.complex128, 13)])
def test_skip_scaling_keeps_bound_planes_and_reuses_them(backend, dtype, N):
    """enable_skip_scal{A,B}: after a whole call the bound plane of each operand persists in its own slot (bit-exact vs the
    oracle's extract), and a second call with skip_scal{A,B} = 1 and a DIFFERENT partner operand reuses planes + shifts of the
    kept operand: equal to the oracle run fed with the kept operand's shifts (gemmul8_real.hpp:82-83,101-104,123-139)."""
    import ctypes as C
    import gemmul8_amd as g
    import gpu_util as gu
    import oracle_lib as ol
    be = getattr(g, backend)
    rng = np.random.default_rng(N)
    m, n, k = 77, 52, 260
    A, B, B2 = rand((m, k), dtype, rng), rand((k, n), dtype, rng), rand((k, n), dtype, rng)
    cplx = np.dtype(dtype).kind == "c"
    lib = g.lib()
    tot, wa, wb = g.work_size(cplx, be, m, n, k, N, True, True)
    work = torch.zeros(tot, dtype=torch.uint8, device="cuda")
    dA, dB, dB2 = gu.to_dev(A), gu.to_dev(B), gu.to_dev(B2)
    dC = torch.zeros((n, m), dtype=dA.dtype, device="cuda")
    one, zero = np.array([1], dtype), np.array([0], dtype)
    st = torch.cuda.current_stream().cuda_stream
    code = g._dtype_code(dA.dtype)

    def call(Bt, skA, skB):
        g.check(lib.gemmul8_gemm(st, code, be, 0, 0, m, n, k, one.ctypes.data, dA.data_ptr(), m, Bt.data_ptr(), k, zero.ctypes.data,
                                 dC.data_ptr(), m, N, 0, work.data_ptr(), None, None, 1, 1, skA, skB, None))
        torch.cuda.synchronize()
        return gu.from_dev(dC).copy()
    C1 = call(dB, 0, 0)
    L = g.Layout()
    g.check(lib.gemmul8_get_layout(code, be, m, n, k, N, work.data_ptr(), None, None, 1, 1, C.byref(L)))
    w = work.cpu().numpy()
    base = work.data_ptr()
    parts = 3 if cplx else 1
    oA, _ = ol.extract_bounds(A, "N", True, be)
    oB, _ = ol.extract_bounds(B, "N", False, be)
    for p in range(parts):
        pa = w[L.A_bound - base + p * L.sizeA:][:L.mp * L.kp].reshape(L.mp, L.kp)[:m, :k]
        pb = w[L.B_bound - base + p * L.sizeB:][:n * L.kp].reshape(n, L.kp)[:, :k]
        assert np.array_equal(pa, oA[p]) and np.array_equal(pb, oB[p]), "kept bound planes differ from the oracle's extract"
    sftA = w[L.sftA - base:][:2 * m].view(np.int16).copy()
    Co1 = ol.gemm(A, B, N, backend=be, sftA_in=sftA, sftB_in=w[L.sftB - base:][:2 * n].view(np.int16).copy())
    assert gu.bits_equal(C1, Co1)
    # second call: A kept (skip), B2 new.  The accurate-mode shift of B2's columns is computed against A's KEPT bound plane.
    C2 = call(dB2, 1, 0)
    w = work.cpu().numpy()
    assert np.array_equal(w[L.sftA - base:][:2 * m].view(np.int16), sftA), "the kept operand's shifts must not change"
    sftB2 = w[L.sftB - base:][:2 * n].view(np.int16).copy()
    oB2, s0B2 = ol.extract_bounds(B2, "N", False, be)
    _, cmax = ol.bound_maxima(oA, oB2, be)
    Co2 = ol.gemm(A, B2, N, backend=be, sftA_in=sftA, sftB_in=sftB2)
    assert gu.bits_equal(C2, Co2)
    if be == g.INT8:  # column shifts from the exact integer maxima: reproduce them through the oracle's finalize
        exp = s0B2.copy()
        ol.lib().oz2_shift_finalize_i8(be, N, n, cmax.ctypes.data_as(C.c_void_p), exp.ctypes.data_as(C.c_void_p))
        gu.shifts_close(sftB2, exp, "sftB (skip A)")


@pytest.mark.parametrize("backend,dtype,N", [("INT8", np.float64, 14), ("INT8", np.complex128, 15), ("FP8", np.float32, 6), ("INT8", np.float32, 7)])
@pytest.mark.parametrize("alpha,beta", [(1, 0), (-1, 1), (-1.5, 1.5)])
def test_device_pointer_scalars_bit_exact(backend, dtype, N, alpha, beta):
    """alpha/beta in DEVICE memory (hipBLAS pointer mode device): always the general fma form, also for alpha = +-1 and
    beta in {0, 1} (inverse_scaling_real.hpp:211-216) -- bit-exact against the oracle's scalar_mode = 1."""
    import gemmul8_amd as g
    import gpu_util as gu
    import oracle_lib as ol
    be = getattr(g, backend)
    rng = np.random.default_rng(3)
    m, n, k = 45, 33, 147
    A, B, C0 = rand((m, k), dtype, rng), rand((k, n), dtype, rng), rand((m, n), dtype, rng)
    if np.dtype(dtype).kind == "c" and alpha == -1.5:
        alpha, beta = -1.5 + 1.2j, 1.5 - 0.7j
    dA, dB, dC = gu.to_dev(A), gu.to_dev(B), gu.to_dev(C0.copy())
    d_al = torch.tensor([alpha], dtype=dA.dtype, device="cuda")
    d_be = torch.tensor([beta], dtype=dA.dtype, device="cuda")
    cplx = np.dtype(dtype).kind == "c"
    tot, _, _ = g.work_size(cplx, be, m, n, k, N)
    work = torch.zeros(tot, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    import ctypes as C
    code = g._dtype_code(dA.dtype)
    for fast in (False, True):
        dC.copy_(gu.to_dev(C0))
        g.check(g.lib().gemmul8_gemm(st, code, be, 0, 0, m, n, k, d_al.data_ptr(), dA.data_ptr(), m, dB.data_ptr(), k, d_be.data_ptr(),
                                     dC.data_ptr(), m, N, int(fast), work.data_ptr(), None, None, 0, 0, 0, 0, None))
        torch.cuda.synchronize()
        L = g.Layout()
        g.check(g.lib().gemmul8_get_layout(code, be, m, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(L)))
        w = work.cpu().numpy()
        base = work.data_ptr()
        sA = w[L.sftA - base:][:2 * m].view(np.int16).copy()
        sB = w[L.sftB - base:][:2 * n].view(np.int16).copy()
        Co = ol.gemm(A, B, N, fastmode=fast, backend=be, alpha=alpha, beta=beta, C0=C0, scalar_mode=1, sftA_in=sA, sftB_in=sB)
        assert gu.bits_equal(gu.from_dev(dC), Co)


@pytest.mark.parametrize("backend,N", [("INT8", 15), ("FP8", 13)])
@pytest.mark.parametrize("fast", [False, True])
def test_kat_result_bit_patterns(backend, N, fast):
    """Regression pin: the result BITS of the reference's known-answer sample (tests/golden/kat_result_bits.json, produced by
    the oracle -- tools/make_kat_result_bits.py) through the HIP path."""
    import gemmul8_amd as g
    import gpu_util as gu
    d = json.load(open(os.path.join(GOLD, "kat_dgemm_4x5x3.json")))
    A = np.array([float.fromhex(x) for x in d["A"]]).reshape((4, 5), order="F")
    B = np.array([float.fromhex(x) for x in d["B"]]).reshape((5, 3), order="F")
    gold = json.load(open(os.path.join(GOLD, "kat_result_bits.json")))[f"{backend}_N{N}_{'fast' if fast else 'accurate'}"]
    Cd, it = gu.hip_gemm(A, B, N, fastmode=fast, backend=getattr(g, backend), want_intermediates=True)
    assert [float(x).hex() for x in Cd.flatten(order="F")] == gold["C"]
    assert it["sftA"].tolist() == gold["sftA"] and it["sftB"].tolist() == gold["sftB"]
    assert it["C_mid"].flatten().tolist() == gold["C_mid"]


@pytest.mark.parametrize("n", [2, 3])
def test_tall_skinny_accurate_mode(n):
    """m = 2.7 M rows, n <= 3: the accurate-mode scratch (row maxima, column maxima, per-row amax bits = 16 m + ... bytes)
    no longer fits the C_hi region alone (4 * mp * n bytes) nor the 32 MiB tail alone -- it spans both, which are adjacent -- and the
    row-strided extract / quantise launches need more than 65535 row tiles (1-D grid).  Sampled rows bit-exact against the oracle fed
    with the device's shifts; every shift within the usual +-1 of the oracle's."""
    import ctypes as C
    import gemmul8_amd as g
    import gpu_util as gu
    import oracle_lib as ol
    m, k, N = 2_700_000, 8, 6
    gen = torch.Generator(device="cuda").manual_seed(n)
    A = (torch.rand((k, m), generator=gen, dtype=torch.float64, device="cuda") - 0.5).contiguous()   # column-major m x k
    B = (torch.rand((n, k), generator=gen, dtype=torch.float64, device="cuda") - 0.5).contiguous()   # column-major k x n
    Cd, _, work = g.gemm(A, B, N, fastmode=False)
    torch.cuda.synchronize()
    L = g.Layout()
    g.check(g.lib().gemmul8_get_layout(g.D, g.INT8, m, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(L)))
    base = work.data_ptr()
    sftA = work[L.sftA - base:L.sftA - base + 2 * m].cpu().numpy().view(np.int16)
    sftB = work[L.sftB - base:L.sftB - base + 2 * n].cpu().numpy().view(np.int16)
    An = np.asfortranarray(A.cpu().numpy().T)
    Bn = np.asfortranarray(B.cpu().numpy().T)
    oA, oB = ol.accurate_shifts(An, Bn, N)
    gu.shifts_close(sftA, oA, "sftA")
    gu.shifts_close(sftB, oB, "sftB")
    rows = np.array([0, 1, 255, 256, 65535 * 16, 65535 * 16 + 17, 1_048_576, m - 257, m - 1] + np.random.default_rng(1).integers(0, m, 40).tolist())
    Co = ol.gemm(An[rows], Bn, N, sftA_in=sftA[rows], sftB_in=sftB)
    got = Cd[:, torch.as_tensor(rows, device="cuda")].cpu().numpy().T
    assert gu.bits_equal(np.ascontiguousarray(got), np.ascontiguousarray(Co))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,N,fast", [(np.float32, 6, False), (np.float64, 13, True), (np.complex64, 8, False)])
def test_fp8_backend_fp6_images_and_e4m3_planes_agree(dtype, N, fast, monkeypatch):
    """FP8 backend, round 5: the residue planes are FP6 panel images of the same integers (gemmul8_layout.lo_format == 1, n >= 64;
    csrc/oz2_gemm_f6.hip) unless GEMMUL8_FP8_PLANES=e4m3 keeps the reference's e4m3 bytes.  Both encodings must be bit-exact against the
    oracle -- planes, C_mid and C -- including a last row block of B that is not a multiple of 16 rows and a second tile column."""
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(606)
    m, n, k = 300, 333, 520
    A, B = rand((m, k), dtype, rng), rand((k, n), dtype, rng)
    for planes, want in (("fp6", 1), ("e4m3", 0)):
        gu.setknob(monkeypatch, "GEMMUL8_FP8_PLANES", planes)
        Cd, it = gu.hip_gemm(A, B, N, fastmode=fast, backend=g.FP8, want_intermediates=True)
        assert it["lo_format"] == want
        gu.parity_case(A, B, N, fast, backend=g.FP8)
    # below 64 columns the images of B's only block would not fit its plane: the e4m3 planes are used
    gu.setknob(monkeypatch, "GEMMUL8_FP8_PLANES", "fp6")
    _, it = gu.hip_gemm(A, B[:, :48].copy(), N, fastmode=fast, backend=g.FP8, want_intermediates=True)
    assert it["lo_format"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,N", [(np.float32, 8), (np.float64, 12), (np.float64, 18)])
@pytest.mark.parametrize("opA,opB", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
def test_fp6_lane_per_fragment_writer_on_ragged_shapes(dtype, N, opA, opB):
    """Real operands of the FP8 backend are quantised by the lane-per-fragment kernel (csrc/oz2_scale.hip quantise_f6_pair_kernel: a lane owns the 32
    consecutive k of one row, the hardware packs the 32 e2m3 codes; row-strided operands by 32 coalesced loads, K-major ones through a wave-private
    LDS transpose).  Shapes that leave every loop ragged -- rows not a multiple of 64 (nor of 16), k not a multiple of 32 / 128 / the 16-byte load,
    odd leading dimensions (element-wise tail loads), Karatsuba moduli (N > 6), 18 moduli (scaled values beyond 2^53: the second reduction step) --
    in all four operand orientations, bit-exact against the oracle (planes, C_mid, C)."""
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(2026)
    from conftest import SOAK
    shapes = ((65, 64, 33), (130, 257, 127)) if N > 12 else ((65, 64, 33), (130, 257, 127), (64, 70, 2049), (300, 96, 515))  # (the oracle's cost grows with N)
    if not SOAK and (opA, opB) in (("T", "N"), ("N", "T")):
        shapes = shapes[:2]   # the two long shapes run in the N/N and T/T orientations (each operand in both forms); GEMMUL8_TEST_SOAK=1: everywhere
    for (m, n, k) in shapes:
        A = rand((m, k) if opA == "N" else (k, m), dtype, rng, phi=1.0)
        B = rand((k, n) if opB == "N" else (n, k), dtype, rng, phi=1.0)
        _, it = gu.hip_gemm(A, B, N, fastmode=True, backend=g.FP8, opA=opA, opB=opB, want_intermediates=True)
        assert it["lo_format"] == 1
        gu.parity_case(A, B, N, True, backend=g.FP8, opA=opA, opB=opB)
    gu.parity_case(A, B, N, False, backend=g.FP8, opA=opA, opB=opB)


@pytest.mark.gpu
def test_fp8_skip_scaling_cached_planes_meet_a_partner_of_another_width():
    """FP8 backend, round 5: whether the residue planes are FP6 panel images depends on n (B's last row block must fit its plane).  With skip-scaling
    enabled an operand's planes outlive the call and may meet a partner of another width -- the reference's use of the flag: one A against changing B
    (gemmul8_real.hpp:82-83,123-139) -- so layouts with skip-scaling enabled keep the e4m3 byte planes: A quantised beside a 96-column B is reused beside
    a 40-column B2 (skip_scalA = 1, separate workA / workB) and the result equals the oracle's, fed with the kept shifts."""
    import ctypes as C
    import gemmul8_amd as g
    import gpu_util as gu
    import oracle_lib as ol
    rng = np.random.default_rng(55)
    m, k, n1, n2, N = 130, 300, 96, 40, 8
    A, B1, B2 = rand((m, k), np.float32, rng), rand((k, n1), np.float32, rng), rand((k, n2), np.float32, rng)
    lib = g.lib()
    dA, dB1, dB2 = gu.to_dev(A), gu.to_dev(B1), gu.to_dev(B2)
    one, zero = np.array([1], np.float32), np.array([0], np.float32)
    st = torch.cuda.current_stream().cuda_stream
    wA_bytes = C.c_size_t(0)
    wB_bytes = C.c_size_t(0)
    tot = lib.gemmul8_work_size(0, g.FP8, m, max(n1, n2), k, N, 1, 1, C.byref(wA_bytes), C.byref(wB_bytes))
    work = torch.zeros(tot, dtype=torch.uint8, device="cuda")
    workA = torch.zeros(wA_bytes.value, dtype=torch.uint8, device="cuda")
    workB = torch.zeros(wB_bytes.value, dtype=torch.uint8, device="cuda")

    def call(dB, n, skA):
        dC = torch.zeros((n, m), dtype=torch.float32, device="cuda")
        g.check(lib.gemmul8_gemm(st, g.S, g.FP8, 0, 0, m, n, k, one.ctypes.data, dA.data_ptr(), m, dB.data_ptr(), k, zero.ctypes.data, dC.data_ptr(), m, N, 1,
                                 work.data_ptr(), workA.data_ptr(), workB.data_ptr(), 1, 1, skA, 0, None))
        torch.cuda.synchronize()
        L = g.Layout()
        g.check(lib.gemmul8_get_layout(g.S, g.FP8, m, n, k, N, work.data_ptr(), workA.data_ptr(), workB.data_ptr(), 1, 1, C.byref(L)))
        assert L.lo_format == 0
        sA = workA.cpu().numpy()[L.sftA - workA.data_ptr():][:2 * m].view(np.int16).copy()
        sB = workB.cpu().numpy()[L.sftB - workB.data_ptr():][:2 * n].view(np.int16).copy()
        return gu.from_dev(dC).copy(), sA, sB
    C1, sA1, sB1 = call(dB1, n1, 0)
    assert gu.bits_equal(C1, ol.gemm(A, B1, N, fastmode=True, backend=g.FP8, sftA_in=sA1, sftB_in=sB1))
    C2, sA2, sB2 = call(dB2, n2, 1)
    assert np.array_equal(sA1, sA2)
    assert gu.bits_equal(C2, ol.gemm(A, B2, N, fastmode=True, backend=g.FP8, sftA_in=sA2, sftB_in=sB2))


@pytest.mark.parametrize("panels", ["3", "3r", "64"])
def test_crt_panels_knob_is_bit_identical(panels, monkeypatch):
    """GEMMUL8_CRT_PANELS=<P>[r] (csrc/oz2_driver.hip; SURVEY 8 f3 by cache residency, measured negative in profiles/r06_panel_crt_cache_residency.txt):
    the real INT8 call as P column panels gemm(p) -> crt(p), optionally all through panel 0's columns of C_mid.  Same bits as the one-launch order, also
    with beta != 0, ragged edges and more panels asked for than there are tile columns."""
    import gpu_util as gu
    rng = np.random.default_rng(31)
    m, n, k = 300, 1000, 260
    A, B, C0 = rand((m, k), np.float64, rng), rand((k, n), np.float64, rng), rand((m, n), np.float64, rng)
    ref = gu.hip_gemm(A, B, 14, alpha=-1.5, beta=0.5, C0=C0)
    gu.setknob(monkeypatch, "GEMMUL8_CRT_PANELS", panels)
    got = gu.hip_gemm(A, B, 14, alpha=-1.5, beta=0.5, C0=C0)
    assert gu.bits_equal(got, ref)
    if not panels.endswith("r"):
        gu.parity_case(A, B, 14, True, alpha=-1.5, beta=0.5, C0=C0)   # every panel wrote its own columns: C_mid is the oracle's, too


@pytest.mark.parametrize("ops", ["NN", "TN", "NT", "TT"])
@pytest.mark.parametrize("backend,dtype,N", [("INT8", np.float64, 14), ("INT8", np.complex64, 7), ("FP8", np.float32, 6), ("FP8", np.complex128, 12)])
def test_scale_launch_folds_are_bit_identical(backend, dtype, N, ops, monkeypatch):
    """Round 6: accurate mode in 6 launches instead of 9 -- row maxima of row-strided operands as partial arrays (both operands in one launch, no atomics),
    both extracts in one launch that also zero-fills the maxima arrays, the shift finalize (scaling_accu_real.hpp:6-18) on the quantise launch.  Every operand
    orientation (both K-major, one, none), both backends incl. the FP6 lane-per-fragment writer: same shifts, planes, C_mid and C as the unfolded sequence
    (GEMMUL8_SCALE_FOLD=0) and as the oracle."""
    import gemmul8_amd as g
    import gpu_util as gu
    be = getattr(g, backend)
    opA, opB = ops
    rng = np.random.default_rng(500 + N)
    m, n, k = 150, 130, 1100
    A = rand((m, k) if opA == "N" else (k, m), dtype, rng)
    B = rand((k, n) if opB == "N" else (n, k), dtype, rng)
    A[3 if opA == "N" else slice(None), slice(None) if opA == "N" else 3] = 0          # an all-zero row of op(A): amax = 0, f(0) = 0
    Cf, itf = gu.hip_gemm(A, B, N, backend=be, opA=opA, opB=opB, want_intermediates=True)
    gu.setknob(monkeypatch, "GEMMUL8_SCALE_FOLD", "0")
    Cu, itu = gu.hip_gemm(A, B, N, backend=be, opA=opA, opB=opB, want_intermediates=True)
    gu.setknob(monkeypatch, "GEMMUL8_SCALE_FOLD", None)
    for key in ("sftA", "sftB", "A_lo", "B_lo", "C_mid"):
        assert np.array_equal(itf[key], itu[key]), key
    assert gu.bits_equal(Cf, Cu)
    gu.parity_case(A, B, N, False, opA=opA, opB=opB, backend=be)
