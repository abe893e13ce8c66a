"""-m gpu: gemmul8_gemm_batched (a strided batch as ONE set of launches: gridDim.z items in every kernel, the items' residue planes
folded into the persistent GEMM kernels' plane sequence) against per-item gemmul8_gemm calls -- bitwise -- and against the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TORCH_DT = {np.float32: torch.float32, np.float64: torch.float64, np.complex64: torch.complex64, np.complex128: torch.complex128}


def rand(shape, dtype, rng):
    x = (rng.random(shape) - 0.5) * np.exp(rng.standard_normal(shape))
    if np.dtype(dtype).kind == "c":
        x = x + 1j * (rng.random(shape) - 0.5) * np.exp(rng.standard_normal(shape))
    return x.astype(dtype)


def run_batched(A, B, C0, N, fast, opA, opB, alpha, beta, strideB_zero=False, backend=None):
    """A: (batch, colsA, rowsA) torch (column-major items), likewise B, C0.  Returns (batched C, per-item C)."""
    import gemmul8_amd as g
    be_ = g.INT8 if backend is None else backend
    batch = A.shape[0]
    lda, ldb, ldc = A.shape[2], B.shape[2], C0.shape[2]
    m, k = (lda, A.shape[1]) if opA == "N" else (A.shape[1], lda)
    n = B.shape[1] if opB == "N" else ldb
    dt = A.dtype
    code = g._dtype_code(dt)
    np_dt = {v: k_ for k_, v in TORCH_DT.items()}[dt]
    al, be = np.array([alpha], np_dt), np.array([beta], np_dt)
    st = torch.cuda.current_stream().cuda_stream
    Cb = C0.clone()
    work = torch.empty(g.lib().gemmul8_work_size_batched(int(dt.is_complex), be_, m, n, k, N, batch), dtype=torch.uint8, device="cuda")
    sB = 0 if strideB_zero else B.shape[1] * B.shape[2]
    g.check(g.lib().gemmul8_gemm_batched(st, code, be_, g.OPS[opA], g.OPS[opB], m, n, k, al.ctypes.data, A.data_ptr(), lda,
                                         A.shape[1] * A.shape[2], B.data_ptr(), ldb, sB, be.ctypes.data, Cb.data_ptr(), ldc,
                                         C0.shape[1] * C0.shape[2], batch, N, int(fast), work.data_ptr()))
    Ci = C0.clone()
    for b in range(batch):
        g.gemm(A[b], B[0 if strideB_zero else b], N, fastmode=fast, backend=be_, opA=opA, opB=opB, alpha=alpha, beta=beta, C_out=Ci[b])
    torch.cuda.synchronize()
    return Cb, Ci


@pytest.mark.parametrize("dtype,N", [(np.float64, 14), (np.float32, 7), (np.complex128, 15), (np.complex64, 8)])
@pytest.mark.parametrize("fast", [False, True])
def test_batched_equals_per_item(dtype, N, fast):
    rng = np.random.default_rng(N + fast)
    batch, m, n, k = 5, 150, 70, 210
    td = TORCH_DT[dtype]
    A = torch.from_numpy(rand((batch, k, m), dtype, rng)).cuda()
    B = torch.from_numpy(rand((batch, n, k), dtype, rng)).cuda()
    C0 = torch.from_numpy(rand((batch, n, m), dtype, rng)).cuda()
    alpha, beta = (-1.5, 0.5) if np.dtype(dtype).kind != "c" else (-1.5 + 0.5j, 0.5 - 0.25j)
    Cb, Ci = run_batched(A, B, C0, N, fast, "N", "N", alpha, beta)
    assert torch.equal(Cb.view(torch.uint8), Ci.view(torch.uint8)), int((Cb != Ci).sum())
    assert td == Cb.dtype


@pytest.mark.parametrize("opA,opB", [("T", "N"), ("N", "T"), ("T", "T")])
@pytest.mark.parametrize("m,n,k,batch", [(37, 41, 300, 3), (300, 530, 64, 2), (256, 256, 256, 9)])
def test_batched_ops_and_shapes(opA, opB, m, n, k, batch):
    rng = np.random.default_rng(m + n + k)
    A = torch.from_numpy(rand((batch,) + ((k, m) if opA == "N" else (m, k)), np.float64, rng)).cuda()
    B = torch.from_numpy(rand((batch,) + ((n, k) if opB == "N" else (k, n)), np.float64, rng)).cuda()
    C0 = torch.from_numpy(rand((batch, n, m), np.float64, rng)).cuda()
    Cb, Ci = run_batched(A, B, C0, 14, False, opA, opB, 1.0, 0.0)
    assert torch.equal(Cb.view(torch.uint8), Ci.view(torch.uint8)), int((Cb != Ci).sum())


def test_batched_shared_operand_and_oracle():
    """strideB = 0 (one B for every item, as torch.matmul broadcasts produce) and one item against the CPU oracle."""
    import gpu_util as gu
    import oracle_lib as ol
    rng = np.random.default_rng(9)
    batch, m, n, k = 4, 45, 33, 147
    An, Bn = rand((batch, k, m), np.float64, rng), rand((1, n, k), np.float64, rng)
    A, B = torch.from_numpy(An).cuda(), torch.from_numpy(Bn).cuda()
    C0 = torch.zeros((batch, n, m), dtype=torch.float64, device="cuda")
    Cb, Ci = run_batched(A, B, C0, 15, True, "N", "N", 1.0, 0.0, strideB_zero=True)
    assert torch.equal(Cb.view(torch.uint8), Ci.view(torch.uint8))
    # fast mode: the shifts depend on the operand only, so the oracle with its own shifts must agree up to the log2 policy;
    # compare against the oracle fed with nothing: values to rounding
    ref = ol.gemm(np.asfortranarray(An[2].T), np.asfortranarray(Bn[0].T), 15, fastmode=True)
    got = Cb[2].cpu().numpy().T
    assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref))


@pytest.mark.parametrize("dtype,N", [(np.float64, 12), (np.float32, 6), (np.complex128, 9), (np.complex64, 7)])
@pytest.mark.parametrize("fast", [False, True])
def test_batched_fp8_backend_equals_per_item(dtype, N, fast):
    """The FP8 backend in the batched entry point (round 3: the items fold into the FP8 kernels' plane sequence like the INT8 ones):
    three / nine GEMMs per modulus with int16 scratch planes per item, float bound maxima, complex bound stages -- bitwise equal to
    per-item gemmul8_gemm calls."""
    import gemmul8_amd as g
    rng = np.random.default_rng(100 + N + fast)
    batch, m, n, k = 4, 150, 70, 210
    A = torch.from_numpy(rand((batch, k, m), dtype, rng)).cuda()
    B = torch.from_numpy(rand((batch, n, k), dtype, rng)).cuda()
    C0 = torch.from_numpy(rand((batch, n, m), dtype, rng)).cuda()
    alpha, beta = (-1.5, 0.5) if np.dtype(dtype).kind != "c" else (-1.5 + 0.5j, 0.5 - 0.25j)
    Cb, Ci = run_batched(A, B, C0, N, fast, "N", "N", alpha, beta, backend=g.FP8)
    assert torch.equal(Cb.view(torch.uint8), Ci.view(torch.uint8)), int((Cb != Ci).sum())


def test_batched_work_size_and_fp8_k_limit():
    import gemmul8_amd as g
    one = np.array([1.0])
    x = torch.zeros(64, dtype=torch.float64, device="cuda")
    rc = g.lib().gemmul8_gemm_batched(None, g.D, g.FP8, 0, 0, 8, 8, 65537, one.ctypes.data, x.data_ptr(), 8, 64, x.data_ptr(), 8, 64,
                                      one.ctypes.data, x.data_ptr(), 8, 64, 1, 8, 0, x.data_ptr())
    assert rc == -2   # FP8: k <= 65536 (exact FP32 accumulation), as in gemmul8_gemm
    assert g.lib().gemmul8_work_size_batched(0, g.INT8, 100, 100, 100, 14, 7) == 7 * g.lib().gemmul8_batched_item_bytes(0, g.INT8, 100, 100, 100, 14) + 256


def test_python_wrapper_broadcasts():
    """gemmul8_amd.gemm_batched: B with a batch dimension of 1 is shared by every item (stride 0)."""
    import gemmul8_amd as g
    rng = np.random.default_rng(21)
    A = torch.from_numpy(rand((6, 96, 130), np.float64, rng)).cuda()    # six 130 x 96 matrices
    B = torch.from_numpy(rand((1, 50, 96), np.float64, rng)).cuda()     # one 96 x 50 matrix
    Cb, _ = g.gemm_batched(A, B, 14)
    for b in range(6):
        Ci, _, _ = g.gemm(A[b], B[0], 14)
        assert torch.equal(Cb[b].view(torch.uint8), Ci.view(torch.uint8)), b


@pytest.mark.parametrize("seed", range(12))
def test_batched_fuzz(seed):
    """Random shapes / ops / types / modes / batch counts: the one-launch-set path against per-item calls, bitwise."""
    rng = np.random.default_rng(1000 + seed)
    dtype = [np.float64, np.float32, np.complex128, np.complex64][seed % 4]
    N = int(rng.integers(2, 14 if np.dtype(dtype).itemsize in (4, 8) and dtype in (np.float32, np.complex64) else 21))
    m, n, k = (int(rng.integers(1, 400)) for _ in range(3))
    batch = int(rng.integers(2, 7))
    opA, opB = rng.choice(["N", "T"] + (["C"] if np.dtype(dtype).kind == "c" else [])), rng.choice(["N", "T"])
    fast = bool(rng.integers(0, 2))
    A = torch.from_numpy(rand((batch,) + ((k, m) if opA == "N" else (m, k)), dtype, rng)).cuda()
    B = torch.from_numpy(rand((batch,) + ((n, k) if opB == "N" else (k, n)), dtype, rng)).cuda()
    C0 = torch.from_numpy(rand((batch, n, m), dtype, rng)).cuda()
    alpha, beta = [(1.0, 0.0), (-1.0, 1.0), (0.75, -0.5)][seed % 3]
    Cb, Ci = run_batched(A, B, C0, N, fast, str(opA), str(opB), alpha, beta)
    assert torch.equal(Cb.view(torch.uint8), Ci.view(torch.uint8)), (dtype, N, m, n, k, batch, opA, opB, fast)


@pytest.mark.parametrize("dtype,m", [(np.float64, 1024), (np.complex128, 512)])
def test_batched_with_dma_crt_kernel(dtype, m, monkeypatch):
    """The LDS-DMA form of the CRT kernel inside a batched launch (item in gridDim.z: workspace and C offsets per item), forced with
    GEMMUL8_CRT_KERNEL=dma on a shape with whole 1024-byte units per column; beta != 0 so that the old C is read per item."""
    import gpu_util as gu
    gu.setknob(monkeypatch, "GEMMUL8_CRT_KERNEL", "dma")
    rng = np.random.default_rng(77)
    batch, n, k = 3, 40, 96
    A = torch.from_numpy(rand((batch, k, m), dtype, rng)).cuda()
    B = torch.from_numpy(rand((batch, n, k), dtype, rng)).cuda()
    C0 = torch.from_numpy(rand((batch, n, m), dtype, rng)).cuda()
    alpha, beta = (-1.5, 0.5) if np.dtype(dtype).kind != "c" else (-1.5 + 0.5j, 0.5 - 0.25j)
    Cb, Ci = run_batched(A, B, C0, 14, False, "N", "N", alpha, beta)
    assert torch.equal(Cb.view(torch.uint8), Ci.view(torch.uint8)), int((Cb != Ci).sum())
