"""-m gpu: the tile-stationary INT8 GEMM kernel with the CRT accumulation inside against the CPU oracle and against
the two-launch path, bit for bit.  LABORATORY code since round 4: both in-kernel CRT forms measured 10-28 % slower than the
two-launch path and live in tools/experiments/fused_crt/ (libgemmul8_lab.so: the product's objects with the INT8
GEMM object replaced by the FUSE = 1 | 2 instantiations, and gemmul8_lab_gemm = gemmul8_scale + gemmul8_lowprec_gemm_crt).
libgemmul8.so exports none of it (last test).  The fused kernel still writes the residue planes to C_mid, so every intermediate
the parity tests look at stays comparable."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "tools", "experiments", "fused_crt", "lib", "libgemmul8_lab.so")


class _LabLib:
    """The laboratory library behind the interface gpu_util drives: gemmul8_gemm is its gemmul8_lab_gemm."""

    def __init__(self, cdll):
        self._l = cdll
        self.gemmul8_gemm = cdll.gemmul8_lab_gemm

    def __getattr__(self, name):
        return getattr(self._l, name)


@pytest.fixture(autouse=True)
def lab_library():
    """Every test of this module runs gpu_util's helpers on libgemmul8_lab.so."""
    import gemmul8_amd as g
    assert os.path.exists(LAB), "laboratory library not built: make -C tools/experiments/fused_crt (done by __graft_entry__.build())"
    product = g.lib()
    L = g.bind(C.CDLL(LAB))
    L.gemmul8_lab_gemm.restype = C.c_int
    L.gemmul8_lab_gemm.argtypes = product.gemmul8_gemm.argtypes
    L.gemmul8_lowprec_gemm_crt.restype = C.c_int
    L.gemmul8_lowprec_gemm_crt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.POINTER(g.Layout),
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.gemmul8_fused_crt_selected.restype = C.c_int
    L.gemmul8_fused_crt_selected.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_uint]
    g._lib = _LabLib(L)
    try:
        yield product
    finally:
        g._lib = product


def rand(shape, dtype, rng, phi=1.0):
    x = (rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))
    return x.astype(dtype)


@pytest.fixture(params=["1", "2"], ids=["producer-crt", "consumer-tail"])
def fused(monkeypatch, request):
    # read by gemmul8_lab_gemm on every call: fused whenever it is legal; 1 = CRT on the producer waves, 2 = CRT tail on the consumer waves
    monkeypatch.setenv("GEMMUL8_FUSED_CRT", request.param)


@pytest.mark.parametrize("dtype,N", [(np.float64, 2), (np.float64, 6), (np.float64, 7), (np.float64, 14), (np.float64, 16), (np.float64, 20),
                                     (np.float32, 2), (np.float32, 7), (np.float32, 13)])
@pytest.mark.parametrize("fast", [False, True])
def test_fused_parity_small(fused, dtype, N, fast):
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(7 * N + fast)
    m, n, k = 37, 41, 300          # odd m: column starts of C are only 8-byte aligned (scalar store path)
    A, B = rand((m, k), dtype, rng), rand((k, n), dtype, rng)
    A[5, :] = 0
    B[:, 7] = 0
    gu.parity_case(A, B, N, fast)


@pytest.mark.parametrize("m,n,k", [(300, 530, 200), (256, 256, 256), (257, 255, 130), (16, 600, 77), (1030, 48, 4500)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fused_parity_shapes(fused, m, n, k, dtype):
    """Several tiles per workgroup row/column, ragged edges, k past the K-step-barrier schedule's limit (ping-pong schedule)."""
    import gpu_util as gu
    rng = np.random.default_rng(m + n + k)
    A, B = rand((m, k), dtype, rng), rand((k, n), dtype, rng)
    gu.parity_case(A, B, 12 if dtype == np.float32 else 15, False)


@pytest.mark.parametrize("opA,opB", [("N", "N"), ("T", "N"), ("N", "T")])
@pytest.mark.parametrize("alpha,beta", [(1, 0), (1, 1), (-1, 0), (-1, 1), (-1.5, 1.5)])
def test_fused_ops_axpby(fused, opA, opB, alpha, beta):
    import gpu_util as gu
    rng = np.random.default_rng(11)
    m, n, k = 144, 80, 190         # m a multiple of 16: vector loads / stores of C
    A = rand((m, k) if opA == "N" else (k, m), np.float64, rng)
    B = rand((k, n) if opB == "N" else (n, k), np.float64, rng)
    C0 = rand((m, n), np.float64, rng)
    gu.parity_case(A, B, 14, False, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0)


@pytest.mark.parametrize("dtype,N", [(np.float64, 14), (np.float32, 7)])
@pytest.mark.parametrize("alpha,beta", [(1, 0), (-1, 1), (-1.5, 1.5)])
def test_fused_device_pointer_scalars_and_ldc(fused, dtype, N, alpha, beta):
    """alpha / beta in device memory (general fma form, oracle scalar_mode = 1) and a C with ldc > m whose padding must survive."""
    import gemmul8_amd as g
    import gpu_util as gu
    import oracle_lib as ol
    rng = np.random.default_rng(5)
    m, n, k, ldc = 150, 70, 210, 163
    A, B, C0 = rand((m, k), dtype, rng), rand((k, n), dtype, rng), rand((m, n), dtype, rng)
    dA, dB = gu.to_dev(A), gu.to_dev(B)
    dC = torch.full((n, ldc), 7.25, dtype=dA.dtype, device="cuda")
    dC[:, :m] = gu.to_dev(C0)
    d_al = torch.tensor([alpha], dtype=dA.dtype, device="cuda")
    d_be = torch.tensor([beta], dtype=dA.dtype, device="cuda")
    tot, _, _ = g.work_size(False, g.INT8, m, n, k, N)
    work = torch.zeros(tot, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    code = g._dtype_code(dA.dtype)
    g.check(g.lib().gemmul8_gemm(st, code, g.INT8, 0, 0, m, n, k, d_al.data_ptr(), dA.data_ptr(), m, dB.data_ptr(), k, d_be.data_ptr(),
                                 dC.data_ptr(), ldc, N, 0, work.data_ptr(), None, None, 0, 0, 0, 0, None))
    torch.cuda.synchronize()
    L = g.Layout()
    g.check(g.lib().gemmul8_get_layout(code, g.INT8, m, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(L)))
    w = work.cpu().numpy()
    base = work.data_ptr()
    sA = w[L.sftA - base:][:2 * m].view(np.int16).copy()
    sB = w[L.sftB - base:][:2 * n].view(np.int16).copy()
    Co = ol.gemm(A, B, N, alpha=alpha, beta=beta, C0=C0, scalar_mode=1, sftA_in=sA, sftB_in=sB)
    got = dC.cpu().numpy()
    assert gu.bits_equal(got[:, :m].T, Co)
    assert np.all(got[:, m:] == 7.25), "padding between the columns of C was written"


@pytest.mark.parametrize("m,n,k,N,dtype", [(4352, 4096, 192, 14, np.float64),     # 272 tiles: two rounds on 256 workgroups
                                            (2048, 2304, 1024, 14, np.float64),
                                            (1024, 1024, 4352, 15, np.float64),    # kp > 4096: ping-pong schedule
                                            (3072, 1536, 512, 7, np.float32)])
def test_fused_equals_two_launch_path(monkeypatch, m, n, k, N, dtype):
    """Sizes beyond the oracle's reach: the fused launch against lowprec_gemm + crt on the same planes -- C_mid and C bitwise."""
    import gemmul8_amd as g
    import gpu_util as gu
    rng = np.random.default_rng(m + k)
    A, B, C0 = rand((m, k), dtype, rng), rand((k, n), dtype, rng), rand((m, n), dtype, rng)
    out = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("GEMMUL8_FUSED_CRT", mode)
        out[mode] = gu.hip_gemm(A, B, N, alpha=-1.5, beta=0.5, C0=C0, want_intermediates=True)
    for mode in ("1", "2"):
        assert np.array_equal(out["0"][1]["C_mid"], out[mode][1]["C_mid"])
        assert gu.bits_equal(out["0"][0], out[mode][0]), mode
    ref = A.astype(np.float64) @ B.astype(np.float64)
    ref = -1.5 * ref + 0.5 * C0
    err = np.max(np.abs(out["1"][0] - ref)) / np.max(np.abs(ref))
    assert err < (1e-5 if dtype == np.float32 else 1e-13), err


def test_fused_selection_rule(monkeypatch):
    """Opt-in: GEMMUL8_FUSED_CRT=1 whenever legal, =auto when the tiles of ONE plane fill the chip about as well as the tiles of all
    planes do; unset = the two-launch path."""
    import gemmul8_amd as g
    sel = g.lib().gemmul8_fused_crt_selected
    monkeypatch.delenv("GEMMUL8_FUSED_CRT", raising=False)
    assert sel(g.D, g.INT8, 8192, 8192, 14) == 0
    monkeypatch.setenv("GEMMUL8_FUSED_CRT", "auto")
    assert sel(g.D, g.INT8, 8192, 8192, 14) == 1 and sel(g.D, g.INT8, 16384, 16384, 16) == 1 and sel(g.S, g.INT8, 4096, 4096, 7) == 1
    assert sel(g.D, g.INT8, 2048, 2048, 14) == 0          # 64 tiles: a quarter of the chip
    monkeypatch.setenv("GEMMUL8_FUSED_CRT", "1")
    assert sel(g.D, g.INT8, 2048, 2048, 14) == 1
    assert sel(g.Z, g.INT8, 8192, 8192, 14) == 0 and sel(g.D, g.FP8, 8192, 8192, 14) == 0
    assert g.lib().gemmul8_lowprec_gemm_crt(None, g.Z, g.INT8, 8, 8, 8, 14, C.byref(g.Layout()), C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), 8) == -3


@pytest.mark.parametrize("alpha", [1.0, -1.5])
def test_fused_beta0_never_reads_c(fused, alpha):
    """beta == 0 must not read C (round-3 ADVICE: the stand-alone CRT kernels had the rule, the in-kernel forms evaluated fma(0, C, alpha*AB)):
    a NaN-filled C comes back finite and bit-equal to the oracle."""
    import gemmul8_amd as g
    import gpu_util as gu
    import oracle_lib as ol
    rng = np.random.default_rng(3)
    m, n, k, N = 144, 80, 190, 14
    A, B = rand((m, k), np.float64, rng), rand((k, n), np.float64, rng)
    C0 = np.full((m, n), np.nan)
    got, inter = gu.hip_gemm(A, B, N, alpha=alpha, beta=0.0, C0=C0, want_intermediates=True)
    assert np.isfinite(got).all()
    want = ol.gemm(A, B, N, alpha=alpha, beta=0.0, C0=np.zeros((m, n)), sftA_in=inter["sftA"], sftB_in=inter["sftB"])
    assert gu.bits_equal(got, want)


def test_product_library_has_no_in_kernel_crt(lab_library):
    """libgemmul8.so instantiates FUSE = 0 only and exports neither laboratory entry point."""
    product = lab_library
    for name in ("gemmul8_lowprec_gemm_crt", "gemmul8_fused_crt_selected", "gemmul8_lab_gemm"):
        assert not hasattr(product, name), name
