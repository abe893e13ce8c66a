"""ctypes front-end for the CPU oracle (oracle/liboz2_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liboz2_oracle.so")

DT = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.complex64): 2, np.dtype(np.complex128): 3}
INT8, FP8 = 0, 1
OPS = {"N": 0, "T": 1, "C": 2}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
        _lib.oz2_num_mat.restype = C.c_uint
        _lib.oz2_work_size.restype = C.c_size_t
        _lib.oz2_work_size.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.c_int, C.c_int,
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _lib.oz2_gemm.restype = C.c_int
        _lib.oz2_gemm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                  C.c_uint, C.c_int, C.c_int] + [C.c_void_p] * 7
        _lib.oz2_gemm_mod.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_uint, C.c_uint]
        _lib.oz2_invscal.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _lib.oz2_quantise.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p,
                                      C.c_size_t, C.c_void_p, C.c_void_p]
        _lib.oz2_extract.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_void_p]
        _lib.oz2_bound_shifts.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib.oz2_bound_maxima_i8.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                             C.c_size_t, C.c_void_p, C.c_void_p]
        _lib.oz2_bound_maxima_f8.argtypes = _lib.oz2_bound_maxima_i8.argtypes
        _lib.oz2_shift_finalize_i8.argtypes = [C.c_int, C.c_uint, C.c_size_t, C.c_void_p, C.c_void_p]
        _lib.oz2_invscal_grouped.argtypes = _lib.oz2_invscal.argtypes + [C.c_uint, C.c_void_p]
        _lib.oz2_crt_partial.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_void_p]
        _lib.oz2_crt_finish.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _lib.oz2_fast_shifts.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p,
                                         C.c_size_t, C.c_void_p]
    return _lib


def num_mat(backend, N):
    return lib().oz2_num_mat(backend, N)


def work_size(cplx, backend, m, n, k, N, enA=False, enB=False):
    wa, wb = C.c_size_t(0), C.c_size_t(0)
    tot = lib().oz2_work_size(int(cplx), backend, m, n, k, N, int(enA), int(enB), C.byref(wa), C.byref(wb))
    return tot, wa.value, wb.value


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def op_dims(op, rows, k):
    """Stored shape (column-major, as a Fortran-ordered numpy array) of an operand whose op() is rows x k."""
    return (rows, k) if op == "N" else (k, rows)


def gemm(A, B, N, fastmode=False, backend=INT8, opA="N", opB="N", alpha=1.0, beta=0.0, C0=None, scalar_mode=0,
         sftA_in=None, sftB_in=None, want_intermediates=False):
    """Run the oracle pipeline.  A, B: Fortran-ordered 2-D arrays as stored (before op).  Returns C (m x n, F-order)
    and, if requested, a dict with sftA, sftB, A_lo, B_lo, C_mid."""
    A = np.asfortranarray(A)
    B = np.asfortranarray(B)
    dt = A.dtype
    assert B.dtype == dt
    m, k = (A.shape if opA == "N" else A.shape[::-1])
    kb, n = (B.shape if opB == "N" else B.shape[::-1])
    assert k == kb
    cplx = dt.kind == "c"
    parts = 3 if cplx else 1
    nm = num_mat(backend, N)
    Cout = np.zeros((m, n), dtype=dt, order="F") if C0 is None else np.asfortranarray(C0).copy(order="F")
    al = np.array([alpha], dtype=dt)
    be = np.array([beta], dtype=dt)
    inter = {}
    sA = sB = Alo = Blo = Cmid = None
    if want_intermediates:
        sA = np.zeros(m, np.int16)
        sB = np.zeros(n, np.int16)
        Alo = np.zeros((parts, nm, m, k), np.uint8)
        Blo = np.zeros((parts, nm, n, k), np.uint8)
        mid_dt = np.int8 if backend == INT8 else np.int16
        Cmid = np.zeros((N, n, m, 2) if cplx else (N, n, m), mid_dt)
    if sftA_in is not None:
        sftA_in = np.ascontiguousarray(sftA_in, np.int16)
    if sftB_in is not None:
        sftB_in = np.ascontiguousarray(sftB_in, np.int16)
    rc = lib().oz2_gemm(DT[dt], backend, OPS[opA], OPS[opB], m, n, k, _p(al), _p(A), A.shape[0], _p(B), B.shape[0],
                        _p(be), _p(Cout), Cout.shape[0], N, int(fastmode), scalar_mode, _p(sftA_in), _p(sftB_in),
                        _p(sA), _p(sB), _p(Alo), _p(Blo), _p(Cmid))
    assert rc == 0
    if want_intermediates:
        inter = dict(sftA=sA, sftB=sB, A_lo=Alo, B_lo=Blo, C_mid=Cmid)
        return Cout, inter
    return Cout


def gemm_embedded(bufA, offA, lda, bufB, offB, ldb, bufC, offC, ldc, m, n, k, N, fastmode=False, backend=INT8, opA="N", opB="N", alpha=1.0,
                  beta=0.0, sftA_in=None, sftB_in=None):
    """The oracle pipeline on SUB-MATRIX VIEWS: op(A), op(B), C live inside larger column-major buffers (1-D numpy arrays of the element
    type) at element offsets off* with leading dimensions ld* >= their row counts -- the calling pattern of the hook's LU / QR trailing
    updates (src/hook.cu:609-730 forwards the caller's lda / ldb / ldc; include/gemmul8.hpp:107-112).  bufC is updated IN PLACE.
    Returns the intermediates dict of `gemm`."""
    dt = bufA.dtype
    assert bufB.dtype == dt and bufC.dtype == dt and bufA.ndim == bufB.ndim == bufC.ndim == 1
    cplx = dt.kind == "c"
    parts = 3 if cplx else 1
    nm = num_mat(backend, N)
    al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
    sA, sB = np.zeros(m, np.int16), np.zeros(n, np.int16)
    Alo = np.zeros((parts, nm, m, k), np.uint8)
    Blo = np.zeros((parts, nm, n, k), np.uint8)
    Cmid = np.zeros((N, n, m, 2) if cplx else (N, n, m), np.int8 if backend == INT8 else np.int16)
    if sftA_in is not None:
        sftA_in = np.ascontiguousarray(sftA_in, np.int16)
    if sftB_in is not None:
        sftB_in = np.ascontiguousarray(sftB_in, np.int16)
    isz = dt.itemsize
    rc = lib().oz2_gemm(DT[dt], backend, OPS[opA], OPS[opB], m, n, k, _p(al), bufA.ctypes.data + offA * isz, lda, bufB.ctypes.data + offB * isz, ldb,
                        _p(be), bufC.ctypes.data + offC * isz, ldc, N, int(fastmode), 0, _p(sftA_in), _p(sftB_in), _p(sA), _p(sB), _p(Alo), _p(Blo), _p(Cmid))
    assert rc == 0
    return dict(sftA=sA, sftB=sB, A_lo=Alo, B_lo=Blo, C_mid=Cmid)


def accurate_shifts(A, B, N, backend=INT8):
    """Accurate-mode shifts only (extract + bound GEMM + finalize, no modular planes): A (m x k), B (k x n) as stored for
    op N/N.  Returns (sftA[m], sftB[n]) NEGATED like the workspace holds them.  Cheap for thin A or thin B."""
    A = np.asfortranarray(A)
    B = np.asfortranarray(B)
    dt = A.dtype
    m, k = A.shape
    _, n = B.shape
    cplx = dt.kind == "c"
    parts = 3 if cplx else 1
    Ab = np.zeros(parts * m * k, np.uint8)
    Bb = np.zeros(parts * n * k, np.uint8)
    sA = np.zeros(m, np.int16)
    sB = np.zeros(n, np.int16)
    L = lib()
    L.oz2_extract(DT[dt], backend, 0, 0, m, k, _p(A), A.shape[0], _p(Ab), _p(sA))
    L.oz2_extract(DT[dt], backend, 1, 0, n, k, _p(B), B.shape[0], _p(Bb), _p(sB))
    L.oz2_bound_shifts(backend, int(cplx), N, m, n, k, _p(Ab), _p(Bb), _p(sA), _p(sB), 1, 1)
    return sA, sB


def extract_bounds(X, op, is_A, backend=INT8):
    """Accurate-mode extract of one operand as stored (before op): returns (planes uint8 [(1|3)][rows][k], sft0 int16[rows]).
    A: rows = m, K-major iff op != N; B: rows = n, K-major iff op == N (scaling_accu_real.hpp:23-136)."""
    X = np.asfortranarray(X)
    dt = X.dtype
    kmajor = (op != "N") if is_A else (op == "N")
    rows, k = (X.shape[1], X.shape[0]) if kmajor else X.shape
    parts = 3 if dt.kind == "c" else 1
    lo = np.zeros((parts, rows, k), np.uint8)
    sft0 = np.zeros(rows, np.int16)
    lib().oz2_extract(DT[dt], backend, int(kmajor), int(op == "C"), rows, k, _p(X), X.shape[0], _p(lo), _p(sft0))
    return lo, sft0


def bound_maxima(Abar, Bbar, backend=INT8, c0=0, c1=None):
    """Row / column maxima of the bound product (find_max.hpp:67-114): int32 for INT8, float32 (inflated) for FP8."""
    parts, m, k = Abar.shape
    n = Bbar.shape[1]
    c1 = n if c1 is None else c1
    dt = np.int32 if backend == INT8 else np.float32
    rmax, cmax = np.zeros(m, dt), np.zeros(n, dt)
    fn = lib().oz2_bound_maxima_i8 if backend == INT8 else lib().oz2_bound_maxima_f8
    fn(int(parts == 3), m, n, k, _p(np.ascontiguousarray(Abar)), _p(np.ascontiguousarray(Bbar)), c0, c1, _p(rmax), _p(cmax))
    return rmax, cmax


def e4m3_decode(b):
    """OCP e4m3fn byte(s) -> float64 (sign ignored: bound planes are non-negative except the complex difference plane)."""
    b = np.asarray(b, np.uint8).astype(np.int64)
    e, mnt = (b >> 3) & 15, b & 7
    v = np.where(e == 0, mnt * 2.0 ** -9, (8 + mnt) * 2.0 ** (e - 10.0))
    return np.where(b & 0x80, -v, v)


def bound_maxima_f8_exact(Abar, Bbar):
    """Exactly accumulated, UN-inflated row / column maxima of the real FP8 bound product (float64 holds every partial sum exactly:
    products are multiples of 2^-18 below 2^16, k <= 65536).  What an accurate-mode bound must not fall below."""
    P = e4m3_decode(Abar[0]) @ e4m3_decode(Bbar[0]).T
    return P.max(axis=1), P.max(axis=0)


def bound_maxima_f8_exact_cplx(Abar, Bbar):
    """Complex FP8 bound, exactly accumulated and UN-inflated, from the three e4m3 bound planes (|Re|, |Im|, RU(|Re| - |Im|)) of each
    operand: element-wise max(T, C1) with C1 = |Ar||Bi| + |Ai||Br| and T = C0 + C1, C0 = D_A D_B (find_max.hpp:117-140 without the
    inflation).  What the device's accurate-mode bound must not fall below, whatever the engine does to the sums."""
    ar, ai, ad = (e4m3_decode(Abar[i]) for i in range(3))
    br, bi, bd = (e4m3_decode(Bbar[i]) for i in range(3))
    C1 = ar @ bi.T + ai @ br.T
    T = ad @ bd.T + C1
    P = np.maximum(T, C1)
    return P.max(axis=1), P.max(axis=0)


FP8_BOUND_SAFE, FP8_BOUND_REFERENCE = 0, 1


def set_fp8_bound_mode(mode):
    """1 = the reference's (k+1)*2^-24 (find_max.hpp:82-96) -- THE ORACLE'S DEFAULT; 0 = the product's engine-safe inflation (the product's
    default: a test comparing it with the oracle selects 0 on both sides), 2 = the product's round-3 default."""
    lib().oz2_set_fp8_bound_mode(int(mode))


def get_fp8_bound_mode():
    return int(lib().oz2_get_fp8_bound_mode())


def fp8_bound_ku(k, mode=0):
    ieee = np.float32(k + 1) * np.float32(2.0 ** -24)
    return float(ieee) if mode == 1 else float(np.float32(1.75 * 2.0 ** -11) + np.float32(4.0) * ieee)
