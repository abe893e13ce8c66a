"""gemmul8_amd -- Python plumbing over the C ABI of libgemmul8.so (MI355X-native Ozaki-II GEMM emulation).

The product is the shared library (hand-written HIP for gfx950, gemmul8_amd/csrc) behind
include/gemmul8_c.h / include/gemmul8.hpp.  This package only (1) loads it with ctypes, (2) passes
torch device pointers / HIP streams through the C ABI, and (3) hosts the moduli-sharded multi-GPU
driver (gemmul8_amd.dist) on torch.distributed.  There is NO CPU fallback: if the library is missing
every entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgemmul8.so")

S, D, Cx, Z = 0, 1, 2, 3
INT8, FP8 = 0, 1
OPS = {"N": 0, "T": 1, "C": 2}


class Layout(C.Structure):
    """Mirror of struct gemmul8_layout (include/gemmul8_c.h)."""
    _fields_ = [
        ("kp", C.c_size_t), ("mp", C.c_size_t), ("num_mat", C.c_size_t), ("parts", C.c_size_t),
        ("sizeA", C.c_size_t), ("sizeB", C.c_size_t), ("sizeC", C.c_size_t),
        ("A_lo", C.c_void_p), ("B_lo", C.c_void_p),
        ("part_strideA", C.c_size_t), ("part_strideB", C.c_size_t),
        ("A_bound", C.c_void_p), ("B_bound", C.c_void_p),
        ("sftA", C.c_void_p), ("sftB", C.c_void_p),
        ("C_mid", C.c_void_p), ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t),
        ("lo_format", C.c_size_t),
    ]


_lib = None

EXPORTS = ["gemmul8_version", "gemmul8_work_size", "gemmul8_gemm", "gemmul8_get_layout", "gemmul8_scale",
           "gemmul8_scale_bounds", "gemmul8_scale_finish", "gemmul8_lowprec_gemm", "gemmul8_crt", "gemmul8_set_fp8_bound_mode",
           "gemmul8_hook_would_emulate", "gemmul8_reload_knobs", "gemmul8_abi_version", "gemmul8_layout_bytes"]

ABI_VERSION = 7  # GEMMUL8_ABI_VERSION of include/gemmul8_c.h this module's struct mirrors were written against


def _bind_hip_runtime():
    """libgemmul8.so carries no DT_NEEDED on the HIP runtime: it must bind to the SAME libamdhip64 the
    process already uses (PyTorch bundles its own copy; two runtimes do not share device pointers).
    Promote the loaded copy to RTLD_GLOBAL, or load the system one if none is mapped yet."""
    try:
        import torch  # noqa: F401  (maps torch/lib/libamdhip64.so)
    except Exception:
        pass
    path = None
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
    if path is None:
        path = "/opt/rocm/lib/libamdhip64.so"
    C.CDLL(path, mode=C.RTLD_GLOBAL)
    return path


def lib():
    """Load libgemmul8.so (fails loudly if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: the HIP extension is not built (python -c 'import __graft_entry__ as g; g.build()')")
    _bind_hip_runtime()
    _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def bind(L):
    """Declare the C-ABI signatures (include/gemmul8_c.h) on a loaded library object: libgemmul8.so, or a laboratory build of it
    (tools/build_probes.sh, tools/experiments/) that a measurement script wants to drive through the same helpers."""
    L.gemmul8_version.restype = C.c_char_p
    L.gemmul8_abi_version.restype = C.c_int
    L.gemmul8_layout_bytes.restype = C.c_size_t
    if L.gemmul8_abi_version() != ABI_VERSION or L.gemmul8_layout_bytes() != C.sizeof(Layout):
        raise RuntimeError(f"libgemmul8.so has ABI version {L.gemmul8_abi_version()} / a {L.gemmul8_layout_bytes()}-byte gemmul8_layout; this module mirrors "
                           f"version {ABI_VERSION} / {C.sizeof(Layout)} bytes: rebuild the library (python -c 'import __graft_entry__ as g; g.build()')")
    L.gemmul8_work_size.restype = C.c_size_t
    L.gemmul8_work_size.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.c_int, C.c_int,
                                    C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.gemmul8_gemm.restype = C.c_int
    L.gemmul8_gemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                               C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                               C.c_size_t, C.c_uint, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.gemmul8_get_layout.restype = C.c_int
    L.gemmul8_get_layout.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Layout)]
    L.gemmul8_scale.restype = C.c_int
    L.gemmul8_scale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint, C.c_int, C.c_uint, C.c_uint,
                                C.POINTER(Layout), C.c_int, C.c_int]
    L.gemmul8_scale_bounds.restype = C.c_int
    L.gemmul8_scale_bounds.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint, C.c_size_t, C.c_size_t,
                                       C.POINTER(Layout), C.c_int, C.c_int]
    L.gemmul8_scale_finish.restype = C.c_int
    L.gemmul8_scale_finish.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint, C.c_int, C.c_uint, C.c_uint,
                                       C.POINTER(Layout), C.c_int, C.c_int]
    L.gemmul8_lowprec_gemm.restype = C.c_int
    L.gemmul8_lowprec_gemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint,
                                       C.c_uint, C.c_uint, C.POINTER(Layout)]
    L.gemmul8_crt.restype = C.c_int
    L.gemmul8_crt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                              C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.gemmul8_work_size_batched.restype = C.c_size_t
    L.gemmul8_work_size_batched.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.c_size_t]
    L.gemmul8_batched_item_bytes.restype = C.c_size_t
    L.gemmul8_batched_item_bytes.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint]
    L.gemmul8_gemm_batched.restype = C.c_int
    L.gemmul8_gemm_batched.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_longlong, C.c_void_p, C.c_size_t, C.c_longlong, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_longlong, C.c_size_t, C.c_uint, C.c_int, C.c_void_p]
    L.gemmul8_set_fp8_bound_mode.restype = C.c_int
    L.gemmul8_set_fp8_bound_mode.argtypes = [C.c_int]
    L.gemmul8_hook_would_emulate.restype = C.c_int
    L.gemmul8_hook_would_emulate.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.c_int, C.c_size_t]
    L.gemmul8_reload_knobs.restype = None
    L.gemmul8_reload_knobs.argtypes = []
    L.gemmul8_add_f64.restype = C.c_int
    L.gemmul8_add_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.gemmul8_add_row_bias.restype = C.c_int
    L.gemmul8_add_row_bias.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    return L


def work_size(is_complex, backend, m, n, k, num_moduli, enA=False, enB=False):
    wa, wb = C.c_size_t(0), C.c_size_t(0)
    tot = lib().gemmul8_work_size(int(is_complex), backend, m, n, k, num_moduli, int(enA), int(enB), C.byref(wa), C.byref(wb))
    return tot, wa.value, wb.value


def check(rc, what="gemmul8"):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc}")


def _dtype_code(t):
    import torch
    return {torch.float32: S, torch.float64: D, torch.complex64: Cx, torch.complex128: Z}[t]


def gemm(A, B, num_moduli, fastmode=False, backend=INT8, opA="N", opB="N", alpha=1.0, beta=0.0, C_out=None, work=None,
         timers=False, stream=None):
    """C = alpha*op(A)*op(B) + beta*C through the C ABI.

    A, B (and C_out) are COLUMN-MAJOR matrices held as torch tensors of shape (cols, rows) -- i.e. the
    tensor is the transpose view of the BLAS matrix, contiguous, so that `ld` = tensor.shape[1].
    Returns (C, timers_ns or None, work)."""
    import numpy as np
    import torch
    assert A.is_cuda and B.is_cuda and A.is_contiguous() and B.is_contiguous()
    dt = A.dtype
    if B.dtype != dt or (C_out is not None and (C_out.dtype != dt or not C_out.is_contiguous())):
        raise TypeError("A, B and C_out must share one dtype and be contiguous")
    lda, ldb = A.shape[1], B.shape[1]
    m, k = (lda, A.shape[0]) if opA == "N" else (A.shape[0], lda)
    kb, n = (ldb, B.shape[0]) if opB == "N" else (B.shape[0], ldb)
    assert k == kb, (k, kb)
    if C_out is None:
        C_out = torch.zeros((n, m), dtype=dt, device=A.device)
    if work is None:
        tot, _, _ = work_size(dt.is_complex, backend, m, n, k, num_moduli)
        work = torch.empty(tot, dtype=torch.uint8, device=A.device)
    np_dt = {torch.float32: np.float32, torch.float64: np.float64, torch.complex64: np.complex64, torch.complex128: np.complex128}[dt]
    al = np.array([alpha], dtype=np_dt)
    be = np.array([beta], dtype=np_dt)
    tm = (C.c_double * 4)() if timers else None
    st = stream if stream is not None else torch.cuda.current_stream(A.device).cuda_stream
    rc = lib().gemmul8_gemm(st, _dtype_code(dt), backend, OPS[opA], OPS[opB], m, n, k, al.ctypes.data, A.data_ptr(), lda,
                            B.data_ptr(), ldb, be.ctypes.data, C_out.data_ptr(), C_out.shape[1], num_moduli, int(fastmode),
                            work.data_ptr(), None, None, 0, 0, 0, 0, tm)
    check(rc, "gemmul8_gemm")
    return C_out, (list(tm) if timers else None), work


def gemm_batched(A, B, num_moduli, fastmode=False, opA="N", opB="N", alpha=1.0, beta=0.0, C_out=None, work=None, stream=None, backend=INT8):
    """A strided batch as ONE set of launches (gemmul8_gemm_batched).  A, B (and C_out): contiguous tensors of shape
    (batch, cols, rows) -- each item a column-major matrix as in `gemm`; a batch dimension of 1 on A or B broadcasts (stride 0).
    Returns (C, work).  Bit-identical to calling `gemm` per item."""
    import numpy as np
    import torch
    assert A.is_cuda and B.is_cuda and A.is_contiguous() and B.is_contiguous() and A.dim() == 3 and B.dim() == 3
    dt = A.dtype
    batch = max(A.shape[0], B.shape[0])
    assert A.shape[0] in (1, batch) and B.shape[0] in (1, batch) and B.dtype == dt
    lda, ldb = A.shape[2], B.shape[2]
    m, k = (lda, A.shape[1]) if opA == "N" else (A.shape[1], lda)
    kb, n = (ldb, B.shape[1]) if opB == "N" else (B.shape[1], ldb)
    assert k == kb, (k, kb)
    if C_out is None:
        C_out = torch.zeros((batch, n, m), dtype=dt, device=A.device)
    assert C_out.is_contiguous() and C_out.dtype == dt and C_out.shape[0] == batch
    if work is None:
        work = torch.empty(lib().gemmul8_work_size_batched(int(dt.is_complex), backend, m, n, k, num_moduli, batch), dtype=torch.uint8, device=A.device)
    np_dt = {torch.float32: np.float32, torch.float64: np.float64, torch.complex64: np.complex64, torch.complex128: np.complex128}[dt]
    al, be = np.array([alpha], dtype=np_dt), np.array([beta], dtype=np_dt)
    st = stream if stream is not None else torch.cuda.current_stream(A.device).cuda_stream
    sa = 0 if A.shape[0] == 1 else A.shape[1] * A.shape[2]
    sb = 0 if B.shape[0] == 1 else B.shape[1] * B.shape[2]
    rc = lib().gemmul8_gemm_batched(st, _dtype_code(dt), backend, OPS[opA], OPS[opB], m, n, k, al.ctypes.data, A.data_ptr(), lda, sa, B.data_ptr(), ldb, sb,
                                    be.ctypes.data, C_out.data_ptr(), C_out.shape[2], C_out.shape[1] * C_out.shape[2], batch, num_moduli,
                                    int(fastmode), work.data_ptr())
    check(rc, "gemmul8_gemm_batched")
    return C_out, work
