"""TEST INFRASTRUCTURE: a gemmul8_dist_engine (include/gemmul8_dist.h) whose compute and memory functions run on the HOST with
the CPU oracle, in the DEVICE's workspace layout (padded planes, scratch carving of gemmul8_get_layout).

It lets the CPU test-suite drive the C++ multi-GPU plans of gemmul8_amd/csrc/oz2_dist.cpp -- partition arithmetic, the
bound all-reduce, the residue exchange, the FP64 partial-sum reduce-scatter, the allgather of C -- at world sizes 2..8 over
gloo, through the very C ABI the product uses.  The product never sees this file: its engine is the HIP one (NULL table).
"""
import ctypes as C

import numpy as np

import gemmul8_amd as g
import oracle_lib as ol
from gemmul8_amd import dist as gd

NP_DT = {0: np.float32, 1: np.float64, 2: np.complex64, 3: np.complex128}
OPS = "NTC"


def _view(ptr, nbytes):
    return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))


def _stored(ptr, ld, dtype, rows, cols):
    """Column-major stored matrix (rows x cols, leading dimension ld) -> Fortran-ordered numpy copy."""
    dt = np.dtype(NP_DT[dtype])
    raw = _view(ptr, ((cols - 1) * ld + rows) * dt.itemsize).view(dt)
    out = np.empty((rows, cols), dt, order="F")
    for c in range(cols):
        out[:, c] = raw[c * ld:c * ld + rows]
    return out


class OracleEngine:
    def __init__(self):
        self.bufs = {}
        self.calls = []
        t = dict(gd.ENGINE_FIELDS)
        self._cb = [t[name](getattr(self, "_" + name)) for name, _ in gd.ENGINE_FIELDS]
        self.table = gd.Engine(*self._cb)

    # ---- memory
    def _alloc(self, nbytes):
        buf = np.full(max(int(nbytes), 1) + 256, 0xC3, np.uint8)   # poisoned, like device memory
        addr = (buf.ctypes.data + 255) // 256 * 256
        self.bufs[addr] = buf
        return addr

    def _release(self, p):
        self.bufs.pop(p, None)

    def _zero(self, p, nbytes, stream):
        _view(p, nbytes)[:] = 0
        return 0

    def _copy(self, dst, src, nbytes, stream):
        C.memmove(dst, src, nbytes)
        return 0

    def _copy2d(self, dst, dpitch, src, spitch, width, height, stream):
        for r in range(height):
            C.memmove(dst + r * dpitch, src + r * spitch, width)
        return 0

    # ---- compute (oracle) on the device layout
    def _operands(self, dtype, opA, opB, m, n, k, A, lda, B, ldb):
        Am = _stored(A, lda, dtype, *((m, k) if opA == 0 else (k, m)))
        Bm = _stored(B, ldb, dtype, *((k, n) if opB == 0 else (n, k)))
        return Am, Bm

    def _scale_bounds(self, stream, dtype, backend, opA, opB, m, n, k, A, lda, B, ldb, N, c0, c1, Lp, skipA, skipB):
        try:
            L = Lp.contents
            self.calls.append(("bounds", m, n, c0, c1))
            Am, Bm = self._operands(dtype, opA, opB, m, n, k, A, lda, B, ldb)
            oA, s0A = ol.extract_bounds(Am, OPS[opA], True, backend)
            oB, s0B = ol.extract_bounds(Bm, OPS[opB], False, backend)
            for p in range(oA.shape[0]):
                pa = _view(L.A_bound + p * L.sizeA, L.mp * L.kp).reshape(L.mp, L.kp)
                pa[:m, :k], pa[:m, k:] = oA[p], 0
                pb = _view(L.B_bound + p * L.sizeB, n * L.kp).reshape(n, L.kp)
                pb[:, :k], pb[:, k:] = oB[p], 0
            _view(L.sftA, 2 * m).view(np.int16)[:] = s0A
            _view(L.sftB, 2 * n).view(np.int16)[:] = s0B
            np_ = (n + 255) // 256 * 256
            mx = _view(L.scratch, 4 * (L.mp + np_)).view(np.int32)
            mx[:] = 0
            rmax, cmax = ol.bound_maxima(oA, oB, backend, c0, c1)
            mx[:m] = rmax.view(np.int32)
            mx[L.mp:L.mp + n] = cmax.view(np.int32)
            return 0
        except Exception as e:
            print("OracleEngine.scale_bounds:", repr(e))
            return 1

    def _scale_finish(self, stream, dtype, backend, opA, opB, m, n, k, A, lda, B, ldb, N, fast, t0, t1, Lp, skipA, skipB):
        try:
            L = Lp.contents
            self.calls.append(("finish", m, n, t0, t1))
            Am, Bm = self._operands(dtype, opA, opB, m, n, k, A, lda, B, ldb)
            lib = ol.lib()
            sA = _view(L.sftA, 2 * m).view(np.int16)
            sB = _view(L.sftB, 2 * n).view(np.int16)
            kmA, kmB = int(opA != 0), int(opB == 0)
            if fast:
                a, b = np.zeros(m, np.int16), np.zeros(n, np.int16)
                lib.oz2_fast_shifts(dtype, backend, N, kmA, m, k, ol._p(Am), Am.shape[0], ol._p(a))
                lib.oz2_fast_shifts(dtype, backend, N, kmB, n, k, ol._p(Bm), Bm.shape[0], ol._p(b))
                sA[:], sB[:] = a, b
            else:
                if backend != ol.INT8:
                    raise NotImplementedError("the CPU test engine finalises INT8 maxima only")
                np_ = (n + 255) // 256 * 256
                mx = _view(L.scratch, 4 * (L.mp + np_)).view(np.int32)
                a, b = sA.copy(), sB.copy()
                lib.oz2_shift_finalize_i8(backend, N, m, ol._p(np.ascontiguousarray(mx[:m])), ol._p(a))
                lib.oz2_shift_finalize_i8(backend, N, n, ol._p(np.ascontiguousarray(mx[L.mp:L.mp + n])), ol._p(b))
                sA[:], sB[:] = a, b
            parts = 3 if dtype >= 2 else 1
            nm = ol.num_mat(backend, N)
            q0 = ol.num_mat(backend, t0) if t0 > 0 else 0
            q1 = ol.num_mat(backend, t1) if t1 > 0 else 0
            for X, op, is_A, rows, km, sft, lo_ptr, size, pstride in ((Am, opA, True, m, kmA, sA, L.A_lo, L.sizeA, L.part_strideA),
                                                                      (Bm, opB, False, n, kmB, sB, L.B_lo, L.sizeB, L.part_strideB)):
                lo = np.zeros((parts, nm, rows, k), np.uint8)
                lib.oz2_quantise(dtype, backend, N, km, int(op == 2), rows, k, ol._p(X), X.shape[0], ol._p(np.ascontiguousarray(sft)), ol._p(lo))
                rows_alloc = L.mp if is_A else n
                for p in range(parts):
                    for q in range(nm):
                        pl = _view(lo_ptr + p * pstride + q * size, rows_alloc * L.kp).reshape(rows_alloc, L.kp)
                        if q0 <= q < q1:
                            pl[:rows, :k], pl[:rows, k:] = lo[p, q], 0
                        else:
                            pl[:] = 0x55   # planes of other ranks' moduli: poisoned, must never be used
            return 0
        except Exception as e:
            print("OracleEngine.scale_finish:", repr(e))
            return 1

    def _lowprec_gemm(self, stream, dtype, backend, m, n, k, N, t0, t1, Lp):
        try:
            L = Lp.contents
            self.calls.append(("lowprec", m, n, t0, t1))
            cplx = dtype >= 2
            parts = 3 if cplx else 1
            nm = ol.num_mat(backend, N)
            Alo = np.zeros((parts, nm, m, k), np.uint8)
            Blo = np.zeros((parts, nm, n, k), np.uint8)
            for p in range(parts):
                for q in range(nm):
                    Alo[p, q] = _view(L.A_lo + p * L.part_strideA + q * L.sizeA, L.mp * L.kp).reshape(L.mp, L.kp)[:m, :k]
                    Blo[p, q] = _view(L.B_lo + p * L.part_strideB + q * L.sizeB, n * L.kp).reshape(n, L.kp)[:, :k]
            mid_dt = np.int8 if backend == ol.INT8 else np.int16
            comps = 2 if cplx else 1
            Cm = np.zeros((N, n, m, comps), mid_dt)
            ol.lib().oz2_gemm_mod(backend, int(cplx), N, m, n, k, ol._p(Alo), ol._p(Blo), ol._p(Cm), t0, t1)
            isz = np.dtype(mid_dt).itemsize * comps
            for t in range(N):
                pc = _view(L.C_mid + t * L.sizeC * isz, L.sizeC * isz).view(mid_dt).reshape(n, L.mp, comps)
                if t0 <= t < t1:
                    pc[:, :m, :] = Cm[t]
                else:
                    pc[:] = 77   # poison
            return 0
        except Exception as e:
            print("OracleEngine.lowprec_gemm:", repr(e))
            return 1

    @staticmethod
    def _gather_mid(backend, cplx, N, m, n, C_mid, ld_mid, plane_stride, t0, t1):
        mid_dt = np.int8 if backend == ol.INT8 else np.int16
        comps = 2 if cplx else 1
        isz = np.dtype(mid_dt).itemsize * comps
        Cm = np.zeros((N, n, m, comps), mid_dt)
        for t in range(t0, t1):
            pc = _view(C_mid + (t - t0) * plane_stride * isz, ((n - 1) * ld_mid + m) * isz).view(mid_dt)
            for j in range(n):
                Cm[t, j] = pc[j * ld_mid * comps:(j * ld_mid + m) * comps].reshape(m, comps)
        return Cm

    def _crt(self, stream, dtype, backend, N, m, n, C_mid, ld_mid, plane_stride, sftA, sftB, alpha, beta, Cp, ldc):
        try:
            self.calls.append(("crt", m, n))
            Cm = self._gather_mid(backend, dtype >= 2, N, m, n, C_mid, ld_mid, plane_stride, 0, N)
            ol.lib().oz2_invscal(dtype, backend, N, m, n, ol._p(Cm), C.c_void_p(sftA), C.c_void_p(sftB), C.c_void_p(alpha), C.c_void_p(beta),
                                 C.c_void_p(Cp), ldc, 0)
            return 0
        except Exception as e:
            print("OracleEngine.crt:", repr(e))
            return 1

    def _crt_partial(self, stream, dtype, backend, N, t0, t1, m, n, C_mid, ld_mid, plane_stride, out_hi, out_lo, ld_out, col_block, block_stride):
        try:
            self.calls.append(("crt_partial", m, n, t0, t1))
            cplx = dtype >= 2
            comps = 2 if cplx else 1
            Cm = self._gather_mid(backend, cplx, N, m, n, C_mid, ld_mid, plane_stride, t0, t1)
            hi = np.zeros((n, m, comps))
            lo = np.zeros((n, m, comps))
            ol.lib().oz2_crt_partial(dtype, backend, N, t0, t1, m, n, ol._p(Cm), ol._p(hi), ol._p(lo))
            for j in range(n):
                b, cin = divmod(j, col_block)
                for src, dst in ((hi, out_hi), (lo, out_lo)):
                    d = _view(dst + 8 * (b * block_stride + cin * ld_out * comps), 8 * m * comps).view(np.float64)
                    d[:] = src[j].reshape(-1)
            return 0
        except Exception as e:
            print("OracleEngine.crt_partial:", repr(e))
            return 1

    def _add_f64(self, stream, dst, src, count):
        self.calls.append(("add_f64", count))
        d = _view(dst, 8 * count).view(np.float64)
        d += _view(src, 8 * count).view(np.float64)
        return 0

    def _crt_finish(self, stream, dtype, backend, N, m, n, in_hi, in_lo, ld_in, sftA, sftB, alpha, beta, Cp, ldc):
        try:
            self.calls.append(("crt_finish", m, n))
            comps = 2 if dtype >= 2 else 1
            hi = np.zeros((n, m, comps))
            lo = np.zeros((n, m, comps))
            for j in range(n):
                hi[j] = _view(in_hi + 8 * j * ld_in * comps, 8 * m * comps).view(np.float64).reshape(m, comps)
                lo[j] = _view(in_lo + 8 * j * ld_in * comps, 8 * m * comps).view(np.float64).reshape(m, comps)
            ol.lib().oz2_crt_finish(dtype, backend, N, m, n, ol._p(hi), ol._p(lo), C.c_void_p(sftA), C.c_void_p(sftB), C.c_void_p(alpha),
                                    C.c_void_p(beta), C.c_void_p(Cp), ldc, 0)
            return 0
        except Exception as e:
            print("OracleEngine.crt_finish:", repr(e))
            return 1
