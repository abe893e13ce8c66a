"""CPU tests of the boundary: libgemmul8.so loads, exports every symbol include/gemmul8_c.h declares,
and its host-only entry points (workSize, argument validation) behave like the reference."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import gemmul8_amd as g
import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol():
    """Every function include/gemmul8_c.h and include/gemmul8_dist.h declare is exported by libgemmul8.so."""
    L = g.lib()
    total = 0
    for header in ("gemmul8_c.h", "gemmul8_dist.h"):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        declared = set(re.findall(r"GEMMUL8_API[^;(]*?\b(gemmul8_[a-z_0-9]+)\s*\(", hdr))
        assert declared, f"no declarations parsed in {header}"
        for name in sorted(declared):
            assert hasattr(L, name), f"libgemmul8.so does not export {name} ({header})"
        if header == "gemmul8_c.h":
            assert set(g.EXPORTS) <= declared
        total += len(declared)
    assert total >= 24


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("backend", [g.INT8, g.FP8])
def test_work_size_matches_reference_formula(cplx, backend):
    """gemmul8::workSize (gemmul8_real.hpp:8-47, gemmul8_complex.hpp:8-47) restated in the oracle."""
    rng = np.random.default_rng(0)
    for _ in range(50):
        m, n, k = (int(x) for x in rng.integers(1, 5000, 3))
        N = int(rng.integers(2, 21))
        enA, enB = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        assert g.work_size(cplx, backend, m, n, k, N, enA, enB) == ol.work_size(cplx, backend, m, n, k, N, enA, enB)
    GiB = 2.0 ** 30
    assert abs(g.work_size(False, g.INT8, 8192, 8192, 8192, 14)[0] / GiB - 2.875) < 1e-3


def test_argument_validation_without_gpu():
    L = g.lib()
    a = np.zeros(4)
    tm = (C.c_double * 4)()
    # num_moduli out of range / null pointers are rejected before any HIP call
    rc = L.gemmul8_gemm(None, g.D, g.INT8, 0, 0, 2, 2, 2, a.ctypes.data, a.ctypes.data, 2, a.ctypes.data, 2, a.ctypes.data,
                        a.ctypes.data, 2, 1, 0, a.ctypes.data, None, None, 0, 0, 0, 0, tm)
    assert rc == -1
    rc = L.gemmul8_gemm(None, g.D, g.INT8, 0, 0, 2, 2, 2, a.ctypes.data, None, 2, a.ctypes.data, 2, a.ctypes.data,
                        a.ctypes.data, 2, 14, 0, a.ctypes.data, None, None, 0, 0, 0, 0, tm)
    assert rc == -2
    rc = L.gemmul8_gemm(None, g.D, g.INT8, 0, 0, 2, 2, (1 << 17) + 1, a.ctypes.data, a.ctypes.data, 2, a.ctypes.data, 2,
                        a.ctypes.data, a.ctypes.data, 2, 14, 0, a.ctypes.data, None, None, 0, 0, 0, 0, tm)
    assert rc == -2


def test_float_types_reject_more_than_13_moduli():
    """Contract of the reference (GEMMul8/include/gemmul8.hpp:30): 2..20 moduli for double / complex-double, 2..13 for float /
    complex-float.  The reference's float pipeline accepts more and overflows (P reaches 2^128 at 16 moduli); the restated oracle
    shows it (inf), so the C ABI refuses instead of returning inf: GEMMUL8_E_NUM_MODULI before any HIP call, every entry point."""
    import oracle_lib as ol
    L = g.lib()
    a = np.zeros(64)
    p = a.ctypes.data
    lay = g.Layout()
    for dt in (g.S, g.Cx):
        for be in (g.INT8, g.FP8):
            for N in (14, 16, 20):
                assert L.gemmul8_gemm(None, dt, be, 0, 0, 2, 2, 2, p, p, 2, p, 2, p, p, 2, N, 0, p, None, None, 0, 0, 0, 0, None) == -1
                assert L.gemmul8_gemm_batched(None, dt, be, 0, 0, 2, 2, 2, p, p, 2, 4, p, 2, 4, p, p, 2, 4, 2, N, 0, p) == -1
                assert L.gemmul8_get_layout(dt, be, 2, 2, 2, N, p, None, None, 0, 0, C.byref(lay)) == -1
                assert L.gemmul8_crt(None, dt, be, N, 2, 2, p, 2, 4, p, p, p, p, p, 2) == -1
    for dt in (g.D, g.Z):
        assert L.gemmul8_get_layout(dt, g.INT8, 2, 2, 2, 20, p, None, None, 0, 0, C.byref(lay)) == 0
    # why: the float pipeline of the reference, restated, overflows beyond 13-15 moduli
    rng = np.random.default_rng(0)
    A, B = (rng.random((5, 24)) - 0.5).astype(np.float32), (rng.random((24, 4)) - 0.5).astype(np.float32)
    assert np.isfinite(ol.gemm(A, B, 13)).all()
    assert not np.isfinite(ol.gemm(A, B, 20)).all()
