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
