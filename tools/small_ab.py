#!/usr/bin/env python3
"""A/B wall time of whole gemmul8_gemm calls (DGEMM, 14 moduli, accurate mode) at launch-bound sizes across builds of libgemmul8.so
loaded in one process and timed interleaved.  usage: python tools/small_ab.py a.so b.so ..."""
import ctypes as C
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gemmul8_amd as g

ref = g.lib()
libs = []
tmp = tempfile.mkdtemp()
for i, pth in enumerate(sys.argv[1:]):
    cp = os.path.join(tmp, f"v{i}.so")
    shutil.copy(pth, cp)
    L = C.CDLL(cp)
    L.gemmul8_gemm.restype = C.c_int
    L.gemmul8_gemm.argtypes = ref.gemmul8_gemm.argtypes
    libs.append(L)
st = torch.cuda.current_stream().cuda_stream
one, zero = np.array([1.0]), np.array([0.0])
for n in (512, 1024, 2048, 4096):
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    Cm = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    tot, _, _ = g.work_size(False, g.INT8, n, n, n, 14)
    work = torch.zeros(tot, dtype=torch.uint8, device="cuda")

    def call(L):
        rc = L.gemmul8_gemm(st, g.D, g.INT8, 0, 0, n, n, n, one.ctypes.data, A.data_ptr(), n, B.data_ptr(), n, zero.ctypes.data, Cm.data_ptr(), n,
                            14, 0, work.data_ptr(), None, None, 0, 0, 0, 0, None)
        assert rc == 0
    reps = 200 if n <= 1024 else 50
    res = [[] for _ in libs]
    for rnd in range(7):
        for i, L in enumerate(libs):
            for _ in range(10):
                call(L)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                call(L)
            torch.cuda.synchronize()
            res[i].append((time.perf_counter() - t) / reps * 1e6)
    print(n, "  ".join(f"{os.path.basename(p)} {sorted(r)[len(r) // 2]:.1f} us" for p, r in zip(sys.argv[1:], res)))
