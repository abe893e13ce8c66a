#!/usr/bin/env python3
"""A/B of the WHOLE emulated call (gemmul8_gemm: bounds, quantise, low-precision GEMMs, CRT) across several builds of libgemmul8.so
loaded in ONE process and timed INTERLEAVED (A,B,C,A,B,C,...) so that box-to-box and power-state drift cancel.
usage: python tools/call_ab.py [--size 8192] [--k 1024,8192] [--moduli 14] [--dtype d|s|z|c] [--fast] [--rounds 9] lib_a.so lib_b.so ..."""
import argparse
import ctypes as C
import os
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gemmul8_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--k", default="8192")
ap.add_argument("--moduli", type=int, default=14)
ap.add_argument("--dtype", default="d")
ap.add_argument("--fast", action="store_true")
ap.add_argument("--backend", default="int8", help="int8 | fp8")
ap.add_argument("--rounds", type=int, default=9)
a = ap.parse_args()
tdt = {"d": torch.float64, "s": torch.float32, "z": torch.complex128, "c": torch.complex64}[a.dtype]
ndt = {"d": np.float64, "s": np.float32, "z": np.complex128, "c": np.complex64}[a.dtype]
ref = g.lib()
tmp = tempfile.mkdtemp()
libs = []
for i, pth in enumerate(a.libs):
    cp = os.path.join(tmp, f"v{i}.so")
    shutil.copy(pth, cp)
    L = C.CDLL(cp)
    L.gemmul8_gemm.restype = C.c_int
    L.gemmul8_gemm.argtypes = ref.gemmul8_gemm.argtypes
    libs.append(L)
n, N = a.size, a.moduli
BACKEND = g.FP8 if a.backend.lower() == "fp8" else 0
st = torch.cuda.current_stream().cuda_stream
al, be = np.array([1.0], dtype=ndt), np.array([0.0], dtype=ndt)
for k in [int(x) for x in a.k.split(",")]:
    A = torch.randn((k, n), dtype=tdt, device="cuda")   # column-major m x k as a (k, m) tensor
    B = torch.randn((n, k), dtype=tdt, device="cuda")   # column-major k x n as a (n, k) tensor
    Cout = torch.zeros((n, n), dtype=tdt, device="cuda")
    tot, _, _ = g.work_size(tdt.is_complex, BACKEND, n, n, k, N)
    work = torch.empty(tot, dtype=torch.uint8, device="cuda")
    dcode = g._dtype_code(tdt)
    ts = [[] for _ in libs]
    for r in range(a.rounds + 2):
        for i, L in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.gemmul8_gemm(st, dcode, BACKEND, g.OPS["N"], g.OPS["N"], n, n, k, al.ctypes.data, A.data_ptr(), n, B.data_ptr(), k,
                                be.ctypes.data, Cout.data_ptr(), n, N, int(a.fast), work.data_ptr(), None, None, 0, 0, 0, 0, None)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0, rc
            if r >= 2:
                ts[i].append(e0.elapsed_time(e1))
    flops = (8 if tdt.is_complex else 2) * n * n * k
    for i, pth in enumerate(a.libs):
        t = sorted(ts[i])
        med = t[len(t) // 2]
        print(f"k={k:6d} {os.path.basename(pth):28s} whole call median {med:8.3f} ms  min {t[0]:8.3f}  -> {flops / med / 1e9:7.1f} TFLOPS", flush=True)
