#!/usr/bin/env python3
"""Emulated vs native GEMM over the shape classes a hooked solver issues (trailing updates: large m = n, small k; panel products:
one small dimension), to calibrate the automatic hook floor (oz2_hook.cpp below_floor).
The CSV feeds tools/fit_floor.py, which fits the cost model the hook evaluates.
usage: python tools/floor_scan.py [--dtype d|s|z|c] [--moduli 10,14,18] > profiles/sweeps/rNN_floor_scan_<dtype>.csv"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gemmul8_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="d")
ap.add_argument("--moduli", default="")
a = ap.parse_args()
tdt = {"d": torch.float64, "s": torch.float32, "z": torch.complex128, "c": torch.complex64}[a.dtype]
cplx = tdt.is_complex
Ns = [int(x) for x in a.moduli.split(",")] if a.moduli else {"d": [10, 14, 18], "s": [5, 7, 9], "z": [10, 14, 18], "c": [5, 7, 9]}[a.dtype]


def timed(fn, reps):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


shapes = []
big = (2048, 4096, 8192) if cplx else (2048, 4096, 8192, 16384)
for mn in big:
    for k in (64, 128, 256, 384, 512, 768, 1024, 2048):
        shapes.append((mn, mn, k))
for m, n in ((16384, 256), (16384, 1024), (8192, 128), (8192, 512), (4096, 1024)):   # tall-skinny outputs and their transposes
    for k in (256, 1024, 4096):
        shapes.append((m, n, k))
        shapes.append((n, m, k))
for mn in (512, 1024):                                                               # small squares, long k
    for k in (1024, 4096, 16384):
        shapes.append((mn, mn, k))


def rnd(shape):
    x = torch.rand(shape, dtype=torch.float64 if tdt in (torch.float64, torch.complex128) else torch.float32, device="cuda") - 0.5
    return torch.complex(x, torch.rand_like(x) - 0.5) if cplx else x


print("dtype,m,n,k,N,fast,native_ms,emulated_ms,native_TFLOPS,emulated_TFLOPS,ratio")
for m, n, k in shapes:
    A = rnd((k, m))   # column-major m x k
    B = rnd((n, k))   # column-major k x n
    Cm = torch.zeros((n, m), dtype=tdt, device="cuda")
    reps = 9 if m * n * k > 1e10 else 25
    tn = timed(lambda: torch.mm(B, A, out=Cm), reps)
    fl = (8.0 if cplx else 2.0) * m * n * k
    for N in Ns:
        tot, _, _ = g.work_size(cplx, g.INT8, m, n, k, N)
        work = torch.empty(tot, dtype=torch.uint8, device="cuda")
        for fast in (0, 1):
            te = timed(lambda: g.gemm(A, B, N, fastmode=bool(fast), C_out=Cm, work=work), reps)
            print(f"{a.dtype},{m},{n},{k},{N},{fast},{tn:.4f},{te:.4f},{fl / tn / 1e9:.1f},{fl / te / 1e9:.1f},{tn / te:.3f}", flush=True)
        del work
    del A, B, Cm
