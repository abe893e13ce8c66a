#!/usr/bin/env python3
"""FP8-backend / INT8-backend time ratio of the emulation at EQUAL moduli count over the hook's shape classes (tools/floor_scan.py's shapes, thinned):
the hook's automatic floor prices an FP8-backend call as the INT8 cost model x one factor (oz2_hook.cpp floor_model_declines).
usage: python tools/fp8_factor_scan.py [--dtype d|s|z|c] > profiles/sweeps/rNN_fp8_factor_<dtype>.csv"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gemmul8_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="d")
a = ap.parse_args()
tdt = {"d": torch.float64, "s": torch.float32, "z": torch.complex128, "c": torch.complex64}[a.dtype]
cplx = tdt.is_complex
Ns = {"d": [12], "s": [6], "z": [12], "c": [6]}[a.dtype]


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
for mn in (2048, 4096, 8192):
    for k in (64, 256, 512, 1024, 2048, 8192):
        shapes.append((mn, mn, k))
for m, n in ((16384, 256), (8192, 512), (4096, 1024)):
    for k in (256, 1024, 4096):
        shapes.append((m, n, k))
        shapes.append((n, m, k))
for mn in (512, 1024):
    for k in (1024, 4096, 16384):
        shapes.append((mn, mn, k))


def rnd(shape):
    x = torch.rand(shape, dtype=torch.float64 if tdt in (torch.float64, torch.complex128) else torch.float32, device="cuda") - 0.5
    return torch.complex(x, torch.rand_like(x) - 0.5) if cplx else x


print("dtype,m,n,k,N,fast,int8_ms,fp8_ms,fp8_over_int8")
for m, n, k in shapes:
    A = rnd((k, m))
    B = rnd((n, k))
    Cm = torch.zeros((n, m), dtype=tdt, device="cuda")
    reps = 9 if m * n * k > 1e10 else 25
    for N in Ns:
        for fast in (0, 1):
            t = {}
            for be in (g.INT8, g.FP8):
                tot, _, _ = g.work_size(cplx, be, m, n, k, N)
                work = torch.empty(tot, dtype=torch.uint8, device="cuda")
                t[be] = timed(lambda: g.gemm(A, B, N, fastmode=bool(fast), C_out=Cm, work=work, backend=be), reps)
                del work
            print(f"{a.dtype},{m},{n},{k},{N},{fast},{t[g.INT8]:.4f},{t[g.FP8]:.4f},{t[g.FP8] / t[g.INT8]:.3f}", flush=True)
    del A, B, Cm
