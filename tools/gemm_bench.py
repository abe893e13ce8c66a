#!/usr/bin/env python3
"""Micro-benchmark of the batched INT8 MFMA GEMM (gemmul8_lowprec_gemm) on random int8 planes.
Usage: python tools/gemm_bench.py [--size 8192] [--moduli 14] [--iters 10]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gemmul8_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--moduli", type=int, default=14)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--k", type=int, default=0, help="inner dimension (default: --size)")
ap.add_argument("--warmup", type=int, default=2)
a = ap.parse_args()
n, N = a.size, a.moduli
k = a.k or n
lib = g.lib()
tot, _, _ = g.work_size(False, g.INT8, n, n, k, N)
work = torch.empty(tot, dtype=torch.uint8, device="cuda")
L = g.Layout()
g.check(lib.gemmul8_get_layout(g.D, g.INT8, n, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(L)))
# random residues in [-127,127] in the A_lo / B_lo regions
offA = L.A_lo - work.data_ptr()
offB = L.B_lo - work.data_ptr()
work[offA:offA + N * L.sizeA] = torch.randint(0, 256, (N * L.sizeA,), dtype=torch.uint8, device="cuda")
work[offB:offB + N * L.sizeB] = torch.randint(0, 256, (N * L.sizeB,), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(a.warmup):
    g.check(lib.gemmul8_lowprec_gemm(st, g.D, g.INT8, n, n, k, N, 0, N, C.byref(L)))
torch.cuda.synchronize()
ts = []
for _ in range(a.iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.check(lib.gemmul8_lowprec_gemm(st, g.D, g.INT8, n, n, k, N, 0, N, C.byref(L)))
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
med = ts[len(ts) // 2]
ops = N * 2.0 * n * n * k
print(f"gemm_i8 MOD: size {n} k {k} moduli {N}: median {med:.3f} ms  min {ts[0]:.3f} ms  -> {ops / med * 1e-9:.0f} TOP/s median, {ops / ts[0] * 1e-9:.0f} best ({ops / med * 1e-9 / 5000 * 100:.1f}% of 5 POP/s)")
