#!/usr/bin/env python3
"""A/B micro-benchmark of gemmul8_lowprec_gemm (INT8, random planes) across several builds of libgemmul8.so loaded in
ONE process and timed INTERLEAVED (A,B,C,A,B,C,...), so that box-to-box and warm-up/power-state drift cancel.
usage: python tools/gemm_ab.py [--k 256,8192] [--rounds 9] lib_a.so lib_b.so ..."""
import argparse
import ctypes as C
import os
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gemmul8_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--moduli", type=int, default=14)
ap.add_argument("--k", default="8192")
ap.add_argument("--rounds", type=int, default=9)
a = ap.parse_args()
n, N = a.size, a.moduli
ref = g.lib()  # binds the HIP runtime, gives layout/work_size
tmp = tempfile.mkdtemp()
libs = []
for i, pth in enumerate(a.libs):
    cp = os.path.join(tmp, f"v{i}.so")  # distinct inode/name so dlopen does not alias
    shutil.copy(pth, cp)
    L = C.CDLL(cp)
    L.gemmul8_lowprec_gemm.restype = C.c_int
    L.gemmul8_lowprec_gemm.argtypes = ref.gemmul8_lowprec_gemm.argtypes
    libs.append(L)
st = torch.cuda.current_stream().cuda_stream
for k in [int(x) for x in a.k.split(",")]:
    tot, _, _ = g.work_size(False, g.INT8, n, n, k, N)
    work = torch.empty(tot, dtype=torch.uint8, device="cuda")
    Lo = g.Layout()
    g.check(ref.gemmul8_get_layout(g.D, g.INT8, n, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(Lo)))
    offA, offB = Lo.A_lo - work.data_ptr(), Lo.B_lo - work.data_ptr()
    work[offA:offA + N * Lo.sizeA] = torch.randint(0, 256, (N * Lo.sizeA,), dtype=torch.uint8, device="cuda")
    work[offB:offB + N * Lo.sizeB] = torch.randint(0, 256, (N * Lo.sizeB,), dtype=torch.uint8, device="cuda")
    ts = [[] for _ in libs]
    for r in range(a.rounds + 2):
        for i, L in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                g.check(L.gemmul8_lowprec_gemm(st, g.D, g.INT8, n, n, k, N, 0, N, C.byref(Lo)))
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                ts[i].append(e0.elapsed_time(e1) / 3)
    for i, pth in enumerate(a.libs):
        t = sorted(ts[i])
        med = t[len(t) // 2]
        print(f"k={k:5d} {os.path.basename(pth):40s} median {med:7.3f} ms  min {t[0]:7.3f}  -> {N * 2.0 * n * n * k / med * 1e-9:6.0f} TOP/s")
    del work
