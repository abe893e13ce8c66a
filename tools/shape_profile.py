#!/usr/bin/env python3
"""One emulated DGEMM shape, both modes, a few calls each: meant to run under `rocprofv3 --kernel-trace --stats` to see which kernels a
small / skinny shape spends its time in.  usage: tools/shape_profile.py m n k [moduli=14] [reps=20]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemmul8_amd as g
m, n, k = (int(x) for x in sys.argv[1:4])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 14
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
A = torch.rand((k, m), dtype=torch.float64, device="cuda") - 0.5   # column-major m x k
B = torch.rand((n, k), dtype=torch.float64, device="cuda") - 0.5   # column-major k x n
Cm = torch.zeros((n, m), dtype=torch.float64, device="cuda")
tot, _, _ = g.work_size(False, g.INT8, m, n, k, N)
work = torch.empty(tot, dtype=torch.uint8, device="cuda")
for fast in (False, True):
    for _ in range(reps):
        g.gemm(A, B, N, fastmode=fast, C_out=Cm, work=work)
torch.cuda.synchronize()
