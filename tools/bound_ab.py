#!/usr/bin/env python3
"""Interleaved A/B of the accurate-mode bound phase (gemmul8_scale_bounds: amax + extract + bound GEMM with the maxima epilogue) across
library builds, DGEMM n^3.  usage: tools/bound_ab.py [--n 1024,2048,4096] lib_a.so lib_b.so ..."""
import argparse, ctypes as C, os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gemmul8_amd as g
ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--n", default="1024,2048,4096,8192")
a = ap.parse_args()
ref = g.lib(); tmp = tempfile.mkdtemp(); libs = []
for i, pth in enumerate(a.libs):
    cp = os.path.join(tmp, f"v{i}.so"); shutil.copy(pth, cp); L = C.CDLL(cp)
    L.gemmul8_scale_bounds.restype = C.c_int; L.gemmul8_scale_bounds.argtypes = ref.gemmul8_scale_bounds.argtypes; libs.append(L)
st = torch.cuda.current_stream().cuda_stream
for n in [int(x) for x in a.n.split(",")]:
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5; B = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    tot, _, _ = g.work_size(False, g.INT8, n, n, n, 14); work = torch.empty(tot, dtype=torch.uint8, device="cuda")
    Lo = g.Layout(); g.check(ref.gemmul8_get_layout(g.D, g.INT8, n, n, n, 14, work.data_ptr(), None, None, 0, 0, C.byref(Lo)))
    ts = [[] for _ in libs]
    for r in range(9):
        for i, L in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.check(L.gemmul8_scale_bounds(st, g.D, g.INT8, 0, 0, n, n, n, A.data_ptr(), n, B.data_ptr(), n, 14, 0, n, C.byref(Lo), 0, 0))
            e1.record(); torch.cuda.synchronize()
            if r >= 2: ts[i].append(e0.elapsed_time(e1) / 5)
    for i, pth in enumerate(a.libs):
        t = sorted(ts[i]); print(f"n={n:5d} {os.path.basename(pth):22s} bounds phase median {t[len(t)//2]*1e3:8.1f} us  min {t[0]*1e3:8.1f}")
