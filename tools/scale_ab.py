#!/usr/bin/env python3
"""A/B timing of the scaling phase (gemmul8_scale: bounds + finish, DGEMM 8192^3 N=14) across builds of libgemmul8.so
loaded in one process (interleaved).  usage: scale_ab.py [fast] [fp8s[:size[:moduli]]] a.so b.so
(fp8s = SGEMM on the FP8 backend, default 16384^3, 6 moduli: the FP6 panel-image writer)"""
import ctypes as C, os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gemmul8_amd as g
args = sys.argv[1:]
fast = 0
if args and args[0] == "fast":
    fast, args = 1, args[1:]
n, N = 8192, 14
dtc, tdt, be = g.D, torch.float64, g.INT8
if args and args[0].startswith("fp8s"):
    f = args[0].split(":")
    n, N = (int(f[1]) if len(f) > 1 else 16384), (int(f[2]) if len(f) > 2 else 6)
    dtc, tdt, be, args = g.S, torch.float32, g.FP8, args[1:]
ref = g.lib()
tmp = tempfile.mkdtemp()
libs = []
for i, pth in enumerate(args):
    cp = os.path.join(tmp, f"v{i}.so"); shutil.copy(pth, cp)
    L = C.CDLL(cp); L.gemmul8_scale.restype = C.c_int; L.gemmul8_scale.argtypes = ref.gemmul8_scale.argtypes; libs.append(L)
tot, _, _ = g.work_size(False, be, n, n, n, N)
work = torch.empty(tot, dtype=torch.uint8, device="cuda")
Lo = g.Layout()
g.check(ref.gemmul8_get_layout(dtc, be, n, n, n, N, work.data_ptr(), None, None, 0, 0, C.byref(Lo)))
gen = torch.Generator(device="cuda").manual_seed(1)
A = torch.rand((n, n), generator=gen, dtype=tdt, device="cuda") - 0.5
B = torch.rand((n, n), generator=gen, dtype=tdt, device="cuda") - 0.5
st = torch.cuda.current_stream().cuda_stream
ts = [[] for _ in libs]
for r in range(12):
    for i, L in enumerate(libs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.check(L.gemmul8_scale(st, dtc, be, 0, 0, n, n, n, A.data_ptr(), n, B.data_ptr(), n, N, fast, 0, N, C.byref(Lo), 0, 0))
        e1.record(); torch.cuda.synchronize()
        if r >= 2: ts[i].append(e0.elapsed_time(e1) / 3)
for i, pth in enumerate(args):
    t = sorted(ts[i]); print(f"{os.path.basename(pth):30s} median {t[len(t)//2]*1e3:8.1f} us  min {t[0]*1e3:8.1f} us")
