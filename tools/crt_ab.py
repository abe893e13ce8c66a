#!/usr/bin/env python3
"""A/B timing of gemmul8_crt across builds of libgemmul8.so loaded in one process (interleaved).  usage: crt_ab.py a.so b.so"""
import ctypes as C, os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gemmul8_amd as g
n, N = 8192, 14
ref = g.lib()
tmp = tempfile.mkdtemp()
libs = []
for i, pth in enumerate(sys.argv[1:]):
    cp = os.path.join(tmp, f"v{i}.so"); shutil.copy(pth, cp)
    L = C.CDLL(cp); L.gemmul8_crt.restype = C.c_int; L.gemmul8_crt.argtypes = ref.gemmul8_crt.argtypes; libs.append(L)
tot, _, _ = g.work_size(False, g.INT8, n, n, n, N)
work = torch.randint(0, 256, (tot,), dtype=torch.uint8, device="cuda")
Lo = g.Layout()
g.check(ref.gemmul8_get_layout(g.D, g.INT8, n, n, n, N, work.data_ptr(), None, None, 0, 0, C.byref(Lo)))
Cm = torch.zeros((n, n), dtype=torch.float64, device="cuda")
one, zero = np.array([1.0]), np.array([0.0])
st = torch.cuda.current_stream().cuda_stream
ts = [[] for _ in libs]
for r in range(12):
    for i, L in enumerate(libs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.check(L.gemmul8_crt(st, g.D, g.INT8, N, n, n, Lo.C_mid, Lo.mp, Lo.sizeC, Lo.sftA, Lo.sftB, one.ctypes.data, zero.ctypes.data, Cm.data_ptr(), n))
        e1.record(); torch.cuda.synchronize()
        if r >= 2: ts[i].append(e0.elapsed_time(e1) / 3)
for i, pth in enumerate(sys.argv[1:]):
    t = sorted(ts[i]); print(f"{os.path.basename(pth):30s} median {t[len(t)//2]*1e3:8.1f} us  min {t[0]*1e3:8.1f} us")
