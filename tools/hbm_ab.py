#!/usr/bin/env python3
"""Interleaved A/B of the HBM-bound phases of one emulated DGEMM (8192^3, 14 moduli by default) across builds of libgemmul8.so loaded
in one process: scale_bounds (amax + extract + bound GEMM), scale_finish (shift_finalize + quantise A + quantise B) and crt, each with
its algorithmic bytes / time.  usage: tools/hbm_ab.py [--n 8192] [--moduli 14] [--phases bounds,finish,crt] a.so b.so ..."""
import argparse, ctypes as C, os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gemmul8_amd as g
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+"); ap.add_argument("--n", type=int, default=8192); ap.add_argument("--moduli", type=int, default=14)
ap.add_argument("--phases", default="bounds,finish,crt"); ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--dtype", default="d", choices=["d", "z", "s", "c"], help="element type (z / c: complex)")
a = ap.parse_args()
n, N = a.n, a.moduli
ref = g.lib(); tmp = tempfile.mkdtemp(); libs = []
for i, pth in enumerate(a.libs):
    cp = os.path.join(tmp, f"v{i}.so"); shutil.copy(pth, cp); L = C.CDLL(cp)
    for f in ("gemmul8_scale_bounds", "gemmul8_scale_finish", "gemmul8_crt"):
        getattr(L, f).restype = C.c_int; getattr(L, f).argtypes = getattr(ref, f).argtypes
    libs.append(L)
DT = {"d": (g.D, torch.float64, False, 8), "z": (g.Z, torch.complex128, True, 16), "s": (g.S, torch.float32, False, 4), "c": (g.Cx, torch.complex64, True, 8)}[a.dtype]
code, tdt, cplx, esz = DT
tot, _, _ = g.work_size(cplx, g.INT8, n, n, n, N)
work = torch.randint(0, 255, (tot,), dtype=torch.uint8, device="cuda")
Lo = g.Layout(); g.check(ref.gemmul8_get_layout(code, g.INT8, n, n, n, N, work.data_ptr(), None, None, 0, 0, C.byref(Lo)))
gen = torch.Generator(device="cuda").manual_seed(1)
rdt = torch.float32 if tdt in (torch.float32, torch.complex64) else torch.float64
def rnd():
    x = torch.rand((n, n), generator=gen, dtype=rdt, device="cuda") - 0.5
    return torch.complex(x, torch.rand((n, n), generator=gen, dtype=rdt, device="cuda") - 0.5).contiguous() if cplx else x
A, B = rnd(), rnd()
Cm = torch.zeros((n, n), dtype=tdt, device="cuda")
npd = {torch.float64: np.float64, torch.float32: np.float32, torch.complex128: np.complex128, torch.complex64: np.complex64}[tdt]
one, zero = np.array([1.0], dtype=npd), np.array([0.0], dtype=npd)
st = torch.cuda.current_stream().cuda_stream
g.check(ref.gemmul8_scale_bounds(st, code, g.INT8, 0, 0, n, n, n, A.data_ptr(), n, B.data_ptr(), n, N, 0, n, C.byref(Lo), 0, 0))
g.check(ref.gemmul8_scale_finish(st, code, g.INT8, 0, 0, n, n, n, A.data_ptr(), n, B.data_ptr(), n, N, 1, 0, N, C.byref(Lo), 0, 0))  # valid shifts for the CRT
calls = {
    "bounds": (lambda L: L.gemmul8_scale_bounds(st, code, g.INT8, 0, 0, n, n, n, A.data_ptr(), n, B.data_ptr(), n, N, 0, n, C.byref(Lo), 0, 0),
               3.0 * esz * n * n + 2.0 * (3 if cplx else 1) * n * n),
    # fast mode = 1 would add the norm kernels; the quantise kernels are the same in both modes: time them with the shifts the bounds left
    "finish": (lambda L: L.gemmul8_scale_finish(st, code, g.INT8, 0, 0, n, n, n, A.data_ptr(), n, B.data_ptr(), n, N, 0, 0, N, C.byref(Lo), 0, 0),
               2 * (esz + N * (3 if cplx else 1)) * n * n),
    "finishA": (lambda L: L.gemmul8_scale_finish(st, code, g.INT8, 0, 0, n, n, n, A.data_ptr(), n, B.data_ptr(), n, N, 0, 0, N, C.byref(Lo), 0, 1),
                (esz + N * (3 if cplx else 1)) * n * n),
    "finishB": (lambda L: L.gemmul8_scale_finish(st, code, g.INT8, 0, 0, n, n, n, A.data_ptr(), n, B.data_ptr(), n, N, 0, 0, N, C.byref(Lo), 1, 0),
                (esz + N * (3 if cplx else 1)) * n * n),
    "crt": (lambda L: L.gemmul8_crt(st, code, g.INT8, N, n, n, Lo.C_mid, Lo.mp, Lo.sizeC, Lo.sftA, Lo.sftB, one.ctypes.data, zero.ctypes.data, Cm.data_ptr(), n),
            (esz + N * (2 if cplx else 1)) * n * n),
}
for ph in a.phases.split(","):
    fn, nbytes = calls[ph]
    ts = [[] for _ in libs]
    for r in range(a.rounds + 2):
        for i, L in enumerate(libs):
            if ph.startswith("finish"):  # every timed call starts from the state the bounds phase leaves (shift_finalize negates in place)
                g.check(ref.gemmul8_scale_bounds(st, code, g.INT8, 0, 0, n, n, n, A.data_ptr(), n, B.data_ptr(), n, N, 0, n, C.byref(Lo), 0, 0))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.check(fn(L)); e1.record(); torch.cuda.synchronize()
            if r >= 2: ts[i].append(e0.elapsed_time(e1))
    for i, pth in enumerate(a.libs):
        t = sorted(ts[i]); med = t[len(t) // 2]
        print(f"{ph:7s} {os.path.basename(pth):24s} median {med*1e3:8.1f} us  min {t[0]*1e3:8.1f} us   {nbytes/med*1e-9:7.2f} TB/s algorithmic")
