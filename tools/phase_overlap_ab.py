#!/usr/bin/env python3
"""Phase overlap, MEASURED: does the CRT of column panel p on a second stream beside the residue GEMMs of panel p + 1 shorten the
call, with the GEMM's persistent grid reduced so that the CRT finds free CUs?  DGEMM m = n = size, INT8 backend, N moduli, the planes of a real
quantise pass.  Arms, interleaved (median of --rounds):
    serial          lowprec_gemm (all planes, all columns) then crt, one stream: the product's order
    gemm-only(G)    the same GEMM launch on G CUs: what giving CUs away costs the GEMM
    split(P)        P column panels, one stream: gemm(0) crt(0) gemm(1) crt(1) ...: what the split itself costs
    overlap(P, G)   gemm(p) on stream 1 with G CUs, crt(p) on stream 2 behind an event: crt(p) runs beside gemm(p + 1)
usage: python tools/phase_overlap_ab.py [--size 8192] [--k 8192,1024] [--moduli 14] [--rounds 7]"""
import argparse
import copy
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gemmul8_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--k", default="8192,1024")
ap.add_argument("--moduli", type=int, default=14)
ap.add_argument("--rounds", type=int, default=7)
a = ap.parse_args()
lib = g.lib()
n, N = a.size, a.moduli
one, zero = np.array([1.0]), np.array([0.0])
s1 = torch.cuda.current_stream()
s2 = torch.cuda.Stream()


def setcus(c):
    if c:
        os.environ["GEMMUL8_GEMM_CUS"] = str(c)
    else:
        os.environ.pop("GEMMUL8_GEMM_CUS", None)
    lib.gemmul8_reload_knobs()


for k in [int(x) for x in a.k.split(",")]:
    torch.manual_seed(k)
    A = torch.rand((k, n), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((n, k), dtype=torch.float64, device="cuda") - 0.5
    Cout = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    tot, _, _ = g.work_size(False, g.INT8, n, n, k, N)
    work = torch.empty(tot, dtype=torch.uint8, device="cuda")
    L = g.Layout()
    g.check(lib.gemmul8_get_layout(g.D, g.INT8, n, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(L)))
    g.check(lib.gemmul8_scale(s1.cuda_stream, g.D, g.INT8, 0, 0, n, n, k, A.data_ptr(), n, B.data_ptr(), k, N, 0, 0, N, C.byref(L), 0, 0))
    torch.cuda.synchronize()

    def panel(p, P):
        c0, c1 = n * p // P, n * (p + 1) // P
        Lp = g.Layout()
        C.memmove(C.byref(Lp), C.byref(L), C.sizeof(g.Layout))
        Lp.B_lo = L.B_lo + c0 * L.kp
        Lp.C_mid = L.C_mid + c0 * L.mp
        return c0, c1, Lp

    def gemm(st, c0, c1, Lp):
        g.check(lib.gemmul8_lowprec_gemm(st.cuda_stream, g.D, g.INT8, n, c1 - c0, k, N, 0, N, C.byref(Lp)))

    def crt(st, c0, c1, Lp):
        g.check(lib.gemmul8_crt(st.cuda_stream, g.D, g.INT8, N, n, c1 - c0, Lp.C_mid, L.mp, L.sizeC, L.sftA, L.sftB + 2 * c0, one.ctypes.data, zero.ctypes.data,
                                Cout.data_ptr() + 8 * c0 * n, n))

    def serial():
        gemm(s1, 0, n, L)
        crt(s1, 0, n, L)

    def gemm_only():
        gemm(s1, 0, n, L)

    def split(P):
        for p in range(P):
            c0, c1, Lp = panel(p, P)
            gemm(s1, c0, c1, Lp)
            crt(s1, c0, c1, Lp)

    def overlap(P):
        evs = []
        for p in range(P):
            c0, c1, Lp = panel(p, P)
            gemm(s1, c0, c1, Lp)
            ev = torch.cuda.Event()
            ev.record(s1)
            s2.wait_event(ev)
            crt(s2, c0, c1, Lp)
        ev = torch.cuda.Event()
        ev.record(s2)
        s1.wait_event(ev)

    arms = [("serial", 0, serial), ("gemm-only(256)", 0, gemm_only), ("gemm-only(240)", 240, gemm_only), ("gemm-only(224)", 224, gemm_only),
            ("split(2)", 0, lambda: split(2)), ("split(4)", 0, lambda: split(4)),
            ("overlap(2, 256)", 0, lambda: overlap(2)), ("overlap(2, 240)", 240, lambda: overlap(2)), ("overlap(2, 224)", 224, lambda: overlap(2)),
            ("overlap(4, 256)", 0, lambda: overlap(4)), ("overlap(4, 240)", 240, lambda: overlap(4)), ("overlap(4, 224)", 224, lambda: overlap(4))]
    # reference result of the serial arm
    setcus(0)
    serial()
    torch.cuda.synchronize()
    Cref = Cout.clone()
    ts = {name: [] for name, _, _ in arms}
    for r in range(a.rounds + 1):
        for name, cus, fn in arms:
            setcus(cus)
            Cout.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s1)
            fn()
            e1.record(s1)
            torch.cuda.synchronize()
            if r >= 1:
                ts[name].append(e0.elapsed_time(e1))
            if r == 1 and not name.startswith("gemm-only"):
                assert torch.equal(Cout, Cref), f"{name}: C differs from the serial arm"
    setcus(0)
    base = sorted(ts["serial"])[len(ts["serial"]) // 2]
    print(f"DGEMM {n} x {n} x {k}, {N} moduli, INT8: lowprec GEMMs + CRT (planes of a real quantise pass), median of {a.rounds}; every arm's C is bit-identical to the serial arm's")
    for name, _, _ in arms:
        t = sorted(ts[name])
        med = t[len(t) // 2]
        print(f"   {name:18s} {med:8.3f} ms   ({(med / base - 1) * 100:+6.2f} % vs serial)" if not name.startswith("gemm-only") else f"   {name:18s} {med:8.3f} ms")
    del work, A, B, Cout, Cref
