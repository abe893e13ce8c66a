#!/usr/bin/env python3
"""SURVEY section 8 f3 by CACHE RESIDENCY: column panels small enough that a panel's C_mid stays in the 256 MiB Infinity
Cache between its residue GEMMs and its CRT.  DGEMM m = n = size, INT8 backend, N moduli, the planes of a real quantise pass.  Arms, interleaved,
median of --rounds, one stream, every arm's C bit-identical to the serial arm's:
    serial       lowprec_gemm (all planes, all columns) then crt: the product's order
    gemm / crt   the two phases alone (all columns)
    split(P)     P column panels: gemm(0) crt(0) gemm(1) crt(1) ...; every panel has its own columns of C_mid (the reference's workspace layout)
    ring(P)      the same, but every panel writes its residues into the columns of panel 0: the strips are overwritten while (hopefully) still
                 dirty in the memory-side cache, so that C_mid never has to reach HBM
each with the residue-store policy of the GEMM epilogue forced non-temporal (nt=1), cache-allocating (nt=0) or left to the library (nt=-).
usage: python tools/panel_crt_ab.py [--size 8192] [--k 128,256,512,1024] [--moduli 14] [--panels 4,8,16,32] [--rounds 7]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gemmul8_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--k", default="128,256,512,1024")
ap.add_argument("--moduli", type=int, default=14)
ap.add_argument("--panels", default="4,8,16,32")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--nt", default="-,0,1")
a = ap.parse_args()
lib = g.lib()
n, N = a.size, a.moduli
one, zero = np.array([1.0]), np.array([0.0])
s1 = torch.cuda.current_stream()


def setnt(v):
    if v == "-":
        os.environ.pop("GEMMUL8_EPI_NT", None)
    else:
        os.environ["GEMMUL8_EPI_NT"] = v
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

    def panel(p, P, ring):
        c0, c1 = n * p // P, n * (p + 1) // P
        Lp = g.Layout()
        C.memmove(C.byref(Lp), C.byref(L), C.sizeof(g.Layout))
        Lp.B_lo = L.B_lo + c0 * L.kp
        Lp.C_mid = L.C_mid + (0 if ring else c0 * L.mp)
        return c0, c1, Lp

    def gemm(c0, c1, Lp):
        g.check(lib.gemmul8_lowprec_gemm(s1.cuda_stream, g.D, g.INT8, n, c1 - c0, k, N, 0, N, C.byref(Lp)))

    def crt(c0, c1, Lp):
        g.check(lib.gemmul8_crt(s1.cuda_stream, g.D, g.INT8, N, n, c1 - c0, Lp.C_mid, L.mp, L.sizeC, L.sftA, L.sftB + 2 * c0, one.ctypes.data, zero.ctypes.data,
                                Cout.data_ptr() + 8 * c0 * n, n))

    def serial():
        gemm(0, n, L)
        crt(0, n, L)

    def split(P, ring):
        for p in range(P):
            c0, c1, Lp = panel(p, P, ring)
            gemm(c0, c1, Lp)
            crt(c0, c1, Lp)

    arms = []
    for nt in a.nt.split(","):
        arms.append((f"serial nt={nt}", nt, serial, True))
        arms.append((f"gemm nt={nt}", nt, lambda: gemm(0, n, L), False))
        for P in [int(x) for x in a.panels.split(",")]:
            arms.append((f"split({P}) nt={nt}", nt, (lambda P=P: split(P, False)), True))
            arms.append((f"ring({P}) nt={nt}", nt, (lambda P=P: split(P, True)), True))
    arms.append(("crt", "-", lambda: crt(0, n, L), False))
    setnt("-")
    serial()
    torch.cuda.synchronize()
    Cref = Cout.clone()
    ts = {name: [] for name, _, _, _ in arms}
    for r in range(a.rounds + 1):
        for name, nt, fn, check in arms:
            setnt(nt)
            if check:
                Cout.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s1)
            fn()
            e1.record(s1)
            torch.cuda.synchronize()
            if r >= 1:
                ts[name].append(e0.elapsed_time(e1))
            if r == 1 and check:
                assert torch.equal(Cout, Cref), f"{name}: C differs from the serial arm"
    setnt("-")
    # leave the serial order in C_mid for whoever reads the workspace next
    base = sorted(ts["serial nt=-"])[len(ts["serial nt=-"]) // 2]
    print(f"DGEMM {n} x {n} x {k}, {N} moduli, INT8: lowprec GEMMs + CRT (planes of a real quantise pass), median of {a.rounds}; C_mid = {N * L.sizeC / 2**20:.0f} MiB; "
          f"every arm's C is bit-identical to the serial arm's")
    for name, _, _, check in arms:
        t = sorted(ts[name])
        med = t[len(t) // 2]
        print(f"   {name:22s} {med:8.3f} ms   ({(med / base - 1) * 100:+6.2f} % vs serial nt=-)" if check else f"   {name:22s} {med:8.3f} ms")
    sys.stdout.flush()
    del work, A, B, Cout, Cref
