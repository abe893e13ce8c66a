#!/usr/bin/env python3
"""A/B micro-benchmark of the FP8 backend's residue GEMMs (gemmul8_lowprec_gemm, backend FP8) across several builds of libgemmul8.so
loaded in ONE process and timed INTERLEAVED (A,B,C,A,B,C,...).  The operand planes are the real ones: the first build quantises random
U(-0.5, 0.5) operands once per plane format (GEMMUL8_FP8_PLANES), every build multiplies the same planes.
usage: python tools/f8_ab.py [--size 8192] [--k 8192] [--moduli 6] [--rounds 7] [--check] [--smi] lib_a.so[:e4m3] lib_b.so ...
       a ':e4m3' suffix runs that build on e4m3 byte planes (the round-4 kernel) instead of FP6 panel images, ':nofuse' on FP6 planes with the
       three products of a modulus as separate launches (GEMMUL8_FP8_FUSED=0)"""
import argparse
import ctypes as C
import os
import re
import shutil
import subprocess
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gemmul8_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--k", default="8192")
ap.add_argument("--moduli", type=int, default=6)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--dtype", default="S", choices=["S", "D"])
ap.add_argument("--check", action="store_true", help="compare the C_mid planes every build writes with the first build's (bit for bit)")
ap.add_argument("--smi", action="store_true", help="sample sclk / socket power (rocm-smi) while each build loops for ~2.5 s")
a = ap.parse_args()
n, N = a.size, a.moduli
ref = g.lib()
dt = g.S if a.dtype == "S" else g.D
tdt = torch.float32 if a.dtype == "S" else torch.float64
tmp = tempfile.mkdtemp()
libs, fmts, nofuse, colblk = [], [], [], []
for i, spec in enumerate(a.libs):
    pth, _, fmt = spec.partition(":")
    cp = os.path.join(tmp, f"v{i}.so")
    shutil.copy(pth, cp)
    L = C.CDLL(cp)
    for fn in ("gemmul8_lowprec_gemm", "gemmul8_scale", "gemmul8_get_layout"):
        getattr(L, fn).restype = C.c_int
        getattr(L, fn).argtypes = getattr(ref, fn).argtypes
    L.gemmul8_reload_knobs.restype = None
    libs.append(L)
    fmts.append("e4m3" if fmt == "e4m3" else "fp6")
    nofuse.append(fmt == "nofuse")
    colblk.append(fmt[2:] if fmt.startswith("cb") else None)   # ':cb<w>' = GEMMUL8_MAP_COLBLOCK=<w> (tile-columns per column block; 0 = full width)
st = torch.cuda.current_stream().cuda_stream
for k in [int(x) for x in a.k.split(",")]:
    torch.manual_seed(k)
    A = (torch.rand((k, n), dtype=tdt, device="cuda") - 0.5)   # column-major m x k  == row-major (k, m)
    B = (torch.rand((n, k), dtype=tdt, device="cuda") - 0.5)   # column-major k x n
    tot, _, _ = g.work_size(False, g.FP8, n, n, k, N)
    works, Ls = {}, {}
    for fmt in sorted(set(fmts)):
        os.environ["GEMMUL8_FP8_PLANES"] = fmt
        L0 = libs[fmts.index(fmt)]
        L0.gemmul8_reload_knobs()
        w = torch.empty(tot, dtype=torch.uint8, device="cuda")
        Lo = g.Layout()
        g.check(L0.gemmul8_get_layout(dt, g.FP8, n, n, k, N, w.data_ptr(), None, None, 0, 0, C.byref(Lo)))
        g.check(L0.gemmul8_scale(st, dt, g.FP8, 0, 0, n, n, k, A.data_ptr(), n, B.data_ptr(), k, N, 1, 0, N, C.byref(Lo), 0, 0))
        torch.cuda.synchronize()
        works[fmt], Ls[fmt] = w, Lo
    for L, fmt, nf, cb in zip(libs, fmts, nofuse, colblk):  # every build reads the knobs once: its own format; ':nofuse' = FP6 planes, three-launch form
        os.environ["GEMMUL8_FP8_PLANES"] = fmt
        os.environ["GEMMUL8_FP8_FUSED"] = "0" if nf else "1"
        if cb is not None:
            os.environ["GEMMUL8_MAP_COLBLOCK"] = cb
        L.gemmul8_reload_knobs()
        os.environ.pop("GEMMUL8_MAP_COLBLOCK", None)
    os.environ.pop("GEMMUL8_FP8_FUSED", None)
    ts = [[] for _ in libs]
    for r in range(a.rounds + 2):
        for i, L in enumerate(libs):
            Lo = Ls[fmts[i]]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                g.check(L.gemmul8_lowprec_gemm(st, dt, g.FP8, n, n, k, N, 0, N, C.byref(Lo)))
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                ts[i].append(e0.elapsed_time(e1) / 2)
    if a.check:
        refC = None
        for i, L in enumerate(libs):
            Lo, w = Ls[fmts[i]], works[fmts[i]]
            offC = Lo.C_mid - w.data_ptr()
            w[offC:offC + 2 * N * Lo.sizeC] = 0x5A
            g.check(L.gemmul8_lowprec_gemm(st, dt, g.FP8, n, n, k, N, 0, N, C.byref(Lo)))
            torch.cuda.synchronize()
            cur = w[offC:offC + 2 * N * Lo.sizeC].clone()
            if refC is None:
                refC = cur
            else:
                print(f"k={k:5d} {a.libs[i]:44s} C_mid planes {'IDENTICAL to' if torch.equal(cur, refC) else 'DIFFER from'} {a.libs[0]}")
        del refC, cur
    if a.smi:
        for i, L in enumerate(libs):
            Lo = Ls[fmts[i]]
            stop = False
            samples = []

            def sample():
                while not stop:
                    try:
                        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
                        m1 = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
                        m2 = re.search(r"Power \(W\): ([0-9.]+)", o)
                        if m1 and m2:
                            samples.append((int(m1.group(1)), float(m2.group(1))))
                    except Exception:
                        pass
            th = threading.Thread(target=sample)
            th.start()
            t0 = time.time()
            while time.time() - t0 < 2.5:
                for _ in range(4):
                    g.check(L.gemmul8_lowprec_gemm(st, dt, g.FP8, n, n, k, N, 0, N, C.byref(Lo)))
                torch.cuda.synchronize()
            stop = True
            th.join()
            busy = [s_ for s_ in samples if s_[1] > 600]
            if busy:
                print(f"k={k:5d} {a.libs[i]:44s} smi: sclk {sum(s_[0] for s_ in busy) / len(busy):6.0f} MHz  power {sum(s_[1] for s_ in busy) / len(busy):6.0f} W  ({len(busy)} samples)")
    gemms = 3 * N  # residue GEMMs per call (three per modulus)
    for i, spec in enumerate(a.libs):
        t = sorted(ts[i])
        med = t[len(t) // 2]
        print(f"k={k:5d} {spec:44s} median {med:8.3f} ms  min {t[0]:8.3f}  -> {gemms * 2.0 * n * n * k / med * 1e-9:6.0f} TOP/s")
    del works, Ls, A, B
