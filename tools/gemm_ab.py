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
ap.add_argument("--check", action="store_true", help="compare the residue planes every build writes with the first build's (bit for bit)")
ap.add_argument("--smi", action="store_true", help="sample sclk / socket power (rocm-smi) while each build loops for ~1.5 s")
ap.add_argument("--fused", action="store_true", help="time gemmul8_lowprec_gemm_crt (GEMMs + CRT in one launch) of every build; the first build's two-launch path (lowprec_gemm + crt) is timed beside it")
ap.add_argument("--range", type=int, default=0, help="operand bytes uniform in [-R, R] instead of all 256 int8 values (energy per MAC depends on the data)")
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
    if a.fused:  # laboratory builds only (tools/experiments/fused_crt/lib/libgemmul8_lab.so and its ablation builds)
        L.gemmul8_lowprec_gemm_crt.restype = C.c_int
        L.gemmul8_lowprec_gemm_crt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.POINTER(g.Layout),
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.gemmul8_crt.restype = C.c_int
    L.gemmul8_crt.argtypes = ref.gemmul8_crt.argtypes
    libs.append(L)
if a.fused:
    libs.append(None)  # the two-launch path of the first build
    a.libs.append("two-launch(" + os.path.basename(a.libs[0]) + ")")
import numpy as np
one, zero = np.array([1.0]), np.array([0.0])
st = torch.cuda.current_stream().cuda_stream
for k in [int(x) for x in a.k.split(",")]:
    tot, _, _ = g.work_size(False, g.INT8, n, n, k, N)
    work = torch.empty(tot, dtype=torch.uint8, device="cuda")
    Lo = g.Layout()
    g.check(ref.gemmul8_get_layout(g.D, g.INT8, n, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(Lo)))
    offA, offB = Lo.A_lo - work.data_ptr(), Lo.B_lo - work.data_ptr()
    def planes(nbytes):
        if a.range:
            return torch.randint(-a.range, a.range + 1, (nbytes,), dtype=torch.int8, device="cuda").view(torch.uint8)
        return torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda")
    work[offA:offA + N * Lo.sizeA] = planes(N * Lo.sizeA)
    work[offB:offB + N * Lo.sizeB] = planes(N * Lo.sizeB)
    Cout = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    offS = Lo.sftA - work.data_ptr()
    work[offS:offS + 2 * Lo.mp] = 0
    offS = Lo.sftB - work.data_ptr()
    work[offS:offS + 2 * n] = 0
    ts = [[] for _ in libs]
    for r in range(a.rounds + 2):
        for i, L in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                if not a.fused:
                    g.check(L.gemmul8_lowprec_gemm(st, g.D, g.INT8, n, n, k, N, 0, N, C.byref(Lo)))
                elif L is not None:
                    g.check(L.gemmul8_lowprec_gemm_crt(st, g.D, g.INT8, n, n, k, N, C.byref(Lo), one.ctypes.data, zero.ctypes.data, Cout.data_ptr(), n))
                else:
                    g.check(libs[0].gemmul8_lowprec_gemm(st, g.D, g.INT8, n, n, k, N, 0, N, C.byref(Lo)))
                    g.check(libs[0].gemmul8_crt(st, g.D, g.INT8, N, n, n, Lo.C_mid, Lo.mp, Lo.sizeC, Lo.sftA, Lo.sftB, one.ctypes.data, zero.ctypes.data,
                                                Cout.data_ptr(), n))
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                ts[i].append(e0.elapsed_time(e1) / 3)
    if a.check and not a.fused:
        offC = Lo.C_mid - work.data_ptr()
        refC = None
        for i, L in enumerate(libs):
            work[offC:offC + N * Lo.sizeC] = 0x5A
            g.check(L.gemmul8_lowprec_gemm(st, g.D, g.INT8, n, n, k, N, 0, N, C.byref(Lo)))
            torch.cuda.synchronize()
            cur = work[offC:offC + N * Lo.sizeC].clone()
            if refC is None:
                refC = cur
            else:
                print(f"k={k:5d} {os.path.basename(a.libs[i]):40s} residue planes {'IDENTICAL to' if torch.equal(cur, refC) else 'DIFFER from'} {os.path.basename(a.libs[0])}")
        del refC, cur
    if a.smi and not a.fused:
        import subprocess, threading, time, re
        for i, L in enumerate(libs):
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
                for _ in range(20):
                    g.check(L.gemmul8_lowprec_gemm(st, g.D, g.INT8, n, n, k, N, 0, N, C.byref(Lo)))
                torch.cuda.synchronize()
            stop = True
            th.join()
            busy = [s_ for s_ in samples if s_[1] > 600]
            if busy:
                print(f"k={k:5d} {os.path.basename(a.libs[i]):40s} smi: sclk {sum(s_[0] for s_ in busy) / len(busy):6.0f} MHz  power {sum(s_[1] for s_ in busy) / len(busy):6.0f} W  ({len(busy)} samples)")
    for i, pth in enumerate(a.libs):
        t = sorted(ts[i])
        med = t[len(t) // 2]
        print(f"k={k:5d} {os.path.basename(pth):40s} median {med:7.3f} ms  min {t[0]:7.3f}  -> {N * 2.0 * n * n * k / med * 1e-9:6.0f} TOP/s")
    del work
