#!/usr/bin/env python3
"""Residue-store policy of the INT8 GEMM epilogue, WHOLE call (gemmul8_gemm: bounds, quantise, GEMMs, CRT), interleaved:
GEMMUL8_EPI_NT unset (the library's rule, oz2_gemm_i8.hip nt_residue_planes) / 0 (cache-allocating stores) / 1 (non-temporal).
usage: python tools/nt_policy_ab.py [--sizes 4096,8192,16384] [--k 128,256,512,1024,2048] [--moduli 14] [--rounds 9] [--fast]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gemmul8_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="4096,8192,16384")
ap.add_argument("--k", default="128,256,512,1024,2048")
ap.add_argument("--moduli", type=int, default=14)
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--fast", action="store_true")
a = ap.parse_args()
lib = g.lib()
N = a.moduli


def setnt(v):
    if v == "-":
        os.environ.pop("GEMMUL8_EPI_NT", None)
    else:
        os.environ["GEMMUL8_EPI_NT"] = v
    lib.gemmul8_reload_knobs()


for n in [int(x) for x in a.sizes.split(",")]:
    for k in [int(x) for x in a.k.split(",")]:
        torch.manual_seed(k)
        A = torch.rand((k, n), dtype=torch.float64, device="cuda") - 0.5
        B = torch.rand((n, k), dtype=torch.float64, device="cuda") - 0.5
        Cout = torch.zeros((n, n), dtype=torch.float64, device="cuda")
        tot, _, _ = g.work_size(False, g.INT8, n, n, k, N)
        work = torch.empty(tot, dtype=torch.uint8, device="cuda")
        ts = {"-": [], "0": [], "1": []}
        for r in range(a.rounds + 2):
            for nt in ts:
                setnt(nt)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.gemm(A, B, N, fastmode=a.fast, C_out=Cout, work=work)
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    ts[nt].append(e0.elapsed_time(e1))
        setnt("-")
        med = {nt: sorted(v)[len(v) // 2] for nt, v in ts.items()}
        print(f"DGEMM {n}^2 x {k:5d}, {N} moduli, {'fast' if a.fast else 'accurate'}: rule {med['-']:7.3f} ms ({2.0 * n * n * k / med['-'] * 1e-9:6.1f} TFLOPS) | "
              f"nt=0 {med['0']:7.3f} ms ({(med['0'] / med['-'] - 1) * 100:+5.1f} %) | nt=1 {med['1']:7.3f} ms ({(med['1'] / med['-'] - 1) * 100:+5.1f} %)", flush=True)
        del A, B, Cout, work
