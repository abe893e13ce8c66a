#!/usr/bin/env python3
"""WHOLE-call A/B (gemmul8_gemm: bounds, quantise, GEMMs, CRT) between settings of the library's testing knobs (csrc/oz2_knobs.hpp), interleaved in
one process, median of --rounds; every arm's C must equal the first arm's bit for bit.
usage: python tools/knob_ab.py --arms "base;GEMMUL8_CRT_PANELS=8;GEMMUL8_CRT_PANELS=8r,GEMMUL8_EPI_NT=0" [--sizes 8192] [--k 128,256] [--moduli 14]
       [--dtype d|s] [--backend int8|fp8] [--fast] [--rounds 9]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gemmul8_amd as g

ap = argparse.ArgumentParser()
ap.add_argument("--arms", required=True)
ap.add_argument("--sizes", default="8192")
ap.add_argument("--k", default="128,256,512,1024")
ap.add_argument("--moduli", type=int, default=14)
ap.add_argument("--dtype", default="d")
ap.add_argument("--backend", default="int8")
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--fast", action="store_true")
a = ap.parse_args()
lib = g.lib()
N = a.moduli
tdt = {"d": torch.float64, "s": torch.float32}[a.dtype]
be = g.FP8 if a.backend == "fp8" else g.INT8
arms = []
for spec in a.arms.split(";"):
    env = dict(kv.split("=", 1) for kv in spec.split(",") if "=" in kv)
    arms.append((spec, env))
allkeys = sorted({k_ for _, e in arms for k_ in e})


def select(env):
    for k_ in allkeys:
        if k_ in env:
            os.environ[k_] = env[k_]
        else:
            os.environ.pop(k_, None)
    lib.gemmul8_reload_knobs()


for n in [int(x) for x in a.sizes.split(",")]:
    for k in [int(x) for x in a.k.split(",")]:
        torch.manual_seed(k)
        A = (torch.rand((k, n), dtype=torch.float64, device="cuda") - 0.5).to(tdt)
        B = (torch.rand((n, k), dtype=torch.float64, device="cuda") - 0.5).to(tdt)
        Cout = torch.zeros((n, n), dtype=tdt, device="cuda")
        tot, _, _ = g.work_size(False, be, n, n, k, N)
        work = torch.empty(tot, dtype=torch.uint8, device="cuda")
        ts = {name: [] for name, _ in arms}
        Cref = None
        for r in range(a.rounds + 2):
            for name, env in arms:
                select(env)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.gemm(A, B, N, fastmode=a.fast, backend=be, C_out=Cout, work=work)
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    ts[name].append(e0.elapsed_time(e1))
                if r == 0:
                    if Cref is None:
                        Cref = Cout.clone()
                    else:
                        assert torch.equal(Cout, Cref), f"{name}: C differs from the first arm's"
        select({})
        base = sorted(ts[arms[0][0]])[len(ts[arms[0][0]]) // 2]
        print(f"{a.dtype.upper()}GEMM {n}^2 x {k}, {N} moduli, {a.backend}, {'fast' if a.fast else 'accurate'} mode, whole call, median of {a.rounds} (bit-identical C):")
        for name, _ in arms:
            med = sorted(ts[name])[len(ts[name]) // 2]
            print(f"   {name:48s} {med:8.3f} ms  {2.0 * n * n * k / med * 1e-9:7.1f} TFLOPS  ({(med / base - 1) * 100:+6.2f} %)", flush=True)
        del A, B, Cout, work, Cref
