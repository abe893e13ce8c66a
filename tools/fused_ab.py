#!/usr/bin/env python3
"""Interleaved A/B of the whole emulated GEMM call with the CRT fused into the tile-stationary INT8 kernel (GEMMUL8_FUSED_CRT=1)
against the two-launch path (=0), same process, same buffers, alternating rounds (box-to-box spread is +-3 %).  Runs on the
LABORATORY library (tools/experiments/fused_crt/lib/libgemmul8_lab.so: the in-kernel CRT is not part of libgemmul8.so).
usage: tools/fused_ab.py [dtype=d|s] [N=14] [shapes m,n,k ...]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gemmul8_amd as g

_product = g.lib()
_lab = g.bind(C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments", "fused_crt", "lib", "libgemmul8_lab.so")))
_lab.gemmul8_lab_gemm.restype = C.c_int
_lab.gemmul8_lab_gemm.argtypes = _product.gemmul8_gemm.argtypes
_lab.gemmul8_gemm = _lab.gemmul8_lab_gemm  # g.gemm() now runs the laboratory pipeline
_lab.gemmul8_fused_crt_selected.restype = C.c_int
_lab.gemmul8_fused_crt_selected.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_uint]
g._lib = _lab

typ = sys.argv[1] if len(sys.argv) > 1 else "d"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 14
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[3:]] or [(8192, 8192, 8192), (8192, 8192, 1024), (8192, 8192, 2048), (8192, 8192, 4096),
                                                                        (4096, 4096, 4096), (16384, 16384, 2048), (6144, 6144, 6144)]
dt = torch.float64 if typ == "d" else torch.float32
for fast in (0, 1):
    for m, n, k in shapes:
        A = torch.rand((k, m), dtype=dt, device="cuda") - 0.5
        B = torch.rand((n, k), dtype=dt, device="cuda") - 0.5
        Cc = torch.zeros((n, m), dtype=dt, device="cuda")
        tot, _, _ = g.work_size(False, g.INT8, m, n, k, N)
        work = torch.empty(tot, dtype=torch.uint8, device="cuda")
        best = {"0": 1e9, "1": 1e9, "2": 1e9}
        for rnd in range(3):
            for mode in ("0", "1", "2"):
                os.environ["GEMMUL8_FUSED_CRT"] = mode
                for _ in range(2):
                    g.gemm(A, B, N, fastmode=bool(fast), C_out=Cc, work=work)
                torch.cuda.synchronize()
                reps = 10 if m * n * k >= 2 ** 36 else 30
                t0 = time.perf_counter()
                for _ in range(reps):
                    g.gemm(A, B, N, fastmode=bool(fast), C_out=Cc, work=work)
                torch.cuda.synchronize()
                best[mode] = min(best[mode], (time.perf_counter() - t0) / reps)
        os.environ.pop("GEMMUL8_FUSED_CRT")
        sel = g.lib().gemmul8_fused_crt_selected(g.D if typ == "d" else g.S, g.INT8, m, n, N)
        fl = 2.0 * m * n * k
        print(f"{typ}gemm {m}x{n}x{k} N={N} fast={fast}: two-launch {best['0'] * 1e3:7.3f} ms ({fl / best['0'] * 1e-12:6.1f} TFLOPS)  fused, CRT on the producer waves {best['1'] * 1e3:7.3f} ms "
              f"({100 * (best['0'] / best['1'] - 1):+5.1f} %)  fused, CRT tail on the consumer waves {best['2'] * 1e3:7.3f} ms ({100 * (best['0'] / best['2'] - 1):+5.1f} %)  "
              f"default rule picks {'fused' if sel else 'two-launch'}", flush=True)
        del A, B, Cc, work
