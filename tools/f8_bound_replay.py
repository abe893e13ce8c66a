#!/usr/bin/env python3
"""Replay fuzz seeds whose FP8 accurate-mode bound came out below the exact sum (tests/test_gpu_fuzz.py) and hold the device's sums against
accumulation models of v_mfma_scale_f32_16x16x128_f8f6f4 (round 4).  usage: python tools/f8_bound_replay.py 7388 10113"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import gemmul8_amd as g
import gpu_util as gu
import oracle_lib as ol
from test_gpu_fuzz import DIMS_K, DIMS_MN, _rand


def case(seed):
    rng = np.random.default_rng(9000 + seed)
    dtype = [np.float64, np.float32, np.complex128, np.complex64][seed % 4]
    backend = g.FP8 if (seed // 4) % 3 == 2 else g.INT8
    is_f32 = dtype in (np.float32, np.complex64)
    N = int(rng.integers(2, 14 if is_f32 else 21))
    fast = bool(rng.integers(0, 2))
    m, n = (int(rng.choice(DIMS_MN)) for _ in range(2))
    k = int(rng.choice(DIMS_K))
    if backend == g.FP8 or np.dtype(dtype).kind == "c":
        m, n = min(m, 257), min(n, 256)
    crt_force = str(rng.choice(["", "dma", "reg"]))
    tile_force = str(rng.choice(["", "128", "256"]))
    nt_force = str(rng.choice(["", "0", "1"]))
    if crt_force == "dma" and backend == g.INT8:
        m, n = (512 if np.dtype(dtype).kind == "c" else 1024), min(n, 64)
    cb_force = str(rng.choice(["", "1", "2", "3"]))
    cplx = np.dtype(dtype).kind == "c"
    opA = str(rng.choice(["N", "T", "C"] if cplx else ["N", "T"]))
    opB = str(rng.choice(["N", "T", "C"] if cplx else ["N", "T"]))
    phi = float(rng.choice([0.0, 1.0, 3.0]))
    A = _rand((m, k) if opA == "N" else (k, m), dtype, rng, phi)
    B = _rand((k, n) if opB == "N" else (n, k), dtype, rng, phi)
    if rng.integers(0, 3) == 0 and m > 2:
        (A if opA == "N" else A.T)[m // 2, :] = 0
    return dict(dtype=dtype, backend=backend, N=N, fast=fast, m=m, n=n, k=k, opA=opA, opB=opB, phi=phi, A=A, B=B)


def model_sum(a, b, bits, mode):
    """one row x one column: groups of 8 consecutive k (inside 128-byte K-steps), products aligned to the group's reference exponent,
    truncated `bits` below it.  mode 'prod': reference = exponent of the largest product; 'opexp': largest (ea + eb) of the operands."""
    k = len(a)
    kp = (k + 127) // 128 * 128
    a = np.concatenate([a, np.zeros(kp - k)])
    b = np.concatenate([b, np.zeros(kp - k)])
    tot = 0.0
    for g0 in range(0, kp, 8):
        p = a[g0:g0 + 8] * b[g0:g0 + 8]
        if not p.any():
            continue
        if mode == "prod":
            e = np.floor(np.log2(p.max()))
        elif mode == "sepmax":  # largest A exponent of the group + largest B exponent of the group (not necessarily of the same product)
            aa, bb = a[g0:g0 + 8], b[g0:g0 + 8]
            if not (aa > 0).any() or not (bb > 0).any():
                continue
            e = np.floor(np.log2(aa.max())) + np.floor(np.log2(bb.max()))
        else:
            nz = (a[g0:g0 + 8] > 0) & (b[g0:g0 + 8] > 0)
            e = np.max(np.floor(np.log2(a[g0:g0 + 8][nz])) + np.floor(np.log2(b[g0:g0 + 8][nz])))
        grid = 2.0 ** (e - bits)
        tot += np.sum(np.floor(p / grid) * grid)
    return tot


for seed in [int(x) for x in sys.argv[1:]]:
    c = case(seed)
    A, B, N = c["A"], c["B"], c["N"]
    print(f"seed {seed}: {np.dtype(c['dtype']).name} backend {'FP8' if c['backend'] else 'INT8'} N={N} m={c['m']} n={c['n']} k={c['k']} op {c['opA']}{c['opB']} phi={c['phi']} fast={c['fast']}")
    if c["backend"] != g.FP8 or np.dtype(c["dtype"]).kind == "c":
        continue
    # device bound maxima (as bounds_case reads them)
    dA, dB = gu.to_dev(A), gu.to_dev(B)
    m, k = (A.shape if c["opA"] == "N" else A.shape[::-1])
    n = B.shape[1] if c["opB"] == "N" else B.shape[0]
    tot, _, _ = g.work_size(False, g.FP8, m, n, k, N)
    work = torch.full((tot,), 0x5A, dtype=torch.uint8, device="cuda")
    L = g.Layout()
    code = g._dtype_code(dA.dtype)
    lib = g.lib()
    g.check(lib.gemmul8_get_layout(code, g.FP8, m, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(L)))
    st = torch.cuda.current_stream().cuda_stream
    g.check(lib.gemmul8_scale_bounds(st, code, g.FP8, g.OPS[c["opA"]], g.OPS[c["opB"]], m, n, k, dA.data_ptr(), dA.shape[1], dB.data_ptr(), dB.shape[1], N, 0, n,
                                     C.byref(L), 0, 0))
    torch.cuda.synchronize()
    w = work.cpu().numpy()
    base = work.data_ptr()
    np_ = (n + 255) // 256 * 256
    mx = w[L.scratch - base:L.scratch - base + 4 * (L.mp + np_)]
    rmax, cmax = mx[:4 * m].view(np.float32).astype(np.float64), mx[4 * L.mp:4 * L.mp + 4 * n].view(np.float32).astype(np.float64)
    oA, _ = ol.extract_bounds(A, c["opA"], True, g.FP8)
    oB, _ = ol.extract_bounds(B, c["opB"], False, g.FP8)
    a = ol.e4m3_decode(oA[0])
    b = ol.e4m3_decode(oB[0])
    P = a @ b.T
    ku = ol.fp8_bound_ku(k, 0)
    ex_r = P.max(axis=1)
    rel = (rmax - ex_r) / np.maximum(ex_r, 1e-300)
    i = int(np.argmin(rel))
    j = int(np.argmax(P[i]))
    print(f"  ku = {ku:.4e} = {ku * 2 ** 13:.3f} * 2^-13; worst row {i}: exact max {ex_r[i]:.9g} (column {j}), device {rmax[i]:.9g}, (dev - exact) / exact = {rel[i]:.3e}")
    print(f"  -> engine loss on that sum, if the maximum is still at column {j}: {1 - rmax[i] / (1 + ku) / ex_r[i]:.4e} = {(1 - rmax[i] / (1 + ku) / ex_r[i]) * 2 ** 13:.3f} * 2^-13")
    print(f"     device sum (un-inflated, approx): {rmax[i] / (1 + ku):.9g}")
    for bits in (13, 12, 14):
        for mode in ("prod", "opexp", "sepmax"):
            ms = model_sum(a[i], b[j], bits, mode)
            print(f"     model {mode:6s} {bits} bits: sum {ms:.9g}  loss {(1 - ms / ex_r[i]) * 2 ** 13:.3f} * 2^-13   inflated {ms * (1 + ku):.9g}")

    # per-row table (n == 1 cases are the informative ones: every row's maximum is its only sum)
    if n == 1:
        def uninflate(d):
            c0 = np.float32(d / (1 + ku))
            for step in range(-6, 7):
                c = np.float32(c0)
                for _ in range(abs(step)):
                    c = np.nextafter(c, np.float32(np.inf if step > 0 else -np.inf), dtype=np.float32)
                v = float(c) + float(np.float32(ku)) * float(c)
                r = np.float32(v)
                if float(r) < v:
                    r = np.nextafter(r, np.float32(np.inf), dtype=np.float32)
                if float(r) == d:
                    return float(c)
            return float(c0)
        print("  row: device sum | exact | loss(dev) and model losses in units of 2^-13 of the exact sum: prod13 sepmax13 sepmax14 sepmax15 sepmax16 | prod13 with subnormal inputs flushed: both, A only, B only")
        for r in range(m):
            if ex_r[r] <= 0:
                continue
            dsum = uninflate(rmax[r])
            ls = [(1 - model_sum(a[r], b[0], bits, mode) / ex_r[r]) * 2 ** 13 for mode, bits in (("prod", 13), ("sepmax", 13), ("sepmax", 14), ("sepmax", 15), ("sepmax", 16))]
            af, bf_ = np.where(a[r] < 2.0 ** -6, 0.0, a[r]), np.where(b[0] < 2.0 ** -6, 0.0, b[0])   # e4m3 subnormal operands flushed to zero
            ls.append((1 - model_sum(af, bf_, 13, "prod") / ex_r[r]) * 2 ** 13)
            ls.append((1 - model_sum(af, b[0], 13, "prod") / ex_r[r]) * 2 ** 13)
            ls.append((1 - model_sum(a[r], bf_, 13, "prod") / ex_r[r]) * 2 ** 13)
            print(f"   {r:3d}: {dsum:.9g} | {ex_r[r]:.9g} | {(1 - dsum / ex_r[r]) * 2 ** 13:7.3f}   " + " ".join(f"{x:7.3f}" for x in ls))
