#!/usr/bin/env python3
"""Exchange variant (A) of the moduli-sharded plan (FP64 partial CRT sums added across ranks) against the
reference-order accumulation: how many output elements change, and by how much, when the num_moduli chains are grouped by rank
(contiguous groups, partials added in rank order).  CPU oracle on a random DGEMM / SGEMM / ZGEMM; integer intermediates are
identical by construction, so only the final CRT step is recomputed.  Output: profiles/archive/r02_variant_a_mismatch.txt"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol


def split_range(total, parts, idx):
    q, r = divmod(total, parts)
    b = idx * q + min(idx, r)
    return b, b + q + (1 if idx < r else 0)


rng = np.random.default_rng(2)
rows = []
for name, dtype, N, m, n, k in (("DGEMM", np.float64, 14, 384, 384, 512), ("DGEMM", np.float64, 16, 384, 384, 512), ("ZGEMM", np.complex128, 20, 192, 192, 256),
                                ("SGEMM", np.float32, 8, 384, 384, 512)):
    A = (rng.random((m, k)) - 0.5).astype(dtype)
    B = (rng.random((k, n)) - 0.5).astype(dtype)
    if np.dtype(dtype).kind == "c":
        A = A + 1j * (rng.random((m, k)) - 0.5)
        B = B + 1j * (rng.random((k, n)) - 0.5)
    for fast in (False, True):
        ref, it = ol.gemm(A, B, N, fastmode=fast, want_intermediates=True)
        code = ol.DT[np.dtype(dtype)]
        al, be = np.array([1], dtype), np.array([0], dtype)
        for world in (2, 4, 8):
            out = np.zeros((m, n), dtype, order="F")
            bounds = np.array([split_range(N, world, r)[0] for r in range(world)] + [N], np.uint32)
            ol.lib().oz2_invscal_grouped(code, 0, N, m, n, ol._p(it["C_mid"]), ol._p(it["sftA"]), ol._p(it["sftB"]), ol._p(al), ol._p(be), ol._p(out), m, 0,
                                         world, ol._p(bounds))
            rt = np.float32 if dtype == np.float32 else np.float64
            a = np.ascontiguousarray(out).view(rt).ravel()
            b = np.ascontiguousarray(ref).view(rt).ravel()
            nbad = int((a != b).sum())
            rel = float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), 1e-300)))
            rows.append(f"{name} {m}x{n}x{k} moduli={N:2d} {'fast' if fast else 'accu'} ranks={world}: {nbad:6d} of {a.size} values differ ({100.0 * nbad / a.size:.3f} %), max rel {rel:.2e}")
            print(rows[-1])
open(os.path.join(ROOT, "profiles", "r02_variant_a_mismatch.txt"), "w").write(
    "# tools/variant_a_mismatch.py: FP64-partial-sum exchange (variant A) vs the reference-order CRT, CPU oracle, INT8 backend, U(-0.5,0.5) inputs\n"
    "# (the residue exchange and the block plan are bit-identical to one GPU by construction: 0 everywhere)\n" + "\n".join(rows) + "\n")
