#!/usr/bin/env python3
"""What the LD_PRELOAD hook reaches inside an unmodified PyTorch process.  Run under
    LD_PRELOAD=gemmul8_amd/lib/libgemmul8_preload.so GEMMUL8_NUM_MOD_D=<N> GEMMUL8_HOOK_STATS=1 [GEMMUL8_HOOK_ROCBLAS=1]
or without LD_PRELOAD for the native numbers.  Prints one JSON line per experiment (tests/test_gpu_hook_reach.py parses them; the hook's own
stats line at exit says how many GEMM calls / flops went through the emulation):
  lu_factor   torch.linalg.lu_factor of an n x n float64 matrix (hipSOLVER / rocSOLVER getrf: its trailing updates are rocBLAS-internal)
  solve       torch.linalg.solve
  blocked_lu  a right-looking blocked LU written with torch ops: panel = torch.linalg.lu_factor on the tall panel, row swaps, triangular
              solve, trailing update C -= L U with torch.addmm-style matmuls -- the HPL shape; the trailing updates are plain DGEMMs
              that reach hipBLAS / hipBLASLt and therefore the hook
Residual: ||P A - L U||_F / (||A||_F n eps)."""
import argparse
import json
import os
import time

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--nb", type=int, default=1024, help="panel width of the blocked LU (= k of its trailing updates)")
ap.add_argument("--what", default="lu_factor,solve,blocked_lu")
a = ap.parse_args()
n, nb = a.n, a.nb
torch.manual_seed(0)
dev = "cuda"
hooked = "gemmul8" in os.environ.get("LD_PRELOAD", "")
A = torch.rand((n, n), dtype=torch.float64, device=dev) - 0.5 + torch.eye(n, dtype=torch.float64, device=dev) * 2.0
eps = 2.0 ** -52


def residual_lu(A0, LU, piv):
    P, L, U = torch.lu_unpack(LU, piv)
    # ||P^T... : torch convention A = P L U
    R = A0 - P @ (L @ U)
    return float((R.norm() / (A0.norm() * n * eps)).item())


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def blocked_lu(A0):
    """Right-looking blocked LU with partial pivoting; returns (LU, perm) with A0[perm] = L U."""
    M = A0.clone()
    perm = torch.arange(n, device=dev)
    for j in range(0, n, nb):
        w = min(nb, n - j)
        LUp, piv = torch.linalg.lu_factor(M[j:, j:j + w])            # tall panel (n - j) x w
        # turn LAPACK pivots into a permutation of the panel's rows and apply it to the whole trailing rows
        p = torch.arange(n - j, device=dev)
        pv = (piv - 1).tolist()
        for i, t in enumerate(pv):
            if t != i:
                tmp = p[i].clone()
                p[i] = p[t]
                p[t] = tmp
        M[j:, :] = M[j:, :][p]
        perm[j:] = perm[j:][p]
        M[j:, j:j + w] = LUp
        if j + w < n:
            L11 = torch.tril(M[j:j + w, j:j + w], -1) + torch.eye(w, dtype=M.dtype, device=dev)
            M[j:j + w, j + w:] = torch.linalg.solve_triangular(L11, M[j:j + w, j + w:], upper=False)   # U12
            M[j + w:, j + w:] -= M[j + w:, j:j + w] @ M[j:j + w, j + w:]                                 # trailing update: DGEMM, k = w
    return M, perm


for what in a.what.split(","):
    if what == "lu_factor":
        dt, (LU, piv) = timed(lambda: torch.linalg.lu_factor(A))
        print(json.dumps({"what": what, "n": n, "hooked": hooked, "ms": dt * 1e3, "tflops": 2 / 3 * n ** 3 / dt * 1e-12, "residual": residual_lu(A, LU, piv)}), flush=True)
    elif what == "solve":
        B = torch.rand((n, 64), dtype=torch.float64, device=dev)
        dt, X = timed(lambda: torch.linalg.solve(A, B))
        r = float(((A @ X - B).norm() / (A.norm() * X.norm() * eps * n)).item())
        print(json.dumps({"what": what, "n": n, "hooked": hooked, "ms": dt * 1e3, "residual": r}), flush=True)
    elif what == "blocked_lu":
        dt, (M, perm) = timed(lambda: blocked_lu(A), reps=1)
        L = torch.tril(M, -1) + torch.eye(n, dtype=M.dtype, device=dev)
        U = torch.triu(M)
        R = A[perm] - L @ U
        r = float((R.norm() / (A.norm() * n * eps)).item())
        upd = sum(2.0 * (n - j - min(nb, n - j)) ** 2 * min(nb, n - j) for j in range(0, n, nb))
        print(json.dumps({"what": what, "n": n, "nb": nb, "hooked": hooked, "ms": dt * 1e3, "tflops": 2 / 3 * n ** 3 / dt * 1e-12, "residual": r,
                          "trailing_update_tflop": upd * 1e-12}), flush=True)
