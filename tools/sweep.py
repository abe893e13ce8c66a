#!/usr/bin/env python3
"""Reference-style sweeps on one MI355X:
  accuracy: 128 x k x 128, k = 2^10..2^16, inputs (U(0,1)-0.5)*exp(phi*N(0,1)), phi in {0, 0.5, 1, 2, 4}
            (testing/test_accuracy.hpp:20-21,67-69; make_matrix.hpp:78): max |C - C_true| / |C_true| of the emulation
            (fast / accurate, several num_moduli) next to the native DGEMM, C_true from 80-bit long-double dot products.
  flops:    m = n and k sweeps (testing/test_flops.hpp:38-56): median-of-10 event timings, TFLOPS = 2mnk/t, per-phase
            times from the call's timers, native DGEMM (rocBLAS via torch) alongside.
  watt:     socket power / shader clock sampled with rocm-smi while the headline configuration loops -> GFLOPS/W.
Writes CSVs under the directory given as argv[1] (default gpurun_out/sweeps)."""
import csv
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gemmul8_amd as g

out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweeps"
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["accuracy", "flops", "watt"]
os.makedirs(out_dir, exist_ok=True)
dev = torch.device("cuda", 0)


def colmajor(x):  # numpy (rows, cols) -> torch tensor (cols, rows) holding the column-major matrix
    return torch.from_numpy(np.ascontiguousarray(x.T)).to(dev)


def accuracy():
    """testing/test_accuracy.hpp:67-208: S / D / C / Z x both backends, 128 x k x 128, phi sweep; one CSV per type and backend.  The moduli
    counts follow testing/common.hpp:38-43 (D / Z: up to 20, S / C: up to 12)."""
    m = n = 128
    types = [("dgemm", np.float64, False), ("sgemm", np.float32, False), ("zgemm", np.float64, True), ("cgemm", np.float32, True)]
    for name, rdt, cplx in types:
        for bname, be in (("int8", g.INT8), ("fp8", g.FP8)):
            rows = []
            rng = np.random.default_rng(2024)
            ks = [1024, 4096, 16384] if cplx else [1024, 4096, 16384, 65536]   # (the 80-bit complex reference product at k = 65536 takes minutes on the host)
            mods = ([8, 10, 12, 14, 16, 18, 20] if rdt == np.float64 else [3, 4, 6, 8, 10, 12])
            for k in ks:
                for phi in [0.0, 0.5, 1.0, 2.0, 4.0]:
                    def rnd(shape):
                        x = ((rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))).astype(rdt)
                        if cplx:
                            x = x + 1j * ((rng.random(shape) - 0.5) * np.exp(phi * rng.standard_normal(shape))).astype(rdt)
                        return x
                    A, B = rnd((m, k)), rnd((k, n))
                    hp = np.clongdouble if cplx else np.longdouble
                    truth = A.astype(hp) @ B.astype(hp)
                    At, Bt = colmajor(A), colmajor(B)
                    nat = torch.matmul(Bt, At).cpu().numpy().T  # (A B)^T in tensor terms
                    den = np.abs(truth)
                    rec = {"k": k, "phi": phi, "native": float(np.max(np.abs(nat - truth) / den))}
                    for N in mods:
                        for fast in [False, True]:
                            Cm, _, _ = g.gemm(At, Bt, N, fastmode=fast, backend=be)
                            got = Cm.cpu().numpy().T
                            rec[f"N{N}_{'fast' if fast else 'accu'}"] = float(np.max(np.abs(got - truth) / den))
                    rows.append(rec)
                    print(name, bname, rec, flush=True)
            with open(os.path.join(out_dir, f"accuracy_{name}_{bname}.csv"), "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                w.writeheader()
                w.writerows(rows)


def timed(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def flops():
    rows = []
    gen = torch.Generator(device=dev).manual_seed(7)
    for mn in [1024, 2048, 4096, 8192, 16384]:
        for k in [1024, 4096, 8192, 16384]:
            if mn == 16384 and k == 16384:
                reps = 3
            else:
                reps = 10
            A = torch.rand((k, mn), generator=gen, dtype=torch.float64, device=dev) - 0.5   # column-major m x k
            B = torch.rand((mn, k), generator=gen, dtype=torch.float64, device=dev) - 0.5   # column-major k x n
            Cm = torch.zeros((mn, mn), dtype=torch.float64, device=dev)
            rec = {"m=n": mn, "k": k}
            fl = 2.0 * mn * mn * k
            rec["native_dgemm_TFLOPS"] = fl / timed(lambda: torch.matmul(B, A), reps) * 1e-9
            for N in [14]:
                tot, _, _ = g.work_size(False, g.INT8, mn, mn, k, N)
                work = torch.empty(tot, dtype=torch.uint8, device=dev)
                for fast in [False, True]:
                    tag = f"N{N}_{'fast' if fast else 'accu'}"
                    ms = timed(lambda: g.gemm(A, B, N, fastmode=fast, C_out=Cm, work=work), reps)
                    rec[tag + "_TFLOPS"] = fl / ms * 1e-9
                    _, tm, _ = g.gemm(A, B, N, fastmode=fast, C_out=Cm, work=work, timers=True)
                    rec[tag + "_phase_ms(scale|gemm|requant|crt)"] = "|".join(f"{x * 1e-6:.3f}" for x in tm)
                del work
            rows.append(rec)
            print(rec, flush=True)
            del A, B, Cm
    with open(os.path.join(out_dir, "flops_dgemm_int8.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


def watt():
    n, N = 8192, 14
    gen = torch.Generator(device=dev).manual_seed(9)
    A = torch.rand((n, n), generator=gen, dtype=torch.float64, device=dev) - 0.5
    B = torch.rand((n, n), generator=gen, dtype=torch.float64, device=dev) - 0.5
    Cm = torch.zeros((n, n), dtype=torch.float64, device=dev)
    tot, _, _ = g.work_size(False, g.INT8, n, n, n, N)
    work = torch.empty(tot, dtype=torch.uint8, device=dev)
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                p = [float(l.split(":")[-1]) for l in o.splitlines() if "Power (W)" in l]
                c = [l.split("(")[-1].split("Mhz")[0] for l in o.splitlines() if "sclk" in l]
                if p:
                    samples.append((p[0], float(c[0]) if c else float("nan")))
            except Exception:
                pass
            time.sleep(0.2)

    rows = []
    for name, fn in [("emulated_dgemm_N14_accu", lambda: g.gemm(A, B, N, C_out=Cm, work=work)), ("native_dgemm", lambda: torch.matmul(B, A))]:
        samples.clear()
        stop.clear()
        th = threading.Thread(target=sampler)
        fn()
        torch.cuda.synchronize()
        th.start()
        t0 = time.perf_counter()
        it = 0
        while time.perf_counter() - t0 < 6.0:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            it += 20
        dt = time.perf_counter() - t0
        stop.set()
        th.join()
        tf = 2.0 * n ** 3 * it / dt * 1e-12
        ps = [s[0] for s in samples[1:]] or [float("nan")]
        cs = [s[1] for s in samples[1:]] or [float("nan")]
        rows.append({"what": name, "TFLOPS": tf, "mean_socket_W": float(np.mean(ps)), "max_socket_W": float(np.max(ps)),
                     "mean_sclk_MHz": float(np.mean(cs)), "GFLOPS_per_W": tf * 1e3 / float(np.mean(ps)), "samples": len(ps)})
        print(rows[-1], flush=True)
    with open(os.path.join(out_dir, "watt_dgemm_8192.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


def types():
    """Every dtype x backend combination at 4096^3 and 8192^3: TFLOPS (complex counts 4x flops, test_flops.hpp:38-40),
    sampled max relative error vs an 80-bit product, native GEMM of the same type alongside."""
    rows = []
    gen = torch.Generator(device=dev).manual_seed(11)
    cases = [("S", torch.float32, g.INT8, 7), ("S", torch.float32, g.FP8, 6), ("D", torch.float64, g.INT8, 14), ("D", torch.float64, g.FP8, 12),
             ("C", torch.complex64, g.INT8, 7), ("C", torch.complex64, g.FP8, 6), ("Z", torch.complex128, g.INT8, 14),
             ("Z", torch.complex128, g.FP8, 12)]
    for n in [4096, 8192]:
        for name, dt, be, N in cases:
            rdt = torch.float32 if dt in (torch.float32, torch.complex64) else torch.float64

            def rnd():
                x = torch.rand((n, n), generator=gen, dtype=rdt, device=dev) - 0.5
                if dt.is_complex:
                    x = torch.complex(x, torch.rand((n, n), generator=gen, dtype=rdt, device=dev) - 0.5)
                return x.contiguous()
            A, B = rnd(), rnd()
            Cm = torch.zeros((n, n), dtype=dt, device=dev)
            tot, _, _ = g.work_size(dt.is_complex, be, n, n, n, N)
            work = torch.empty(tot, dtype=torch.uint8, device=dev)
            fl = (4 if dt.is_complex else 1) * 2.0 * n ** 3
            rec = {"gemm": name + "GEMM", "n": n, "backend": "INT8" if be == g.INT8 else "FP8", "num_moduli": N}
            for fast in [False, True]:
                ms = timed(lambda: g.gemm(A, B, N, fastmode=fast, backend=be, C_out=Cm, work=work), 5, 1)
                rec[("fast" if fast else "accu") + "_TFLOPS"] = fl / ms * 1e-9
            g.gemm(A, B, N, fastmode=False, backend=be, C_out=Cm, work=work)
            rows_i = np.arange(0, n, n // 24)[:24]
            cols_i = np.arange(5, n, n // 24)[:24]
            hp = np.clongdouble if dt.is_complex else np.longdouble
            ref = A[:, rows_i].cpu().numpy().T.astype(hp) @ B[cols_i, :].cpu().numpy().T.astype(hp)
            rec["accu_max_rel_err"] = float(np.max(np.abs(Cm[cols_i][:, rows_i].cpu().numpy().T - ref) / np.abs(ref)))
            nat_ms = timed(lambda: torch.matmul(B, A), 5, 1)
            Cn = torch.matmul(B, A)
            rec["native_TFLOPS"] = fl / nat_ms * 1e-9
            rec["native_max_rel_err"] = float(np.max(np.abs(Cn[cols_i][:, rows_i].cpu().numpy().T - ref) / np.abs(ref)))
            rows.append(rec)
            print(rec, flush=True)
            del A, B, Cm, Cn, work
    with open(os.path.join(out_dir, "types_backends.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


for wname in which:
    {"accuracy": accuracy, "flops": flops, "watt": watt, "types": types}[wname]()
