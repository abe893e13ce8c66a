#!/usr/bin/env python3
"""Print the kernel timeline (start offset, duration, gap to the previous kernel) of ONE emulated DGEMM call from a rocprofv3
kernel trace.  usage: rocprofv3 --kernel-trace --output-format csv -d out -o r -- python tools/gap_probe.py run <n>; then
python tools/gap_probe.py show out/r_kernel_trace.csv"""
import csv, sys
if sys.argv[1] == "run":
    import torch
    import gemmul8_amd as g
    n = int(sys.argv[2])
    gen = torch.Generator(device="cuda").manual_seed(3)
    A = torch.rand((n, n), generator=gen, dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((n, n), generator=gen, dtype=torch.float64, device="cuda") - 0.5
    Cm = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    tot, _, _ = g.work_size(False, g.INT8, n, n, n, 14)
    work = torch.empty(tot, dtype=torch.uint8, device="cuda")
    for _ in range(5):
        g.gemm(A, B, 14, C_out=Cm, work=work)
    torch.cuda.synchronize()
else:
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if "oz2" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]]
    last = rows[-12:]
    t0 = int(last[0]["Start_Timestamp"])
    prev_end = None
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:7.1f} us  {r['Kernel_Name'][:60]}")
        prev_end = e
