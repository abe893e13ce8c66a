# Yardstick only (NOT used by the product): vendor INT8 GEMM throughput through torch._int_mm (hipBLASLt) on the same shape.
import torch, time
n = 8192
a = torch.randint(-127, 127, (n, n), dtype=torch.int8, device="cuda")
b = torch.randint(-127, 127, (n, n), dtype=torch.int8, device="cuda")
for f in (lambda: torch._int_mm(a, b), lambda: torch._int_mm(a, b.t())):
    try:
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"torch._int_mm {n}^3: {ms:.3f} ms -> {2*n**3/ms*1e-9:.0f} TOP/s")
    except Exception as ex:
        print("failed:", str(ex)[:200])
x = torch.randn((n, n), dtype=torch.bfloat16, device="cuda"); y = torch.randn((n, n), dtype=torch.bfloat16, device="cuda")
for _ in range(3): x @ y
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): x @ y
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"bf16 matmul {n}^3: {ms:.3f} ms -> {2*n**3/ms*1e-9:.0f} TFLOP/s")
d = torch.randn((n, n), dtype=torch.float64, device="cuda"); e = torch.randn((n, n), dtype=torch.float64, device="cuda")
for _ in range(2): d @ e
torch.cuda.synchronize()
e0.record()
for _ in range(3): d @ e
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"native fp64 matmul {n}^3: {ms:.3f} ms -> {2*n**3/ms*1e-9:.1f} TFLOP/s")
