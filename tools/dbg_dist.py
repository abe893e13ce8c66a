import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
import gemmul8_amd as g
from gemmul8_amd import dist as gd
dist.init_process_group("gloo")
rank = dist.get_rank(); world = dist.get_world_size()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]); N = 14
gA = torch.Generator(device=dev).manual_seed(12345); gB = torch.Generator(device=dev).manual_seed(54321)
A = torch.rand((n, n), generator=gA, dtype=torch.float64, device=dev) - 0.5
B = torch.rand((n, n), generator=gB, dtype=torch.float64, device=dev) - 0.5
Cm = torch.zeros((n, n), dtype=torch.float64, device=dev)
plan = gd.ShardedGemm(g.D, g.INT8, n, n, n, N, device=dev)
plan.run(A, B, Cm)
torch.cuda.synchronize(); dist.barrier()
eng = plan.eng
# single-GPU reference with intermediates
tot, _, _ = g.work_size(False, g.INT8, n, n, n, N)
work = torch.zeros(tot, dtype=torch.uint8, device=dev)
ref, _, work = g.gemm(A, B, N, work=work)
torch.cuda.synchronize()
L = g.Layout(); g.check(g.lib().gemmul8_get_layout(g.D, g.INT8, n, n, n, N, work.data_ptr(), None, None, 0, 0, C.byref(L)))
def view(wk, base, ptr, nb): off = ptr - base; return wk[off:off+nb]
sA_ref = view(work, work.data_ptr(), L.sftA, 2*n).view(torch.int16)
sB_ref = view(work, work.data_ptr(), L.sftB, 2*n).view(torch.int16)
sA = view(eng.work, eng.work.data_ptr(), eng.L.sftA, 2*n).view(torch.int16)
sB = view(eng.work, eng.work.data_ptr(), eng.L.sftB, 2*n).view(torch.int16)
print(rank, "sftA equal", bool(torch.equal(sA, sA_ref)), "sftB equal", bool(torch.equal(sB, sB_ref)), "ndiffB", int((sB != sB_ref).sum()))
# residue planes in recv vs reference C_mid
ncols = plan.c1 - plan.c0
for t in (0, 6, 7, 13):
    mine = plan._recv_slot(t)
    refp = view(work, work.data_ptr(), L.C_mid + (t * L.sizeC + plan.c0 * L.mp), ncols * L.mp)
    print(rank, "plane", t, "equal", bool(torch.equal(mine, refp)))
mycols = Cm[plan.c0:plan.c1]
print(rank, "my C block equal", bool(torch.equal(mycols, ref[plan.c0:plan.c1])))
dist.barrier(); dist.destroy_process_group()
