"""optimizer-side streams (xl_sumsq, xl_adamw) timed alone at the step's size (202 M parameters)."""
import sys
import torch
sys.path.insert(0, ".")
from xlxmert_amd.ops import HipOps
ops = HipOps(torch.bfloat16)
n = 202_400_000 // 256 * 256
p, g, m, v = (torch.randn(n, device="cuda") * 0.02 for _ in range(4))
v.abs_()
pc = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
flags = torch.ones(n // 256, dtype=torch.uint8, device="cuda")
ss = torch.zeros(1, device="cuda"); lrs = torch.tensor([1e-4, 0.1, 0.001, 0.0], device="cuda")
def adam(): ops.adamw(p, g, m, v, pc, flags, ss, lrs, n, 0.9, 0.999, 1e-6, 0.01, 1.0)
sc = ops.sumsq_scratch("cuda")
def sumsq(): ops.sumsq(g, ss, n, sc)
for name, f, byts in (("adamw", adam, 30 * n), ("sumsq", sumsq, 4 * n)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10
    print(f"{name}: {t * 1e3:.0f} us  {byts / t / 1e9:.2f} TB/s")
