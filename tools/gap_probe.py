import os, sys
sys.path.insert(0, ".")
import torch
from xlxmert_amd.ops import HipOps
M, N, K = 16384, 768, 768
ops = HipOps(torch.bfloat16); ops.set_gemm_pingpong(2)
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); res = torch.randn(M, N, device="cuda").bfloat16()
bias = torch.randn(N, device="cuda")
def run(): ops.gemm(A, B, C, bias, res, None, M, N, K, K, K, N, ldr=N, epilogue=2)
for _ in range(3): run()
torch.cuda.synchronize()
bufs = [torch.zeros(4 * 8192 + 12 * 8192, dtype=torch.int64, device="cuda") for _ in range(4)]
for b in bufs:
    ops.gemm_trace(b); run()
ops.gemm_trace(None)
torch.cuda.synchronize()
ts = [b[:4 * 8192].view(-1, 4).cpu() for b in bufs]
ts = [t[t[:, 0] > 0].double() / 100.0 for t in ts]
for i in range(1, 4):
    print(f"launch {i}: prev last end -> first start {ts[i][:,0].min() - ts[i-1][:,3].max():6.2f} us; prev span {ts[i-1][:,3].max() - ts[i-1][:,0].min():6.2f}; start spread {ts[i][:,0].max() - ts[i][:,0].min():5.2f}")
