"""split-K-with-epilogue launches against the unsplit ones on the shapes the bs-64 VQA engine runs (debug)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlxmert_amd.ops import HipOps
ops = HipOps(torch.bfloat16)
ops.gemm_workspace(256)
g = torch.Generator().manual_seed(1)
for M, N, K, epi in [(4096, 768, 3072, 2), (1024, 768, 3072, 2), (1280, 768, 3072, 2), (4096, 768, 2304, 0), (4096, 768, 3072, 0), (3840, 768, 3072, 2), (4352, 768, 3072, 2)]:
    A = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
    B = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    outs = []
    for mode in (0, 1, 1):
        ops.set_gemm_split_epi(mode)
        C = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
        ops.gemm(A, B, C, bias, res if epi == 2 else None, None, M, N, K, K, K, N, ldr=N, ldx=N, a_kmajor=1, b_kmajor=1, epilogue=epi)
        torch.cuda.synchronize()
        outs.append(C.float())
    d = (outs[0] - outs[1]).abs()
    print(M, N, K, epi, "max diff", d.max().item(), "frac differing", (d > 0).float().mean().item(), "rerun equal", torch.equal(outs[1], outs[2]),
          "worst rows", d.max(1).values.topk(3).indices.tolist())
