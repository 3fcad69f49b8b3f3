"""Section cycle counters of the relay kernel (library built with -DXL_RELAY_PROFILE: tools/build_relay_prof.sh, run with
XL_LIB=xlxmert_amd/libxlxmert_hip_prof.so).  Usage: python tools/relay_trace.py M N K bk epi"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xlxmert_amd.ops import HipOps

M, N, K, bk, epi = [int(a) for a in sys.argv[1:6]]
ops = HipOps(torch.bfloat16)
ops.set_gemm_relay(2)
dev = "cuda"
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
B = (torch.randn((N, K) if bk else (K, N), device=dev) * 0.05).to(torch.bfloat16)
C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.randn(N, device=dev) if bk else None
res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == 2 else None
aux = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi in (6, 7) else None
trace = torch.zeros(4 * 8192 + 12 * 8192, dtype=torch.int64, device=dev)
def run():
    ops.gemm(A, B, C, bias, res, aux, M, N, K, K, K if bk else N, N, ldr=N, ldx=N, a_kmajor=1, b_kmajor=bk, epilogue=epi, p_drop=0.1 if epi == 2 else 0.0, seed=5)
for _ in range(3):
    run()
torch.cuda.synchronize()
ops.gemm_trace(trace)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); run(); e.record()
torch.cuda.synchronize()
ops.gemm_trace(None)
print(f"launch {s.elapsed_time(e) * 1e3:.1f} us (instrumented)")
t = trace[:256 * 16].view(256, 2, 8).cpu().double()
t = t[t[:, 0, 2] + t[:, 1, 2] > 0]
for g in (0, 1):
    c = t[:, g]
    st = c[:, 2].clamp(min=1)
    print(f"group {g}: compute stages/wg {c[:, 2].mean():.1f}: cycles per stage {(c[:, 0] / st).mean():.0f} (barrier wait {(c[:, 1] / st).mean():.0f})")
    sup = c[:, 3]
    n_iv = (sup > 0).float().sum()
    print(f"         support per wg: total {c[:, 3].mean():.0f} cycles = DMA issue {c[:, 4].mean():.0f} + vmcnt wait {c[:, 5].mean():.0f} + chunk {c[:, 6].mean():.0f} + barrier wait {c[:, 7].mean():.0f}")
