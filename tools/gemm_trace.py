"""Per-workgroup timeline of the ping-pong GEMM kernel (xl_gemm_trace): when blocks start, how long prologue / K loop /
epilogue take.  Usage: python tools/gemm_trace.py M N K ak bk epi [out_f32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xlxmert_amd.ops import HipOps

M, N, K, ak, bk, epi = [int(a) for a in sys.argv[1:7]]
of32 = len(sys.argv) > 7 and sys.argv[7] == "1"
ops = HipOps(torch.bfloat16)
ops.set_gemm_pingpong(2)
dev = "cuda"
A = torch.randn((M, K) if ak else (K, M), device=dev).to(torch.bfloat16)
B = torch.randn((N, K) if bk else (K, N), device=dev).to(torch.bfloat16)
C = torch.zeros(M, N, device=dev, dtype=torch.float32 if of32 else torch.bfloat16)
bias = torch.randn(N, device=dev) if ak else None
res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == 2 else None
aux = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi in (1, 3, 6, 7) else None
lda, ldb = (K if ak else M), (K if bk else N)
trace = torch.zeros(4 * 8192 + 12 * 8192, dtype=torch.int64, device=dev)
def run():
    ops.gemm(A, B, C, bias, res, aux, M, N, K, lda, ldb, N, ldr=N, ldx=N, a_kmajor=ak, b_kmajor=bk, out_f32=of32, epilogue=epi)
for _ in range(3):
    run()
torch.cuda.synchronize()
ops.gemm_trace(trace)
run()
torch.cuda.synchronize()
ops.gemm_trace(None)
sec = trace[4 * 8192:].view(-1, 2, 6).cpu()
t = trace[:4 * 8192].view(-1, 4).cpu()
t = t[t[:, 0] > 0].double() / 100.0          # us
t0 = t[:, 0].min()
print(f"{len(t)} workgroups; kernel span {t[:, 3].max() - t0:.1f} us")
order = t[:, 0].argsort()
t = t[order]
for name, a, b in (("start offset", None, 0), ("prologue", 0, 1), ("k loop", 1, 2), ("epilogue", 2, 3)):
    d = (t[:, b] - t0) if a is None else (t[:, b] - t[:, a])
    print(f"  {name:12s} mean {d.mean():7.2f}  min {d.min():7.2f}  p50 {d.median():7.2f}  max {d.max():7.2f} us")
n = len(t)
for lo in range(0, n, 256):
    seg = t[lo:lo + 256]
    print(f"  blocks {lo:4d}..{lo + len(seg) - 1:4d}: start {seg[:, 0].min() - t0:6.1f}..{seg[:, 0].max() - t0:6.1f}  "
          f"end {seg[:, 3].min() - t0:6.1f}..{seg[:, 3].max() - t0:6.1f} us")

if sec.sum() > 0:          # library built with -DXL_PP_PROFILE: per-section cycles of wave 0 (group 0) and wave 4 (group 1)
    nb = len(t)
    nph = 2 * ((K + 63) // 64)
    for gi, name in ((0, "waves 0-3"), (1, "waves 4-7")):
        c = sec[:nb, gi].double().mean(0)
        print(f"  {name}: per phase pair (cycles): L(P0) {c[0] * 2 / nph:6.0f}  L(P1) {c[1] * 2 / nph:6.0f}  vmcnt wait {c[5] / nph:6.0f}  "
              f"barrier-1 wait {c[2] / nph:6.0f}  M {c[3] / nph:6.0f}  barrier-2 wait {c[4] / nph:6.0f}")
