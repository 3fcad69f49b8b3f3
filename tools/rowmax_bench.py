import sys, torch
sys.path.insert(0, ".")
from xlxmert_amd.ops import HipOps
ops = HipOps(torch.bfloat16)
M, K, N, Np = 16384, 2048, 10000, 10240
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(Np, K, device="cuda").bfloat16()
bias = torch.randn(Np, device="cuda"); C = torch.zeros(M, N, device="cuda")
ws = torch.zeros((Np // 64) * M * 4, device="cuda")
prob, arg, lse = torch.zeros(M, device="cuda"), torch.zeros(M, dtype=torch.int32, device="cuda"), torch.zeros(M, device="cuda")
def t(f, name):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    print(f"{name}: {s.elapsed_time(e) / 10 * 1e3:.1f} us")
t(lambda: ops.gemm(A, B, C, bias, None, None, M, N, K, K, K, N, out_f32=True), "logits gemm fp32 out")
t(lambda: ops.ce_fwd_bwd(C, None, None, None, None, lse, arg, prob, M, N, N, N, 1.0), "ce argmax pass")
t(lambda: ops.gemm(A, B, None, bias, None, ws, M, Np, K, K, K, Np, epilogue=5), "rowmax gemm")
t(lambda: ops.rowmax_combine(ws, Np // 64, M, prob, arg, lse), "rowmax combine")
