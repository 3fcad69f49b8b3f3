"""Reference point for the hand-written GEMM kernels: the same contractions through torch.matmul (hipBLASLt / rocBLAS on ROCm),
plain epilogue on both sides, bf16, isolated launches.  Usage: python tools/blas_compare.py"""
import sys
import torch
sys.path.insert(0, ".")
from xlxmert_amd.ops import HipOps, EPI_NONE

ops = HipOps(torch.bfloat16)
ML, MV, MX = 5120, 16384, 21504
SHAPES = [("vis qkv  NT", MV, 2304, 768, 1), ("x qkv    NT", MX, 2304, 768, 1), ("lang qkv NT", ML, 2304, 768, 1),
          ("vis out  NT", MV, 768, 768, 1), ("vis ffn1 NT", MV, 3072, 768, 1), ("vis ffn2 NT", MV, 768, 3072, 1),
          ("lang ffn2 NT", ML, 768, 3072, 1), ("logits   NT", 8448, 10000, 2048, 1),
          ("vis dx1  NN", MV, 768, 3072, 0), ("vis dpre NN", MV, 3072, 768, 0), ("vis dxqkv NN", MV, 768, 2304, 0),
          ("dfeat    NN", 8448, 2048, 10000, 0)]


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


tx = tb = 0.0
for name, M, N, K, bk in SHAPES:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = torch.randn((N, K) if bk else (K, N), device="cuda").to(torch.bfloat16)
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    Bt = B.t() if bk else B
    us_x = timed(lambda: ops.gemm(A, B, C, None, None, None, M, N, K, K, (K if bk else N), N, b_kmajor=bk, epilogue=EPI_NONE))
    us_b = timed(lambda: torch.matmul(A, Bt, out=C))
    f = 2.0 * M * N * K
    tx += us_x; tb += us_b
    print(f"{name:13s} M={M:6d} N={N:6d} K={K:6d}   xl_gemm {us_x:7.1f} us {f / us_x / 1e6:7.1f} TF/s   torch.matmul {us_b:7.1f} us {f / us_b / 1e6:7.1f} TF/s")
print(f"sum: xl_gemm {tx:.0f} us, torch.matmul {tb:.0f} us")
