"""Time the dense contractions of one training step in isolation (bf16, B=256 shapes).  Usage:
   [XL_GEMM_TILE=128|256] [XL_GEMM_ABLATE=n] python tools/gemm_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xlxmert_amd.ops import HipOps, EPI_NONE, EPI_GELU_DG as EPI_GELU, EPI_RESIDUAL, EPI_MULAUX as EPI_DGELU      # the FFN pair as the step runs it

ops = HipOps(torch.bfloat16)
dev = "cuda"
if os.environ.get("XL_GEMM_SLABS", "1") != "0":
    ops.gemm_workspace(256)          # slab workspace of the current stream: tail split (and, XL_GEMM_WGRAD_SLABS=1, weight gradients)
ML, MV, MX = 5120, 16384, 21504
SHAPES = [  # name, M, N, K, ak, bk, epi, out_f32
    ("vis qkv  NT", MV, 2304, 768, 1, 1, EPI_NONE, False),
    ("x   qkv  NT", MX, 2304, 768, 1, 1, EPI_NONE, False),
    ("lang qkv NT", ML, 2304, 768, 1, 1, EPI_NONE, False),
    ("vis out  NT", MV, 768, 768, 1, 1, EPI_RESIDUAL, False),
    ("lang out NT", ML, 768, 768, 1, 1, EPI_RESIDUAL, False),
    ("vis ffn1 NT", MV, 3072, 768, 1, 1, EPI_GELU, False),
    ("lang ffn1 NT", ML, 3072, 768, 1, 1, EPI_GELU, False),
    ("vis ffn2 NT", MV, 768, 3072, 1, 1, EPI_RESIDUAL, False),
    ("lang ffn2 NT", ML, 768, 3072, 1, 1, EPI_RESIDUAL, False),
    ("visn_fc  NT", MV, 768, 2048, 1, 1, EPI_NONE, False),
    ("feat     NT", MV, 2048, 768, 1, 1, EPI_NONE, False),
    ("logits   NT", MV, 10000, 2048, 1, 1, EPI_NONE, True),
    ("vis dpre NN", MV, 3072, 768, 1, 0, EPI_DGELU, False),
    ("vis dx1  NN", MV, 768, 3072, 1, 0, EPI_RESIDUAL, False),
    ("vis dctx NN", MV, 768, 768, 1, 0, EPI_NONE, False),
    ("vis dxqkv NN", MV, 768, 2304, 1, 0, EPI_RESIDUAL, False),
    ("lang dxqkv NN", ML, 768, 2304, 1, 0, EPI_RESIDUAL, False),
    ("dfeat    NN", MV, 2048, 10000, 1, 0, EPI_RESIDUAL, False),
    ("vis dWo  TN", 768, 768, MV, 0, 0, EPI_NONE, True),
    ("vis dWqkv TN", 2304, 768, MV, 0, 0, EPI_NONE, True),
    ("vis dW1  TN", 3072, 768, MV, 0, 0, EPI_NONE, True),
    ("vis dW2  TN", 768, 3072, MV, 0, 0, EPI_NONE, True),
    ("lang dW1 TN", 3072, 768, ML, 0, 0, EPI_NONE, True),
    ("lang dWo TN", 768, 768, ML, 0, 0, EPI_NONE, True),
    ("x dWqkv  TN", 2304, 768, MX, 0, 0, EPI_NONE, True),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
tot_t = tot_f = 0.0
for name, M, N, K, ak, bk, epi, of32 in SHAPES:
    if only and only not in name:
        continue
    A = torch.randn((M, K) if ak else (K, M), device=dev).to(torch.bfloat16)
    B = torch.randn((N, K) if bk else (K, N), device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if of32 else torch.bfloat16)
    bias = torch.randn(N, device=dev) if not (ak == 0) else None
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == EPI_RESIDUAL else None
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi in (EPI_GELU, EPI_DGELU) else None
    lda, ldb = (K if ak else M), (K if bk else N)
    def run():
        ops.gemm(A, B, C, bias, res, aux, M, N, K, lda, ldb, N, ldr=N, ldx=N, a_kmajor=ak, b_kmajor=bk, out_f32=of32, epilogue=epi)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    s.record()
    for _ in range(reps):
        run()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    tf = 2.0 * M * N * K / us / 1e6
    tot_t += us; tot_f += 2.0 * M * N * K
    print(f"{name:14s} M={M:6d} N={N:6d} K={K:6d}  {us:8.1f} us  {tf:7.1f} TF/s")
print(f"sum {tot_t:.0f} us, {tot_f / tot_t / 1e6:.1f} TF/s aggregate")
