"""Paired launches (xl_gemm_pair) against their two separate launches, in isolation, on the step's shapes: visual side 16384 rows,
language side 3328 packed rows.  us per repetition: visual alone | language alone | both back to back | paired.
   python tools/pair_bench.py [ML]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xlxmert_amd.ops import HipOps, GemmCall, EPI_NONE, EPI_GELU_DG, EPI_RESIDUAL, EPI_MULAUX

ops = HipOps(torch.bfloat16)
dev = "cuda"
MV = 16384
ML = int(sys.argv[1]) if len(sys.argv) > 1 else 3328
SHAPES = [  # name, N, K, bk, epi, p_drop, colsum
    ("qkv      NT", 2304, 768, 1, EPI_NONE, 0.0, False),
    ("out      NT", 768, 768, 1, EPI_RESIDUAL, 0.1, False),
    ("ffn1     NT", 3072, 768, 1, EPI_GELU_DG, 0.0, False),
    ("ffn2     NT", 768, 3072, 1, EPI_RESIDUAL, 0.1, False),
    ("dctx     NN", 768, 768, 0, EPI_NONE, 0.0, False),
    ("dxqkv    NN", 768, 2304, 0, EPI_RESIDUAL, 0.0, False),
    ("dpre     NN", 3072, 768, 0, EPI_MULAUX, 0.0, True),
    ("dx1      NN", 768, 3072, 0, EPI_RESIDUAL, 0.0, False),
]


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


tot = [0.0, 0.0, 0.0, 0.0]
for name, N, K, bk, epi, p_drop, colsum in SHAPES:
    calls = []
    for M in (MV, ML):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        B = (torch.randn((N, K) if bk else (K, N), device=dev) * 0.05).to(torch.bfloat16)
        C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(N, device=dev) if bk else None
        res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == EPI_RESIDUAL else None
        aux = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi in (EPI_GELU_DG, EPI_MULAUX) else None
        cs = torch.zeros(N, device=dev) if colsum else None
        ws = torch.zeros(ops.workspace_floats(N), device=dev) if colsum else None
        calls.append(GemmCall(A, B, C, bias, res, aux, M, N, K, K, K if bk else N, N, ldr=N, ldx=N, a_kmajor=1, b_kmajor=bk,
                              epilogue=epi, p_drop=p_drop, seed=5, colsum=cs, ws=ws))
    tv = timed(lambda: ops.gemm(*calls[0].a, **calls[0].kw))
    tl = timed(lambda: ops.gemm(*calls[1].a, **calls[1].kw))
    tb = timed(lambda: (ops.gemm(*calls[0].a, **calls[0].kw), ops.gemm(*calls[1].a, **calls[1].kw)))
    tp = timed(lambda: ops.gemm_pair(calls[0], calls[1]))
    for i, t in enumerate((tv, tl, tb, tp)):
        tot[i] += t
    fl = 2.0 * (MV + ML) * N * K
    print(f"{name} N={N:5d} K={K:5d}  vis {tv:7.1f}  lang {tl:7.1f}  both {tb:7.1f}  pair {tp:7.1f} us  ({fl / tp / 1e6:6.1f} TF/s paired)")
print(f"sum: vis {tot[0]:.0f}  lang {tot[1]:.0f}  both {tot[2]:.0f}  pair {tot[3]:.0f} us")
