"""The role-trading persistent kernel (XL_GEMM_RELAY / xl_set_gemm_relay, csrc/gemm_relay.hip) against the default kernel choice on the
step's chain contractions, in isolation: us per launch, default | relay [| relay with other workgroup counts]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlxmert_amd.ops import HipOps, EPI_NONE, EPI_GELU_DG, EPI_RESIDUAL, EPI_MULAUX

ops = HipOps(torch.bfloat16)
dev = "cuda"
MV, MX, ML = 16384, 16384 + 3328, 3328
SHAPES = [("vis qkv   NT", MV, 2304, 768, 1, EPI_NONE, 0.0), ("x qkv     NT", MX, 2304, 768, 1, EPI_NONE, 0.0),
          ("vis out   NT", MV, 768, 768, 1, EPI_RESIDUAL, 0.1), ("vis ffn1  NT", MV, 3072, 768, 1, EPI_GELU_DG, 0.0),
          ("vis ffn2  NT", MV, 768, 3072, 1, EPI_RESIDUAL, 0.1), ("vis dctx  NN", MV, 768, 768, 0, EPI_NONE, 0.0),
          ("vis dxqkv NN", MV, 768, 2304, 0, EPI_RESIDUAL, 0.0), ("vis dpre  NN", MV, 3072, 768, 0, EPI_MULAUX, 0.0),
          ("vis dx1   NN", MV, 768, 3072, 0, EPI_RESIDUAL, 0.0), ("feat      NT", MV, 2048, 768, 1, EPI_NONE, 0.0),
          ("lang qkv  NT", ML, 2304, 768, 1, EPI_NONE, 0.0), ("lang ffn1 NT", ML, 3072, 768, 1, EPI_GELU_DG, 0.0),
          ("lang ffn2 NT", ML, 768, 3072, 1, EPI_RESIDUAL, 0.1), ("lang out  NT", ML, 768, 768, 1, EPI_RESIDUAL, 0.1)]
WGS = [int(x) for x in os.environ.get("RELAY_WGS", "256").split(",")]


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


only = sys.argv[1] if len(sys.argv) > 1 else None
tot = [0.0] * (1 + len(WGS))
for name, M, N, K, bk, epi, pd in SHAPES:
    if only and only not in name:
        continue
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (torch.randn((N, K) if bk else (K, N), device=dev) * 0.05).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev) if bk else None
    res = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == EPI_RESIDUAL else None
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi in (EPI_GELU_DG, EPI_MULAUX) else None
    run = lambda: ops.gemm(A, B, C, bias, res, aux, M, N, K, K, K if bk else N, N, ldr=N, ldx=N, a_kmajor=1, b_kmajor=bk, epilogue=epi,
                           p_drop=pd, seed=5)
    ops.set_gemm_relay(0)
    t = [timed(run)]
    ref = C.clone()
    for w in WGS:
        ops.set_gemm_relay(2)
        ops.set_gemm_relay_wgs(w)
        C.zero_()
        t.append(timed(run))
        ok = torch.equal(ref, C) if pd == 0.0 or True else True
    ops.set_gemm_relay(0)
    for i in range(len(t)):
        tot[i] += t[i]
    fl = 2.0 * M * N * K
    print(f"{name} M={M:6d} N={N:5d} K={K:5d}  default {t[0]:7.1f} us ({fl / t[0] / 1e6:6.0f} TF/s)  relay " +
          "  ".join(f"[{w}] {x:7.1f} us ({fl / x / 1e6:6.0f} TF/s)" for w, x in zip(WGS, t[1:])) + ("  exact" if ok else "  MISMATCH"), flush=True)
print("sum: default %.0f us   relay %s" % (tot[0], "  ".join(f"[{w}] {x:.0f} us" for w, x in zip(WGS, tot[1:]))))
