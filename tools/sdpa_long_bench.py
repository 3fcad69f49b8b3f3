"""Long-text attention (nq = nk = L in 65..512, language self-attention shape; and L x 64 / 64 x L cross shapes): the MFMA kernels
(default) against the plain long-sequence fallback (XL_SDPA_LONG_MFMA=0: run the script twice), us per launch at B = 64, H = 12."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlxmert_amd.ops import HipOps
ops = HipOps(torch.bfloat16)
B, H, dh = 64, 12, 64
d = H * dh
print("XL_SDPA_LONG_MFMA =", os.environ.get("XL_SDPA_LONG_MFMA", "1"))
for nq, nk in ((128, 128), (256, 256), (512, 512), (128, 64), (64, 128)):
    g = torch.Generator().manual_seed(0)
    qkv_q = torch.randn(B * nq, 3 * d, generator=g).bfloat16().cuda()
    qkv_k = torch.randn(B * nk, 3 * d, generator=g).bfloat16().cuda()
    dqkv_q, dqkv_k = torch.zeros_like(qkv_q), torch.zeros_like(qkv_k)
    o = torch.zeros(B * nq, d, dtype=torch.bfloat16, device="cuda"); dout = torch.randn_like(o)
    lse = torch.zeros(B * H * nq, device="cuda")
    ws = torch.zeros(ops.workspace_floats(d), device="cuda")
    sc = 1.0 / math.sqrt(dh)
    def fwd(): ops.sdpa_fwd(qkv_q, qkv_k[:, d:], qkv_k[:, 2 * d:], None, o, lse, B, H, nq, nk, dh, 3 * d, 3 * d, 3 * d, d, sc, p_drop=0.1, seed=3)
    def bwd(): ops.sdpa_bwd(qkv_q, qkv_k[:, d:], qkv_k[:, 2 * d:], None, dout, lse, dqkv_q, dqkv_k[:, d:], dqkv_k[:, 2 * d:],
                            B, H, nq, nk, dh, 3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d, sc, p_drop=0.1, seed=3, ws=ws)
    for name, f in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(2): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): f()
        e.record(); torch.cuda.synchronize()
        fl = (4 if name == "fwd" else 12) * B * H * nq * nk * dh          # (bwd: 2 recomputed + 4 gradient products + the second q-side pass)
        print(f"{nq}x{nk} {name}: {s.elapsed_time(e) / 5 * 1e3:9.1f} us  ({fl / (s.elapsed_time(e) / 5 * 1e-3) / 1e12:6.1f} TFLOP/s)")
