"""attention core (xl_sdpa_fwd / xl_sdpa_bwd) timed alone on the four shapes of the step."""
import math
import sys
import torch
sys.path.insert(0, ".")
from xlxmert_amd.ops import HipOps
ops = HipOps(torch.bfloat16)
B, H, dh = 256, 12, 64
d = H * dh
for nq, nk, masked in ((64, 64, False), (20, 20, True), (64, 20, True), (20, 64, False)):
    g = torch.Generator().manual_seed(0)
    qkv_q = torch.randn(B * nq, 3 * d, generator=g).bfloat16().cuda()
    qkv_k = torch.randn(B * nk, 3 * d, generator=g).bfloat16().cuda()
    dqkv_q, dqkv_k = torch.zeros_like(qkv_q), torch.zeros_like(qkv_k)
    o = torch.zeros(B * nq, d, dtype=torch.bfloat16, device="cuda"); dout = torch.randn_like(o)
    lse = torch.zeros(B * H * nq, device="cuda")
    km = torch.ones(B, nk, dtype=torch.uint8, device="cuda") if masked else None
    bg, ws = torch.zeros(3 * d, device="cuda"), torch.zeros(ops.workspace_floats(d), device="cuda")
    sc = 1.0 / math.sqrt(dh)
    bits = torch.zeros(ops.sdpa_keep_bits_bytes(B, H, nq, nk, dh) // 4, dtype=torch.int32, device="cuda")
    for pd, kb in ((0.0, None), (0.1, None), (0.1, bits)):         # kb: the forward saves its dropout decisions, the backward tests bits
        def fwd(): ops.sdpa_fwd(qkv_q, qkv_k[:, d:], qkv_k[:, 2 * d:], km, o, lse, B, H, nq, nk, dh, 3 * d, 3 * d, 3 * d, d, sc,
                                p_drop=pd, seed=3, keep_bits=kb)
        def bwd(): ops.sdpa_bwd(qkv_q, qkv_k[:, d:], qkv_k[:, 2 * d:], km, dout, lse, dqkv_q, dqkv_k[:, d:], dqkv_k[:, 2 * d:],
                                B, H, nq, nk, dh, 3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d, sc, p_drop=pd, seed=3, keep_bits=kb)
        def bwdb(): ops.sdpa_bwd(qkv_q, qkv_k[:, d:], qkv_k[:, 2 * d:], km, dout, lse, dqkv_q, dqkv_k[:, d:], dqkv_k[:, 2 * d:],
                                 B, H, nq, nk, dh, 3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d, sc, p_drop=pd, seed=3,
                                 bias_grad=bg, ws=ws, keep_bits=kb)
        label = f"{pd}{' saved bits' if kb is not None else ''}"
        for name, f in (("fwd", fwd), ("bwd", bwd), ("bwd+bias", bwdb)):
            for _ in range(3): f()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): f()
            e.record(); torch.cuda.synchronize()
            print(f"{nq}x{nk} p={label} {name}: {s.elapsed_time(e) / 20 * 1e3:.1f} us")
