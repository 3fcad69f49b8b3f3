"""xl_embed_bwd timed alone at the step's shape (B=256, L=20, d=768; token ids as the synthetic batches draw them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.ops import HipOps
from xlxmert_amd.trainer import synthetic_batch, word_order_of
cfg = XLxmertConfig()
ops = HipOps(torch.bfloat16)
B, L, N = 256, 20, 768
b = synthetic_batch(cfg, B, L, 8, seed=1)
ids = b["input_ids"].cuda()
order = word_order_of(b["input_ids"]).cuda()
tt = torch.zeros_like(ids)
dpre = torch.randn(B * L, N, device="cuda").bfloat16()
dw = torch.zeros(cfg.vocab_size, N, device="cuda")
dp = torch.zeros(cfg.max_position_embeddings, N, device="cuda")
dt = torch.zeros(2, N, device="cuda")
for name, t, o in (("tt=0, sorted rows", tt, order), ("tt=1 on half, sorted rows", (torch.arange(B * L, device="cuda").view(B, L) % 2).long(), order),
                   ("tt=None, sorted rows", None, order), ("tt=0, no row order (scanning kernel)", tt, None)):
    for _ in range(3):
        ops.embed_bwd(dpre, ids, t, dw, dp, dt, B, L, N, order=o)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.embed_bwd(dpre, ids, t, dw, dp, dt, B, L, N, order=o)
    e.record(); torch.cuda.synchronize()
    print(name, f"{s.elapsed_time(e) / 20 * 1e3:.1f} us per call (scatter + position + type kernels)")
