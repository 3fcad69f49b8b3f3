"""eager vs launch-plan steps: per-step loss / gradient norm, and the host time of one step's enqueue on an idle GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.trainer import PretrainStep, synthetic_batch
cfg = XLxmertConfig()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(1)
mask_feat = torch.randn(cfg.visual_feat_dim, generator=g).relu() * 0.1
cent = torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu()
batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=50 + i).items()} for i in range(3)]
for mode in ("eager", "plan"):
    tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda", seed=3, lr=1e-4, total_steps=100,
                      train_dropout=True, plan=(mode == "plan"))
    tr.store.view("mask_feat").copy_(mask_feat)
    tr.set_centroids(cent)
    out = []
    for t in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        l = tr.step(batches[t % 3])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append((l[0].item(), tr.grad_norm(), (t1 - t0) * 1e3, (t2 - t0) * 1e3, tr.engine.n_mrows))
    print(mode)
    for t, (l, n, h, w, m) in enumerate(out):
        print(f"  step {t}: loss {l:.5f}  |g| {n:.4f}  host {h:.2f} ms  wall {w:.2f} ms  rows {m}")
