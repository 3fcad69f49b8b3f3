"""Upper bound of what a packed language path (no [PAD] rows) could save: the same step at text length 20 (the contract: lengths
U{6..20}, a third of the B x 20 language rows are padding) and at 13 (the mean real length), same batch otherwise."""
import sys, time
import torch
sys.path.insert(0, ".")
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.trainer import PretrainStep, synthetic_batch
cfg = XLxmertConfig()
B = 256
for L in (20, 13):
    tr = PretrainStep(cfg, B, L, 64, dtype=torch.bfloat16, device="cuda", seed=9595, total_steps=1000, train_dropout=True,
                      plan=True, drop_grads=True, overlap_optimizer=True)
    g = torch.Generator().manual_seed(9595)
    tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
    batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, L, 8, seed=9595 + i).items()} for i in range(4)]
    for i in range(6): tr.step(batches[i % 4])
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(20): tr.step(batches[i % 4])
    torch.cuda.synchronize()
    print(f"text length {L}: {(time.perf_counter() - t) / 20 * 1e3:.2f} ms per step")
    del tr
