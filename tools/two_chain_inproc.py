"""Schedule probe: do TWO independent half-batch training chains in one process (two PretrainStep objects at bs 128, each with its own
library context and its own set of streams, stepped alternately by one host thread) finish more examples per second than ONE chain
at bs 256?  Not a training mode (two models): it bounds what splitting the step into two pipelined half-batches could give, before
anybody builds the gradient accumulation that would need.
   python tools/two_chain_inproc.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.trainer import PretrainStep, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = XLxmertConfig()
g = torch.Generator().manual_seed(9595)
cents = torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu()


def make(B, stream):
    with torch.cuda.stream(stream):
        tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda:0", seed=9595, total_steps=1000, train_dropout=True,
                          plan=True, drop_grads=True, overlap_optimizer=True)
        tr.set_centroids(cents)
        batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=9595 + i).items()} for i in range(4)]
    return tr, batches


def run(chains, n):
    """chains: [(trainer, batches, stream)]; n steps of each, alternately; examples per second over all chains"""
    for i in range(12):
        for tr, bs, st in chains:
            with torch.cuda.stream(st):
                tr.step(bs[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        for tr, bs, st in chains:
            with torch.cuda.stream(st):
                tr.step(bs[i % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return sum(tr.engine.B for tr, _, _ in chains) * n / dt, dt / n * 1e3


s0 = torch.cuda.Stream()
tr, bs = make(256, s0)
v, ms = run([(tr, bs, s0)], steps)
print(f"one chain  bs 256:        {v:9.1f} examples/s  {ms:7.3f} ms per round")
del tr, bs
torch.cuda.empty_cache()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a, ba = make(128, s1)
v, ms = run([(a, ba, s1)], steps)
print(f"one chain  bs 128:        {v:9.1f} examples/s  {ms:7.3f} ms per round")
b, bb = make(128, s2)
v, ms = run([(a, ba, s1), (b, bb, s2)], steps)
print(f"two chains bs 128 + 128:  {v:9.1f} examples/s  {ms:7.3f} ms per round")
v, ms = run([(a, ba, s1)], steps)
print(f"one chain  bs 128 again:  {v:9.1f} examples/s  {ms:7.3f} ms per round")
