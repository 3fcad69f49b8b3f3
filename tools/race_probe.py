"""Is a step's result independent of the relative timing of its streams?  The same four training steps (tiny bf16 model, dropout on, launch
plan) from the same state, N times in one process, with random host-side stalls and filler kernels on the engine's side streams between the
steps: every repetition must give the same losses / gradient norm / parameters up to fp32-atomics noise (1e-6).  A missing stream dependency
shows up as an occasional LARGE deviation (1e-4 .. 1e-3)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.engine import reserve_streams
from xlxmert_amd.trainer import PretrainStep, synthetic_batch

cfg = XLxmertConfig(vocab_size=200, hidden_size=128, num_attention_heads=2, intermediate_size=256, max_position_embeddings=32,
                    visual_feat_dim=64, num_clusters=96, l_layers=2, x_layers=2, r_layers=2)
g = torch.Generator().manual_seed(3)
cents = torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu()
B = 8
batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=40 + i).items()} for i in range(4)]
streams = reserve_streams("cuda:0")
rng = random.Random(int(os.environ.get("SEED", "1")))
PLAN = os.environ.get("PLAN", "1") == "1"
SLABS = os.environ.get("SLABS", "1") == "1"
junk = torch.zeros(1 << 22, device="cuda")


def run(perturb):
    tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda", seed=5, lr=1e-3, total_steps=100, train_dropout=True,
                      plan=PLAN, drop_grads=False)
    tr.set_centroids(cents)
    tr.ops.set_gemm_wgrad_slabs(1 if SLABS else 0)
    out = []
    for i in range(4):
        if perturb:
            for st in streams:
                if rng.random() < 0.7:
                    with torch.cuda.stream(st):
                        torch.cuda._sleep(rng.randrange(10000, 3000000))
                        junk.add_(1.0)
            if rng.random() < 0.5:
                time.sleep(rng.random() * 0.003)
            if rng.random() < 0.3:
                torch.cuda._sleep(rng.randrange(10000, 2000000))
        out.append(tr.step(batches[i]).clone())
        if i == 0:
            tr.sync()
            g1 = tr.store.grad.clone()
            names = {n: (m.offset, m.shape) for n, m in tr.store.index.items()}
    tr.sync()
    return out, tr.store.master.clone(), tr.grad_norm(), g1, names


ref = run(False)
worst = 0.0
for rep in range(int(os.environ.get("REPS", "40"))):
    got = run(rep % 2 == 1)
    dl = max((a - b).abs().max().item() for a, b in zip(ref[0], got[0]))
    dp = (ref[1] - got[1]).abs().max().item()
    dn = abs(ref[2] - got[2]) / ref[2]
    worst = max(worst, dn)
    flag = "  <-- LARGE" if dn > 1e-5 or dp > 1e-6 else ""
    print(f"rep {rep:2d} perturb {rep % 2}: loss dev {dl:.2e}  param dev {dp:.2e}  grad-norm dev {dn:.2e}{flag}")
    if flag and not globals().get("_shown"):
        _shown = True
        d = (ref[3] - got[3]).abs()
        print("   step-1 gradient buffers differ in", int((d > 0).sum()), "elements; max", d.max().item())
        for n, (off, shape) in got[4].items():
            k = 1
            for x in shape:
                k *= x
            dd = d[off:off + k]
            if dd.max().item() > 1e-7 * max(1.0, ref[3][off:off + k].abs().max().item()):
                print(f"     {n:60s} max diff {dd.max().item():.3e} of {ref[3][off:off + k].abs().max().item():.3e}  ({int((dd > 0).sum())} elements)")
print("worst gradient-norm deviation", worst)
