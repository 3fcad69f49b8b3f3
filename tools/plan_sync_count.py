"""Count the stream hand-overs of one recorded training step by kind and stream (tools/sync_cost_probe.py: an event record costs the
recording stream ~3 us, ~5 us when another stream waits for it; a wait for an already-signalled event is free)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.engine import reserve_streams
from xlxmert_amd.trainer import PretrainStep, synthetic_batch

reserve_streams("cuda:0")
cfg = XLxmertConfig()
B = 256
tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda:0", seed=9595, total_steps=100000, train_dropout=True,
                  plan=True, drop_grads=True, overlap_optimizer=True)
g = torch.Generator().manual_seed(9595)
tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=9595 + i).items()} for i in range(2)]
captured = []
orig = tr.ops.lib.make_plan
def make_plan(calls):
    captured.append(list(calls))
    return orig(calls)
tr.ops.lib.make_plan = make_plan
for i in range(4):
    tr.step(batches[i % 2])
torch.cuda.synchronize()
calls = captured[-1]
main = torch.cuda.current_stream().cuda_stream
side, dwv, dwl = [s.cuda_stream for s in reserve_streams("cuda:0")]
names = {main: "main", side: "lang", dwv: "dw_v", dwl: "dw_l"}
cnt = collections.Counter()
kern = collections.Counter()
for name, args in calls:
    if name == "xl_stream_fork":
        cnt[("record", names.get(args[1], hex(args[1] or 0)))] += 1
        cnt[("wait", names.get(args[2], hex(args[2] or 0)))] += 1
        cnt[("fork", names.get(args[1], "?") + "->" + names.get(args[2], "?"))] += 1
    elif name == "xl_event_record":
        cnt[("record", names.get(args[1], hex(args[1] or 0)))] += 1
    elif name == "xl_stream_wait":
        cnt[("wait", names.get(args[1], hex(args[1] or 0)))] += 1
    elif name not in ("xl_ctx_bind", "xl_set_step_seed_ptr", "xl_set_deferred_reduce"):
        kern[names.get(args[-1], "other")] += 1
print(len(calls), "calls in the plan")
for k in sorted(cnt):
    print(f"  {k[0]:7s} {k[1]:14s} {cnt[k]}")
print("launch-type calls per stream:", dict(kern))
