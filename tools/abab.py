"""In-process A/B/A/B of two library configurations on the plan-replayed bs-256 training step: same lease, same process, same trainer,
alternating arms (default 5 alternations), every arm measured after a settling run; paired deltas (B - mean of the neighbouring A's)
with their spread.  The ONLY accepted evidence for switching a kernel-choice default (VERDICT r05 item 3 ii): boxes differ by 1 ms and
drift by up to 1 ms within minutes, so numbers from different gpurun calls -- or from one call without alternation -- are not comparable.

    python tools/abab.py --b relay=1                      # A = library defaults, B = xl_set_gemm_relay(1)
    python tools/abab.py --a tile192=0 --b tile192=1 --alternations 6 --steps 40
Setter names: HipOps.set_gemm_<name> (relay, relay_wgs, q, duo, persistent, tile192, pingpong, split_epi, pair, wgrad_slabs), engine
attributes (keep_bits: the attention forward saves its dropout decisions for the backward; last_ffn: the last cross layer's visual
feed-forward block on the masked rows only), or env:NAME=value
for switches the library reads from the environment at first use (only effective for contexts created afterwards: not supported here).
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.engine import reserve_streams
from xlxmert_amd.trainer import PretrainStep, synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--a", default="", help="comma-separated name=value setters of arm A (default: library defaults)")
ap.add_argument("--b", required=True)
ap.add_argument("--alternations", type=int, default=5)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--settle", type=int, default=20)
args = ap.parse_args()

reserve_streams("cuda:0")
cfg = XLxmertConfig()
B = 256
tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda:0", seed=9595, total_steps=100000, train_dropout=True,
                  plan=True, drop_grads=True, overlap_optimizer=True)
g = torch.Generator().manual_seed(9595)
tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=9595 + i).items()} for i in range(4)]


def parse(spec):
    out = []
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        out.append((k, int(v)))
    return out


A, Bs = parse(args.a), parse(args.b)
names = sorted({k for k, _ in A} | {k for k, _ in Bs})
ENGINE_ATTRS = {"keep_bits": ("use_keep_bits", True), "last_ffn": ("compact_last_ffn", True)}       # engine attributes read at every launch (name -> (attribute, default))
DEFAULTS = {"relay": 0, "relay_wgs": 256, "q": 0, "duo": 1, "persistent": 0, "tile192": 0, "pingpong": 1, "split_epi": 0, "pair": 1, "wgrad_slabs": 0}


def apply(arm):
    vals = {k: (ENGINE_ATTRS[k][1] if k in ENGINE_ATTRS else DEFAULTS[k]) for k in names}
    vals.update(dict(arm))
    for k, v in vals.items():
        if k in ENGINE_ATTRS:
            setattr(tr.engine, ENGINE_ATTRS[k][0], bool(v))
        else:
            getattr(tr.ops, "set_gemm_" + k)(v)
    tr._plans.clear()                   # the kernel choice is frozen in a recorded plan


def run(arm):
    apply(arm)
    for i in range(8 + args.settle):
        tr.step(batches[i % 4])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(args.steps):
        tr.step(batches[i % 4])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / args.steps * 1e3


a_ms, b_ms = [run(A)], []
for i in range(args.alternations):
    b_ms.append(run(Bs))
    a_ms.append(run(A))
    print(f"  A {a_ms[-2]:7.3f}   B {b_ms[-1]:7.3f}   A {a_ms[-1]:7.3f}   paired delta {b_ms[-1] - 0.5 * (a_ms[-2] + a_ms[-1]):+.3f} ms", flush=True)
deltas = [b_ms[i] - 0.5 * (a_ms[i] + a_ms[i + 1]) for i in range(len(b_ms))]
out = {"A": args.a or "defaults", "B": args.b, "a_ms": [round(x, 3) for x in a_ms], "b_ms": [round(x, 3) for x in b_ms],
       "paired_delta_ms": [round(d, 3) for d in deltas], "mean_delta_ms": round(statistics.mean(deltas), 3),
       "stdev_delta_ms": round(statistics.stdev(deltas), 3) if len(deltas) > 1 else None,
       "a_drift_ms": round(max(a_ms) - min(a_ms), 3), "steps_per_arm": args.steps, "settle_steps": args.settle}
print(json.dumps(out))
