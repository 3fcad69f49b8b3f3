"""Where would a faster kernel family move the STEP?  Upper bounds by ablation: the same plan-replayed bs-256 step with one family of
launches left out (its consumers read what the warm-up steps left in the buffers, so the data stay realistic; the results are
meaningless, the timing is what is measured).  A family whose removal shortens the step by x ms cannot give back more than x.

    python tools/sensitivity.py [--steps 40] [--only name,name]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.engine import reserve_streams
from xlxmert_amd.trainer import PretrainStep, synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--only", default="")
ap.add_argument("--settle", type=int, default=20)
ap.add_argument("--max-drift", type=float, default=0.15)
args = ap.parse_args()

reserve_streams("cuda:0")
cfg = XLxmertConfig()
B = 256
tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda:0", seed=9595, total_steps=100000, train_dropout=True,
                  plan=True, drop_grads=True, overlap_optimizer=True)
g = torch.Generator().manual_seed(9595)
tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=9595 + i).items()} for i in range(4)]
ops = tr.ops


def lang(tag):
    return tag.startswith("l") or (tag.startswith("x") and tag.endswith("l"))


def vis(tag):
    return tag.startswith("r") or (tag.startswith("x") and tag.endswith("v"))


def cross(tag):
    return tag.startswith("x") and tag[1:].isdigit()


# name -> predicate(op name, block tag, positional args) -> skip?
V = {
    "base": lambda op, tag, a: False,
    "no_lang_deepK_gemm": lambda op, tag, a: op == "gemm" and lang(tag) and a[8] >= 2304 and a[7] == 768,
    "no_lang_gemm": lambda op, tag, a: op == "gemm" and lang(tag),
    "no_lang_anything": lambda op, tag, a: lang(tag) and op in ("gemm", "sdpa_fwd", "sdpa_bwd", "layernorm_fwd", "layernorm_bwd", "gemm_wgrad_group"),
    "no_lang_small": lambda op, tag, a: lang(tag) and "+" not in tag and op in ("sdpa_fwd", "sdpa_bwd", "layernorm_fwd", "layernorm_bwd"),
    "no_lang_small_paired": lambda op, tag, a: lang(tag) and "+" not in tag and op in ("sdpa_fwd", "sdpa_bwd", "layernorm_fwd", "layernorm_bwd")
    and not (tag.startswith("l") and int(tag[1:]) < 4),
    "no_lang_sdpa_paired": lambda op, tag, a: lang(tag) and "+" not in tag and op in ("sdpa_fwd", "sdpa_bwd") and not (tag.startswith("l") and int(tag[1:]) < 4),
    "no_lang_ln_paired": lambda op, tag, a: lang(tag) and "+" not in tag and op in ("layernorm_fwd", "layernorm_bwd") and not (tag.startswith("l") and int(tag[1:]) < 4),
    "no_langstack_gemm": lambda op, tag, a: op == "gemm" and tag.startswith("l"),
    "no_langstack_anything": lambda op, tag, a: tag.startswith("l") and op in ("gemm", "sdpa_fwd", "sdpa_bwd", "layernorm_fwd", "layernorm_bwd", "gemm_wgrad_group"),
    "no_xlang_gemm": lambda op, tag, a: op == "gemm" and tag.startswith("x") and tag.endswith("l"),
    "no_embed_bwd": lambda op, tag, a: op == "embed_bwd",
    "no_lang_wgrad": lambda op, tag, a: op == "gemm_wgrad_group" and lang(tag),
    "no_vis_wgrad": lambda op, tag, a: op == "gemm_wgrad_group" and not lang(tag),
    "no_ln_fwd": lambda op, tag, a: op == "layernorm_fwd",
    "no_ln_bwd": lambda op, tag, a: op == "layernorm_bwd",
    "no_sdpa_fwd": lambda op, tag, a: op == "sdpa_fwd",
    "no_sdpa_bwd": lambda op, tag, a: op == "sdpa_bwd",
    "no_adamw": lambda op, tag, a: op in ("adamw", "sumsq"),
    "no_head": lambda op, tag, a: tag == "head" and op in ("gemm", "ce_fwd_bwd", "colsum", "layernorm_fwd", "layernorm_bwd", "gelu_bwd", "gemm_wgrad_group"),
    "no_ce": lambda op, tag, a: op == "ce_fwd_bwd",
    "no_cross_gemm": lambda op, tag, a: op == "gemm" and cross(tag),
    "no_vis_ffn_gemm": lambda op, tag, a: op == "gemm" and vis(tag) and (a[7] == 3072 or a[8] == 3072),
    "no_vis_gemm": lambda op, tag, a: op == "gemm" and vis(tag),
    "no_gather_scatter_dropout": lambda op, tag, a: op in ("gather_rows", "scatter_rows", "dropout", "codebook_gather"),
}
names = [n for n in V if not args.only or n in args.only.split(",") or n == "base"]
orig = {}
for op in ("gemm", "sdpa_fwd", "sdpa_bwd", "layernorm_fwd", "layernorm_bwd", "gemm_wgrad_group", "adamw", "sumsq", "ce_fwd_bwd", "colsum",
           "gelu_bwd", "gather_rows", "scatter_rows", "dropout", "codebook_gather", "embed_bwd"):
    orig[op] = getattr(ops, op)
cur = {"pred": V["base"], "skipped": 0}


def wrap(op):
    f = orig[op]

    def w(*a, **kw):
        if cur["pred"](op, getattr(ops, "block", "") or "", a):
            cur["skipped"] += 1
            return
        return f(*a, **kw)
    return w


for op in orig:
    setattr(ops, op, wrap(op))


# An ablated step trains on garbage: after a few optimizer passes the parameters hold NaN / zeros, every later contraction runs on
# degenerate operand bits, draws less power and clocks higher -- round 5's "1.08 ms drift inside one run" (and the 1.25 ms step in
# profiles/r06b/sensitivity.txt right after no_sdpa_fwd) was exactly that, not the box.  Every run therefore starts from a snapshot.
st = tr.store
SNAP = {k: getattr(st, k).clone() for k in ("master", "compute", "exp_avg", "exp_avg_sq") if getattr(st, k, None) is not None}
SNAP_STEP = tr.step_dev.clone() if hasattr(tr, "step_dev") else None


def restore():
    for k, v in SNAP.items():
        getattr(st, k).copy_(v)
    if SNAP_STEP is not None:
        tr.step_dev.copy_(SNAP_STEP)


def run(name):
    cur["pred"], cur["skipped"] = V[name], 0
    restore()
    tr._plans.clear()
    for i in range(8):                  # re-record every masked-row geometry of the four batches
        tr.step(batches[i % 4])
    skipped = cur["skipped"] // 8
    for i in range(args.settle):        # the power controller needs ~17 steps after any pause to settle (bench.py ms_per_step_series)
        tr.step(batches[i % 4])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(args.steps):
        tr.step(batches[i % 4])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / args.steps * 1e3, skipped


# Every arm is BRACKETED by two base runs and its delta is taken against the mean of its own bracket; an arm whose two brackets differ by
# more than --max-drift ms is flagged (round 5's table compared every arm with ONE base run taken minutes earlier: the box drifted by
# 1 ms inside the run and half the rows were drift, not signal).
res = {}
arms = [n for n in names if n != "base"]
prev, _ = run("base")
print(f"{'base':28s} {prev:7.3f} ms", flush=True)
for name in arms:
    ms, sk = run(name)
    nxt, _ = run("base")
    drift = nxt - prev
    ref = 0.5 * (prev + nxt)
    ok = abs(drift) <= args.max_drift
    res[name] = {"ms": round(ms, 3), "base_before": round(prev, 3), "base_after": round(nxt, 3), "delta_ms": round(ms - ref, 3),
                 "bracket_drift_ms": round(drift, 3), "accepted": ok, "launches_skipped_per_step": sk}
    print(f"{name:28s} {ms:7.3f} ms  ({ms - ref:+.3f} vs its bracket {prev:.3f} / {nxt:.3f}{'' if ok else '  REJECTED: drift'})  skipped/step {sk}", flush=True)
    prev = nxt
print(json.dumps(res))
