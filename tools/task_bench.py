"""Throughput of the scope-table 'next' rows on one MI355X (synthetic data, bf16, random-init weights):
VQA fine-tune step (BASELINE config 3), 4-step Mask-Predict sampling (config 4), word_mask / matched pretraining steps.
Usage: python tools/task_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import lxmert_oracle as O
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.trainer import PretrainStep, word_rows_of

cfg, oc = XLxmertConfig(), O.OracleConfig()
dev = "cuda"


def timed(fn, n=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def cuda(d):
    return {k: v.cuda() for k, v in d.items()}


for B in (128, 512):
    tr = PretrainStep(cfg, B, 20, 64, device=dev, task="vqa", num_answers=3129, train_dropout=True, total_steps=1000)
    batch = cuda(O.make_vqa_inputs(oc, 3129, 1, B, 20, 8))
    dt = timed(lambda: tr.step(batch))
    print(f"vqa step        bs {B:4d}: {dt * 1e3:7.2f} ms  {B / dt:9.0f} examples/s")
    del tr
for task in ("word_mask", "matched"):
    B = 256
    tr = PretrainStep(cfg, B, 20, 64, device=dev, task=task, train_dropout=True, total_steps=1000)
    g = torch.Generator().manual_seed(0)
    tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
    inp = O.make_inputs(oc, 2, B, 20, 8)
    wl, ml = O.make_lang_task_labels(oc, inp["input_ids"], 3)
    batch = cuda({"input_ids": inp["input_ids"], "visual_pos": inp["visual_pos"], "cluster_ids": inp["cluster_ids"],
                  "word_labels": wl, "matched_labels": ml})
    batch["word_rows"] = word_rows_of(wl)          # from the loader, on the host: decoder + loss on the labelled rows only
    dt = timed(lambda: tr.step(batch))
    print(f"{task:10s} step  bs {B:4d}: {dt * 1e3:7.2f} ms  {B / dt:9.0f} examples/s")
    del tr
from xlxmert_amd.engine import Engine
from xlxmert_amd.ops import HipOps
from xlxmert_amd.params import ParamStore
from xlxmert_amd.trainer import init_reference_weights
for B in (64, 256):
    store = ParamStore(cfg, dev, torch.bfloat16, task="vis_mask")
    init_reference_weights(store, 1)
    g = torch.Generator().manual_seed(0)
    store.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
    eng = Engine(cfg, store, HipOps(torch.bfloat16), B, 20, 64, need_lang=False)
    eng.sync_compute_weights()
    inp = O.make_inputs(oc, 4, B, 20, 8)
    eng.set_inputs(inp["input_ids"].cuda(), inp["attention_mask"].cuda(), None, inp["visual_pos"].cuda(),
                   cluster_ids=torch.zeros(B, 64, dtype=torch.long, device=dev), vis_mask=torch.ones(B, 64, dtype=torch.bool, device=dev))
    dt = timed(lambda: eng.sample_codes_nar(4))
    print(f"sampler T=4     bs {B:4d}: {dt * 1e3:7.2f} ms  {B / dt:9.0f} images/s (codes for the GAN decoder)")
    del eng, store
