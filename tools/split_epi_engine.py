"""bs-64 VQA forward logits under different kernel choices (same weights, same inputs): how far apart do bf16 re-associations land? (debug)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.engine import Engine
from xlxmert_amd.ops import HipOps
from xlxmert_amd.params import ParamStore
from xlxmert_amd.trainer import init_reference_weights
cfg = XLxmertConfig()
L, V, A, B = 20, 64, 3129, 64
store = ParamStore(cfg, "cuda", torch.bfloat16, task="vqa", num_answers=A)
init_reference_weights(store, 7)
g = torch.Generator().manual_seed(3)
lens = torch.randint(8, L + 1, (B,), generator=g)
inp = {"input_ids": torch.randint(1000, 20000, (B, L), generator=g), "attention_mask": (torch.arange(L)[None, :] < lens[:, None]).long(),
       "visual_pos": torch.rand(B, V, 4, generator=g), "visual_feats": torch.randn(B, V, 2048, generator=g).relu()}
outs = {}
for name, setup in [("default", lambda o: None), ("split_epi", lambda o: o.set_gemm_split_epi(1)), ("no_pingpong", lambda o: o.set_gemm_pingpong(0)),
                    ("duo_all", lambda o: o.lib.call("xl_set_gemm_duo", 2)), ("default2", lambda o: None)]:
    ops = HipOps(torch.bfloat16)
    setup(ops)
    eng = Engine(cfg, store, ops, B, L, V, need_lang=True)
    eng.sync_compute_weights()
    eng.set_inputs(inp["input_ids"].cuda(), inp["attention_mask"].cuda(), None, inp["visual_pos"].cuda(), visual_feats=inp["visual_feats"].cuda())
    outs[name] = eng.vqa_forward().float().clone()
    torch.cuda.synchronize()
store32 = ParamStore(cfg, "cuda", torch.float32, task="vqa", num_answers=A)
init_reference_weights(store32, 7)
eng = Engine(cfg, store32, HipOps(torch.float32), B, L, V, need_lang=True)
eng.sync_compute_weights()
eng.set_inputs(inp["input_ids"].cuda(), inp["attention_mask"].cuda(), None, inp["visual_pos"].cuda(), visual_feats=inp["visual_feats"].cuda())
f32 = eng.vqa_forward().float().clone()
torch.cuda.synchronize()
for k, v in outs.items():
    d = (v - f32)
    print(f"{k:12s} against the fp32 path: max {d.abs().max().item():.5f}  rms {d.pow(2).mean().sqrt().item():.6f}")
ref = outs["default"]
for k, v in outs.items():
    print(f"{k:12s} max |diff to default| {(v - ref).abs().max().item():.5f}   (logit scale {ref.abs().max().item():.3f})")
