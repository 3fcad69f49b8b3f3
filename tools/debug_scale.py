"""GPU debug: per-tensor gradient error of the fp32 and bf16 engines vs the CPU oracle at d=768 (1/1/1 layers)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import lxmert_oracle as O
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.params import ParamStore
from xlxmert_amd.trainer import PretrainStep, synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lx = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,1,1").split(",")]
torch.set_num_threads(16)
keys = ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size", "max_position_embeddings",
        "type_vocab_size", "l_layers", "x_layers", "r_layers", "visual_feat_dim", "visual_pos_dim", "num_clusters")
cfg = XLxmertConfig(l_layers=lx[0], x_layers=lx[1], r_layers=lx[2])
oc = O.OracleConfig(**{k: getattr(cfg, k) for k in keys})
sd = O.make_state_dict(oc, 5)
batch = synthetic_batch(cfg, B, 20, 8, seed=11)
t = time.time()
leaf = {k: v.clone().requires_grad_(v.is_floating_point() and k != "vis_emb.weight") for k, v in sd.items()}
leaf["obj_predict_head.out_cluster.weight"] = leaf["vis_emb.weight"]
ref = O.xlxmert_vis_mask_forward(leaf, oc, batch["input_ids"], batch["visual_pos"], batch["attention_mask"],
                                 batch["cluster_ids"], batch["vis_mask"], batch["obj_labels"])
ref["total_loss"].backward()
print("oracle", time.time() - t, "s; losses", ref["obj_loss"].item(), ref["feat_loss"].item(), flush=True)
tot = sum((v.grad.double() ** 2).sum() for k, v in leaf.items() if v.grad is not None and k != "obj_predict_head.out_cluster.weight") ** 0.5
print("oracle grad norm", tot.item())
for dtype in (torch.float32, torch.bfloat16):
    store = ParamStore(cfg, "cuda:0", dtype)
    store.load_named(sd)
    tr = PretrainStep(cfg, B, 20, 64, dtype=dtype, device="cuda:0", store=store)
    dev = {k: v.cuda() for k, v in batch.items()}
    tr.engine.set_inputs(dev["input_ids"], dev["attention_mask"], dev["token_type_ids"], dev["visual_pos"],
                         cluster_ids=dev["cluster_ids"], vis_mask=dev["vis_mask"], obj_labels=dev["obj_labels"])
    losses = tr.engine.vis_mask_forward_backward()
    torch.cuda.synchronize()
    print(dtype, "losses", losses[:2].tolist())
    rows = []
    for k, v in leaf.items():
        if v.grad is None or k == "obj_predict_head.out_cluster.weight":
            continue
        got = store.gview(k).cpu().double()
        rn = v.grad.double().norm().item()
        rel = (got - v.grad.double()).norm().item() / max(rn, 1e-12)
        rows.append((rel, k, rn, got.norm().item()))
    rows.sort(reverse=True)
    for r in rows[:14]:
        print("   rel %.3e  %-70s ref|g| %.3e got|g| %.3e" % r)
    g = store.grad[:store.n_used]
    print("   flat grad norm", g.double().norm().item(), "finite", torch.isfinite(g).all().item())
