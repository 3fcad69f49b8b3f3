"""Drop-in API surface on the GPU: reference class / kwarg / attribute / state-dict names over the HIP engine."""
import pytest
import torch
import torch.nn.functional as F

import lxmert_oracle as O
from _util import golden_cfg, golden_inputs, load_golden, maxdiff

pytestmark = pytest.mark.gpu
CFG_KEYS = ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size", "max_position_embeddings",
            "type_vocab_size", "l_layers", "x_layers", "r_layers", "visual_feat_dim", "visual_pos_dim", "num_clusters")


def make_model(g, dtype=torch.float32):
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.modeling import XLxmertForPretraining
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in CFG_KEYS})
    sd = O.make_state_dict(oc, int(g["seed"]))
    m = XLxmertForPretraining(cfg, device="cuda", dtype=dtype)
    ckpt = {"module." + k: v for k, v in sd.items()}            # published checkpoints carry the DDP prefix
    missing, unexpected = m.load_state_dict(ckpt)
    assert not unexpected, unexpected
    m.eval()
    return m, oc, sd


def test_pretraining_wrapper_forward_backward_and_state_dict():
    g = load_golden("tiny_222")
    m, oc, sd = make_model(g)
    inp = {k: v.cuda() for k, v in golden_inputs(g).items()}
    keys = set(m.state_dict().keys())
    assert {"mask_feat", "vis_emb.weight", "bert.embeddings.word_embeddings.weight", "bert.encoder.visn_fc.box_fc.weight",
            "bert.encoder.layer.0.attention.self.query.weight", "bert.encoder.x_layers.1.visual_attention.att.key.bias",
            "bert.encoder.r_layers.1.output.LayerNorm.weight", "bert.pooler.dense.weight",
            "obj_predict_head.transform.dense.weight", "obj_predict_head.out_cluster.weight",
            "obj_predict_head.out_cluster.bias"} <= keys
    assert m.obj_predict_head.out_cluster.weight is m.vis_emb.weight and not m.vis_emb.weight.requires_grad
    names = {n for n, _ in m.named_parameters()}
    assert "bert.encoder.layer.1.intermediate.dense.weight" in names and "mask_feat" in names
    m.zero_grad()
    out = m(input_ids=inp["input_ids"], visual_pos=inp["visual_pos"], attention_mask=inp["attention_mask"],
            cluster_ids=inp["cluster_ids"], vis_mask=inp["vis_mask"], token_type_ids=inp["token_type_ids"],
            return_dict=True, label_dict={"obj_labels": inp["obj_labels"], "feat_labels": m.vis_emb(inp["cluster_ids"])},
            task="vis_mask")
    assert set(out) == {"obj_loss", "feat_loss", "vis_loss", "total_loss"}
    assert abs(out["obj_loss"].item() - g["obj_loss"].item()) < 1e-4
    assert abs(out["total_loss"].item() - g["total_loss"].item()) < 1e-4
    out["total_loss"].backward()
    params = dict(m.named_parameters())
    for k in [str(n) for n in g["grad_names"]]:
        assert maxdiff(params[k].grad.cpu(), g["grad:" + k]) < 1e-4, k


def test_lxmert_model_and_head_through_autograd():
    """`.bert(...)` -> `(lang, vis, pooled)` and `.obj_predict_head(vis)` -> {'feat','obj'}, losses in plain torch."""
    g = load_golden("tiny_222")
    m, oc, sd = make_model(g)
    inp = {k: v.cuda() for k, v in golden_inputs(g).items()}
    feats = m.vis_emb(inp["cluster_ids"])
    B, V, _ = feats.shape
    feats = torch.where(inp["vis_mask"].view(B, V, 1), m.mask_feat.detach().view(1, 1, -1), feats)
    with pytest.raises(ValueError, match="visual_pos"):
        m.bert(input_ids=inp["input_ids"], visual_feats=feats)
    m.zero_grad()
    out = m.bert(input_ids=inp["input_ids"], visual_feats=feats, visual_pos=inp["visual_pos"],
                 attention_mask=inp["attention_mask"], token_type_ids=inp["token_type_ids"], return_dict=True)
    lang, vis, pooled = out[0], out[1], out[2]
    assert out.vision_output is vis and out.pooled_output is pooled
    real = golden_inputs(g)["attention_mask"].reshape(-1)
    assert maxdiff(lang.detach().cpu().reshape(len(real), -1)[real], torch.from_numpy(g["lang"]).reshape(len(real), -1)[real]) < 1e-4
    assert maxdiff(vis.detach().cpu(), g["vis"]) < 1e-4 and maxdiff(pooled.detach().cpu(), g["pooled"]) < 1e-4
    head = m.obj_predict_head(vis, out_keys=["obj", "feat"])
    assert maxdiff(head["obj"].detach().cpu(), g["obj"]) < 1e-3 and maxdiff(head["feat"].detach().cpu(), g["feat"]) < 1e-4
    obj_loss = F.cross_entropy(head["obj"].view(B * V, -1), inp["obj_labels"].flatten())
    fl = F.smooth_l1_loss(head["feat"], m.vis_emb(inp["cluster_ids"]), reduction="none").mean(2)
    fl = ((fl * inp["vis_mask"]).sum(1) / inp["vis_mask"].sum(1).clamp(min=1)).mean()
    assert abs(obj_loss.item() - g["obj_loss"].item()) < 1e-4 and abs(fl.item() - g["feat_loss"].item()) < 1e-4
    (obj_loss + fl).backward()
    params = dict(m.named_parameters())
    for k in [str(n) for n in g["grad_names"]]:
        if k == "mask_feat":
            continue                      # visual_feats were built outside the module in this test
        assert maxdiff(params[k].grad.cpu(), g["grad:" + k]) < 1e-4, k


def test_vqa_model_dropin_matches_reference_fixture():
    """the nn.Module surface of tasks/vqa_model.py -- reference-layout state dict in, {'logit'} out, and the
    reference's own training idiom (BCEWithLogitsLoss on the logit, .backward(), param.grad) gives the fixture's gradients."""
    import lxmert_oracle as O
    from _util import golden_cfg, golden_inputs, load_golden, maxdiff
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.modeling import VQAModel
    g = load_golden("vqa_tiny")
    oc = golden_cfg(g)
    A = int(g["num_answers"])
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "l_layers", "x_layers",
                                                      "r_layers", "visual_feat_dim", "visual_pos_dim", "num_clusters")})
    m = VQAModel(cfg, A, dtype=torch.float32).eval()
    sd = O.make_vqa_state_dict(oc, A, int(g["seed"]))
    missing, _ = m.load_state_dict({"module." + k: v for k, v in sd.items()})
    assert not missing
    inp = {k: v.cuda() for k, v in golden_inputs(g).items()}
    m.zero_grad()
    out = m(input_ids=inp["input_ids"], visual_feats=inp["visual_feats"], visual_pos=inp["visual_pos"],
            attention_mask=inp["input_ids"] > 0)
    assert maxdiff(out["logit"].cpu(), g["logit"]) < 1e-4
    loss = torch.nn.BCEWithLogitsLoss()(out["logit"], inp["targets"])
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    for k in [str(n) for n in g["grad_names"]]:
        ref = torch.from_numpy(g["grad:" + k])
        assert maxdiff(params[k].grad.cpu(), ref) <= 1e-4 * max(1.0, ref.abs().max().item()), k
    assert set(m.state_dict().keys()) >= {"answer_head.logit_fc.3.bias", "bert.pooler.dense.weight"}


def test_sample_codes_dropin_matches_reference_fixture():
    """SURVEY 8f N2 through the module surface: XLxmertForPretraining.sample_codes == the reference loop's final codes."""
    import lxmert_oracle as O
    from _util import golden_cfg, load_golden, maxdiff
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.modeling import XLxmertForPretraining
    g = load_golden("sampler_tiny")
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "l_layers", "x_layers",
                                                      "r_layers", "visual_feat_dim", "visual_pos_dim", "num_clusters")})
    m = XLxmertForPretraining(cfg, dtype=torch.float32)
    m.load_state_dict(O.make_state_dict(oc, int(g["seed"])))
    grid = int(g["grid"])
    code, ids = m.sample_codes(torch.from_numpy(g["in_input_ids"]).cuda(), n_steps=int(g["n_steps"]), grid_size=grid)
    B = code.shape[0]
    ref = torch.from_numpy(g["code"]).permute(0, 2, 1).reshape(B, -1, grid, grid)
    assert maxdiff(code.cpu(), ref) == 0.0
