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


@pytest.mark.parametrize("cls_name", ["VQAModel", "GQAModel"])
def test_vqa_model_dropin_matches_reference_fixture(cls_name):
    """the nn.Module surface of tasks/vqa_model.py -- reference-layout state dict in, {'logit'} out, and the
    reference's own training idiom (BCEWithLogitsLoss on the logit, .backward(), param.grad) gives the fixture's gradients.
    GQAModel (ref tasks/gqa_model.py:7-72) is the same module under the name the GQA driver imports."""
    import lxmert_oracle as O
    import xlxmert_amd.modeling as M
    from _util import golden_cfg, golden_inputs, load_golden, maxdiff
    from xlxmert_amd.config import XLxmertConfig
    VQAModel = getattr(M, cls_name)
    assert issubclass(M.GQAModel, M.VQAModel)
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


def test_nlvr2_model_dropin_matches_reference_fixture():
    """SURVEY 8f N1, NLVR2: the nn.Module surface of tasks/nlvr2_model.py -- [P,2,V,F] pairs in, {'logit': [P,2]} out, and the
    reference's training idiom (CrossEntropyLoss, tasks/nlvr2.py:72) gives the fixture's gradients."""
    import lxmert_oracle as O
    from _util import golden_cfg, golden_inputs, load_golden, maxdiff
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.modeling import NLVR2Model
    g = load_golden("nlvr2_tiny")
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "l_layers", "x_layers",
                                                      "r_layers", "visual_feat_dim", "visual_pos_dim", "num_clusters")})
    m = NLVR2Model(cfg, dtype=torch.float32).eval()
    missing, _ = m.load_state_dict(O.make_nlvr2_state_dict(oc, int(g["seed"])))
    assert not missing
    inp = {k: v.cuda() for k, v in golden_inputs(g).items()}
    m.zero_grad()
    out = m(input_ids=inp["input_ids"], visual_feats=inp["visual_feats"], visual_pos=inp["visual_pos"],
            attention_mask=inp["input_ids"] > 0)
    assert out["logit"].shape == (inp["labels"].shape[0], 2)
    assert maxdiff(out["logit"].cpu(), g["logit"]) < 1e-4
    loss = torch.nn.CrossEntropyLoss()(out["logit"], inp["labels"])
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    for k in [str(n) for n in g["grad_names"]]:
        ref = torch.from_numpy(g["grad:" + k])
        assert maxdiff(params[k].grad.cpu(), ref) <= 1e-4 * max(1.0, ref.abs().max().item()), k
    assert tuple(m.state_dict()["answer_head.logit_fc.0.weight"].shape) == (2 * cfg.hidden_size, 2 * cfg.hidden_size)


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


# ---------------------------------------------------------------- SURVEY 8f N3 through the reference's nn.Module surface
def _qa_model(g, dtype=torch.float32):
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.modeling import XLxmertForPretraining
    oc = golden_cfg(g)
    A = int(g["num_qa_labels"])
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in CFG_KEYS}, task_qa=True, num_qa_labels=A)
    m = XLxmertForPretraining(cfg, device="cuda", dtype=dtype).eval()
    sd = O.make_qa_state_dict(oc, A, int(g["seed"]))
    missing, unexpected = m.load_state_dict({"module." + k: v for k, v in sd.items()})
    assert not missing and not unexpected, (missing, unexpected)
    return m, sd


def _qa_call(m, g, task):
    inp = {k: v.cuda() for k, v in golden_inputs(g).items()}
    ld = {"qa_labels": inp["qa_labels"]}
    if task == "vis_mask":
        ld.update(obj_labels=inp["obj_labels"], feat_labels=m.vis_emb(inp["cluster_ids"]))
    elif task == "word_mask":
        ld["word_labels"] = inp["word_labels"]
    elif task == "matched":
        ld["matched_labels"] = inp["matched_labels"]
    return m(input_ids=inp["input_ids"], visual_pos=inp["visual_pos"], attention_mask=inp["attention_mask"],
             cluster_ids=inp["cluster_ids"], vis_mask=inp["vis_mask"], token_type_ids=inp["token_type_ids"],
             return_dict=True, label_dict=ld, task=task)


@pytest.mark.parametrize("task", ["qa", "vis_mask", "word_mask", "matched"])
def test_pretraining_module_dispatches_every_task_like_the_reference(task):
    """XLxmertForPretraining.forward(task=...) on a task_qa model (ref lxrt/modeling.py:85-90, 211-235, 292-304): `.cls`,
    `.answer_head` with the App. C keys, out_dict keys / losses / qa_pred and, after total_loss.backward(), the reference's
    gradients (tests/golden/qa_tasks_tiny.npz = the reference's own branches)."""
    from _util import slice_idx
    g = load_golden("qa_tasks_tiny")
    m, sd = _qa_model(g)
    keys = set(m.state_dict().keys())
    assert {"cls.predictions.bias", "cls.predictions.transform.dense.weight", "cls.predictions.transform.LayerNorm.bias",
            "cls.predictions.decoder.weight", "cls.seq_relationship.weight", "answer_head.logit_fc.0.weight",
            "answer_head.logit_fc.2.bias", "answer_head.logit_fc.3.weight", "bert.pooler.dense.bias"} <= keys
    names = {n for n, _ in m.named_parameters()}
    assert {"cls.seq_relationship.bias", "cls.predictions.transform.dense.bias", "answer_head.logit_fc.3.bias"} <= names
    m.zero_grad()
    out = _qa_call(m, g, task)
    want = {"qa": {"qa_loss"}, "vis_mask": {"obj_loss", "feat_loss", "vis_loss", "qa_loss"}, "word_mask": {"lm_loss", "qa_loss"},
            "matched": {"matched_loss", "qa_loss"}}[task] | {"qa_pred", "total_loss"}
    assert set(out) == want, set(out) ^ want
    for k in want - {"qa_pred"}:
        assert abs(out[k].item() - float(g[f"{task}:{k}"])) < 2e-5, k
    assert (out["qa_pred"].cpu().numpy() == g[f"{task}:qa_pred"]).all()
    out["total_loss"].backward()
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    tied = "cls.predictions.decoder.weight"
    for k in [str(n) for n in g[task + ":grad_names"] if str(n) != tied]:
        got = params[k].grad.cpu()
        if f"{task}:grad:{k}" in g:
            ref = torch.from_numpy(g[f"{task}:grad:{k}"])
            assert maxdiff(got, ref) <= 1e-4 * max(1.0, ref.abs().max().item()), k
        else:
            ref = torch.from_numpy(g[f"{task}:gslice:{k}"])
            assert maxdiff(got.reshape(-1)[torch.from_numpy(slice_idx(got.numel()))], ref) <= 1e-4 * max(1.0, ref.abs().max().item()), k
    # tensors outside the branch (e.g. the codebook head in task 'qa') got nothing
    if task == "qa":
        assert params["obj_predict_head.linear_feat.weight"].grad.abs().max().item() == 0.0
        assert params["cls.seq_relationship.weight"].grad.abs().max().item() == 0.0


def test_module_backward_accumulates_until_zero_grad():
    """--update > 1 idiom: two forward/backward calls without zero_grad() leave the SUM of both gradients (autograd semantics);
    a scaled loss scales only what that call adds."""
    g = load_golden("qa_tasks_tiny")
    m, sd = _qa_model(g)
    m.zero_grad()
    _qa_call(m, g, "vis_mask")["total_loss"].backward()
    g1 = m._store.grad[:m._store.n_used].clone()
    _qa_call(m, g, "word_mask")["total_loss"].backward()
    g12 = m._store.grad[:m._store.n_used].clone()
    m.zero_grad()
    _qa_call(m, g, "word_mask")["total_loss"].backward()
    g2 = m._store.grad[:m._store.n_used].clone()
    torch.cuda.synchronize()
    assert g1.abs().max().item() > 0 and maxdiff(g12.cpu(), (g1 + g2).cpu()) < 1e-5 * max(1.0, g12.abs().max().item())
    (0.5 * _qa_call(m, g, "word_mask")["total_loss"]).backward()
    torch.cuda.synchronize()
    assert maxdiff(m._store.grad[:m._store.n_used].cpu(), (1.5 * g2).cpu()) < 1e-5 * max(1.0, g2.abs().max().item())


def test_pretraining_module_without_qa_head_and_default_tasks():
    """the canonical configuration (pretrain.bash: Mask_LM + Matched + Obj prediction, no --taskQA): `.cls` exists, no
    `.answer_head`; word_mask / matched dispatch matches the reference fixture of that configuration."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.modeling import XLxmertForPretraining
    g = load_golden("lang_tasks_tiny")
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in CFG_KEYS})
    m = XLxmertForPretraining(cfg, device="cuda", dtype=torch.float32).eval()
    assert hasattr(m, "cls") and not hasattr(m, "answer_head")
    m.load_state_dict(O.make_cls_state_dict(oc, int(g["seed"])))
    inp = {k: v.cuda() for k, v in golden_inputs(g).items()}
    for task, key in (("word_mask", "word_labels"), ("matched", "matched_labels")):
        m.zero_grad()
        out = m(input_ids=inp["input_ids"], visual_pos=inp["visual_pos"], attention_mask=inp["attention_mask"],
                cluster_ids=inp["cluster_ids"], vis_mask=inp["vis_mask"], token_type_ids=inp["token_type_ids"],
                return_dict=True, label_dict={key: inp[key]}, task=task)
        assert abs(out["total_loss"].item() - float(g[task + ":loss"])) < 2e-5
        out["total_loss"].backward()
        torch.cuda.synchronize()
        params = dict(m.named_parameters())
        for k in [str(n) for n in g[task + ":grad_names"] if str(n) != "cls.predictions.decoder.weight"]:
            ref = torch.from_numpy(g[f"{task}:grad:{k}"])
            assert maxdiff(params[k].grad.cpu(), ref) <= 1e-4 * max(1.0, ref.abs().max().item()), (task, k)
    with pytest.raises(ValueError):
        m(input_ids=inp["input_ids"], visual_pos=inp["visual_pos"], cluster_ids=inp["cluster_ids"], task="attr_mask", label_dict={})


def test_bert_forward_options_match_reference_fixture():
    """LxmertModel.forward(visual_attention_mask=, output_hidden_states=True) through the module API, and a loss that reads all
    three outputs (the pooled_output gradient included) through autograd: outputs, every hidden state and every gradient
    against the reference's own LxmertModel (fixture vismask_tiny)."""
    g = load_golden("vismask_tiny")
    m, oc, sd = make_model(g)
    m.train(False)
    t = lambda k: torch.from_numpy(g[k]).cuda()
    out = m.bert(input_ids=t("in_input_ids"), visual_feats=t("in_visual_feats"), visual_pos=t("in_visual_pos"),
                 attention_mask=t("in_attention_mask"), visual_attention_mask=t("in_visual_attention_mask"),
                 token_type_ids=t("in_token_type_ids"), output_hidden_states=True, return_dict=True)
    real = t("in_attention_mask").bool()
    assert maxdiff(out.language_output[real].cpu(), t("lang")[real].cpu()) < 1e-4
    assert maxdiff(out.vision_output.cpu(), g["vis"]) < 1e-4 and maxdiff(out.pooled_output.cpu(), g["pooled"]) < 1e-4
    assert len(out.language_hidden_states) == oc.l_layers + oc.x_layers and len(out.vision_hidden_states) == oc.r_layers + oc.x_layers
    for i, h in enumerate(out.language_hidden_states):
        assert maxdiff(h[real].cpu(), t(f"lang_h{i}")[real].cpu()) < 1e-4, i
    for i, h in enumerate(out.vision_hidden_states):
        assert maxdiff(h.cpu(), g[f"vis_h{i}"]) < 1e-4, i
    m.zero_grad()
    vm = t("in_visual_attention_mask")
    loss = (out.language_output * t("w_lang") * real[..., None]).sum() + (out.vision_output * t("w_vis") * vm[..., None]).sum() + \
        (out.pooled_output * t("w_pooled")).sum()
    assert abs(loss.item() - g["loss"].item()) < 1e-4 * max(1.0, abs(g["loss"].item()))
    loss.backward()
    params = dict(m.named_parameters())
    for k in [str(n) for n in g["grad_names"]]:
        ref = torch.from_numpy(g["grad:" + k]).double()
        got = params[k].grad.double().cpu()
        assert (got - ref).norm().item() <= 1e-4 * max(ref.norm().item(), 1e-3), k
    # tuple form: HF appends the two hidden-state tuples
    tup = m.bert(input_ids=t("in_input_ids"), visual_feats=t("in_visual_feats"), visual_pos=t("in_visual_pos"),
                 attention_mask=t("in_attention_mask"), output_hidden_states=True, return_dict=False)
    assert len(tup) == 5 and len(tup[3]) == oc.l_layers + oc.x_layers


def test_bert_inputs_embeds_through_autograd():
    """LxmertModel.forward(inputs_embeds=leaf) through the module API: outputs and d(inputs_embeds) via autograd (fixture
    embeds_tiny from the reference's own LxmertModel)."""
    g = load_golden("embeds_tiny")
    m, oc, sd = make_model(g)
    m.train(False)
    t = lambda k: torch.from_numpy(g[k]).cuda()
    emb = t("in_inputs_embeds").clone().requires_grad_(True)
    out = m.bert(inputs_embeds=emb, visual_feats=t("in_visual_feats"), visual_pos=t("in_visual_pos"),
                 attention_mask=t("in_attention_mask"), token_type_ids=t("in_token_type_ids"), return_dict=True)
    real = t("in_attention_mask").bool()
    assert maxdiff(out.language_output[real].cpu(), t("lang")[real].cpu()) < 1e-4
    assert maxdiff(out.vision_output.cpu(), g["vis"]) < 1e-4 and maxdiff(out.pooled_output.cpu(), g["pooled"]) < 1e-4
    m.zero_grad()
    loss = (out.language_output * t("w_lang") * real[..., None]).sum() + (out.vision_output * t("w_vis")).sum() + \
        (out.pooled_output * t("w_pooled")).sum()
    loss.backward()
    ref = t("d_inputs_embeds").double()
    assert (emb.grad.double() - ref).norm().item() <= 1e-4 * ref.norm().item()
    with pytest.raises(ValueError):
        m.bert(input_ids=t("in_input_ids"), inputs_embeds=emb, visual_feats=t("in_visual_feats"), visual_pos=t("in_visual_pos"))


def test_reference_written_checkpoint_through_the_module_api(tmp_path):
    """SURVEY 8f N4 on the device: tests/golden/ckpt_tiny_LXRT.pth (the reference model's own state_dict() behind `module.`,
    oracle/gen_golden.py::gen_ckpt) -> io.load_state_dict -> XLxmertForPretraining.load_state_dict(strict=True): nothing
    missing, nothing unexpected, the reference's outputs reproduced; state_dict() -> save_checkpoint -> the reference's own
    loader semantics gives the same tensors back."""
    import os
    from _util import GOLDEN
    from xlxmert_amd import io as xio
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.modeling import XLxmertForPretraining
    g = load_golden("ckpt_tiny_io")
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in CFG_KEYS})
    cfg.task_mask_lm, cfg.task_matched = True, True
    m = XLxmertForPretraining(cfg, device="cuda", dtype=torch.float32)
    path = os.path.join(GOLDEN, "ckpt_tiny_LXRT.pth")
    sd = xio.load_state_dict(path)
    missing, unexpected = m.load_state_dict(torch.load(path), strict=True)        # the raw file: `module.` keys
    assert missing == [] and unexpected == []
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.eval()
    inp = {k: v.cuda() for k, v in golden_inputs(g).items()}
    with torch.no_grad():
        feats = torch.where(inp["vis_mask"][..., None], m.mask_feat.view(1, 1, -1), m.vis_emb(inp["cluster_ids"]))
        lang, vis, pooled = m.bert(input_ids=inp["input_ids"], visual_feats=feats, visual_pos=inp["visual_pos"],
                                   attention_mask=inp["attention_mask"], token_type_ids=inp["token_type_ids"])
        head = m.obj_predict_head(vis, out_keys=["obj", "feat"])
    real = inp["attention_mask"].reshape(-1).bool().cpu()
    B, L = inp["input_ids"].shape
    assert maxdiff(lang.float().cpu().view(B * L, -1)[real], torch.from_numpy(g["lang"]).view(B * L, -1)[real]) < 1e-4
    assert maxdiff(vis.float().cpu(), g["vis"]) < 1e-4 and maxdiff(pooled.float().cpu(), g["pooled"]) < 1e-4
    assert maxdiff(head["obj"].cpu(), g["obj"]) < 2e-4 and maxdiff(head["feat"].cpu(), g["feat"]) < 1e-4
    out = m(input_ids=inp["input_ids"], visual_pos=inp["visual_pos"], attention_mask=inp["attention_mask"],
            cluster_ids=inp["cluster_ids"], vis_mask=inp["vis_mask"], token_type_ids=inp["token_type_ids"],
            label_dict={"obj_labels": inp["obj_labels"], "feat_labels": m.vis_emb(inp["cluster_ids"])}, task="vis_mask")
    assert abs(out["obj_loss"].item() - g["obj_loss"].item()) < 1e-4 and abs(out["feat_loss"].item() - g["feat_loss"].item()) < 1e-4
    back = xio.load_state_dict_reference_semantics(xio.save_checkpoint(m, str(tmp_path), "Epoch01"))
    assert set(back) == set(sd)
    for k, v in sd.items():
        assert torch.equal(back[k], v), k


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_output_attentions_through_the_module_api(dtype, tol):
    """LxmertModel.forward(output_attentions=True) (HF:691-704): language / vision / cross-encoder attention probabilities with
    HF's field names, tuple position and shapes, against the reference's own outputs (fixture attn_tiny); hidden states and
    attentions together in the tuple form."""
    g = load_golden("attn_tiny")
    m, oc, sd = make_model(g, dtype)
    t = lambda k: torch.from_numpy(g[k]).cuda()
    kw = dict(input_ids=t("in_input_ids"), visual_feats=t("in_visual_feats").to(dtype), visual_pos=t("in_visual_pos"),
              attention_mask=t("in_attention_mask"), visual_attention_mask=t("in_visual_attention_mask"),
              token_type_ids=t("in_token_type_ids"))
    with torch.no_grad():
        out = m.bert(output_attentions=True, **kw)
        tup = m.bert(output_attentions=True, output_hidden_states=True, return_dict=False, **kw)
    assert len(tup) == 8 and len(tup[5]) == int(g["n_lang_att"]) and len(tup[6]) == int(g["n_vis_att"]) and len(tup[7]) == int(g["n_x_att"])
    real = t("in_attention_mask").bool()
    assert maxdiff(out.vision_output.float().cpu(), g["vis"]) < (1e-4 if dtype == torch.float32 else 0.15)
    for name, got in (("lang_att", out.language_attentions), ("vis_att", out.vision_attentions), ("x_att", out.cross_encoder_attentions)):
        for i, a in enumerate(got):
            ref = t(f"{name}{i}")
            assert a.shape == ref.shape and a.dtype == torch.float32
            msk = torch.ones_like(ref, dtype=torch.bool) if name == "vis_att" else real[:, None, :, None].expand_as(ref)
            assert maxdiff(a[msk].cpu(), ref[msk].cpu()) < tol, (name, i)


# ---------------------------------------------------------------- SURVEY 8f N2 through the reference's class
def _imggen_model(g):
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.modeling import ImggenModel
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in CFG_KEYS})
    grid = int(g["grid"])
    m = ImggenModel(cfg, args=None, num_clusters=oc.num_clusters, dtype=torch.float32, grid_size=grid)
    assert not hasattr(m, "cls") and not hasattr(m, "answer_head") and m.G is None and m.vis_emb is None
    with pytest.raises(RuntimeError):
        m.sample_image_NAR(torch.from_numpy(g["in_input_ids"]).cuda())         # no codebook yet
    sd = O.make_state_dict(oc, int(g["seed"]))
    m.load_state_dict({k: v for k, v in sd.items() if k not in ("vis_emb.weight", "obj_predict_head.out_cluster.weight")})
    m.set_visual_embedding(sd["vis_emb.weight"].numpy())                        # ref :28-39 (np.ndarray accepted)
    assert m.obj_predict_head.out_cluster.weight is m.vis_emb.weight
    with pytest.raises(RuntimeError):
        m.sample_image_NAR(torch.from_numpy(g["in_input_ids"]).cuda(), n_steps=1)          # no generator yet
    m.set_image_generator(lambda x: x)                                          # identity "GAN": the image IS the code grid
    return m, grid


def _as_image(code, grid):
    """what ImggenModel returns for an identity generator: denorm(code as [B, F, g, g]) on the host"""
    c = torch.as_tensor(code)
    B = c.shape[0]
    return ((c.permute(0, 2, 1).reshape(B, -1, grid, grid) + 1) / 2).clamp(0, 1)


def test_imggen_model_nar_matches_reference_fixture():
    """ImggenModel.sample_image_NAR (ref tasks/imggen_model.py:169-257) through the class: final image, every intermediate image
    (return_intermediate: the running codes after each Mask-Predict step, from the fixture's per-step predictions and masks), a
    tokenizer callable with LxmertTokenizer's call signature, and n_steps=None -> grid ** 2 steps."""
    g = load_golden("sampler_tiny")
    m, grid = _imggen_model(g)
    ids = torch.from_numpy(g["in_input_ids"]).cuda()
    T = int(g["n_steps"])
    img = m.sample_image_NAR(ids, n_steps=T)
    assert img.device.type == "cpu" and maxdiff(img, _as_image(g["code"], grid)) == 0.0
    steps = m.sample_image_NAR(ids, n_steps=T, return_intermediate=True)
    assert isinstance(steps, list) and len(steps) == T and maxdiff(steps[-1], img) == 0.0
    cent = m.vis_emb.weight.cpu()
    cur = torch.zeros_like(torch.from_numpy(g["step_pred_ids"][0]))
    for i in range(T):
        mask = torch.from_numpy(g["step_masks"][i]).bool()
        cur = torch.where(mask, torch.from_numpy(g["step_pred_ids"][i]), cur)
        assert maxdiff(steps[i], _as_image(cent[cur], grid)) == 0.0, i

    class Tok:                                  # LxmertTokenizer's call signature (ref :56-58): sentences -> .input_ids
        def __call__(self, sentences, max_length=None, truncation=None, return_tensors=None):
            assert max_length == 20 and truncation is True and return_tensors == "pt" and len(sentences) == ids.shape[0]
            return type("Enc", (), {"input_ids": ids.cpu()})()
    with pytest.raises(RuntimeError):
        m.sample_image_NAR(["a", "b", "c"], n_steps=T)                          # strings need a tokenizer
    m.tokenizer = Tok()
    assert maxdiff(m.sample_image_NAR(["a", "b", "c"], n_steps=T), img) == 0.0
    full = m.sample_image_NAR(ids)                                              # n_steps=None: grid ** 2 refinement steps (ref :191-192)
    assert full.shape == img.shape and int(m.bert._engine.vmask.sum().item()) == ids.shape[0] * int(1 / grid ** 2 * grid ** 2)


@pytest.mark.parametrize("mode", ["confidence", "tlbr", "random"])
def test_imggen_model_ar_matches_reference_fixture(mode):
    """ImggenModel.sample_image_AR (ref :49-167): position_confidence (default) / position_TLBR / position_random(seed=7: the
    reference's random.Random(seed).shuffle order) give the fixture's codes; the intermediate images follow the fixture's masks."""
    g = load_golden("sampler_ar_tiny")
    m, grid = _imggen_model(g)
    ids = torch.from_numpy(g["in_input_ids"]).cuda()
    kw = {"confidence": {}, "tlbr": dict(position_TLBR=True), "random": dict(position_random=True, seed=7)}[mode]
    img = m.sample_image_AR(ids, **kw)
    assert maxdiff(img, _as_image(g["code_" + mode], grid)) == 0.0
    assert maxdiff(m.vis_emb.weight[m.code_ids].cpu(), g["code_" + mode]) == 0.0       # the chosen ids ARE those codes
    steps = m.sample_image_AR(ids, return_intermediate=True, **kw)
    V = grid * grid
    assert len(steps) == V and maxdiff(steps[-1], img) == 0.0
    # after step i the positions still masked hold mask_feat, the others their final code (a position is filled once)
    mf, final = m.mask_feat.detach().cpu(), torch.from_numpy(g["code_" + mode])
    for i in (0, V // 2, V - 2):
        mask = torch.from_numpy(g["step_masks_" + mode][i]).bool()
        code = torch.where(mask[..., None], mf.view(1, 1, -1), final)
        assert maxdiff(steps[i], _as_image(code, grid)) == 0.0, i
    with pytest.raises(ValueError):
        m.sample_image_AR(ids, position_confidence=False)
