"""Engine sequencing (forward chain + hand-derived backward chain) checked on CPU against the oracle,
with the kernels replaced by tests/fake_ops.FakeOps (test infrastructure, see its header)."""
import pytest
import torch

import lxmert_oracle as O
from _util import golden_cfg, golden_inputs, load_golden, maxdiff
from fake_ops import FakeOps
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.engine import Engine
from xlxmert_amd.params import ParamStore


def make_engine(g, need_lang, dtype=torch.float32, row_pad=None):
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                      "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                      "l_layers", "x_layers", "r_layers", "visual_feat_dim",
                                                      "visual_pos_dim", "num_clusters")})
    sd = O.make_state_dict(oc, int(g["seed"]))
    inp = golden_inputs(g)
    B, L = inp["input_ids"].shape
    V = inp["cluster_ids"].shape[1]
    store = ParamStore(cfg, "cpu", dtype, task="vis_mask" if not need_lang else "all")
    store.load_named(sd)
    eng = Engine(cfg, store, FakeOps(dtype), B, L, V, need_lang=need_lang)
    if row_pad is not None:
        eng.ROW_PAD = row_pad
    eng.sync_compute_weights()
    eng.set_inputs(inp["input_ids"], inp["attention_mask"], inp["token_type_ids"], inp["visual_pos"],
                   cluster_ids=inp["cluster_ids"], vis_mask=inp["vis_mask"], obj_labels=inp["obj_labels"])
    return eng, oc, sd, inp


@pytest.mark.parametrize("name", ["tiny_222", "tiny_955"])
def test_forward_matches_golden(name):
    g = load_golden(name)
    eng, oc, sd, inp = make_engine(g, need_lang=True)
    lang, vis, pooled = eng.encoder_forward()
    feat, logits = eng.head_forward()
    B, L = inp["input_ids"].shape
    real = inp["attention_mask"].reshape(-1)
    assert maxdiff(lang.view(B * L, -1)[real], torch.from_numpy(g["lang"]).view(B * L, -1)[real]) < 5e-5
    assert maxdiff(vis.view(g["vis"].shape), g["vis"]) < 5e-5
    assert maxdiff(pooled, g["pooled"]) < 5e-5
    assert maxdiff(feat.view(g["feat"].shape), g["feat"]) < 5e-5
    assert maxdiff(logits.view(g["obj"].shape), g["obj"]) < 1e-4
    losses = eng.losses_forward_backward(want_grad=False)
    assert abs(losses[0].item() - g["obj_loss"].item()) < 2e-5
    assert abs(losses[1].item() - g["feat_loss"].item()) < 2e-5


@pytest.mark.parametrize("row_pad", [None, 1, 8])
def test_vis_mask_step_gradients(row_pad):
    """row_pad: granule of the masked-row head's row list (default 256 = every row at this size; 1 = the exact list;
    8 = a list padded with -1 entries, which must change nothing)"""
    g = load_golden("tiny_222")
    eng, oc, sd, inp = make_engine(g, need_lang=False, row_pad=row_pad)
    n_masked = int((inp["vis_mask"].reshape(-1) != 0).sum())
    if row_pad is not None:
        assert eng.n_mrows == (n_masked + row_pad - 1) // row_pad * row_pad < eng.MV
        if row_pad == 8:
            assert eng.n_mrows > n_masked and (eng.mrows[n_masked:eng.n_mrows] == -1).all()
    losses = eng.vis_mask_forward_backward()
    # with a row list shorter than all rows the LAST cross layer's visual feed-forward block ran on the masked rows only
    # (Engine.encoder_forward ffn_rows): the reference's losses and gradients below are unchanged by it
    if row_pad is not None:
        assert eng._ffn_rows_run is not None and eng.x_layers[-1]["ffn_v"].rows == eng.n_mrows
    else:
        assert eng._ffn_rows_run is None and eng.x_layers[-1]["ffn_v"].rows is None
    assert abs(losses[0].item() - g["obj_loss"].item()) < 2e-5
    assert abs(losses[1].item() - g["feat_loss"].item()) < 2e-5
    st = eng.store
    names = [str(n) for n in g["grad_names"]]
    for k in names:
        got = st.gview(k)
        assert maxdiff(got, g["grad:" + k]) < 3e-5, k
    # tensors the reference leaves without a gradient sit outside the optimizer range
    used = {m.name for u in st.units if u.used for m in u.members}
    assert used == set(names), used ^ set(names)


@pytest.mark.parametrize("need_lang", [False, True])
@pytest.mark.parametrize("name", ["tiny_222", "tiny_955"])
def test_last_visual_ffn_on_masked_rows_only_changes_nothing(name, need_lang):
    """The masked-visual-token step reads the vision output at the masked rows only, so the last cross layer's visual
    feed-forward block runs on those rows (gathered, with -1 pad entries in the row list) and hands its output to the head
    compact; XL_COMPACT_LAST_FFN=0 / Engine.compact_last_ffn = False computes every row as the reference does.  Same losses, same
    gradients -- also in the engine that keeps the language side of the last cross layer alive; the encoder's plain forward is
    untouched (vision output on every row); a backward from a vision-output gradient is refused after a row-subset forward."""
    g = load_golden(name)
    res = {}
    for compact in (False, True):
        eng, oc, sd, inp = make_engine(g, need_lang=need_lang, row_pad=8 if name == "tiny_222" else 1)     # (with / without pad entries)
        eng.compact_last_ffn = compact
        assert 0 < eng.n_mrows < eng.MV
        losses = eng.vis_mask_forward_backward().clone()
        assert (eng._ffn_rows_run is not None) == compact
        res[compact] = (losses, eng.store.grad[:eng.store.n_used].clone())
        if compact and need_lang:
            with pytest.raises(RuntimeError, match="masked rows only"):
                eng.backward_from_outputs(d_vis=torch.zeros(eng.MV, eng.d))
        lang, vis, pooled = eng.encoder_forward()               # the plain forward: every row again
        assert eng._ffn_rows_run is None and maxdiff(vis.view(g["vis"].shape), g["vis"]) < 5e-5
    assert maxdiff(res[False][0], res[True][0]) < 1e-6
    assert maxdiff(res[False][1], res[True][1]) < 2e-6


def test_need_lang_engine_gives_same_vis_grads():
    """Full (lang+vis+pooled) engine with a zero language gradient == the dead-branch-eliminated engine."""
    g = load_golden("tiny_222")
    eng, oc, sd, inp = make_engine(g, need_lang=True)
    eng.encoder_forward()
    eng.head_forward()
    eng.store.grad.zero_()
    eng.losses_forward_backward(True)
    eng.GA.zero_()
    eng.head_backward(eng.vr(eng.GA))
    eng.encoder_backward(have_lang_grad=True)
    for k in [str(n) for n in g["grad_names"]]:
        assert maxdiff(eng.store.gview(k), g["grad:" + k]) < 3e-5, k


def test_dropout_backward_matches_directional_derivative():
    """Training mode (hidden + attention-probability dropout, counter-based masks): the hand-derived backward agrees
    with a central finite difference of the loss along a random parameter direction (masks are a pure function of
    (seed, index), so the loss is smooth in the parameters)."""
    g = load_golden("tiny_222")

    def run(master=None, grads=True):
        oc = golden_cfg(g)
        cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                          "intermediate_size", "max_position_embeddings",
                                                          "type_vocab_size", "l_layers", "x_layers", "r_layers",
                                                          "visual_feat_dim", "visual_pos_dim", "num_clusters")})
        inp = golden_inputs(g)
        B, L = inp["input_ids"].shape
        store = ParamStore(cfg, "cpu", torch.float32)
        store.load_named(O.make_state_dict(oc, int(g["seed"])))
        if master is not None:
            store.master.copy_(master)
        eng = Engine(cfg, store, FakeOps(torch.float32), B, L, inp["cluster_ids"].shape[1], need_lang=False,
                     train_dropout=True)
        assert eng.p_hid == 0.1 and eng.p_attn == 0.1
        eng.set_step_seed(7)
        eng.set_inputs(inp["input_ids"], inp["attention_mask"], inp["token_type_ids"], inp["visual_pos"],
                       cluster_ids=inp["cluster_ids"], vis_mask=inp["vis_mask"], obj_labels=inp["obj_labels"])
        if grads:
            losses = eng.vis_mask_forward_backward()
        else:
            eng.encoder_forward(want_pooled=False)
            eng.head_forward()
            losses = eng.losses_forward_backward(want_grad=False)
        return losses[0].double().item() + losses[1].double().item(), store

    loss0, st0 = run()
    assert abs(loss0 - (g["obj_loss"].item() + g["feat_loss"].item())) > 1e-3      # dropout really changed the forward
    torch.manual_seed(0)
    gflat = st0.grad[:st0.n_used].double()
    u = torch.randn(st0.n_used, dtype=torch.float64)
    u[gflat == 0] = 0                               # frozen rows / padding carry no gradient
    u = u / u.norm()
    eps = 2e-2
    mp, mm = st0.master.clone(), st0.master.clone()
    mp[:st0.n_used] += (eps * u).float()
    mm[:st0.n_used] -= (eps * u).float()
    lp, _ = run(mp, grads=False)
    lm, _ = run(mm, grads=False)
    fd = (lp - lm) / (2 * eps)
    an = (gflat * u).sum().item()
    assert abs(fd - an) <= 5e-3 * max(1.0, abs(an)), (fd, an)


def make_vqa_engine(g, ops, device="cpu", dtype=torch.float32):
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                      "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                      "l_layers", "x_layers", "r_layers", "visual_feat_dim",
                                                      "visual_pos_dim", "num_clusters")})
    A = int(g["num_answers"])
    sd = O.make_vqa_state_dict(oc, A, int(g["seed"]))
    inp = golden_inputs(g)
    B, L = inp["input_ids"].shape
    V = inp["visual_feats"].shape[1]
    store = ParamStore(cfg, device, dtype, task="vqa", num_answers=A)
    store.load_named(sd)
    eng = Engine(cfg, store, ops, B, L, V, need_lang=True)
    eng.sync_compute_weights()
    dev = torch.device(device)
    eng.set_inputs(inp["input_ids"].to(dev), inp["attention_mask"].to(dev), None, inp["visual_pos"].to(dev),
                   visual_feats=inp["visual_feats"].to(dev))
    return eng, inp


def test_vqa_step_vs_reference_fixture():
    """engine sequencing of the VQA fine-tune step (answer head + pooler backward + full encoder backward
    with d(language_output)) against the fixture generated by the reference's VQAModel."""
    g = load_golden("vqa_tiny")
    eng, inp = make_vqa_engine(g, FakeOps(torch.float32))
    loss = eng.vqa_forward_backward(inp["targets"])
    assert maxdiff(eng.answer.logit, g["logit"]) < 5e-5
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    names = [str(n) for n in g["grad_names"]]
    for k in names:
        ref = torch.from_numpy(g["grad:" + k])
        got = eng.store.gview(k)
        if "embeddings" in k and k.endswith("embeddings.weight"):
            ref = ref.clone()
            ref[0] = got[0]            # padding_idx = 0 rows get no gradient in the reference either (they are zero there)
        assert maxdiff(got, ref) <= 1e-4 * max(1.0, ref.abs().max().item()), k
    used = {m.name for u in eng.store.units if u.used for m in u.members}
    assert set(names) <= used and not any(n.startswith("obj_predict_head") or n == "mask_feat" for n in used)


def make_nlvr2_engine(g, ops, device="cpu", dtype=torch.float32):
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                      "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                      "l_layers", "x_layers", "r_layers", "visual_feat_dim",
                                                      "visual_pos_dim", "num_clusters")})
    sd = O.make_nlvr2_state_dict(oc, int(g["seed"]))
    inp = golden_inputs(g)
    B, L = inp["input_ids"].shape                      # B = 2 P encoder rows
    P, _, V, Fd = inp["visual_feats"].shape
    store = ParamStore(cfg, device, dtype, task="nlvr2")
    store.load_named(sd)
    eng = Engine(cfg, store, ops, B, L, V, need_lang=True)
    eng.sync_compute_weights()
    dev = torch.device(device)
    eng.set_inputs(inp["input_ids"].to(dev), inp["attention_mask"].to(dev), None, inp["visual_pos"].reshape(B, V, -1).to(dev),
                   visual_feats=inp["visual_feats"].reshape(B, V, Fd).to(dev))
    return eng, inp


def check_nlvr2_grads(eng, g, tol):
    names = [str(n) for n in g["grad_names"]]
    for k in names:
        ref = torch.from_numpy(g["grad:" + k])
        got = eng.store.gview(k).float().cpu()
        if "embeddings" in k and k.endswith("embeddings.weight"):
            ref = ref.clone()
            ref[0] = got[0]
        assert maxdiff(got, ref) <= tol * max(1.0, ref.abs().max().item()), k
    used = {m.name for u in eng.store.units if u.used for m in u.members}
    assert set(names) == used                          # the optimizer range is exactly the reference's grad-carrying set


def test_nlvr2_step_vs_reference_fixture():
    """SURVEY 8f N1, NLVR2 variant: 2P encoder rows, pooled_output read as [P, 2d] by the pair head, CrossEntropyLoss over
    the 2 classes, backward through head, pooler and the whole encoder -- against the reference's NLVR2Model.forward."""
    g = load_golden("nlvr2_tiny")
    eng, inp = make_nlvr2_engine(g, FakeOps(torch.float32))
    loss = eng.nlvr2_forward_backward(inp["labels"])
    assert eng.answer.logit.shape == (inp["labels"].shape[0], 2)
    assert maxdiff(eng.answer.logit, g["logit"]) < 5e-5
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    check_nlvr2_grads(eng, g, 1e-4)


def make_sampler_engine(g, ops, device="cpu", dtype=torch.float32):
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                      "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                      "l_layers", "x_layers", "r_layers", "visual_feat_dim",
                                                      "visual_pos_dim", "num_clusters")})
    sd = O.make_state_dict(oc, int(g["seed"]))
    ids = torch.from_numpy(g["in_input_ids"])
    B, L = ids.shape
    grid = int(g["grid"])
    V = grid * grid
    store = ParamStore(cfg, device, dtype, task="vis_mask")
    store.load_named(sd)
    eng = Engine(cfg, store, ops, B, L, V, need_lang=False)
    eng.sync_compute_weights()
    dev = torch.device(device)
    pos = torch.from_numpy(O.box_position(grid)).unsqueeze(0).expand(B, -1, -1)
    eng.set_inputs(ids.to(dev), (ids > 0).to(dev), None, pos.to(dev), cluster_ids=torch.zeros(B, V, dtype=torch.long, device=dev),
                   vis_mask=torch.ones(B, V, dtype=torch.bool, device=dev))
    return eng, sd


def test_sampler_loop_vs_reference_fixture():
    """the engine's on-device Mask-Predict loop reproduces the reference modules' codes exactly (fp32)."""
    g = load_golden("sampler_tiny")
    eng, sd = make_sampler_engine(g, FakeOps(torch.float32))
    cid, code, prob = eng.sample_codes_nar(int(g["n_steps"]))
    assert maxdiff(code.view(g["code"].shape), g["code"]) == 0.0
    assert maxdiff(prob.view(g["step_pred_prob"][-1].shape), g["step_pred_prob"][-1]) < 1e-5
    assert torch.equal(eng.vmask.long().cpu(), torch.from_numpy(g["step_masks"][-1]))


def make_lang_task_engine(g, task, ops, device="cpu", dtype=torch.float32):
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                      "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                      "l_layers", "x_layers", "r_layers", "visual_feat_dim",
                                                      "visual_pos_dim", "num_clusters")})
    sd = O.make_cls_state_dict(oc, int(g["seed"]))
    inp = golden_inputs(g)
    B, L = inp["input_ids"].shape
    V = inp["cluster_ids"].shape[1]
    store = ParamStore(cfg, device, dtype, task=task)
    store.load_named(sd)
    eng = Engine(cfg, store, ops, B, L, V, need_lang=True)
    eng.sync_compute_weights()
    dev = torch.device(device)
    eng.set_inputs(inp["input_ids"].to(dev), inp["attention_mask"].to(dev), inp["token_type_ids"].to(dev),
                   inp["visual_pos"].to(dev), cluster_ids=inp["cluster_ids"].to(dev))
    return eng, inp


def labelled_rows(word_labels):
    """what a data loader hands over with the labels: flat indices b*L+l of the positions that carry one"""
    return (word_labels.reshape(-1) >= 0).nonzero().reshape(-1).to(torch.int32).cpu()


def check_lang_task(g, task, eng, inp, loss_tol, grad_tol, dev="cpu", rows=False):
    labels = inp["word_labels" if task == "word_mask" else "matched_labels"].to(dev)
    if task == "word_mask":
        loss = eng.word_mask_forward_backward(labels, labelled_rows(labels) if rows else None)
        assert (eng.lang_heads.n_rows > 0) == rows
    else:
        loss = eng.matched_forward_backward(labels)
    assert abs(loss.item() - float(g[task + ":loss"])) < loss_tol
    names = [str(n) for n in g[task + ":grad_names"]]
    used = {m.name for u in eng.store.units if u.used for m in u.members}
    assert set(names) == used, (sorted(set(names) ^ used))[:6]      # optimizer range == the reference's grad-carrying set
    for k in names:
        ref = torch.from_numpy(g[task + ":grad:" + k])
        got = eng.store.gview(k).cpu()
        assert maxdiff(got, ref) <= grad_tol * max(1.0, ref.abs().max().item()), k


@pytest.mark.parametrize("task", ["word_mask", "matched"])
def test_language_pretraining_steps_vs_reference_fixture(task):
    """MLM (tied decoder: d(word embeddings) = embedding scatter + decoder weight gradient) and matched head
    steps against the fixture generated by the reference's own branches."""
    g = load_golden("lang_tasks_tiny")
    eng, inp = make_lang_task_engine(g, task, FakeOps(torch.float32))
    check_lang_task(g, task, eng, inp, 5e-6, 1e-4)
    if task == "word_mask":             # decoder + loss on the labelled rows only: same loss, same gradients
        check_lang_task(g, task, eng, inp, 5e-6, 1e-4, rows=True)


def make_qa_engine(g, task, ops, device="cpu", dtype=torch.float32, store_task=None):
    """engine over a task_qa pretraining model (tests/golden/qa_tasks_tiny.npz): store task = the step's task, or "all"."""
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                      "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                      "l_layers", "x_layers", "r_layers", "visual_feat_dim",
                                                      "visual_pos_dim", "num_clusters")})
    A = int(g["num_qa_labels"])
    sd = O.make_qa_state_dict(oc, A, int(g["seed"]))
    inp = golden_inputs(g)
    B, L = inp["input_ids"].shape
    V = inp["cluster_ids"].shape[1]
    store = ParamStore(cfg, device, dtype, task=store_task or task, num_answers=A)
    store.load_named(sd)
    eng = Engine(cfg, store, ops, B, L, V, need_lang=True)
    eng.sync_compute_weights()
    return eng, inp


def run_qa_task(eng, inp, task, dev="cpu"):
    x = {k: v.to(dev) for k, v in inp.items()}
    base = (x["input_ids"], x["attention_mask"], x["token_type_ids"], x["visual_pos"])
    ql = x["qa_labels"]
    if task == "vis_mask":
        eng.set_inputs(*base, cluster_ids=x["cluster_ids"], vis_mask=x["vis_mask"], obj_labels=x["obj_labels"])
        losses = eng.vis_mask_forward_backward(True, qa_labels=ql)
        return losses[0] + losses[1] + eng.answer.loss[0]
    eng.set_inputs(*base, cluster_ids=x["cluster_ids"])                 # un-masked codebook features (ref modeling.py:190-193)
    if task == "qa":
        return eng.qa_forward_backward(ql)[0]
    if task == "word_mask":
        return eng.word_mask_forward_backward(x["word_labels"], qa_labels=ql)[0] + eng.answer.loss[0]
    return eng.matched_forward_backward(x["matched_labels"], qa_labels=ql)[0] + eng.answer.loss[0]


def check_qa_task(g, task, eng, inp, loss_tol, grad_tol, dev="cpu", exact_set=True):
    from _util import slice_idx
    total = run_qa_task(eng, inp, task, dev)
    if dev != "cpu":
        torch.cuda.synchronize()
    assert abs(total.item() - float(g[task + ":total_loss"])) < loss_tol, (total.item(), float(g[task + ":total_loss"]))
    assert abs(eng.answer.loss[0].item() - float(g[task + ":qa_loss"])) < loss_tol
    if task == "qa":
        assert (eng.answer.row_argmax.cpu().numpy() == g["qa:qa_pred"]).all()
    names = [str(n) for n in g[task + ":grad_names"]]
    if exact_set:
        used = {m.name for u in eng.store.units if u.used for m in u.members}
        assert set(names) == used, (sorted(set(names) ^ used))[:6]  # optimizer range == the reference's grad-carrying set
    for k in names:
        got = eng.store.gview(k).cpu()
        if f"{task}:grad:{k}" in g:
            ref = torch.from_numpy(g[f"{task}:grad:{k}"])
            assert maxdiff(got, ref) <= grad_tol * max(1.0, ref.abs().max().item()), k
        else:
            ref = torch.from_numpy(g[f"{task}:gslice:{k}"])
            smp = got.reshape(-1)[torch.from_numpy(slice_idx(got.numel()))]
            assert maxdiff(smp, ref) <= grad_tol * max(1.0, ref.abs().max().item()), k
            assert abs(got.double().norm().item() - g[f"{task}:gnorm:{k}"].item()) <= 10 * grad_tol * max(1.0, g[f"{task}:gnorm:{k}"].item()), k


@pytest.mark.parametrize("task", ["qa", "vis_mask", "word_mask", "matched"])
def test_qa_branch_steps_vs_reference_fixture(task):
    """SURVEY 8f N3, QA branch: a model built with task_qa adds the answer-head CE to every task's loss (ref
    lxrt/modeling.py:89-90, 292-304); fixture = the reference's own branches."""
    g = load_golden("qa_tasks_tiny")
    eng, inp = make_qa_engine(g, task, FakeOps(torch.float32))
    check_qa_task(g, task, eng, inp, 1e-5, 1e-4)


def check_ar_sampler(g, eng, mode):
    trace = []
    cid, code, prob = eng.sample_codes_ar(None, mode, positions=g["random_positions"].tolist(), trace=trace)
    assert maxdiff(code.float().cpu().view(g["code_" + mode].shape), g["code_" + mode]) == 0.0
    ref = torch.from_numpy(g["step_masks_" + mode])
    for i, m in enumerate(trace):
        assert torch.equal(m.cpu(), ref[i]), (mode, i)


@pytest.mark.parametrize("mode", ["confidence", "tlbr", "random"])
def test_ar_sampler_vs_reference_fixture(mode):
    """SURVEY 8f N2, autoregressive variant: same codes and the same fill order (mask after every step) as the reference loop."""
    g = load_golden("sampler_ar_tiny")
    eng, sd = make_sampler_engine(g, FakeOps(torch.float32))
    check_ar_sampler(g, eng, mode)


def test_fused_predict_equals_logits_path_bf16(monkeypatch):
    """Sampler on the bf16 path: the codebook contraction ending in the row-max epilogue (no logits in memory) gives the codes,
    masks and probabilities of the path that materialises the fp32 logits (engine sequencing + the host restatement of
    XL_EPI_ROWMAX / xl_rowmax_combine; the kernels themselves are compared on the GPU).  4 images x 8x8 grid = 256 rows, a
    100-entry codebook padded to 256."""
    g = load_golden("sampler_tiny")
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                      "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                      "l_layers", "x_layers", "r_layers", "visual_feat_dim",
                                                      "visual_pos_dim", "num_clusters")})
    sd = O.make_state_dict(oc, int(g["seed"]))
    B, L, grid = 4, 8, 8
    ids = torch.from_numpy(g["in_input_ids"])[:1].expand(B, -1).clone()
    ids[1:, 2] = (ids[1:, 2] + torch.arange(1, B)) % (cfg.vocab_size - 1) + 1
    pos = torch.from_numpy(O.box_position(grid)).unsqueeze(0).expand(B, -1, -1)
    outs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("XL_FUSED_PREDICT", fused)
        store = ParamStore(cfg, "cpu", torch.bfloat16, task="vis_mask")
        store.load_named(sd)
        eng = Engine(cfg, store, FakeOps(torch.bfloat16), B, L, grid * grid, need_lang=False)
        eng.sync_compute_weights()
        eng.set_inputs(ids, ids > 0, None, pos, cluster_ids=torch.zeros(B, grid * grid, dtype=torch.long),
                       vis_mask=torch.ones(B, grid * grid, dtype=torch.bool))
        assert eng.fused_predict_available() == (fused == "1")
        cid, code, prob = eng.sample_codes_nar(3)
        outs[fused] = (cid.clone(), prob.clone(), eng.vmask.clone(), [c for c in eng.ops.calls if c[0] == "gemm" and c[-1] == 5])
    assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][2], outs["0"][2])
    assert (outs["1"][1] - outs["0"][1]).abs().max().item() < 1e-5
    assert len(outs["1"][3]) == 3 and not outs["0"][3]


def make_vismask_engine(g, ops, device="cpu", dtype=torch.float32):
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                      "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                      "l_layers", "x_layers", "r_layers", "visual_feat_dim",
                                                      "visual_pos_dim", "num_clusters")})
    sd = O.make_state_dict(oc, int(g["seed"]))
    t = lambda k: torch.from_numpy(g[k]).to(device)
    B, L = g["in_input_ids"].shape
    V = g["in_visual_feats"].shape[1]
    store = ParamStore(cfg, device, dtype, task="all")
    store.load_named(sd)
    eng = Engine(cfg, store, ops, B, L, V, need_lang=True)
    eng.sync_compute_weights()
    eng.set_inputs(t("in_input_ids"), t("in_attention_mask"), t("in_token_type_ids"), t("in_visual_pos"),
                   visual_feats=t("in_visual_feats").to(dtype), visual_attention_mask=t("in_visual_attention_mask"))
    return eng


def check_vismask(g, eng, tol, gtol):
    """forward (outputs + every hidden state) and backward (gradients of the fixture's linear functional of the three outputs,
    through the pooler and the masked attention) against the reference's LxmertModel."""
    dev = eng.dev
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    B, L, V, d = eng.B, eng.L, eng.V, eng.d
    lang, vis, pooled = eng.encoder_forward(want_pooled=True)
    real = t("in_attention_mask").bool()
    assert maxdiff(lang.view(B, L, d)[real].float(), t("lang")[real]) < tol
    assert maxdiff(vis.view(B, V, d).float(), t("vis")) < tol and maxdiff(pooled.float(), t("pooled")) < tol
    lh, vh = eng.hidden_states()
    assert len(lh) == eng.cfg.l_layers + eng.cfg.x_layers and len(vh) == eng.cfg.r_layers + eng.cfg.x_layers
    for i, h in enumerate(lh):
        assert maxdiff(h.view(B, L, d)[real].float(), t(f"lang_h{i}")[real]) < tol, i
    for i, h in enumerate(vh):
        assert maxdiff(h.view(B, V, d).float(), t(f"vis_h{i}")) < tol, i
    eng.store.grad.zero_()
    vm = t("in_visual_attention_mask")
    eng.backward_from_outputs((t("w_lang") * real[..., None]).to(eng.cdtype), (t("w_vis") * vm[..., None]).to(eng.cdtype),
                              t("w_pooled").to(eng.cdtype))
    worst = 0.0
    for k in [str(n) for n in g["grad_names"]]:
        ref = t("grad:" + k).double()
        got = eng.store.gview(k).double()
        worst = max(worst, (got - ref).norm().item() / max(ref.norm().item(), 1e-3))      # key biases: exact-zero gradient
    assert worst < gtol, worst


def test_visual_attention_mask_hidden_states_and_pooled_gradient():
    g = load_golden("vismask_tiny")
    check_vismask(g, make_vismask_engine(g, FakeOps(torch.float32)), 5e-5, 1e-4)


def check_inputs_embeds(g, ops, device="cpu", dtype=torch.float32, tol=5e-5, gtol=1e-4):
    """Engine with inputs_embeds instead of input_ids: outputs, d(inputs_embeds), every parameter gradient; the word-embedding
    table gets no gradient (fixture embeds_tiny, from the reference's LxmertModel)."""
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                      "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                      "l_layers", "x_layers", "r_layers", "visual_feat_dim",
                                                      "visual_pos_dim", "num_clusters")})
    t = lambda k: torch.from_numpy(g[k]).to(device)
    B, L, d = g["in_inputs_embeds"].shape
    V = g["in_visual_feats"].shape[1]
    store = ParamStore(cfg, device, dtype, task="all")
    store.load_named(O.make_state_dict(oc, int(g["seed"])))
    eng = Engine(cfg, store, ops, B, L, V, need_lang=True)
    eng.sync_compute_weights()
    eng.set_inputs(None, t("in_attention_mask"), t("in_token_type_ids"), t("in_visual_pos"), visual_feats=t("in_visual_feats").to(dtype),
                   inputs_embeds=t("in_inputs_embeds"))
    lang, vis, pooled = eng.encoder_forward(want_pooled=True)
    real = t("in_attention_mask").bool()
    assert maxdiff(lang.view(B, L, d)[real].float(), t("lang")[real]) < tol
    assert maxdiff(vis.view(B, V, d).float(), t("vis")) < tol and maxdiff(pooled.float(), t("pooled")) < tol
    eng.store.grad.zero_()
    eng.backward_from_outputs((t("w_lang") * real[..., None]).to(eng.cdtype), t("w_vis").to(eng.cdtype), t("w_pooled").to(eng.cdtype))
    ref = t("d_inputs_embeds").double()
    assert (eng.d_inputs_embeds().double() - ref).norm().item() <= gtol * ref.norm().item()
    for k in [str(n) for n in g["grad_names"]]:
        ref = t("grad:" + k).double()
        assert (eng.store.gview(k).double() - ref).norm().item() <= gtol * max(ref.norm().item(), 1e-3), k
    assert eng.store.gview("bert.embeddings.word_embeddings.weight").abs().max().item() == 0.0
    # ... and the engine goes back to input_ids on the next call
    eng.set_inputs(torch.zeros(B, L, dtype=torch.long, device=device), t("in_attention_mask"), t("in_token_type_ids"), t("in_visual_pos"),
                   visual_feats=t("in_visual_feats").to(dtype))
    assert not eng.embeds_mode


def test_inputs_embeds_vs_reference_fixture():
    check_inputs_embeds(load_golden("embeds_tiny"), FakeOps(torch.float32))


@pytest.mark.parametrize("row_pad", [1, 8, 256])
def test_packed_language_rows_equal_the_dense_path(monkeypatch, row_pad):
    """Engine(pack_lang=True): the language side runs on the real tokens only (row list + per-example offsets, the packed count
    rounded up to `row_pad` with zero rows) -- outputs of the real rows, losses and EVERY gradient equal the dense path's and
    the reference fixture's; the [PAD] rows of language_output come back as zeros."""
    monkeypatch.setattr(Engine, "ROW_PAD", row_pad)
    g = load_golden("tiny_222")
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "l_layers", "x_layers",
                                                      "r_layers", "visual_feat_dim", "visual_pos_dim", "num_clusters")})
    sd = O.make_state_dict(oc, int(g["seed"]))
    inp = golden_inputs(g)
    B, L = inp["input_ids"].shape
    V = inp["cluster_ids"].shape[1]
    am = inp["attention_mask"].bool()
    n_real = int(am.sum())
    assert n_real < B * L                                        # the fixture has [PAD] positions
    res = {}
    for pack in (False, True):
        store = ParamStore(cfg, "cpu", torch.float32, task="all")
        store.load_named(sd)
        eng = Engine(cfg, store, FakeOps(torch.float32), B, L, V, need_lang=True, pack_lang=pack)
        eng.sync_compute_weights()
        kw = dict(lang_rows=am.reshape(-1).nonzero().reshape(-1),
                  lang_off=torch.cat([torch.zeros(1, dtype=torch.int64), am.sum(1).cumsum(0)]).to(torch.int32)) if pack else {}
        eng.set_inputs(inp["input_ids"], inp["attention_mask"], inp["token_type_ids"], inp["visual_pos"],
                       cluster_ids=inp["cluster_ids"], vis_mask=inp["vis_mask"], obj_labels=inp["obj_labels"], **kw)
        assert eng.packed == pack
        if pack:
            assert eng.ML == min(eng.MLc, (n_real + row_pad - 1) // row_pad * row_pad) and eng.MX == eng.MV + eng.ML
            assert (eng.lrows[n_real:eng.ML] == -1).all()
        lang, vis, pooled = eng.encoder_forward()
        lang, vis, pooled = lang.clone(), vis.clone(), pooled.clone()
        lh, vh = eng.hidden_states()
        lh = [t.clone() for t in lh]
        eng.store.grad.zero_()
        eng.vis_mask_forward_backward()
        # a gradient for the language output too (as the language tasks produce): d_lang random at real rows
        gl = torch.randn(B * L, cfg.hidden_size, generator=torch.Generator().manual_seed(1)) * am.reshape(-1, 1)
        eng.backward_from_outputs(d_lang=gl.view(B, L, -1), d_vis=None, d_pooled=torch.ones(B, cfg.hidden_size))
        res[pack] = (lang, vis, pooled, lh, eng.losses.clone(), eng.store.grad.clone())
    (l0, v0, p0, h0, ls0, g0), (l1, v1, p1, h1, ls1, g1) = res[False], res[True]
    real = am.reshape(-1)
    assert maxdiff(l0[real], l1[real]) < 2e-6 and maxdiff(v0, v1) < 2e-6 and maxdiff(p0, p1) < 2e-6
    assert l1[~real].abs().max().item() == 0.0                   # [PAD] rows of language_output: zeros
    assert maxdiff(l1.view(B * L, -1)[real], torch.from_numpy(g["lang"]).view(B * L, -1)[real]) < 5e-5
    for a, b in zip(h0, h1):
        assert maxdiff(a[real], b[real]) < 2e-6 and b[~real].abs().max().item() == 0.0
    assert maxdiff(ls0, ls1) < 1e-6
    assert maxdiff(g0, g1) < 5e-6, maxdiff(g0, g1)


def check_paired_blocks(ops_factory, device, tol):
    """Engine(pair_blocks=True) -- the visual / language sub-blocks of the cross layers and a visual + a language layer of the two
    stacks in lock step on one stream, their contractions two per launch (xl_gemm_pair) -- against the two-stream engine: outputs,
    hidden states, losses and EVERY gradient, with the language lane's own kernels on this stream (XL_PAIR_SIDE=0) and on the
    language stream (1).  tiny_955 has 9 language against 5 visual layers: language layers 0-3 run unpaired ahead of the pairs."""
    g = load_golden("tiny_955")
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "l_layers", "x_layers",
                                                      "r_layers", "visual_feat_dim", "visual_pos_dim", "num_clusters")})
    sd = O.make_state_dict(oc, int(g["seed"]))
    inp = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in golden_inputs(g).items()}
    B, L = inp["input_ids"].shape
    V = inp["cluster_ids"].shape[1]
    res = {}
    for mode in ("two_streams", "paired", "paired_side"):
        store = ParamStore(cfg, device, torch.float32, task="all")
        store.load_named(sd)
        eng = Engine(cfg, store, ops_factory(), B, L, V, need_lang=True)
        eng.pair_blocks, eng.pair_side = mode != "two_streams", mode == "paired_side"
        eng.sync_compute_weights()
        eng.set_inputs(inp["input_ids"], inp["attention_mask"], inp["token_type_ids"], inp["visual_pos"],
                       cluster_ids=inp["cluster_ids"], vis_mask=inp["vis_mask"], obj_labels=inp["obj_labels"])
        assert len(eng._stack_pairs()) == (min(cfg.l_layers, cfg.r_layers) if eng.pair_blocks else 0)
        lang, vis, pooled = (t.clone() for t in eng.encoder_forward())
        lh, vh = eng.hidden_states()
        hs = [t.clone() for t in lh + vh]
        eng.store.grad.zero_()
        eng.vis_mask_forward_backward()
        gl = torch.randn(B * L, cfg.hidden_size, generator=torch.Generator().manual_seed(1)).to(device) * (inp["attention_mask"].reshape(-1, 1) != 0)
        eng.backward_from_outputs(d_lang=gl.view(B, L, -1), d_vis=None, d_pooled=torch.ones(B, cfg.hidden_size, device=device))
        if device != "cpu":
            torch.cuda.synchronize()
        res[mode] = (lang, vis, pooled, hs, eng.losses.clone(), eng.store.grad.clone())
    ref = res["two_streams"]
    real = (inp["attention_mask"].reshape(-1) != 0)
    assert maxdiff(ref[0].view(B * L, -1)[real], torch.from_numpy(g["lang"]).to(device).view(B * L, -1)[real]) < 1e-4
    for mode in ("paired", "paired_side"):
        got = res[mode]
        for a, b in zip(ref[:3], got[:3]):
            assert maxdiff(a, b) <= tol, mode
        for a, b in zip(ref[3], got[3]):
            assert maxdiff(a, b) <= tol, mode
        assert maxdiff(ref[4], got[4]) <= tol and maxdiff(ref[5], got[5]) <= 10 * tol, (mode, maxdiff(ref[5], got[5]))


def test_paired_blocks_equal_the_two_stream_engine():
    check_paired_blocks(lambda: FakeOps(torch.float32), "cpu", 0.0)


def check_attention_probs(g, eng, tol):
    """Engine.attention_probs() against the reference's LxmertModel(output_attentions=True) (fixture attn_tiny): per language
    layer [B,H,L,L], per visual layer [B,H,V,V] (ragged visual mask), per cross layer [B,H,L,V]; rows of [PAD] queries are
    excluded (the reference leaves don't-care values there, the packed path zeros)."""
    t = lambda k: torch.from_numpy(g[k]).to(eng.dev)
    eng.encoder_forward(want_pooled=True)
    la, va, xa = eng.attention_probs()
    assert (len(la), len(va), len(xa)) == (int(g["n_lang_att"]), int(g["n_vis_att"]), int(g["n_x_att"]))
    real = t("in_attention_mask").bool()                       # [B, L]
    for i, a in enumerate(la):
        ref = t(f"lang_att{i}")
        m = real[:, None, :, None].expand_as(ref)
        assert maxdiff(a[m], ref[m]) < tol, ("lang", i)
        assert a.shape == ref.shape and abs(a.sum(-1)[real[:, None, :].expand(-1, a.shape[1], -1)] - 1).max().item() < 10 * tol
    for i, a in enumerate(va):
        assert maxdiff(a, t(f"vis_att{i}")) < tol, ("vis", i)
    for i, a in enumerate(xa):
        ref = t(f"x_att{i}")
        m = real[:, None, :, None].expand_as(ref)
        assert maxdiff(a[m], ref[m]) < tol, ("cross", i)


@pytest.mark.parametrize("pack", [False, True])
def test_attention_probs_match_reference(pack, monkeypatch):
    monkeypatch.setenv("XL_PACK_LANG", "1" if pack else "0")
    g = load_golden("attn_tiny")
    eng = make_vismask_engine(g, FakeOps(torch.float32))
    assert eng.packed == pack
    check_attention_probs(g, eng, 2e-6)
