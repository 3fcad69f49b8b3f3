"""CPU oracle for the X-LXMERT hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) restatement of the arithmetic the
reference executes on the path named by BASELINE.json's north_star.  It is the
checker for the HIP kernels; nothing in the product package (`xlxmert_amd/`)
imports it.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import or execute it.

Where the arithmetic lives.  The reference (`/root/reference/x-lxmert/src/lxrt/modeling.py`)
imports its encoder from the un-vendored dependency `transformers`
(pin: transformers==4.1.1, `/root/reference/requirements.txt:11`; the container
has 5.15.0).  Citations below:
  ref:<file>:<lines>  -> file under /root/reference/
  HF:<lines>          -> transformers/models/lxmert/modeling_lxmert.py (5.15.0 as
                         installed; the 4.1.1 math is identical on this path, see
                         SURVEY.md section 8c)

Parity pin.  The reference ships no tests or golden vectors ("parity unpinned" by
the reference itself).  This restatement is pinned against outputs of the
reference's own classes imported in the build container
(`oracle/gen_golden.py` -> `tests/golden/*.npz`, checked by
`tests/test_oracle_golden.py`).

All functions take a flat state dict (`{name: tensor}` with the reference's key
layout) and are differentiable through torch autograd, so
the same restatement yields the gradient goldens.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class OracleConfig:
    """Subset of LxmertConfig (HFcfg:72-101) that the path reads."""
    vocab_size: int = 30522
    hidden_size: int = 768
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    l_layers: int = 9
    x_layers: int = 5
    r_layers: int = 5
    visual_feat_dim: int = 2048
    visual_pos_dim: int = 4
    num_clusters: int = 10000
    layer_norm_eps: float = 1e-12       # HF:188,273,335,460,464,580 (literal 1e-12)


# --------------------------------------------------------------------------- inputs
def box_position(grid_size: int = 8) -> np.ndarray:
    """ref:x-lxmert/src/utils.py:75-85 -- (x0,y0,x1,y1) of each grid cell, row-major."""
    n = grid_size * grid_size
    boxes = np.zeros((n, 4), dtype=np.float32)
    for i in range(grid_size):
        for j in range(grid_size):
            boxes[i * grid_size + j] = (j / grid_size, i / grid_size,
                                        (j + 1) / grid_size, (i + 1) / grid_size)
    return boxes


def extended_mask(attention_mask: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """HF:758-766 -- [B,L] {0,1} -> additive [B,1,1,L]: (1-m)*finfo(dtype).min."""
    m = attention_mask[:, None, None, :].to(dtype)
    return (1.0 - m) * torch.finfo(dtype).min


# --------------------------------------------------------------------------- storage-precision emulation (analysis only)
# The HIP engine's bf16 mode stores every activation and activation gradient in bf16 (fp32 accumulation inside a contraction,
# fp32 LayerNorm statistics, fp32 weight gradients).  EMU["mode"] = "bf16" replays that on the CPU: every value the engine
# WRITES TO MEMORY goes through round-to-bf16 on the way forward and its gradient through round-to-bf16 on the way back;
# contraction operands (weights too) are bf16.  "bf16_fp32res" is SURVEY section 7's precision mode: the same, except that the
# residual stream -- the pre-LayerNorm sums and the LayerNorm outputs as RESIDUAL operands -- stays fp32 (a contraction still
# reads its bf16 rounding).  None (default): exact fp32, the oracle proper; the golden fixtures are only ever compared in that
# mode.  tests/test_precision_emulation_cpu.py uses the two modes to say how much of the bf16 path's gradient error an fp32
# residual stream would remove.
EMU = {"mode": None}


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _act(x):
    """a value the engine stores as an activation (and whose gradient it stores as one)"""
    return _RoundSTE.apply(x) if EMU["mode"] else x


def _res(x):
    """a value of the residual stream: bf16 in the engine as built, fp32 in the blueprint's precision mode"""
    return _RoundSTE.apply(x) if EMU["mode"] == "bf16" else x


# --------------------------------------------------------------------------- blocks
def _linear(sd, prefix, x):
    if EMU["mode"]:          # bf16 operands (the compute copy of the weight), fp32 accumulate, fp32 bias
        w = sd[prefix + ".weight"]
        return F.linear(_RoundSTE.apply(x), w + (w.to(torch.bfloat16).to(w.dtype) - w).detach(), sd[prefix + ".bias"])
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def _layer_norm(sd, prefix, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def embeddings(sd, cfg, input_ids, token_type_ids, prefix="bert.embeddings", inputs_embeds=None):
    """HF:191-214 -- LN(word[ids] + pos[0..L-1] + type[tt]); dropout is identity (eval).  `inputs_embeds` [B, L, d] replaces
    the word lookup (HF:199-206)."""
    B, L = token_type_ids.shape
    pos_ids = torch.arange(L, device=token_type_ids.device)[None, :].expand(B, L)
    # all three tables are nn.Embedding(..., padding_idx=0) (HF:184-186): forward is a plain gather,
    # but row 0 of each table receives NO gradient -- i.e. [PAD] word, position 0 ([CLS]) and
    # token-type 0 (every token; ref lxmert_pretrain.py:200) are frozen rows.
    word = inputs_embeds if inputs_embeds is not None else F.embedding(input_ids, sd[prefix + ".word_embeddings.weight"], padding_idx=0)
    e = (word
         + F.embedding(pos_ids, sd[prefix + ".position_embeddings.weight"], padding_idx=0)
         + F.embedding(token_type_ids, sd[prefix + ".token_type_embeddings.weight"], padding_idx=0))
    return _res(_layer_norm(sd, prefix + ".LayerNorm", e, cfg.layer_norm_eps))


def attention(sd, cfg, prefix, hidden, context, mask_add=None):
    """HF:238-266 -- multi-head softmax(QK^T/sqrt(dh) + mask) V, heads merged."""
    H = cfg.num_attention_heads
    dh = cfg.hidden_size // H
    B, nq, _ = hidden.shape
    nk = context.shape[1]
    q = _act(_linear(sd, prefix + ".query", hidden)).view(B, nq, H, dh).transpose(1, 2)
    k = _act(_linear(sd, prefix + ".key", context)).view(B, nk, H, dh).transpose(1, 2)
    v = _act(_linear(sd, prefix + ".value", context)).view(B, nk, H, dh).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
    if mask_add is not None:
        s = s + mask_add
    p = _act(F.softmax(s, dim=-1))                      # (the probabilities are a bf16 MFMA operand in the engine)
    o = _act(torch.matmul(p, v))
    return o.permute(0, 2, 1, 3).contiguous().view(B, nq, H * dh)


def attention_output(sd, cfg, prefix, x, input_tensor):
    """HF:276-280 -- LN(dense(x) + input)."""
    return _res(_layer_norm(sd, prefix + ".LayerNorm", _res(_linear(sd, prefix + ".dense", x) + input_tensor),
                            cfg.layer_norm_eps))


def self_attention_layer(sd, cfg, prefix, x, mask_add):
    """HF:304-316."""
    return attention_output(sd, cfg, prefix + ".output",
                            attention(sd, cfg, prefix + ".self", x, x, mask_add), x)


def cross_attention_layer(sd, cfg, prefix, x, ctx, ctx_mask_add):
    """HF:289-295."""
    return attention_output(sd, cfg, prefix + ".output",
                            attention(sd, cfg, prefix + ".att", x, ctx, ctx_mask_add), x)


def intermediate(sd, cfg, prefix, x):
    """HF:325-328 -- exact-erf GELU (ACT2FN['gelu'])."""
    return _act(F.gelu(_linear(sd, prefix + ".dense", x)))


def output(sd, cfg, prefix, h, input_tensor):
    """HF:338-342."""
    return _res(_layer_norm(sd, prefix + ".LayerNorm", _res(_linear(sd, prefix + ".dense", h) + input_tensor),
                            cfg.layer_norm_eps))


def lxmert_layer(sd, cfg, prefix, x, mask_add=None):
    """HF:352-358 -- self-attention block then FFN block."""
    a = self_attention_layer(sd, cfg, prefix + ".attention", x, mask_add)
    return output(sd, cfg, prefix + ".output", intermediate(sd, cfg, prefix + ".intermediate", a), a)


def x_layer(sd, cfg, prefix, lang, lang_mask_add, vis, vis_mask_add=None):
    """HF:417-449 -- ONE shared `visual_attention` used in both directions, both reading
    the pre-update inputs (HF:386-397); then per-stream self-attention and FFN."""
    l1 = cross_attention_layer(sd, cfg, prefix + ".visual_attention", lang, vis, vis_mask_add)
    v1 = cross_attention_layer(sd, cfg, prefix + ".visual_attention", vis, lang, lang_mask_add)
    l2 = self_attention_layer(sd, cfg, prefix + ".lang_self_att", l1, lang_mask_add)
    v2 = self_attention_layer(sd, cfg, prefix + ".visn_self_att", v1, vis_mask_add)
    l3 = output(sd, cfg, prefix + ".lang_output", intermediate(sd, cfg, prefix + ".lang_inter", l2), l2)
    v3 = output(sd, cfg, prefix + ".visn_output", intermediate(sd, cfg, prefix + ".visn_inter", v2), v2)
    return l3, v3


def visual_feature_encoder(sd, cfg, visual_feats, visual_pos, prefix="bert.encoder.visn_fc"):
    """HF:468-476 -- (LN(Wf f + bf) + LN(Wp p + bp)) / 2."""
    x = _layer_norm(sd, prefix + ".visn_layer_norm", _linear(sd, prefix + ".visn_fc", visual_feats),
                    cfg.layer_norm_eps)
    y = _layer_norm(sd, prefix + ".box_layer_norm", _linear(sd, prefix + ".box_fc", visual_pos),
                    cfg.layer_norm_eps)
    return _res((x + y) / 2)


def encoder(sd, cfg, lang, lang_mask_add, visual_feats, visual_pos, vis_mask_add=None,
            prefix="bert.encoder", return_hidden=False):
    """HF:498-557 -- visn_fc; l_layers x lang; r_layers x vis; x_layers x cross."""
    vis = visual_feature_encoder(sd, cfg, visual_feats, visual_pos, prefix + ".visn_fc")
    lang_h, vis_h = [], []
    for i in range(cfg.l_layers):
        lang = lxmert_layer(sd, cfg, f"{prefix}.layer.{i}", lang, lang_mask_add)
        lang_h.append(lang)
    for i in range(cfg.r_layers):
        vis = lxmert_layer(sd, cfg, f"{prefix}.r_layers.{i}", vis, vis_mask_add)
        vis_h.append(vis)
    for i in range(cfg.x_layers):
        lang, vis = x_layer(sd, cfg, f"{prefix}.x_layers.{i}", lang, lang_mask_add, vis, vis_mask_add)
        lang_h.append(lang)
        vis_h.append(vis)
    if return_hidden:
        return lang, vis, lang_h, vis_h
    return lang, vis


def pooler(sd, cfg, lang, prefix="bert.pooler"):
    """HF:566-572 -- tanh(dense(lang[:,0]))."""
    return torch.tanh(_linear(sd, prefix + ".dense", lang[:, 0]))


def lxmert_model(sd, cfg, input_ids, visual_feats, visual_pos, attention_mask=None,
                 token_type_ids=None, prefix="bert", return_hidden=False, visual_attention_mask=None, inputs_embeds=None):
    """HF:691-822 -- returns (language_output, vision_output, pooled_output).
    `visual_attention_mask` (HF:760-770: the same additive extension as the language mask, applied to the visual KEYS of the
    visual self-attention and of the language -> vision cross-attention) is None in every reference caller
    (ref:x-lxmert/src/pretrain/lxmert_pretrain.py:207); pinned by tests/golden/vismask_tiny.npz."""
    dtype = sd[prefix + ".embeddings.word_embeddings.weight"].dtype
    shape = input_ids.shape if input_ids is not None else inputs_embeds.shape[:-1]          # HF:735-741
    if attention_mask is None:
        attention_mask = torch.ones(shape, dtype=torch.long)
    if token_type_ids is None:
        token_type_ids = torch.zeros(shape, dtype=torch.long)
    mask_add = extended_mask(attention_mask, dtype)
    emb = embeddings(sd, cfg, input_ids, token_type_ids, prefix + ".embeddings", inputs_embeds)
    vis_mask_add = extended_mask(visual_attention_mask, dtype) if visual_attention_mask is not None else None
    out = encoder(sd, cfg, emb, mask_add, visual_feats.to(dtype), visual_pos.to(dtype), vis_mask_add,
                  prefix + ".encoder", return_hidden)
    lang, vis = out[0], out[1]
    pooled = pooler(sd, cfg, lang, prefix + ".pooler")
    if return_hidden:
        return lang, vis, pooled, out[2], out[3]
    return lang, vis, pooled


# --------------------------------------------------------------------------- language pretraining heads
def lm_prediction_head(sd, cfg, lang, prefix="cls.predictions"):
    """HF:589-599 LxmertLMPredictionHead -- decoder(LN(gelu(dense(x)))) + bias, decoder weight TIED to the word embeddings
    (transformers 4.1.1 ctor: `self.decoder.weight = lxmert_model_embedding_weights`; ref lxrt/modeling.py:86)."""
    h = _layer_norm(sd, prefix + ".transform.LayerNorm", F.gelu(_linear(sd, prefix + ".transform.dense", lang)), 1e-12)
    return F.linear(h, sd["bert.embeddings.word_embeddings.weight"]) + sd[prefix + ".bias"]


def qa_loss_term(sd, cfg, pooled, qa_labels):
    """ref:x-lxmert/src/lxrt/modeling.py:292-304: `if self.task_qa:` (the model was built with the QA head, ref :89-90) the
    answer-head cross-entropy over `label_dict['qa_labels']` (ignore_index -100) is added to total_loss in EVERY task
    branch -- the condition is on the model, not on the `task` argument."""
    score = visual_answer_head(sd, cfg, pooled)
    return F.cross_entropy(score.view(-1, score.shape[-1]), qa_labels.view(-1)), score


def xlxmert_word_mask_forward(sd, cfg, input_ids, visual_pos, attention_mask, cluster_ids, word_labels, token_type_ids=None,
                              qa_labels=None):
    """ref:x-lxmert/src/lxrt/modeling.py:154-225, task == 'word_mask': un-masked centroid features in, MLM CE (ignore -100)."""
    feats = codebook_features(sd, cluster_ids, None)
    lang, vis, pooled = lxmert_model(sd, cfg, input_ids, feats, visual_pos, attention_mask, token_type_ids)
    scores = lm_prediction_head(sd, cfg, lang)
    lm_loss = F.cross_entropy(scores.view(-1, cfg.vocab_size), word_labels.view(-1))
    out = {"lm_loss": lm_loss, "total_loss": lm_loss, "scores": scores}
    if qa_labels is not None:
        out["qa_loss"], out["qa_score"] = qa_loss_term(sd, cfg, pooled, qa_labels)
        out["total_loss"] = lm_loss + out["qa_loss"]
    return out


def xlxmert_matched_forward(sd, cfg, input_ids, visual_pos, attention_mask, cluster_ids, matched_labels, token_type_ids=None,
                            qa_labels=None):
    """ref:x-lxmert/src/lxrt/modeling.py:154-235, task == 'matched': seq_relationship(pooled_output), 2-way CE (HF:648-657)."""
    feats = codebook_features(sd, cluster_ids, None)
    lang, vis, pooled = lxmert_model(sd, cfg, input_ids, feats, visual_pos, attention_mask, token_type_ids)
    score = _linear(sd, "cls.seq_relationship", pooled)
    loss = F.cross_entropy(score.view(-1, 2), matched_labels.view(-1))
    out = {"matched_loss": loss, "total_loss": loss, "score": score}
    if qa_labels is not None:
        out["qa_loss"], out["qa_score"] = qa_loss_term(sd, cfg, pooled, qa_labels)
        out["total_loss"] = loss + out["qa_loss"]
    return out


def xlxmert_qa_forward(sd, cfg, input_ids, visual_pos, attention_mask, cluster_ids, qa_labels, token_type_ids=None):
    """ref:x-lxmert/src/lxrt/modeling.py:154-210, 292-306, task == 'qa' on a task_qa model: un-masked centroid features in
    (the [MASK] substitution is for task == 'vis_mask' only, :190-193), total_loss = qa_loss; qa_pred = argmax (:300)."""
    feats = codebook_features(sd, cluster_ids, None)
    lang, vis, pooled = lxmert_model(sd, cfg, input_ids, feats, visual_pos, attention_mask, token_type_ids)
    loss, score = qa_loss_term(sd, cfg, pooled, qa_labels)
    return {"qa_loss": loss, "total_loss": loss, "qa_score": score, "qa_pred": score.argmax(1)}


def make_qa_state_dict(cfg, num_qa_labels, seed, perturb=True):
    """make_cls_state_dict(cfg, seed) + the deterministic answer head of make_vqa_state_dict (same recipe, seed + 7)."""
    sd = make_cls_state_dict(cfg, seed, perturb)
    vq = make_vqa_state_dict(cfg, num_qa_labels, seed, perturb)
    for name, _ in answer_head_shapes(cfg, num_qa_labels):
        sd[name] = vq[name]
    return sd


def make_qa_labels(num_qa_labels, B, seed):
    """one answer id per example, ~1 in 4 without an answer (-100: ignored by the loss, ref lxmert_pretrain.py:186-190)."""
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, num_qa_labels, size=B, dtype=np.int64)
    lab[rng.random(B) < 0.25] = -100
    if (lab == -100).all():
        lab[0] = 0
    return torch.from_numpy(lab)


def cls_head_shapes(cfg):
    d = cfg.hidden_size
    return [("cls.predictions.transform.dense.weight", (d, d)), ("cls.predictions.transform.dense.bias", (d,)),
            ("cls.predictions.transform.LayerNorm.weight", (d,)), ("cls.predictions.transform.LayerNorm.bias", (d,)),
            ("cls.predictions.bias", (cfg.vocab_size,)),
            ("cls.seq_relationship.weight", (2, d)), ("cls.seq_relationship.bias", (2,))]


def make_cls_state_dict(cfg, seed, perturb=True):
    """make_state_dict(cfg, seed) + deterministic `cls.*` heads (seed + 13); `cls.predictions.decoder.weight` is the word
    embedding matrix (tied)."""
    sd = make_state_dict(cfg, seed, perturb)
    rng = np.random.default_rng(seed + 13)
    for name, shape in cls_head_shapes(cfg):
        if name.endswith("LayerNorm.weight"):
            w = np.ones(shape, np.float32) + (0.1 * rng.standard_normal(shape, dtype=np.float32) if perturb else 0.0)
        elif len(shape) == 1:
            w = 0.05 * rng.standard_normal(shape, dtype=np.float32) if perturb else np.zeros(shape, np.float32)
        else:
            w = 0.02 * rng.standard_normal(shape, dtype=np.float32)
        sd[name] = torch.from_numpy(np.asarray(w, np.float32))
    sd["cls.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    return sd


def make_lang_task_labels(cfg, input_ids, seed):
    """word_labels: the token id at ~25 % of the real, non-special positions, -100 elsewhere (the loss ignores -100; the
    reference's data code writes -1, which its own CrossEntropyLoss would reject); matched_labels: 0/1 per example."""
    rng = np.random.default_rng(seed)
    ids = input_ids.numpy()
    lab = np.full(ids.shape, -100, np.int64)
    for b in range(ids.shape[0]):
        real = np.flatnonzero(ids[b] > 0)[1:-1]
        pick = real[rng.random(len(real)) < 0.25]
        if len(pick) == 0 and len(real):
            pick = real[:1]
        lab[b, pick] = ids[b, pick]
    return torch.from_numpy(lab), torch.from_numpy(rng.integers(0, 2, size=ids.shape[0], dtype=np.int64))


# --------------------------------------------------------------------------- iterative sampler
def sample_codes_nar(sd, cfg, input_ids, n_steps, grid_size=8, return_trace=False):
    """ref:x-lxmert/src/tasks/imggen_model.py:169-243 (sample_image_NAR, up to the hand-off to the frozen GAN generator).
    Returns (code [B,V,F], code_ids [B,V], pred_prob [B,V]); the code tensor only ever holds mask_feat or centroid rows,
    so it is tracked here both ways."""
    B = input_ids.shape[0]
    V = grid_size ** 2
    dtype = sd["vis_emb.weight"].dtype
    visual_pos = torch.from_numpy(box_position(grid_size)).unsqueeze(0).expand(B, -1, -1).to(dtype)
    trace = []
    with torch.no_grad():
        for i in range(n_steps):
            n_mask = int((n_steps - i) / n_steps * V)                                   # :201-202
            if i == 0:
                vis_mask = torch.ones(B, V, dtype=torch.long)                           # :204-206
                code = torch.zeros(B, V, cfg.visual_feat_dim, dtype=dtype)
                code_ids = torch.zeros(B, V, dtype=torch.long)
            else:
                _, lowest_arg = pred_prob.topk(n_mask, dim=1, largest=False)            # :208-212
                vis_mask = torch.zeros(B, V, dtype=torch.long)
                vis_mask.scatter_(1, lowest_arg, 1)
            m3 = vis_mask.view(B, V, 1).bool()
            code = torch.where(m3, sd["mask_feat"].view(1, 1, -1).to(dtype), code)      # :215-218
            _, vis, _ = lxmert_model(sd, cfg, input_ids, code, visual_pos, input_ids > 0)   # :221-227
            _, obj = visual_obj_head(sd, cfg, vis)                                      # :228-229
            pred_prob, pred_code_id = torch.softmax(obj, dim=2).max(dim=2)              # :232-235
            code = torch.where(m3, sd["vis_emb.weight"][pred_code_id], code)            # :238-243
            code_ids = torch.where(vis_mask.bool(), pred_code_id, code_ids)
            if return_trace:
                trace.append((vis_mask.clone(), pred_code_id.clone(), pred_prob.clone()))
    if return_trace:
        return code, code_ids, pred_prob, trace
    return code, code_ids, pred_prob


def sample_codes_ar(sd, cfg, input_ids, n_steps=None, grid_size=8, mode="confidence", positions=None):
    """ref:x-lxmert/src/tasks/imggen_model.py:49-153 (sample_image_AR up to the GAN hand-off).  mode: "confidence"
    (position_confidence, the default), "tlbr" (position_TLBR) or "random" (position_random; `positions` = the order
    popped from the END of the host-shuffled list, ref :78-91, 104-108)."""
    B = input_ids.shape[0]
    V = grid_size ** 2
    n_steps = V if n_steps is None else n_steps
    dtype = sd["vis_emb.weight"].dtype
    visual_pos = torch.from_numpy(box_position(grid_size)).unsqueeze(0).expand(B, -1, -1).to(dtype)
    visited = torch.zeros(B, V)
    positions = list(positions) if positions is not None else None
    with torch.no_grad():
        for i in range(n_steps):
            if i == 0:
                vis_mask = torch.ones(B, V, dtype=torch.long)
                code = torch.zeros(B, V, cfg.visual_feat_dim, dtype=dtype)
            if mode == "random":
                cur = positions.pop() % V
                vis_mask[:, cur] = 1
            elif mode == "tlbr":
                cur = i
            code = torch.where(vis_mask.view(B, V, 1).bool(), sd["mask_feat"].view(1, 1, -1).to(dtype), code)
            _, vis, _ = lxmert_model(sd, cfg, input_ids, code, visual_pos, input_ids > 0)
            _, obj = visual_obj_head(sd, cfg, vis)
            pred_prob, pred_code_id = torch.softmax(obj, dim=2).max(dim=2)
            pred_code = sd["vis_emb.weight"][pred_code_id]
            if mode in ("tlbr", "random"):
                update = torch.zeros(B, V, dtype=torch.bool)
                update[:, cur] = True
                vis_mask[:, cur] = 0
            else:
                _p = pred_prob.masked_fill(visited.bool(), -10000)
                _, top_arg = _p.topk(1, dim=1, largest=True)
                update = torch.zeros(B, V, dtype=torch.long)
                update.scatter_(1, top_arg, 1)
                vis_mask.scatter_(1, top_arg, 0)
                visited.scatter_(1, top_arg, 1)
            code = torch.where(update.view(B, V, 1).bool(), pred_code, code)
    return code, vis_mask


# --------------------------------------------------------------------------- VQA / GQA fine-tune head
def visual_answer_head(sd, cfg, pooled, prefix="answer_head.logit_fc"):
    """HF:602-614 LxmertVisualAnswerHead -- Linear(d, 2d) -> GeLU -> LayerNorm(2d, eps=1e-12) -> Linear(2d, num_answers)."""
    h = F.gelu(_linear(sd, prefix + ".0", pooled))
    h = _layer_norm(sd, prefix + ".2", h, 1e-12)
    return _linear(sd, prefix + ".3", h)


def vqa_forward(sd, cfg, input_ids, visual_feats, visual_pos, attention_mask=None, targets=None, token_type_ids=None):
    """ref:x-lxmert/src/tasks/vqa_model.py:22-72 (logit = answer_head(pooled_output) on REAL grid features) and
    ref:x-lxmert/src/tasks/vqa.py:166-187 (attention_mask = input_ids > 0; loss = BCEWithLogitsLoss()(logit, target))."""
    if attention_mask is None:
        attention_mask = input_ids > 0
    lang, vis, pooled = lxmert_model(sd, cfg, input_ids, visual_feats, visual_pos, attention_mask, token_type_ids)
    logit = visual_answer_head(sd, cfg, pooled)
    out = {"logit": logit, "pooled": pooled}
    if targets is not None:
        out["loss"] = F.binary_cross_entropy_with_logits(logit, targets)
    return out


def answer_head_shapes(cfg, num_answers):
    d = cfg.hidden_size
    p = "answer_head.logit_fc"
    return [(p + ".0.weight", (2 * d, d)), (p + ".0.bias", (2 * d,)), (p + ".2.weight", (2 * d,)), (p + ".2.bias", (2 * d,)),
            (p + ".3.weight", (num_answers, 2 * d)), (p + ".3.bias", (num_answers,))]


def make_vqa_state_dict(cfg, num_answers, seed, perturb=True):
    """encoder weights of make_state_dict(cfg, seed) + a deterministic answer head (same recipe, seed + 7)."""
    sd = make_state_dict(cfg, seed, perturb)
    rng = np.random.default_rng(seed + 7)
    for name, shape in answer_head_shapes(cfg, num_answers):
        if name.endswith(".2.weight"):
            w = np.ones(shape, np.float32) + (0.1 * rng.standard_normal(shape, dtype=np.float32) if perturb else 0.0)
        elif len(shape) == 1:
            w = 0.05 * rng.standard_normal(shape, dtype=np.float32) if perturb else np.zeros(shape, np.float32)
        else:
            w = 0.02 * rng.standard_normal(shape, dtype=np.float32)
        sd[name] = torch.from_numpy(np.asarray(w, np.float32))
    return sd


def make_vqa_inputs(cfg, num_answers, seed, B, L=20, grid=8):
    """Synthetic VQA batch (ref vqa_data.py:225-262): real grid features relu(N(0,1)) [B,V,F], boxes, word ids with PAD = 0,
    soft target scores in {0, .3, .6, .9, 1} on a few answers per question."""
    base = make_inputs(cfg, seed, B, L, grid)
    rng = np.random.default_rng(seed + 11)
    V = grid * grid
    feats = np.maximum(rng.standard_normal((B, V, cfg.visual_feat_dim), dtype=np.float32), 0.0)
    tgt = np.zeros((B, num_answers), np.float32)
    for b in range(B):
        for a in rng.permutation(num_answers)[: int(rng.integers(1, 4))]:
            tgt[b, a] = rng.choice(np.array([0.3, 0.6, 0.9, 1.0], np.float32))
    return {"input_ids": base["input_ids"], "attention_mask": base["input_ids"] > 0, "visual_pos": base["visual_pos"],
            "visual_feats": torch.from_numpy(feats), "targets": torch.from_numpy(tgt)}


# --------------------------------------------------------------------------- NLVR2 fine-tune head
def nlvr2_forward(sd, cfg, input_ids, visual_feats, visual_pos, attention_mask=None, labels=None, token_type_ids=None):
    """ref:x-lxmert/src/tasks/nlvr2_model.py:50-93: visual_feats [P, 2, V, F] / visual_pos [P, 2, V, 4] are flattened to 2P
    encoder rows (input_ids already holds every statement twice, `:35-48`), pooled_output [2P, d] is viewed as [P, 2d] and
    fed to the answer head; ref:x-lxmert/src/tasks/nlvr2.py:72 trains it with CrossEntropyLoss over the 2 classes.
    The published class builds `logit_fc` with a d-wide input and then calls an `answer_head` it never defines (SURVEY
    App. A #14); the head used here is the one its forward needs: Linear(2d, 2d) -> GeLU -> LayerNorm(2d) -> Linear(2d, 2),
    HF's LxmertVisualAnswerHead with a 2d-wide first Linear, under the name the forward calls."""
    P, two, V, Fd = visual_feats.shape
    assert two == 2
    if attention_mask is None:
        attention_mask = input_ids > 0
    lang, vis, pooled = lxmert_model(sd, cfg, input_ids, visual_feats.reshape(P * 2, V, Fd), visual_pos.reshape(P * 2, V, -1),
                                     attention_mask, token_type_ids)
    logit = visual_answer_head(sd, cfg, pooled.reshape(P, 2 * cfg.hidden_size))
    out = {"logit": logit, "pooled": pooled}
    if labels is not None:
        out["loss"] = F.cross_entropy(logit, labels)
    return out


def make_nlvr2_state_dict(cfg, seed, perturb=True):
    """make_vqa_state_dict(cfg, 2, seed) with the head's first Linear widened to (2d, 2d) (drawn from seed + 9)."""
    sd = make_vqa_state_dict(cfg, 2, seed, perturb)
    d = cfg.hidden_size
    rng = np.random.default_rng(seed + 9)
    sd["answer_head.logit_fc.0.weight"] = torch.from_numpy(0.02 * rng.standard_normal((2 * d, 2 * d), dtype=np.float32))
    return sd


def make_nlvr2_inputs(cfg, seed, P, L=20, grid=8):
    """Synthetic NLVR2 batch (ref nlvr2_data.py: one statement, two images, a True/False label): P statements -> input_ids
    [2P, L] with every statement repeated for its two images, features [P, 2, V, F], boxes [P, 2, V, 4], labels [P]."""
    base = make_inputs(cfg, seed, P, L, grid)
    rng = np.random.default_rng(seed + 13)
    V = grid * grid
    feats = np.maximum(rng.standard_normal((P, 2, V, cfg.visual_feat_dim), dtype=np.float32), 0.0)
    ids = base["input_ids"].repeat_interleave(2, dim=0)
    pos = base["visual_pos"].unsqueeze(1).repeat(1, 2, 1, 1).contiguous()
    labels = torch.from_numpy(rng.integers(0, 2, size=(P,)).astype(np.int64))
    return {"input_ids": ids, "attention_mask": ids > 0, "visual_pos": pos, "visual_feats": torch.from_numpy(feats),
            "labels": labels}


# --------------------------------------------------------------------------- head + losses
def head_transform(sd, cfg, x, prefix="obj_predict_head.transform"):
    """HF:582-586 -- LN(gelu(dense(x)))."""
    return _act(_layer_norm(sd, prefix + ".LayerNorm", _act(F.gelu(_linear(sd, prefix + ".dense", x))),
                            cfg.layer_norm_eps))


def visual_obj_head(sd, cfg, vis, prefix="obj_predict_head"):
    """ref:x-lxmert/src/lxrt/modeling.py:38-53 (cluster mode) -- returns (feat, obj_logits).
    out_cluster.weight is the frozen centroid matrix (tied, ref :150-151)."""
    h = head_transform(sd, cfg, vis, prefix + ".transform")
    feat = _act(_linear(sd, prefix + ".linear_feat", h))
    if EMU["mode"]:
        c = sd[prefix + ".out_cluster.weight"]
        obj = F.linear(feat, c.to(torch.bfloat16).to(c.dtype), sd[prefix + ".out_cluster.bias"])       # fp32 logits from bf16 operands
    else:
        obj = F.linear(feat, sd[prefix + ".out_cluster.weight"], sd[prefix + ".out_cluster.bias"])
    return feat, obj


def codebook_features(sd, cluster_ids, vis_mask=None):
    """ref:x-lxmert/src/lxrt/modeling.py:185-193 -- centroid lookup + [MASK]-feature substitution."""
    feats = sd["vis_emb.weight"][cluster_ids]
    if vis_mask is not None:
        B, V = cluster_ids.shape
        feats = torch.where(vis_mask.view(B, V, 1).bool(),
                            sd["mask_feat"].view(1, 1, -1).to(feats.dtype), feats)
    return feats


def vis_mask_losses(cfg, feat, obj, obj_labels, feat_labels, vis_mask):
    """ref:x-lxmert/src/lxrt/modeling.py:237-290 -- CE(mean over labels != -100) + masked
    SmoothL1 feature regression."""
    B, V, K = obj.shape
    obj_loss = F.cross_entropy(obj.view(B * V, K), obj_labels.flatten())          # ignore_index=-100
    fl = F.smooth_l1_loss(feat, feat_labels.view(B, V, -1), reduction="none").mean(dim=2)
    fl = (fl * vis_mask).sum(dim=1)
    n_mask = vis_mask.sum(dim=1).clamp(min=1)
    feat_loss = (fl / n_mask).mean()
    return obj_loss, feat_loss


def xlxmert_vis_mask_forward(sd, cfg, input_ids, visual_pos, attention_mask, cluster_ids, vis_mask,
                             obj_labels, feat_labels=None, token_type_ids=None, return_all=False, qa_labels=None):
    """ref:x-lxmert/src/lxrt/modeling.py:154-308, task == 'vis_mask'.
    `feat_labels` defaults to the un-masked centroid features (ref lxmert_pretrain.py:177-179)."""
    feats = codebook_features(sd, cluster_ids, vis_mask)
    lang, vis, pooled = lxmert_model(sd, cfg, input_ids, feats, visual_pos, attention_mask, token_type_ids)
    feat, obj = visual_obj_head(sd, cfg, vis)
    if feat_labels is None:
        feat_labels = sd["vis_emb.weight"][cluster_ids]
    obj_loss, feat_loss = vis_mask_losses(cfg, feat, obj, obj_labels, feat_labels, vis_mask)
    out = {"obj_loss": obj_loss, "feat_loss": feat_loss, "vis_loss": obj_loss + feat_loss,
           "total_loss": obj_loss + feat_loss}
    if qa_labels is not None:               # task_qa model (ref lxrt/modeling.py:292-304)
        out["qa_loss"], out["qa_score"] = qa_loss_term(sd, cfg, pooled, qa_labels)
        out["total_loss"] = out["total_loss"] + out["qa_loss"]
    if return_all:
        out.update(lang=lang, vis=vis, pooled=pooled, feat=feat, obj=obj)
    return out


# --------------------------------------------------------------------------- step semantics
def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics (ref lxmert_pretrain.py:343-353):
    total L2 norm; scale = max_norm/(norm+1e-6) clamped to <= 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, [g * coef for g in grads]


def adamw_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0,
                 correct_bias=True):
    """transformers==4.1.1 `optimization.AdamW.step` (the class no longer exists in 5.15.0;
    SURVEY.md section 8a row A16).  eps is added to sqrt(v) BEFORE bias correction; decoupled
    weight decay is applied AFTER the Adam update with the plain lr.  Returns (p, m, v)."""
    m = m * beta1 + g * (1.0 - beta1)
    v = v * beta2 + g * g * (1.0 - beta2)
    denom = v.sqrt() + eps
    step_size = lr
    if correct_bias:
        step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p = p - step_size * (m / denom)
    if weight_decay > 0.0:
        p = p - lr * weight_decay * p
    return p, m, v


def linear_schedule(step, warmup_steps, total_steps):
    """transformers.get_linear_schedule_with_warmup lambda (ref lxmert_pretrain.py:138-139)."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))


# --------------------------------------------------------------------------- deterministic weights
def param_shapes(cfg: OracleConfig):
    """State-dict layout on the path, in a fixed order."""
    d, dff, F_, K = cfg.hidden_size, cfg.intermediate_size, cfg.visual_feat_dim, cfg.num_clusters
    out = [("mask_feat", (F_,)), ("vis_emb.weight", (K, F_)),
           ("bert.embeddings.word_embeddings.weight", (cfg.vocab_size, d)),
           ("bert.embeddings.position_embeddings.weight", (cfg.max_position_embeddings, d)),
           ("bert.embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, d)),
           ("bert.embeddings.LayerNorm.weight", (d,)), ("bert.embeddings.LayerNorm.bias", (d,)),
           ("bert.encoder.visn_fc.visn_fc.weight", (d, F_)), ("bert.encoder.visn_fc.visn_fc.bias", (d,)),
           ("bert.encoder.visn_fc.visn_layer_norm.weight", (d,)), ("bert.encoder.visn_fc.visn_layer_norm.bias", (d,)),
           ("bert.encoder.visn_fc.box_fc.weight", (d, cfg.visual_pos_dim)), ("bert.encoder.visn_fc.box_fc.bias", (d,)),
           ("bert.encoder.visn_fc.box_layer_norm.weight", (d,)), ("bert.encoder.visn_fc.box_layer_norm.bias", (d,))]

    def att(p, self_name):
        r = []
        for n in ("query", "key", "value"):
            r += [(f"{p}.{self_name}.{n}.weight", (d, d)), (f"{p}.{self_name}.{n}.bias", (d,))]
        r += [(f"{p}.output.dense.weight", (d, d)), (f"{p}.output.dense.bias", (d,)),
              (f"{p}.output.LayerNorm.weight", (d,)), (f"{p}.output.LayerNorm.bias", (d,))]
        return r

    def ffn(pi, po):
        return [(f"{pi}.dense.weight", (dff, d)), (f"{pi}.dense.bias", (dff,)),
                (f"{po}.dense.weight", (d, dff)), (f"{po}.dense.bias", (d,)),
                (f"{po}.LayerNorm.weight", (d,)), (f"{po}.LayerNorm.bias", (d,))]

    for stack, n in (("layer", cfg.l_layers), ("r_layers", cfg.r_layers)):
        for i in range(n):
            p = f"bert.encoder.{stack}.{i}"
            out += att(p + ".attention", "self") + ffn(p + ".intermediate", p + ".output")
    for i in range(cfg.x_layers):
        p = f"bert.encoder.x_layers.{i}"
        out += att(p + ".visual_attention", "att") + att(p + ".lang_self_att", "self") + att(p + ".visn_self_att", "self")
        out += ffn(p + ".lang_inter", p + ".lang_output") + ffn(p + ".visn_inter", p + ".visn_output")
    out += [("bert.pooler.dense.weight", (d, d)), ("bert.pooler.dense.bias", (d,)),
            ("obj_predict_head.transform.dense.weight", (d, d)), ("obj_predict_head.transform.dense.bias", (d,)),
            ("obj_predict_head.transform.LayerNorm.weight", (d,)), ("obj_predict_head.transform.LayerNorm.bias", (d,)),
            ("obj_predict_head.linear_feat.weight", (F_, d)), ("obj_predict_head.linear_feat.bias", (F_,)),
            ("obj_predict_head.out_cluster.bias", (K,))]
    return out


def make_state_dict(cfg: OracleConfig, seed: int, perturb: bool = True, dtype=torch.float32):
    """Deterministic weights from numpy's PCG64 stream (stable across numpy versions), so that
    fixtures need to store only the seed.  `perturb=False` gives the reference init
    (N(0,0.02) matrices, LN (1,0), zero biases, zero mask_feat); `perturb=True` also
    randomises biases / LN affine / mask_feat so that every term is exercised.
    Centroids = relu(N(0,1))."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape in param_shapes(cfg):
        if name == "vis_emb.weight":
            w = np.maximum(rng.standard_normal(shape, dtype=np.float32), 0.0)
        elif name.endswith("LayerNorm.weight") or name.endswith("layer_norm.weight"):
            w = np.ones(shape, np.float32)
            if perturb:
                w += 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif len(shape) == 1:
            w = np.zeros(shape, np.float32)
            if perturb:
                w += 0.05 * rng.standard_normal(shape, dtype=np.float32)
        else:
            w = 0.02 * rng.standard_normal(shape, dtype=np.float32)
        sd[name] = torch.from_numpy(w).to(dtype)
    sd["obj_predict_head.out_cluster.weight"] = sd["vis_emb.weight"]      # tied (ref modeling.py:150-151)
    return sd


def make_inputs(cfg: OracleConfig, seed: int, B: int, L: int = 20, grid: int = 8, ragged: bool = True,
                mask_mode: str = "predict"):
    """Synthetic batch with the reference's layout:
    input_ids [B,L] int64 ([CLS]=101 first, [SEP]=102 last real, 0 = PAD), attention_mask = ids > 0,
    cluster_ids [B,V] int64, vis_mask [B,V] bool (`predict`: n_masks ~ U{1..V} per example,
    ref lxmert_data.py:414-419; `bernoulli`: p=0.15, ref :457-458), obj_labels = cluster_ids with
    -100 where unmasked (ref lxmert_pretrain.py:163-166), visual_pos = box_position(grid)."""
    rng = np.random.default_rng(seed)
    V = grid * grid
    lo = min(1000, cfg.vocab_size // 2)
    ids = rng.integers(lo, cfg.vocab_size, size=(B, L), dtype=np.int64)
    ids[:, 0] = min(101, cfg.vocab_size - 2)
    for b in range(B):
        n = int(rng.integers(min(6, L), L + 1)) if ragged else L
        ids[b, n - 1] = min(102, cfg.vocab_size - 1)
        ids[b, n:] = 0
    cid = rng.integers(0, cfg.num_clusters, size=(B, V), dtype=np.int64)
    vm = np.zeros((B, V), dtype=bool)
    for b in range(B):
        if mask_mode == "predict":
            n = int(rng.integers(1, V + 1))
            vm[b, rng.permutation(V)[:n]] = True
        else:
            vm[b] = rng.random(V) < 0.15
    lab = cid.copy()
    lab[~vm] = -100
    pos = np.broadcast_to(box_position(grid)[None], (B, V, 4)).copy()
    t = torch.from_numpy
    return {"input_ids": t(ids), "attention_mask": t(ids > 0), "token_type_ids": torch.zeros(B, L, dtype=torch.long),
            "cluster_ids": t(cid), "vis_mask": t(vm), "obj_labels": t(lab), "visual_pos": t(pos)}
