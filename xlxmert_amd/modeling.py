"""Drop-in model classes for the X-LXMERT hot path, keeping the reference's API surface:

  * `LxmertModel.forward(input_ids, visual_feats, visual_pos, attention_mask, visual_attention_mask, token_type_ids,
    inputs_embeds, output_attentions, output_hidden_states, return_dict)` -> index-able
    `(language_output, vision_output, pooled_output)`                                   (HF:691-822)
  * `LxmertVisualObjHead.forward(hidden_states, out_keys=[]) -> {'feat', 'obj'}`        (ref lxrt/modeling.py:38-53)
  * `XLxmertForPretraining` with `.bert`, `.obj_predict_head`, `.mask_feat`, `.vis_emb`, `set_visual_embedding()`,
    `forward(..., cluster_ids, vis_mask, label_dict, task='vis_mask') -> dict of losses` (ref lxrt/modeling.py:56-308)
  * `state_dict()` / `load_state_dict()` with the reference's key layout, incl. the DDP
    `module.` prefix convention of the published checkpoints (ref utils.py:42-49)
  * legacy aliases `LXRTEncoder`, `LXRTModel` for the original LXMERT names used by BASELINE.json

All parameters are views into one flat buffer (params.ParamStore); compute runs through engine.Engine on the HIP kernels.
Autograd integration: each forward is ONE torch.autograd.Function whose backward runs the engine's hand-derived backward
and deposits parameter gradients directly into `param.grad` (views of the flat gradient buffer).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .config import XLxmertConfig
from .engine import Engine
from .ops import HipOps
from .params import ParamStore


class LxmertModelOutput(tuple):
    """Index-able (language_output, vision_output, pooled_output) with the HF field names."""

    def __new__(cls, lang, vis, pooled):
        obj = super().__new__(cls, (lang, vis, pooled))
        obj.language_output, obj.vision_output, obj.pooled_output = lang, vis, pooled
        return obj


class _Named(nn.Module):
    """A module whose parameters are views of the ParamStore (registered under the reference's attribute names)."""

    def _bind(self, store, prefix, names):
        for full in names:
            assert full.startswith(prefix)
            rel = full[len(prefix):].lstrip(".")
            mod = self
            parts = rel.split(".")
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, nn.Module())
                mod = mod._modules[p]
            param = nn.Parameter(store.view(full), requires_grad=True)
            param.grad = store.gview(full)
            mod.register_parameter(parts[-1], param)


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, anchor):
        eng = model._engine
        lang, vis, pooled = eng.encoder_forward(want_pooled=True)
        ctx.model = model
        B, L, V, d = eng.B, eng.L, eng.V, eng.d
        return lang.view(B, L, d).clone(), vis.view(B, V, d).clone(), pooled.view(B, d).clone()

    @staticmethod
    def backward(ctx, d_lang, d_vis, d_pooled):
        model = ctx.model
        eng = model._engine
        if d_pooled is not None and d_pooled.abs().sum().item() != 0:
            raise NotImplementedError("gradient through pooled_output: N1 (VQA head) is a later row of the scope table")
        ML = eng.ML
        eng.GA.zero_()
        if d_lang is not None:
            eng.GA[:ML].copy_(d_lang.reshape(ML, eng.d))
        if d_vis is not None:
            eng.GA[ML:].copy_(d_vis.reshape(eng.MV, eng.d))
        eng.encoder_backward(have_lang_grad=True)      # vec-type gradients accumulate: call model.zero_grad() per step
        return None, None


class LxmertModel(_Named):
    """Embeddings + LxmertEncoder (9 language / 5 visual / 5 cross layers) + pooler (HF:675-822)."""

    def __init__(self, config: XLxmertConfig, store=None, device=None, dtype=torch.bfloat16, task="all"):
        super().__init__()
        self.config = config
        dev = torch.device(device if device is not None else "cuda")
        self._owns_store = store is None
        self._store = store if store is not None else ParamStore(config, dev, dtype, task=task)
        self._ops = HipOps(self._store.compute_dtype)
        self._engine = None
        self._geom = None
        names = [n for n in self._store.names() if n.startswith("bert.")]
        self._bind(self._store, "bert", names)
        self.dtype = self._store.compute_dtype
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)

    def _engine_for(self, B, L, V):
        if self._geom != (B, L, V, self.training):
            self._engine = Engine(self.config, self._store, self._ops, B, L, V, need_lang=True,
                                  train_dropout=self.training)
            self._geom = (B, L, V, self.training)
        self._engine.sync_compute_weights()
        return self._engine

    def forward(self, input_ids=None, visual_feats=None, visual_pos=None, attention_mask=None,
                visual_attention_mask=None, token_type_ids=None, inputs_embeds=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, **kwargs):
        # argument validation mirrors HF:731-744
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        if input_ids is None and inputs_embeds is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        if visual_feats is None:
            raise ValueError("`visual_feats` cannot be `None`")
        if visual_pos is None:
            raise ValueError("`visual_pos` cannot be `None`")
        if inputs_embeds is not None:
            raise NotImplementedError("inputs_embeds: the path always starts from input_ids (every reference caller does)")
        if visual_attention_mask is not None:
            raise NotImplementedError("visual_attention_mask is None in every reference caller "
                                      "(ref lxmert_pretrain.py:207, tasks/vqa.py:176-181)")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("attention maps / hidden states are not materialised by the fused kernels")
        B, L = input_ids.shape
        V = visual_feats.shape[1]
        eng = self._engine_for(B, L, V)
        eng.set_inputs(input_ids, attention_mask, token_type_ids, visual_pos, visual_feats=visual_feats)
        eng.use_codebook = False
        if torch.is_grad_enabled():
            lang, vis, pooled = _EncoderFn.apply(self, self._anchor)
        else:
            l_, v_, p_ = eng.encoder_forward(want_pooled=True)
            lang, vis, pooled = l_.view(B, L, -1).clone(), v_.view(B, V, -1).clone(), p_.view(B, -1).clone()
        out = LxmertModelOutput(lang, vis, pooled)
        return out if return_dict in (None, True) else tuple(out)


LXRTModel = LxmertModel          # legacy names (original LXMERT code base / BASELINE.json wording)
LXRTEncoder = LxmertModel


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, vis, want_obj):
        eng = head._bert._engine
        eng.X[-1][eng.ML:].copy_(vis.reshape(eng.MV, eng.d))
        eng.vis_final = eng.X[-1][eng.ML:]
        feat, logits = eng.head_forward(want_logits=want_obj)
        ctx.head = head
        B, V = eng.B, eng.V
        return feat.view(B, V, -1).float().clone(), logits.view(B, V, -1).clone()

    @staticmethod
    def backward(ctx, d_feat, d_obj):
        eng = ctx.head._bert._engine
        MV = eng.MV
        eng.dlogits.zero_()
        if d_obj is not None:
            eng.dlogits[:, :eng.K].copy_(d_obj.reshape(MV, eng.K))
        eng.with_feat_loss = d_feat is not None
        if d_feat is not None:
            eng.dfeat.copy_(d_feat.reshape(MV, eng.F))
        d_vis = torch.empty(MV, eng.d, dtype=eng.cdtype, device=eng.dev)
        eng.head_backward(d_vis)
        return None, d_vis.view(eng.B, eng.V, eng.d), None


class LxmertVisualObjHead(_Named):
    """Cluster-codebook head (ref lxrt/modeling.py:8-53): transform -> linear_feat -> out_cluster (frozen centroids)."""

    def __init__(self, config, bert):
        super().__init__()
        self.config = config
        object.__setattr__(self, "_bert", bert)
        store = bert._store
        names = [n for n in store.names() if n.startswith("obj_predict_head.")]
        self._bind(store, "obj_predict_head", names)
        self.visual_losses = {"obj": {"shape": (-1,), "num": config.num_clusters},
                              "feat": {"shape": (-1, config.visual_feat_dim), "num": config.visual_feat_dim}}
        self.cluster_out = config.num_clusters > 0

    def forward(self, hidden_states, out_keys=()):
        eng = self._bert._engine
        assert eng is not None, "run .bert first: the head shares the engine's batch geometry"
        if torch.is_grad_enabled() and hidden_states.requires_grad:
            feat, obj = _HeadFn.apply(self, hidden_states, True)
        else:
            eng.X[-1][eng.ML:].copy_(hidden_states.reshape(eng.MV, eng.d))
            eng.vis_final = eng.X[-1][eng.ML:]
            f, o = eng.head_forward(True)
            feat, obj = f.view(eng.B, eng.V, -1).float().clone(), o.view(eng.B, eng.V, -1).clone()
        output = {}
        if "feat" in self.visual_losses or "feat" in out_keys:
            output["feat"] = feat
        if "obj" in self.visual_losses or "obj" in out_keys:
            output["obj"] = obj
        return output


class _VisMaskStepFn(torch.autograd.Function):
    """XLxmertForPretraining vis_mask branch as one fused forward+backward (what the trainer also runs)."""

    @staticmethod
    def forward(ctx, model, anchor, feat_loss):
        eng = model.bert._engine
        eng.encoder_forward(want_pooled=False)
        eng.head_forward()
        losses = eng.losses_forward_backward(True, feat_loss)
        ctx.model = model
        return losses[:2].clone()

    @staticmethod
    def backward(ctx, d_losses):
        eng = ctx.model.bert._engine
        s = d_losses.tolist()
        if abs(s[0] - s[1]) > 1e-12 and eng.with_feat_loss:
            raise NotImplementedError("obj_loss and feat_loss must be weighted equally (total_loss = obj + feat)")
        # gradients ACCUMULATE into the flat buffer (several forward/backward calls per update, --update > 1): clearing is
        # zero_grad()'s job, as with autograd; the deferred column reductions are switched on and off inside this backward
        eng.begin_backward()
        eng.head_backward(eng.GA[eng.ML:])
        eng.encoder_backward(False)
        if s[0] != 1.0:
            st = ctx.model._store
            st.grad[:st.n_used].mul_(s[0])
        return None, None, None


class XLxmertForPretraining(nn.Module):
    """ref lxrt/modeling.py:56-308 (task == 'vis_mask'; the word_mask / matched / qa branches are scope-table row N3)."""

    def __init__(self, config: XLxmertConfig, num_clusters=None, device=None, dtype=torch.bfloat16):
        super().__init__()
        if num_clusters is not None:
            config.num_clusters = num_clusters
        self.config = config
        dev = torch.device(device if device is not None else "cuda")
        self._store = ParamStore(config, dev, dtype, task="vis_mask")
        self.bert = LxmertModel(config, store=self._store, device=dev)
        self.obj_predict_head = LxmertVisualObjHead(config, self.bert)
        self.mask_feat = nn.Parameter(self._store.view("mask_feat"))
        self.mask_feat.grad = self._store.gview("mask_feat")
        self.vis_emb = None
        self.task_obj_predict = True
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        from .trainer import init_reference_weights
        init_reference_weights(self._store, seed=0)

    # ---- reference API
    def set_visual_embedding(self, centroids):
        """ref lxrt/modeling.py:140-151: frozen nn.Embedding over the k-means centroids, tied to out_cluster.weight."""
        import numpy as np
        if isinstance(centroids, np.ndarray):
            centroids = torch.from_numpy(centroids)
        self._store.set_centroids(centroids)
        self.vis_emb = nn.Embedding.from_pretrained(self._store.centroids, freeze=True)
        self.obj_predict_head.out_cluster.weight = self.vis_emb.weight        # tied, frozen (ref :150-151)

    def zero_grad(self, set_to_none=False):
        """Gradients live in the flat buffer that every `param.grad` views: clear it in place (never detach the views)."""
        self._store.grad.zero_()
        for name, p_ in self.named_parameters():
            if p_.grad is None and name in self._store.index:
                p_.grad = self._store.gview(name)

    def state_dict(self, *args, prefix="", **kwargs):
        sd = OrderedDict((prefix + k, v.detach()) for k, v in self._store.named_state().items())
        return sd

    def load_state_dict(self, state_dict, strict=False):
        """Accepts the reference layout, with or without the DDP `module.` prefix (ref utils.py:42-49)."""
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        missing = self._store.load_named(sd, strict=strict)
        if self.vis_emb is None and self._store.centroids is not None:
            self.set_visual_embedding(self._store.centroids)
        unexpected = [k for k in sd if k not in self._store.index and k not in ("vis_emb.weight",
                                                                               "obj_predict_head.out_cluster.weight")]
        return missing, unexpected

    def forward(self, input_ids=None, visual_feats=None, visual_pos=None, attention_mask=None,
                visual_attention_mask=None, cluster_ids=None, vis_mask=None, token_type_ids=None, inputs_embeds=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, label_dict=None,
                task="vis_mask", **kwargs):
        if task != "vis_mask":
            raise NotImplementedError(f"task {task!r}: only the masked-visual-token branch is on the hot path (N3 next)")
        if self.vis_emb is None:
            raise RuntimeError("call set_visual_embedding(centroids) first (ref lxrt/modeling.py:185-186)")
        B, L = input_ids.shape
        V = cluster_ids.shape[1]
        eng = self._step_engine(B, L, V)
        labels = label_dict["obj_labels"]
        # feature regression iff the caller supplies its targets: the reference trainer adds label_dict['feat_labels'] (the
        # real grid features) exactly when 'feat' is in --visualLosses (lxmert_pretrain.py:177-179); the canonical recipe
        # (scripts/pretrain.bash:15, --visualLosses obj) has no feature loss.  (The published model code keys the branch on
        # visual_obj_loss and then fails on the missing label: SURVEY App. A item 10.)
        feat_labels = label_dict.get("feat_labels")
        eng.set_inputs(input_ids, attention_mask, token_type_ids, visual_pos, cluster_ids=cluster_ids, vis_mask=vis_mask,
                       obj_labels=labels, feat_labels=feat_labels)
        feat_loss = feat_labels is not None
        if torch.is_grad_enabled():
            losses = _VisMaskStepFn.apply(self, self._anchor, feat_loss)
        else:
            eng.encoder_forward(want_pooled=False)
            eng.head_forward()
            losses = eng.losses_forward_backward(False, feat_loss)[:2].clone()
        obj_loss, feat_l = losses[0], losses[1]
        if not feat_loss:
            return {"obj_loss": obj_loss.detach(), "vis_loss": obj_loss.detach(), "total_loss": obj_loss}
        total = obj_loss + feat_l
        return {"obj_loss": obj_loss.detach(), "feat_loss": feat_l.detach(), "vis_loss": total.detach(), "total_loss": total}

    @torch.no_grad()
    def sample_codes(self, input_ids, n_steps=4, grid_size=8):
        """The device part of ImggenModel.sample_image_NAR (ref tasks/imggen_model.py:169-254): Mask-Predict sampling of
        the grid codes, returned as the generator's input `[B, feat_dim, grid, grid]` (fp32) plus the chosen code ids.
        Tokenisation (before) and the frozen GAN `G(code)` + denorm (after) stay with the caller, as in the reference."""
        import numpy as np
        if self.vis_emb is None:
            raise RuntimeError("call set_visual_embedding(centroids) first")
        was_training = self.training
        self.eval()
        B, L = input_ids.shape
        V = grid_size * grid_size
        eng = self._step_engine(B, L, V)
        pos = np.zeros((V, 4), np.float32)                      # ref utils.box_position
        for i in range(grid_size):
            for j in range(grid_size):
                pos[i * grid_size + j] = [j / grid_size, i / grid_size, (j + 1) / grid_size, (i + 1) / grid_size]
        dev = input_ids.device
        eng.set_inputs(input_ids, input_ids > 0, None, torch.from_numpy(pos).to(dev).unsqueeze(0).expand(B, -1, -1),
                       cluster_ids=torch.zeros(B, V, dtype=torch.long, device=dev),
                       vis_mask=torch.ones(B, V, dtype=torch.bool, device=dev))
        cid, code, _ = eng.sample_codes_nar(n_steps)
        out = code.view(B, V, -1).permute(0, 2, 1).reshape(B, -1, grid_size, grid_size).float()
        self.train(was_training)
        return out, cid.clone()

    def _step_engine(self, B, L, V):
        key = (B, L, V, self.training, "step")
        if self.bert._geom != key:
            self.bert._engine = Engine(self.config, self._store, self.bert._ops, B, L, V, need_lang=False,
                                       train_dropout=self.training)
            self.bert._geom = key
        self.bert._engine.sync_compute_weights()
        return self.bert._engine


# ---------------------------------------------------------------------------------------------- VQA / GQA
class _VqaFn(torch.autograd.Function):
    """VQAModel.forward as one engine forward; backward takes d(logit) from whatever loss the caller applied
    (BCEWithLogitsLoss in the reference, tasks/vqa.py:187) and runs answer head + pooler + encoder backward."""

    @staticmethod
    def forward(ctx, model, anchor):
        eng = model.bert._engine
        logit = eng.vqa_forward()
        ctx.model = model
        return logit.clone()

    @staticmethod
    def backward(ctx, d_logit):
        eng = ctx.model.bert._engine
        ans = eng.answer
        ans.dlogit.zero_()
        ans.dlogit[:, :ans.A].copy_(d_logit)
        eng.GA.zero_()
        cls_rows = eng.lang_final.view(eng.B, eng.L * eng.d)[:, :eng.d]
        ans.bwd(eng.pooled, cls_rows, eng.GA[:eng.ML].view(eng.B, eng.L * eng.d)[:, :eng.d])
        eng.encoder_backward(True)            # gradients accumulate into the flat buffer: call model.zero_grad() per step
        return None, None


class LxmertVisualAnswerHead(_Named):
    """HF:602-614; parameters `logit_fc.{0,2,3}.{weight,bias}` are views of the flat parameter buffer."""

    def __init__(self, store):
        super().__init__()
        self._bind(store, "answer_head", [n for n in store.names() if n.startswith("answer_head.")])


class VQAModel(nn.Module):
    """ref tasks/vqa_model.py:7-72 (also the GQA model, tasks/gqa_model.py): `.bert` + `.answer_head`, forward returns
    {'logit': [B, num_answers] fp32}.  Unlike the published class this one can be constructed (its ctor reads
    `config.num_answers` before setting it and calls `_init_weights` on a non-existent attribute)."""

    def __init__(self, config: XLxmertConfig, num_answers, num_clusters=-1, device=None, dtype=torch.bfloat16):
        super().__init__()
        self.config, self.num_answers = config, num_answers
        dev = torch.device(device if device is not None else "cuda")
        self._store = ParamStore(config, dev, dtype, task="vqa", num_answers=num_answers)
        self.bert = LxmertModel(config, store=self._store, device=dev)
        self.answer_head = LxmertVisualAnswerHead(self._store)
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        from .trainer import init_reference_weights
        init_reference_weights(self._store, seed=0)

    def zero_grad(self, set_to_none=False):
        self._store.grad.zero_()
        for name, p_ in self.named_parameters():
            if p_.grad is None and name in self._store.index:
                p_.grad = self._store.gview(name)

    def state_dict(self, *args, prefix="", **kwargs):
        keep = ("bert.", "answer_head.")
        return OrderedDict((prefix + k, v.detach()) for k, v in self._store.named_state().items() if k.startswith(keep))

    def load_state_dict(self, state_dict, strict=False):
        """reference layout, with or without the DDP `module.` prefix; a pretraining checkpoint (no answer head: the
        reference's load_lxmert_qa path, tasks/vqa.py:55-62) loads the encoder and leaves the head at its init."""
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        missing = self._store.load_named(sd, strict=False)
        missing = [k for k in missing if k.startswith(("bert.", "answer_head."))]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:8]}...")
        return missing, [k for k in sd if k not in self._store.index]

    def forward(self, input_ids=None, visual_feats=None, visual_pos=None, attention_mask=None, visual_attention_mask=None,
                token_type_ids=None, inputs_embeds=None, return_dict=True):
        if visual_attention_mask is not None or inputs_embeds is not None:
            raise NotImplementedError("visual_attention_mask / inputs_embeds are None in every reference caller")
        B, L = input_ids.shape
        V = visual_feats.shape[1]
        key = (B, L, V, self.training, "vqa")
        if self.bert._geom != key:
            self.bert._engine = Engine(self.config, self._store, self.bert._ops, B, L, V, need_lang=True,
                                       train_dropout=self.training)
            self.bert._geom = key
        eng = self.bert._engine
        eng.sync_compute_weights()
        eng.set_inputs(input_ids, attention_mask, token_type_ids, visual_pos, visual_feats=visual_feats)
        logit = _VqaFn.apply(self, self._anchor) if torch.is_grad_enabled() else eng.vqa_forward().clone()
        return {"logit": logit}
