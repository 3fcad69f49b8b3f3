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
        obj.language_hidden_states, obj.vision_hidden_states = None, None        # filled when output_hidden_states is asked for
        obj.language_attentions, obj.vision_attentions, obj.cross_encoder_attentions = None, None, None      # output_attentions
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
    def forward(ctx, model, anchor, inputs_embeds=None):
        eng = model._engine
        lang, vis, pooled = eng.encoder_forward(want_pooled=True)
        ctx.model = model
        ctx.emb_dtype = inputs_embeds.dtype if inputs_embeds is not None and inputs_embeds.requires_grad else None
        ctx.set_materialize_grads(False)            # outputs the loss does not read arrive as None in backward
        B, L, V, d = eng.B, eng.L, eng.V, eng.d
        return lang.view(B, L, d).clone(), vis.view(B, V, d).clone(), pooled.view(B, d).clone()

    @staticmethod
    def backward(ctx, d_lang, d_vis, d_pooled):
        # vec-type gradients accumulate into param.grad (views of the flat buffer): call model.zero_grad() per step
        eng = ctx.model._engine
        eng.backward_from_outputs(d_lang, d_vis, d_pooled)
        return None, None, (eng.d_inputs_embeds().to(ctx.emb_dtype) if ctx.emb_dtype is not None else None)


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
        self._advance_seed(self._engine)
        return self._engine

    _train_calls = 0

    def _advance_seed(self, eng):
        """training mode: every forward draws fresh dropout masks -- the step part of the seeds (device memory, read by the
        kernels of this forward AND of its backward) advances once per call, as torch's generator state does in the reference."""
        if eng.p_hid > 0 or eng.p_attn > 0:
            self._train_calls += 1
            eng.set_step_seed(self._train_calls)

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
        B, L = input_ids.shape if input_ids is not None else inputs_embeds.shape[:-1]          # HF:735-741
        V = visual_feats.shape[1]
        eng = self._engine_for(B, L, V)
        eng.set_inputs(input_ids, attention_mask, token_type_ids, visual_pos, visual_feats=visual_feats,
                       visual_attention_mask=visual_attention_mask, inputs_embeds=inputs_embeds)
        eng.use_codebook = False
        if torch.is_grad_enabled():
            lang, vis, pooled = _EncoderFn.apply(self, self._anchor, inputs_embeds)
        else:
            l_, v_, p_ = eng.encoder_forward(want_pooled=True)
            lang, vis, pooled = l_.view(B, L, -1).clone(), v_.view(B, V, -1).clone(), p_.view(B, -1).clone()
        out = LxmertModelOutput(lang, vis, pooled)
        if output_hidden_states:        # HF:806-822: (language_hidden_states, vision_hidden_states); copies without a gradient path
            lh, vh = eng.hidden_states()
            out.language_hidden_states = tuple(h.view(B, L, -1).detach().clone() for h in lh)
            out.vision_hidden_states = tuple(h.view(B, V, -1).detach().clone() for h in vh)
        if output_attentions:           # HF:806-822; fp32 [B, H, nq, nk] copies without a gradient path, recomputed on demand
            la, va, xa = eng.attention_probs()
            out.language_attentions, out.vision_attentions, out.cross_encoder_attentions = tuple(la), tuple(va), tuple(xa)
        if return_dict in (None, True):
            return out
        t = tuple(out)
        if output_hidden_states:
            t = t + (out.language_hidden_states, out.vision_hidden_states)
        if output_attentions:
            t = t + (out.language_attentions, out.vision_attentions, out.cross_encoder_attentions)
        return t


LXRTModel = LxmertModel          # legacy names (original LXMERT code base / BASELINE.json wording)
LXRTEncoder = LxmertModel


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, vis, want_obj):
        eng = head._bert._engine
        eng.vr(eng.X[-1]).copy_(vis.reshape(eng.MV, eng.d))
        eng.vis_final = eng.vr(eng.X[-1])
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
            eng.vr(eng.X[-1]).copy_(hidden_states.reshape(eng.MV, eng.d))
            eng.vis_final = eng.vr(eng.X[-1])
            f, o = eng.head_forward(True)
            feat, obj = f.view(eng.B, eng.V, -1).float().clone(), o.view(eng.B, eng.V, -1).clone()
        output = {}
        if "feat" in self.visual_losses or "feat" in out_keys:
            output["feat"] = feat
        if "obj" in self.visual_losses or "obj" in out_keys:
            output["obj"] = obj
        return output


class _TaskStepFn(torch.autograd.Function):
    """One branch of XLxmertForPretraining.forward as engine.task_forward (encoder, heads, losses) with engine.task_backward
    as its backward; `losses` come back as one tensor in the order of `keys`."""

    @staticmethod
    def forward(ctx, model, anchor, task, kw, keys):
        eng = model.bert._engine
        out = eng.task_forward(task, **kw)
        keys.extend(out.keys())
        ctx.model = model
        return torch.cat([out[k].reshape(1) for k in keys]).clone()

    @staticmethod
    def backward(ctx, d_losses):
        eng = ctx.model.bert._engine
        s = d_losses.tolist()
        if max(s) - min(s) > 1e-12 * max(1.0, abs(s[0])):
            raise NotImplementedError("the branch's losses must be weighted equally (total_loss is their plain sum)")
        # gradients ACCUMULATE into the flat buffer (several forward/backward calls per update, --update > 1): clearing is
        # zero_grad()'s job, as with autograd; the deferred column reductions are switched on and off inside this backward
        st = ctx.model._store
        if s[0] != 1.0:             # a scaled loss (gradient accumulation / loss scaling): scale what this call adds
            keep = st.grad[:st.n_used].clone()
            st.grad[:st.n_used].zero_()
            eng.task_backward()
            st.grad[:st.n_used].mul_(s[0]).add_(keep)
        else:
            eng.task_backward()
        return None, None, None, None, None


class LxmertPreTrainingHeads(_Named):
    """HF:648-657 `cls`: predictions (LxmertLMPredictionHead, decoder tied to the word embeddings) + seq_relationship;
    parameters are views of the flat buffer, compute runs in engine.LangHeads."""

    def __init__(self, store):
        super().__init__()
        self._bind(store, "cls", [n for n in store.names() if n.startswith("cls.")])


class XLxmertForPretraining(nn.Module):
    """ref lxrt/modeling.py:56-308: `.bert`, `.cls` (task_mask_lm or task_matched), `.obj_predict_head` (task_obj_predict),
    `.answer_head` (task_qa), `.mask_feat`, `.vis_emb`; forward(task = 'vis_mask' | 'word_mask' | 'matched' | 'qa')."""

    def __init__(self, config: XLxmertConfig, num_clusters=None, device=None, dtype=torch.bfloat16):
        super().__init__()
        if num_clusters is not None:
            config.num_clusters = num_clusters
        self.config = config
        self.task_mask_lm, self.task_matched = config.task_mask_lm, config.task_matched
        self.task_obj_predict, self.task_qa = config.task_obj_predict, config.task_qa
        self.num_qa_labels = config.num_qa_labels
        dev = torch.device(device if device is not None else "cuda")
        multi = self.task_mask_lm or self.task_matched or self.task_qa
        # one parameter set for every branch the model was built for; a model with only the codebook head keeps the
        # dead-branch-free layout of the masked-visual-token step
        self._store = ParamStore(config, dev, dtype, task="all" if multi else "vis_mask",
                                 num_answers=self.num_qa_labels if self.task_qa else 0)
        self.bert = LxmertModel(config, store=self._store, device=dev)
        if self.task_mask_lm or self.task_matched:
            self.cls = LxmertPreTrainingHeads(self._store)
        self.obj_predict_head = LxmertVisualObjHead(config, self.bert)
        if self.task_qa:
            self.answer_head = LxmertVisualAnswerHead(self._store)
        self.mask_feat = nn.Parameter(self._store.view("mask_feat"))
        self.mask_feat.grad = self._store.gview("mask_feat")
        self.vis_emb = None
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
        # aliases of tensors stored once: the frozen codebook (vis_emb == out_cluster.weight) and the MLM decoder, tied to the
        # word embeddings (transformers 4.1.1 LxmertPreTrainingHeads(config, embedding_weight), ref lxrt/modeling.py:86)
        alias = ("vis_emb.weight", "obj_predict_head.out_cluster.weight", "cls.predictions.decoder.weight")
        unexpected = [k for k in sd if k not in self._store.index and k not in alias]
        return missing, unexpected

    def forward(self, input_ids=None, visual_feats=None, visual_pos=None, attention_mask=None,
                visual_attention_mask=None, cluster_ids=None, vis_mask=None, token_type_ids=None, inputs_embeds=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, label_dict=None,
                task="vis_mask", **kwargs):
        if task not in ("vis_mask", "word_mask", "matched", "qa"):
            raise ValueError(f"task must be one of 'word_mask', 'vis_mask', 'matched', 'qa' (got {task!r})")
        if self.vis_emb is None:
            raise RuntimeError("call set_visual_embedding(centroids) first (ref lxrt/modeling.py:185-186)")
        if inputs_embeds is not None or visual_attention_mask is not None or output_attentions or output_hidden_states:
            raise NotImplementedError("inputs_embeds / visual_attention_mask / attention maps / hidden states: None in every "
                                      "reference caller of this model (ref lxmert_pretrain.py:201-223)")
        label_dict = label_dict or {}
        B, L = input_ids.shape
        V = cluster_ids.shape[1]
        eng = self._step_engine(B, L, V)
        kw = {}
        if task == "vis_mask":
            # feature regression iff the caller supplies its targets: the reference trainer adds label_dict['feat_labels']
            # (the real grid features) exactly when 'feat' is in --visualLosses (lxmert_pretrain.py:177-179); the canonical
            # recipe (scripts/pretrain.bash:15, --visualLosses obj) has no feature loss.  (The published model code keys the
            # branch on visual_obj_loss and then fails on the missing label: SURVEY App. A item 10.)
            feat_labels = label_dict.get("feat_labels")
            eng.set_inputs(input_ids, attention_mask, token_type_ids, visual_pos, cluster_ids=cluster_ids, vis_mask=vis_mask,
                           obj_labels=label_dict["obj_labels"], feat_labels=feat_labels)
            kw["feat_loss"] = feat_labels is not None
        else:
            if task == "word_mask":
                if not self.task_mask_lm:
                    raise RuntimeError("task 'word_mask' on a model built without task_mask_lm (no `cls` head)")
                kw["word_labels"] = label_dict["word_labels"]
            elif task == "matched":
                if not self.task_matched:
                    raise RuntimeError("task 'matched' on a model built without task_matched (no `cls` head)")
                kw["matched_labels"] = label_dict["matched_labels"]
            # the [MASK]-feature substitution belongs to task == 'vis_mask' only (ref lxrt/modeling.py:190-193)
            eng.set_inputs(input_ids, attention_mask, token_type_ids, visual_pos, cluster_ids=cluster_ids)
        if self.task_qa:                    # `if self.task_qa:` -- on the MODEL, not the task argument (ref :292)
            kw["qa_labels"] = label_dict["qa_labels"]
        elif task == "qa":
            return {"total_loss": torch.zeros((), device=input_ids.device)}         # the reference adds nothing either
        keys = []
        if torch.is_grad_enabled():
            losses = _TaskStepFn.apply(self, self._anchor, task, kw, keys)
        else:
            out = eng.task_forward(task, want_grad=False, **kw)
            keys = list(out.keys())
            losses = torch.cat([out[k].reshape(1) for k in keys]).clone()
        out_dict = {k: losses[i].detach() for i, k in enumerate(keys)}
        if task == "vis_mask":
            out_dict["vis_loss"] = sum(losses[i] for i, k in enumerate(keys) if k in ("obj_loss", "feat_loss")).detach()
        if self.task_qa:
            out_dict["qa_pred"] = eng.answer.row_argmax.long().clone()          # ref :300
        out_dict["total_loss"] = losses.sum()
        return out_dict

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
            self.bert._engine = Engine(self.config, self._store, self.bert._ops, B, L, V,
                                       need_lang=self._store.task != "vis_mask", train_dropout=self.training)
            self.bert._geom = key
        self.bert._engine.sync_compute_weights()
        self.bert._advance_seed(self.bert._engine)
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
        eng.zero_out_grads(eng.GA)
        cls_rows, d_cls = eng._cls_views(eng.GA)
        ans.bwd(eng.pooled, cls_rows, d_cls)
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

    _task = "vqa"

    def __init__(self, config: XLxmertConfig, num_answers, num_clusters=-1, device=None, dtype=torch.bfloat16):
        super().__init__()
        self.config, self.num_answers = config, num_answers
        dev = torch.device(device if device is not None else "cuda")
        self._store = ParamStore(config, dev, dtype, task=self._task, num_answers=num_answers)
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
        if inputs_embeds is not None and inputs_embeds.requires_grad:
            raise NotImplementedError("d(inputs_embeds) through the task wrappers: use .bert directly (no reference caller does)")
        B, L = input_ids.shape if input_ids is not None else inputs_embeds.shape[:-1]
        V = visual_feats.shape[1]
        key = (B, L, V, self.training, self._task)
        if self.bert._geom != key:
            self.bert._engine = Engine(self.config, self._store, self.bert._ops, B, L, V, need_lang=True,
                                       train_dropout=self.training)
            self.bert._geom = key
        eng = self.bert._engine
        eng.sync_compute_weights()
        self.bert._advance_seed(eng)
        eng.set_inputs(input_ids, attention_mask, token_type_ids, visual_pos, visual_feats=visual_feats,
                       visual_attention_mask=visual_attention_mask, inputs_embeds=inputs_embeds)
        logit = _VqaFn.apply(self, self._anchor) if torch.is_grad_enabled() else eng.vqa_forward().clone()
        return {"logit": logit}


class GQAModel(VQAModel):
    """ref tasks/gqa_model.py:7-72: the same module as VQAModel under the name the GQA driver imports (tasks/gqa.py:70,150:
    `.bert` + `.answer_head`, BCEWithLogitsLoss on the soft targets, forward returns {'logit': [B, num_answers]})."""


class NLVR2Model(VQAModel):
    """ref tasks/nlvr2_model.py:7-93: `.bert` + `.answer_head`; forward takes visual_feats [P, 2, V, F], visual_pos
    [P, 2, V, 4] and input_ids [2P, L] (every statement repeated for its two images), flattens the pairs, and feeds
    pooled_output viewed as [P, 2d] to the head; returns {'logit': [P, 2] fp32}.  The published class cannot run (its ctor
    builds `logit_fc` with a d-wide input, its forward calls an undefined `answer_head` on the 2d-wide vector); this one has
    the head that forward needs -- Linear(2d, 2d) -> GeLU -> LayerNorm(2d) -> Linear(2d, 2) -- under `.answer_head`."""
    _task = "nlvr2"

    def __init__(self, config: XLxmertConfig, num_answers=2, num_clusters=-1, device=None, dtype=torch.bfloat16):
        super().__init__(config, num_answers, num_clusters, device, dtype)

    def forward(self, input_ids=None, visual_feats=None, visual_pos=None, attention_mask=None, visual_attention_mask=None,
                token_type_ids=None, inputs_embeds=None, return_dict=True):
        P, n_images, V, Fd = visual_feats.shape
        assert n_images == 2                                                             # ref :64
        return super().forward(input_ids, visual_feats.reshape(P * 2, V, Fd), visual_pos.reshape(P * 2, V, -1), attention_mask,
                               visual_attention_mask, token_type_ids, inputs_embeds, return_dict)


# ---------------------------------------------------------------------------------------------- image generation
class ImggenModel(XLxmertForPretraining):
    """ref tasks/imggen_model.py:11-257: `.bert`, `.obj_predict_head`, `.mask_feat`, `.vis_emb`, `set_visual_embedding`,
    `set_image_generator(G)`, `sample_image_NAR` (Mask-Predict, :169-257) and `sample_image_AR` (one grid position per step:
    position_random / position_TLBR / position_confidence, :49-167).  The loops run on the device without a host round trip
    (Engine.sample_codes_nar / sample_codes_ar); tokenisation in front and the frozen GAN generator `G` + denorm behind stay
    stock PyTorch, as in the reference.

    `sentences`: a list of strings (needs `tokenizer`: any callable with LxmertTokenizer's call signature -- the reference
    downloads 'unc-nlp/lxmert-base-uncased' in its constructor, :26, which an offline machine cannot) or, beyond the reference, an
    int64 tensor of input ids [B, L] (0 = [PAD]).  grid_size / the code width follow the config (the reference hardcodes 8 / 2048)."""

    def __init__(self, config: XLxmertConfig, args=None, num_clusters=10000, device=None, dtype=torch.bfloat16, tokenizer=None,
                 grid_size=8):
        import copy
        config = copy.copy(config)
        config.task_mask_lm = config.task_matched = config.task_qa = False          # the published class has the codebook head only
        config.task_obj_predict = True
        super().__init__(config, num_clusters=num_clusters, device=device, dtype=dtype)
        self.args, self.tokenizer, self.grid_size = args, tokenizer, grid_size
        self.G = None

    def set_image_generator(self, generator):
        """ref :41-42: any callable [B, code_dim, g, g] fp32 -> image batch (image_generator.src.layers.Generator in the reference)"""
        self.G = generator

    @staticmethod
    def denorm(x):
        """(-1, 1) => (0, 1)   (ref :44-47)"""
        return ((x + 1) / 2).clamp(0, 1)

    def _input_ids(self, sentences, max_text_length):
        if torch.is_tensor(sentences):
            return sentences.to(self._store.device)
        if self.tokenizer is None:
            raise RuntimeError("ImggenModel got sentences but no tokenizer: pass tokenizer=LxmertTokenizer.from_pretrained(...) "
                               "(the reference downloads one in its constructor) or hand over input ids")
        ids = self.tokenizer(sentences, max_length=max_text_length, truncation=True, return_tensors="pt").input_ids      # ref :56-58
        return ids.to(self._store.device)

    def _prepare(self, input_ids):
        import numpy as np
        if self.vis_emb is None:
            raise RuntimeError("call set_visual_embedding(centroids) first")
        self.eval()                                                 # ref :54, :180
        g = self.grid_size
        B, L = input_ids.shape
        V = g * g
        eng = self._step_engine(B, L, V)
        pos = np.zeros((V, 4), np.float32)                          # ref utils.box_position
        for i in range(g):
            for j in range(g):
                pos[i * g + j] = [j / g, i / g, (j + 1) / g, (i + 1) / g]
        dev = input_ids.device
        eng.set_inputs(input_ids, input_ids > 0, None, torch.from_numpy(pos).to(dev).unsqueeze(0).expand(B, -1, -1),
                       cluster_ids=torch.zeros(B, V, dtype=torch.long, device=dev), vis_mask=torch.ones(B, V, dtype=torch.bool, device=dev))
        return eng, B, V

    def _image(self, code, B):
        """code [B*V, F] -> G(code as [B, F, g, g]) -> denorm -> host (ref :160-165, :250-256)"""
        if self.G is None:
            raise RuntimeError("call set_image_generator(G) first (ref tasks/imggen_model.py:41)")
        g = self.grid_size
        x = code.view(B, g * g, -1).permute(0, 2, 1).reshape(B, -1, g, g).float()
        return self.denorm(self.G(x)).cpu()

    @torch.no_grad()
    def sample_image_NAR(self, sentences, max_text_length=20, n_steps=None, return_intermediate=False):
        """ref :169-257.  n_steps=None -> grid_size ** 2 (ref :191-192)."""
        eng, B, V = self._prepare(self._input_ids(sentences, max_text_length))
        n_steps = V if n_steps is None else n_steps
        imgs = []
        hook = (lambda i: imgs.append(self._image(eng.materialise_codes(), B))) if return_intermediate else None
        _, code, _ = eng.sample_codes_nar(n_steps, on_step=hook)
        self.code_ids = eng.cid.clone()                             # the chosen codebook ids [B, V] (beyond the reference: handy)
        return imgs if return_intermediate else self._image(code, B)

    @torch.no_grad()
    def sample_image_AR(self, sentences, max_text_length=20, position_random=False, position_TLBR=False, position_confidence=True,
                        n_steps=None, seed=None, return_intermediate=False):
        """ref :49-167: the three position policies with the reference's precedence (random, else TLBR, else confidence) and its
        host-side order for `position_random` (random.Random(seed).shuffle, extended for n_steps > grid ** 2, :77-89)."""
        import random
        eng, B, V = self._prepare(self._input_ids(sentences, max_text_length))
        n_steps = V if n_steps is None else n_steps
        positions = None
        if position_random:
            mode = "random"
            positions = list(range(V))
            (random.Random(seed) if seed is not None else random).shuffle(positions)
            if n_steps > V:
                extra = list(range(n_steps - V))
                (random.Random(seed) if seed is not None else random).shuffle(extra)
                positions = extra + positions
        elif position_TLBR:
            mode = "tlbr"
        elif position_confidence:
            mode = "confidence"
        else:
            raise ValueError("sample_image_AR: one of position_random / position_TLBR / position_confidence must be set")
        imgs = []
        hook = (lambda i: imgs.append(self._image(eng.materialise_codes(masked=True), B))) if return_intermediate else None
        _, code, _ = eng.sample_codes_ar(n_steps, mode, positions=positions, on_step=hook)
        self.code_ids = eng.cid.clone()
        return imgs if return_intermediate else self._image(code, B)
