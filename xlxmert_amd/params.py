"""Flat parameter storage for the X-LXMERT hot path.

All trainable tensors of the path live in ONE fp32 master buffer (plus same-shaped gradient, Adam m/v
buffers and a compute-dtype copy), laid out so that
  * query/key/value weights (and biases) of an attention block are adjacent -> one fused [3d, d] operand;
  * used parameters are ordered by when backward finishes them (head first, embeddings last), so finished gradients
    form a growing prefix and the gradient exchange overlaps with backward on contiguous buckets; the used range of
    the gradient buffer is zeroed once per step (weight gradients accumulate through split-K atomics, vectors through
    two-stage column reductions);
  * parameters that get no gradient on a masked-visual-token step (pooler and the language side of the last
    cross layer -- SURVEY.md section 0.6 V3) sit behind `n_used`: the optimizer and the gradient exchange
    run over [0, n_used) only (the reference's AdamW skips tensors whose .grad is None).
State-dict names are the reference's; every named tensor is a view into the flat buffer.
"""
from dataclasses import dataclass, field

import torch

CHUNK = 256          # allocation granule (elements); AdamW weight-decay flags are per chunk


@dataclass
class Member:
    name: str
    shape: tuple
    offset: int = 0          # element offset in the flat buffer


@dataclass
class Unit:
    members: list
    region: str              # "mat" | "vec"
    used: bool = True
    offset: int = 0
    numel: int = 0
    padded: int = 0
    decay: bool = False


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


def _decays(name):
    """ref lxmert_pretrain.py:125-135: no_decay = ["bias", "LayerNorm.weight"] (substring match; so the answer head's
    LayerNorm, called `logit_fc.2.weight`, DOES decay -- reproduced)."""
    return not ("bias" in name or "LayerNorm.weight" in name)


def build_units(cfg, task="vis_mask", num_answers=0, pair=False):
    """task: "vis_mask" (masked-visual-token pretraining step) or "vqa" (VQA/GQA fine-tune: real features in, pooled output
    -> LxmertVisualAnswerHead, ref tasks/vqa_model.py:7-72; the codebook head and mask_feat are not part of that model).
    pair=True is the NLVR2 model (ref tasks/nlvr2_model.py:7-93): the same layout, but the head reads the concatenated
    pooled outputs of a statement's two images, so its first Linear is (2d, 2d).
    Pretraining tasks ("vis_mask", "word_mask", "matched", "qa", "all") with num_answers > 0 describe a model built with
    `task_qa` (ref lxrt/modeling.py:89-90): the answer head over pooled_output joins EVERY task's loss (ref :292-304), so the
    pooler, the language side of the last cross layer and `answer_head.*` are live in every branch."""
    d, dff, F, P = cfg.hidden_size, cfg.intermediate_size, cfg.visual_feat_dim, cfg.visual_pos_dim
    units = []
    # which heads a step of this task reads (everything else gets no gradient in the reference and must stay untouched):
    #   "word_mask" / "matched": the language pretraining branches (ref lxrt/modeling.py:211-235) -- language output or
    #   pooled_output only; un-masked codebook features in.  "all": every head of the pretraining model is live.
    pretrain = task in ("vis_mask", "all")
    qa = task != "vqa" and num_answers > 0

    def U(region, used, *members):
        units.append(Unit([Member(n, tuple(s)) for n, s in members], region, used))

    def att(p, self_name, used=True):
        U("mat", used, *[(f"{p}.{self_name}.{n}.weight", (d, d)) for n in ("query", "key", "value")])
        U("vec", used, *[(f"{p}.{self_name}.{n}.bias", (d,)) for n in ("query", "key", "value")])
        U("mat", used, (f"{p}.output.dense.weight", (d, d)))
        U("vec", used, (f"{p}.output.dense.bias", (d,)))
        U("vec", used, (f"{p}.output.LayerNorm.weight", (d,)))
        U("vec", used, (f"{p}.output.LayerNorm.bias", (d,)))

    def ffn(pi, po, used=True):
        U("mat", used, (f"{pi}.dense.weight", (dff, d)))
        U("vec", used, (f"{pi}.dense.bias", (dff,)))
        U("mat", used, (f"{po}.dense.weight", (d, dff)))
        U("vec", used, (f"{po}.dense.bias", (d,)))
        U("vec", used, (f"{po}.LayerNorm.weight", (d,)))
        U("vec", used, (f"{po}.LayerNorm.bias", (d,)))

    U("vec", pretrain, ("mask_feat", (F,)))
    e = "bert.embeddings"
    U("vec", True, (f"{e}.word_embeddings.weight", (cfg.vocab_size, d)))
    U("vec", True, (f"{e}.position_embeddings.weight", (cfg.max_position_embeddings, d)))
    U("vec", True, (f"{e}.token_type_embeddings.weight", (cfg.type_vocab_size, d)))
    U("vec", True, (f"{e}.LayerNorm.weight", (d,)))
    U("vec", True, (f"{e}.LayerNorm.bias", (d,)))
    v = "bert.encoder.visn_fc"
    U("mat", True, (f"{v}.visn_fc.weight", (d, F)))
    U("vec", True, (f"{v}.visn_fc.bias", (d,)))
    U("vec", True, (f"{v}.visn_layer_norm.weight", (d,)))
    U("vec", True, (f"{v}.visn_layer_norm.bias", (d,)))
    U("vec", True, (f"{v}.box_fc.weight", (d, P)))
    U("vec", True, (f"{v}.box_fc.bias", (d,)))
    U("vec", True, (f"{v}.box_layer_norm.weight", (d,)))
    U("vec", True, (f"{v}.box_layer_norm.bias", (d,)))
    for stack, n in (("layer", cfg.l_layers), ("r_layers", cfg.r_layers)):
        for i in range(n):
            p = f"bert.encoder.{stack}.{i}"
            att(p + ".attention", "self")
            ffn(p + ".intermediate", p + ".output")
    for i in range(cfg.x_layers):
        p = f"bert.encoder.x_layers.{i}"
        # vis_mask never reads the language output of the LAST cross layer
        lang_used = qa or not (task == "vis_mask" and i == cfg.x_layers - 1)
        # ... and the VQA step never reads the VISUAL output of the last cross layer (only pooled_output): its visual
        # self-attention / FFN get no gradient in the reference, so the optimizer must not touch them
        vis_used = not (task in ("vqa", "word_mask", "matched", "qa") and i == cfg.x_layers - 1)
        att(p + ".visual_attention", "att")
        att(p + ".lang_self_att", "self", lang_used)
        att(p + ".visn_self_att", "self", vis_used)
        ffn(p + ".lang_inter", p + ".lang_output", lang_used)
        ffn(p + ".visn_inter", p + ".visn_output", vis_used)
    pooled_used = qa or task in ("vqa", "matched", "all")
    U("mat", pooled_used, ("bert.pooler.dense.weight", (d, d)))
    U("vec", pooled_used, ("bert.pooler.dense.bias", (d,)))
    h = "obj_predict_head"
    U("mat", pretrain, (f"{h}.transform.dense.weight", (d, d)))
    U("vec", pretrain, (f"{h}.transform.dense.bias", (d,)))
    U("vec", pretrain, (f"{h}.transform.LayerNorm.weight", (d,)))
    U("vec", pretrain, (f"{h}.transform.LayerNorm.bias", (d,)))
    U("mat", pretrain, (f"{h}.linear_feat.weight", (F, d)))
    U("vec", pretrain, (f"{h}.linear_feat.bias", (F,)))
    U("vec", pretrain, (f"{h}.out_cluster.bias", (cfg.num_clusters,)))
    if task in ("word_mask", "matched", "all"):
        c = "cls"                                        # LxmertPreTrainingHeads (HF:589-657); decoder.weight is tied to
        mlm = task in ("word_mask", "all")               # the word embeddings and therefore not a unit of its own
        U("mat", mlm, (f"{c}.predictions.transform.dense.weight", (d, d)))
        U("vec", mlm, (f"{c}.predictions.transform.dense.bias", (d,)))
        U("vec", mlm, (f"{c}.predictions.transform.LayerNorm.weight", (d,)))
        U("vec", mlm, (f"{c}.predictions.transform.LayerNorm.bias", (d,)))
        U("vec", mlm, (f"{c}.predictions.bias", (cfg.vocab_size,)))
        rel = task in ("matched", "all")
        U("vec", rel, (f"{c}.seq_relationship.weight", (2, d)))
        U("vec", rel, (f"{c}.seq_relationship.bias", (2,)))
    if task == "vqa" or qa:
        a = "answer_head.logit_fc"                       # nn.Sequential indices of HF:606-611
        U("mat", True, (f"{a}.0.weight", (2 * d, 2 * d if pair else d)))
        U("vec", True, (f"{a}.0.bias", (2 * d,)))
        U("vec", True, (f"{a}.2.weight", (2 * d,)))
        U("vec", True, (f"{a}.2.bias", (2 * d,)))
        U("mat", True, (f"{a}.3.weight", (num_answers, 2 * d)))
        U("vec", True, (f"{a}.3.bias", (num_answers,)))
    return units


def _backward_rank(cfg, name):
    """position of a tensor's block in the backward pass (stable sort keeps the order inside a block)."""
    if name.startswith("obj_predict_head.") or name.startswith("answer_head.") or name.startswith("cls."):
        return 0
    if ".x_layers." in name:
        return 1 + (cfg.x_layers - 1 - int(name.split(".x_layers.")[1].split(".")[0]))
    base = 1 + cfg.x_layers
    if ".r_layers." in name:
        return base + (cfg.r_layers - 1 - int(name.split(".r_layers.")[1].split(".")[0]))
    base += cfg.r_layers
    if ".encoder.layer." in name:
        return base + (cfg.l_layers - 1 - int(name.split(".encoder.layer.")[1].split(".")[0]))
    base += cfg.l_layers
    if name.startswith("bert.pooler."):
        return 0               # only used on the pooled-output tasks, where it is finished right after the head
    if name.startswith("bert.embeddings."):
        return base + 1         # end of the language stream's backward (right after language layer 0)
    return base + 2             # visn_fc, mask_feat: finished by the very last kernels of the visual stream


class ParamStore:
    """Flat fp32 master parameters + gradients + Adam state + compute-dtype copy, on one device."""

    def __init__(self, cfg, device, compute_dtype=torch.bfloat16, task="vis_mask", num_answers=0):
        # "nlvr2" = the VQA layout with a pair head (two images per statement, ref tasks/nlvr2_model.py:50-86)
        self.pair = task == "nlvr2"
        if self.pair:
            task, num_answers = "vqa", (num_answers or 2)
        self.cfg, self.device, self.compute_dtype, self.task = cfg, torch.device(device), compute_dtype, task
        self.num_answers = num_answers
        units = build_units(cfg, task, num_answers, self.pair)
        # used tensors in the order backward FINISHES them (head, cross layers N..0, visual layers | language layers,
        # embeddings | visual feature encoder): the main stream completes a growing prefix and, at the very end, the tail;
        # the language stream completes the block in between layer by layer (language_range), so the data-parallel
        # exchange can start on contiguous buckets of either range while backward is still running.
        order = sorted([u for u in units if u.used], key=lambda u: _backward_rank(cfg, u.members[0].name)) \
            + [u for u in units if not u.used]
        off = 0
        self.index = {}
        for u in order:
            u.offset = off
            rel = 0
            for m in u.members:
                m.offset = off + rel
                rel += _numel(m.shape)
                self.index[m.name] = m
            u.numel = rel
            u.padded = (rel + CHUNK - 1) // CHUNK * CHUNK
            u.decay = _decays(u.members[0].name)
            off += u.padded
        self.units = order
        self.n_total = off
        self.n_mat = 0          # the whole used range is zeroed once per step (weight gradients accumulate: split-K)
        self.n_used = sum(u.padded for u in order if u.used)
        dev = self.device
        self.master = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.compute = (self.master if compute_dtype == torch.float32
                        else torch.zeros(self.n_total, dtype=compute_dtype, device=dev))
        self.exp_avg = None
        self.exp_avg_sq = None
        flags = torch.zeros(self.n_total // CHUNK, dtype=torch.uint8)
        for u in order:
            if u.decay:
                flags[u.offset // CHUNK:(u.offset + u.padded) // CHUNK] = 1
        self.decay_flags = flags.to(dev)
        self._task_flags = {}
        self._kept = set()
        self.master_partial = False     # sharded exchange, gather='bf16': the master matrices are whole on their owner only (trainer)
        # frozen centroid codebook (vis_emb.weight == obj_predict_head.out_cluster.weight, ref modeling.py:140-151)
        self.centroids = None          # fp32 [K, F]
        self.centroids_c = None        # compute dtype

    def mark_overwritten(self, ranges):
        """ranges [(lo, hi)] of the flat gradient buffer that every backward OVERWRITES (weight gradients with one contribution per
        step, xl_gemm_wgrad_group overwrite_mask): the optimizer pass need not clear them -- bit 2 of the per-chunk flags for every
        256-element chunk that lies inside one of them.  In place: recorded launch plans hold the flag tensors' pointers."""
        new = [r for r in ranges if r not in self._kept]
        if not new:
            return False
        self._kept.update(new)
        fl = torch.zeros(self.n_total // CHUNK, dtype=torch.uint8)
        for lo, hi in new:
            c0, c1 = (lo + CHUNK - 1) // CHUNK, hi // CHUNK
            if c1 > c0:
                fl[c0:c1] = 4
        fl = fl.to(self.device)
        self.decay_flags |= fl
        for t in self._task_flags.values():
            t |= fl
        return True             # (the caller drops launch plans recorded under the old set: they may ACCUMULATE into a range that
                                #  the optimizer pass has just stopped clearing -- ADVICE r5)

    def fp32_read_index(self, lo, hi):
        """int32 positions, relative to `lo`, of the elements of the flat range [lo, hi) that the kernels READ IN FP32 from the master
        buffer (view(): biases, LayerNorm affines, box_fc, mask_feat, ...) -- everything that is not a matrix read through the
        compute-dtype copy (cview(): Linear weights and embedding tables; box_fc.weight [d, 4] is read in fp32 by the feature encoder).  The sharded exchange's bf16 all-gather carries the
        matrices; these elements travel in a small fp32 side car (trainer gather="bf16")."""
        parts = []
        for name, m in self.index.items():
            n = _numel(m.shape)
            matrix = len(m.shape) == 2 and name.endswith("weight") and "box_fc" not in name      # Linear weights, embedding tables
            a, b = max(lo, m.offset), min(hi, m.offset + n)
            if not matrix and a < b:
                parts.append(torch.arange(a - lo, b - lo, dtype=torch.int32))
        if not parts:
            return torch.zeros(0, dtype=torch.int32, device=self.device)
        return torch.cat(parts).sort().values.to(self.device)

    def task_flags(self, task):
        """per-chunk optimizer flags for one step of `task` on a multi-task ("all") store: bit 0 = weight decay, bit 1 = skip
        (tensors that get no gradient in that branch of the reference: .grad stays None and AdamW leaves them alone)."""
        if task not in self._task_flags:
            active = {m.name for u in build_units(self.cfg, task, self.num_answers, self.pair) if u.used for m in u.members}
            fl = self.decay_flags.clone()          # (with the keep bits marked so far; later ones are OR-ed into every clone)
            for u in self.units:
                if not all(m.name in active for m in u.members):
                    fl[u.offset // CHUNK:(u.offset + u.padded) // CHUNK] |= 2
            self._task_flags[task] = fl
        return self._task_flags[task]

    # ---- views
    def view(self, name, buf=None):
        m = self.index[name]
        buf = self.master if buf is None else buf
        return buf[m.offset:m.offset + _numel(m.shape)].view(m.shape)

    def cview(self, name):
        return self.view(name, self.compute)

    def gview(self, name):
        return self.view(name, self.grad)

    def fused(self, names, buf):
        """[sum(rows), cols] view over adjacent members (query/key/value)."""
        ms = [self.index[n] for n in names]
        for a, b in zip(ms[:-1], ms[1:]):
            assert b.offset == a.offset + _numel(a.shape), "members are not adjacent"
        n = sum(_numel(m.shape) for m in ms)
        flat = buf[ms[0].offset:ms[0].offset + n]
        return flat.view(-1, ms[0].shape[1]) if len(ms[0].shape) == 2 else flat

    def language_range(self):
        """[lo, hi) of the gradients the language stream finishes on its own (language layers, then the embeddings): a
        contiguous block between the visual layers and the visual feature encoder, exchanged as a second growing range."""
        lo = self.range_of("bert.encoder.layer.")[0]
        hi = self.range_of("bert.embeddings.")[1]
        return lo, hi

    def heads_end(self):
        """end of the block of head gradients (everything with _backward_rank 0: codebook head, cls.*, answer head, pooler)."""
        return max(u.offset + u.padded for u in self.units if u.used and _backward_rank(self.cfg, u.members[0].name) == 0)

    def range_of(self, prefix):
        """[lo, hi) element range (incl. padding) covered by the units whose first member starts with `prefix`."""
        us = [u for u in self.units if u.used and u.members[0].name.startswith(prefix)]
        return min(u.offset for u in us), max(u.offset + u.padded for u in us)

    @staticmethod
    def group_key(name):
        """parameter group in the order the FORWARD first reads it (the optimizer pass updates group by group in that order when
        it runs behind the step, trainer overlap_optimizer): visual feature encoder, embeddings, language / visual / cross layer
        i, heads (everything on top of the encoder)."""
        for prefix, kind in (("bert.encoder.layer.", "lang"), ("bert.encoder.r_layers.", "vis"), ("bert.encoder.x_layers.", "x")):
            if name.startswith(prefix):
                return (kind, int(name[len(prefix):].split(".")[0]))
        if name.startswith("bert.embeddings."):
            return "emb"
        if name.startswith("bert.encoder.visn_fc.") or name == "mask_feat":
            return "visn"
        return "heads"

    def forward_groups(self):
        """[(key, lo, hi)]: contiguous element ranges of the used part of the flat buffers, one or more per group, listed in the
        order the forward consumes the groups (language and visual stacks interleaved, as they run side by side)."""
        runs = []
        for u in self.units:
            if not u.used:
                continue
            k = self.group_key(u.members[0].name)
            if runs and runs[-1][0] == k and runs[-1][2] == u.offset:
                runs[-1][2] = u.offset + u.padded
            else:
                runs.append([k, u.offset, u.offset + u.padded])
        cfg = self.cfg
        order = ["visn", "emb"]
        li = vi = 0
        while li < cfg.l_layers or vi < cfg.r_layers:       # ~2 language layers per visual layer: what the two streams consume
            for _ in range(2):
                if li < cfg.l_layers:
                    order.append(("lang", li)); li += 1
            if vi < cfg.r_layers:
                order.append(("vis", vi)); vi += 1
        order += [("x", i) for i in range(cfg.x_layers)] + ["heads"]
        rank = {k: i for i, k in enumerate(order)}
        assert all(r[0] in rank for r in runs), [r[0] for r in runs if r[0] not in rank]
        return [tuple(r) for r in sorted(runs, key=lambda r: rank[r[0]])]

    def names(self):
        return list(self.index.keys())

    def set_centroids(self, centroids):
        c = torch.as_tensor(centroids, dtype=torch.float32).to(self.device).contiguous()
        assert c.shape == (self.cfg.num_clusters, self.cfg.visual_feat_dim), c.shape
        if self.centroids is not None:
            # keep the device buffers (kernels recorded in a launch plan, engines' views and the modules' tied
            # vis_emb.weight / out_cluster.weight all hold their addresses): overwrite in place
            self.centroids.copy_(c)
            if self.centroids_c is not self.centroids:
                self.centroids_c.copy_(c)
            return
        self.centroids = c
        self.centroids_c = c if self.compute_dtype == torch.float32 else c.to(self.compute_dtype)

    def ensure_adam_state(self):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros(self.n_used, dtype=torch.float32, device=self.device)
            self.exp_avg_sq = torch.zeros(self.n_used, dtype=torch.float32, device=self.device)

    def load_named(self, sd, strict=False):
        """Copy tensors of a reference-layout state dict into the master buffer (keys not on the path are ignored)."""
        missing = []
        with torch.no_grad():
            for name in self.index:
                if name in sd:
                    self.view(name).copy_(torch.as_tensor(sd[name]).to(self.device, torch.float32))
                else:
                    missing.append(name)
            for k in ("vis_emb.weight", "obj_predict_head.out_cluster.weight"):
                if k in sd:
                    self.set_centroids(sd[k])
                    break
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:8]}...")
        return missing

    def named_state(self):
        if self.master_partial:
            raise RuntimeError("the fp32 master copy of the matrices is current on the owning rank only (sharded exchange with "
                               "gather='bf16'): call PretrainStep.gather_state() on EVERY rank before reading parameters "
                               "(state_dict / save_checkpoint / verify_replicas)")
        out = {n: self.view(n) for n in self.index}
        if "cls.predictions.bias" in self.index:
            out["cls.predictions.decoder.weight"] = self.view("bert.embeddings.word_embeddings.weight")      # tied
        if self.centroids is not None:
            out["vis_emb.weight"] = self.centroids
            out["obj_predict_head.out_cluster.weight"] = self.centroids
        return out
