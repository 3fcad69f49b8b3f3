"""Step semantics and the data-parallel exchange, on CPU: kernels replaced by FakeOps,
collectives over gloo with world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import lxmert_oracle as O
from fake_ops import FakeOps
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.params import ParamStore
from xlxmert_amd.trainer import PretrainStep, linear_schedule, synthetic_batch

TINY = dict(vocab_size=100, hidden_size=64, num_attention_heads=4, intermediate_size=128, max_position_embeddings=32,
            visual_feat_dim=32, num_clusters=56, l_layers=2, x_layers=2, r_layers=1)


def oracle_cfg(cfg):
    return O.OracleConfig(**{k: getattr(cfg, k) for k in ("vocab_size", "hidden_size", "num_attention_heads",
                                                         "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                         "l_layers", "x_layers", "r_layers", "visual_feat_dim",
                                                         "visual_pos_dim", "num_clusters")})


def make_step(cfg, B, L, grid, seed=3, **kw):
    store = ParamStore(cfg, "cpu", torch.float32, task="vis_mask")
    sd = O.make_state_dict(oracle_cfg(cfg), seed)
    store.load_named(sd)
    kw.setdefault("visual_losses", "obj,feat")      # what the fixtures / the oracle's total_loss hold (the model code computes both)
    return PretrainStep(cfg, B, L, grid * grid, dtype=torch.float32, device="cpu", store=store,
                        ops=FakeOps(torch.float32), total_steps=10, **kw), sd


def oracle_grads(cfg, sd, batch):
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and k != "vis_emb.weight") for k, v in sd.items()}
    sd["obj_predict_head.out_cluster.weight"] = sd["vis_emb.weight"]
    out = O.xlxmert_vis_mask_forward(sd, oracle_cfg(cfg), batch["input_ids"], batch["visual_pos"], batch["attention_mask"],
                                     batch["cluster_ids"], batch["vis_mask"], batch["obj_labels"])
    out["total_loss"].backward()
    return {k: v.grad for k, v in sd.items() if v.grad is not None}, out


@pytest.mark.parametrize("head_pad", [256, 8, 4])
def test_two_steps_match_closed_form(monkeypatch, head_pad):
    """head_pad < 256: the masked-row head really compacts at this size, its padded row count changes from step 1 to step 2, and
    this model flips the backward scratch generation an odd number of times per step: the head's gathered input must be the buffer
    its forward wrote, not whatever set tmp() hands out in the backward (a latent bug found with exactly this case: the transform
    weight's gradient was wrong from the second step on for models with an odd number of visual-side layers)."""
    from xlxmert_amd.engine import Engine
    monkeypatch.setattr(Engine, "ROW_PAD", head_pad)
    cfg = XLxmertConfig(**TINY)
    B, L, grid = 3, 8, 4
    tr, sd = make_step(cfg, B, L, grid, lr=1e-2, weight_decay=0.01, warmup_ratio=0.2)
    ref = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in ref.items()}
    v2 = {k: torch.zeros_like(v) for k, v in ref.items()}
    for t in (1, 2):
        batch = synthetic_batch(cfg, B, L, grid, seed=100 + t)
        losses = tr.step(batch)
        grads, out = oracle_grads(cfg, ref, batch)
        assert abs(losses[0].item() - out["obj_loss"].item()) < 3e-5
        assert abs(losses[1].item() - out["feat_loss"].item()) < 3e-5
        names = sorted(grads)
        norm, clipped = O.clip_grad_norm([grads[k] for k in names], 1.0)
        assert abs(tr.grad_norm() - norm.item()) < 1e-4 * max(1.0, norm.item())
        lr = 1e-2 * linear_schedule(t - 1, 2, 10)
        for k, g in zip(names, clipped):
            wd = 0.0 if ("bias" in k or "LayerNorm.weight" in k) else 0.01
            ref[k], m[k], v2[k] = O.adamw_update(ref[k], g, m[k], v2[k], t, lr, weight_decay=wd)
            ref[k] = ref[k].detach()
        for k in names:
            d = (tr.store.view(k) - ref[k]).abs().max().item()
            assert d < 2e-5, (t, k, d)
    # tensors without a gradient are never touched (the reference's AdamW skips grad-None tensors)
    assert torch.equal(tr.store.view("bert.pooler.dense.weight"), sd["bert.pooler.dense.weight"])


def test_text_longer_than_64_tokens_step_matches_oracle():
    """--max_text_length above 64 (ref param.py:140): the engine's buffers, packed rows and attention calls at L = 72 (language
    self-attention 72 x 72, cross-attention 72 x 16 and 16 x 72) -- one step against the oracle's gradients (host restatement of the
    kernels here; the long-sequence kernels themselves against the same restatement in tests/test_hip_kernels.py)."""
    cfg = XLxmertConfig(**dict(TINY, max_position_embeddings=128))
    B, L, grid = 2, 72, 4
    tr, sd = make_step(cfg, B, L, grid, lr=1e-2)
    batch = synthetic_batch(cfg, B, L, grid, seed=94, ragged=False)          # every sentence 72 tokens long
    assert int(batch["attention_mask"].sum(1).min()) == 72
    losses = tr.step(batch)
    grads, out = oracle_grads(cfg, sd, batch)
    assert abs(losses[0].item() - out["obj_loss"].item()) < 3e-5
    for k, g in grads.items():
        if k != "obj_predict_head.out_cluster.weight":
            assert (tr.store.gview(k) - g).abs().max().item() < 3e-5, k


def test_visual_losses_default_is_obj_only_and_feat_labels_are_used():
    """--visualLosses obj (scripts/pretrain.bash:15) is the default: no feature loss, gradients = those of obj_loss alone.
    With "obj,feat" and batch["feat_labels"] (the real grid features, ref lxmert_pretrain.py:177-179) the SmoothL1 term
    regresses onto THOSE targets (ref lxrt/modeling.py:273-287), not onto the centroids."""
    cfg = XLxmertConfig(**TINY)
    oc = oracle_cfg(cfg)
    B, L, grid = 3, 8, 4
    batch = synthetic_batch(cfg, B, L, grid, seed=77)

    def oracle(feat_labels, with_feat):
        leaf = {k: v.clone().requires_grad_(v.is_floating_point() and k != "vis_emb.weight")
                for k, v in O.make_state_dict(oc, 3).items()}
        leaf["obj_predict_head.out_cluster.weight"] = leaf["vis_emb.weight"]
        out = O.xlxmert_vis_mask_forward(leaf, oc, batch["input_ids"], batch["visual_pos"], batch["attention_mask"],
                                         batch["cluster_ids"], batch["vis_mask"], batch["obj_labels"], feat_labels=feat_labels)
        (out["total_loss"] if with_feat else out["obj_loss"]).backward()
        return out, {k: v.grad for k, v in leaf.items() if v.grad is not None and k != "obj_predict_head.out_cluster.weight"}

    # default: obj only
    tr, sd = make_step(cfg, B, L, grid, visual_losses="obj")
    assert tr.feat_loss is False and PretrainStep.__init__.__defaults__ is not None
    tr.engine.set_inputs(batch["input_ids"], batch["attention_mask"], None, batch["visual_pos"], cluster_ids=batch["cluster_ids"],
                         vis_mask=batch["vis_mask"], obj_labels=batch["obj_labels"])
    losses = tr.engine.vis_mask_forward_backward(tr.feat_loss)
    out, grads = oracle(None, False)
    assert abs(losses[0].item() - out["obj_loss"].item()) < 3e-5 and losses[1].item() == 0.0
    for k, g in grads.items():
        assert (tr.store.gview(k) - g).abs().max().item() < 2e-5, k
    # obj,feat with explicit targets
    tgt = torch.randn(B, grid * grid, cfg.visual_feat_dim, generator=torch.Generator().manual_seed(5)).relu()
    tr, sd = make_step(cfg, B, L, grid, visual_losses="obj,feat")
    b2 = dict(batch, feat_labels=tgt)
    losses = tr.step(b2)
    out, grads = oracle(tgt, True)
    assert abs(losses[1].item() - out["feat_loss"].item()) < 3e-5
    out_c, _ = oracle(None, True)
    assert abs(out["feat_loss"].item() - out_c["feat_loss"].item()) > 1e-3        # the targets really differ from the centroids


@pytest.mark.parametrize("grouped", [False, True])
def test_vqa_two_steps_match_closed_form(grouped):
    """VQA fine-tune step semantics (BCE loss, clip 1.0, 4.1.1-AdamW incl. the decayed `logit_fc.2.weight`
    LayerNorm, linear schedule) against the oracle, two consecutive updates."""
    cfg = XLxmertConfig(**TINY)
    oc = oracle_cfg(cfg)
    B, L, grid, A = 3, 8, 4, 29
    store = ParamStore(cfg, "cpu", torch.float32, task="vqa", num_answers=A)
    sd = O.make_vqa_state_dict(oc, A, 5)
    store.load_named(sd)
    tr = PretrainStep(cfg, B, L, grid * grid, dtype=torch.float32, device="cpu", store=store, ops=FakeOps(torch.float32),
                      total_steps=10, lr=1e-2, weight_decay=0.01, warmup_ratio=0.2, task="vqa", num_answers=A,
                      overlap_optimizer=grouped)        # grouped: the optimizer pass issued group by group in forward order
    assert (tr._opt_groups is not None) == grouped and tr.opt_stream is None
    names = [n for n in store.index if store.index[n] is not None and
             any(m.name == n for u in store.units if u.used for m in u.members)]
    ref = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in ref.items()}
    v2 = {k: torch.zeros_like(v) for k, v in ref.items()}
    for t in (1, 2):
        batch = O.make_vqa_inputs(oc, A, 200 + t, B, L, grid)
        loss = tr.step(batch)
        leaf = {k: v.clone().requires_grad_(True) for k, v in ref.items()}
        out = O.vqa_forward(leaf, oc, batch["input_ids"], batch["visual_feats"], batch["visual_pos"], targets=batch["targets"])
        assert abs(loss.item() - out["loss"].item()) < 3e-6
        out["loss"].backward()
        gn = sorted(k for k in leaf if leaf[k].grad is not None)
        assert gn == sorted(names)          # the optimizer range == the reference's set of grad-carrying tensors
        norm, clipped = O.clip_grad_norm([leaf[k].grad for k in gn], 1.0)
        assert abs(tr.grad_norm() - norm.item()) < 1e-4 * max(1.0, norm.item())
        lr = 1e-2 * linear_schedule(t - 1, 2, 10)
        for k, g in zip(gn, clipped):
            wd = 0.0 if ("bias" in k or "LayerNorm.weight" in k) else 0.01
            ref[k], m[k], v2[k] = O.adamw_update(ref[k], g, m[k], v2[k], t, lr, weight_decay=wd)
            ref[k] = ref[k].detach()
        for k in gn:
            d = (tr.store.view(k) - ref[k]).abs().max().item()
            assert d < 2e-5, (t, k, d)
    dead = "bert.encoder.x_layers.1.visn_inter.dense.weight"       # visual side of the last cross layer: never updated
    assert torch.equal(tr.store.view(dead), sd[dead])


def test_nlvr2_step_matches_closed_form():
    """SURVEY 8f N1, NLVR2: one fine-tune step (pairs flattened by the trainer, CE over 2 classes, clip, AdamW) vs the oracle."""
    cfg = XLxmertConfig(**TINY)
    oc = oracle_cfg(cfg)
    P, L, grid = 3, 8, 4
    sd = O.make_nlvr2_state_dict(oc, 5)
    tr = PretrainStep(cfg, 2 * P, L, grid * grid, dtype=torch.float32, device="cpu", ops=FakeOps(torch.float32),
                      total_steps=10, lr=1e-2, weight_decay=0.01, warmup_ratio=0.2, task="nlvr2")
    tr.store.load_named(sd)
    tr.engine.sync_compute_weights()
    batch = O.make_nlvr2_inputs(oc, 77, P, L, grid)
    loss = tr.step(batch)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.nlvr2_forward(leaf, oc, batch["input_ids"], batch["visual_feats"], batch["visual_pos"], labels=batch["labels"])
    assert abs(loss.item() - out["loss"].item()) < 3e-6
    out["loss"].backward()
    gn = sorted(k for k in leaf if leaf[k].grad is not None)
    used = sorted(m.name for u in tr.store.units if u.used for m in u.members)
    assert gn == used
    norm, clipped = O.clip_grad_norm([leaf[k].grad for k in gn], 1.0)
    assert abs(tr.grad_norm() - norm.item()) < 1e-4 * max(1.0, norm.item())
    lr = 1e-2 * linear_schedule(0, 2, 10)
    for k, g in zip(gn, clipped):
        wd = 0.0 if ("bias" in k or "LayerNorm.weight" in k) else 0.01
        ref, _, _ = O.adamw_update(sd[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, lr, weight_decay=wd)
        assert (tr.store.view(k) - ref.detach()).abs().max().item() < 2e-5, k


@pytest.mark.parametrize("grouped", [False, True])
def test_task_round_robin_on_one_parameter_set(grouped):
    """vis_mask -> word_mask -> matched -> vis_mask on ONE parameter set (ref lxmert_pretrain.py:296-298).
    The reference sets .grad = None after every step, so AdamW touches only the tensors of the step's branch and each
    tensor keeps its own update count (bias correction); checked against the oracle with that per-tensor state."""
    cfg = XLxmertConfig(**TINY)
    oc = oracle_cfg(cfg)
    B, L, grid = 3, 8, 4
    store = ParamStore(cfg, "cpu", torch.float32, task="all")
    sd = O.make_cls_state_dict(oc, 11)
    store.load_named(sd)
    tr = PretrainStep(cfg, B, L, grid * grid, dtype=torch.float32, device="cpu", store=store, ops=FakeOps(torch.float32),
                      total_steps=10, lr=1e-2, weight_decay=0.01, warmup_ratio=0.2, task="all", visual_losses="obj,feat",
                      overlap_optimizer=grouped)      # grouped: per-chunk skip flags and update counts sliced per parameter group
    ref = {k: v.clone() for k, v in sd.items() if k != "cls.predictions.decoder.weight"}
    ref["obj_predict_head.out_cluster.weight"] = ref["vis_emb.weight"]
    m = {k: torch.zeros_like(v) for k, v in ref.items()}
    v2 = {k: torch.zeros_like(v) for k, v in ref.items()}
    nstep = {k: 0 for k in ref}
    for t, task in enumerate(["vis_mask", "word_mask", "matched", "vis_mask"], start=1):
        batch = synthetic_batch(cfg, B, L, grid, seed=300 + t)
        wl, ml = O.make_lang_task_labels(oc, batch["input_ids"], 400 + t)
        batch["word_labels"], batch["matched_labels"] = wl, ml
        loss = tr.step(batch, task=task)
        leaf = {k: v.clone().requires_grad_(v.is_floating_point() and k not in ("vis_emb.weight", "obj_predict_head.out_cluster.weight"))
                for k, v in ref.items()}
        leaf["obj_predict_head.out_cluster.weight"] = leaf["vis_emb.weight"]
        if task == "vis_mask":
            out = O.xlxmert_vis_mask_forward(leaf, oc, batch["input_ids"], batch["visual_pos"], batch["attention_mask"],
                                             batch["cluster_ids"], batch["vis_mask"], batch["obj_labels"])
            assert abs(loss[0].item() - out["obj_loss"].item()) < 3e-5
        elif task == "word_mask":
            out = O.xlxmert_word_mask_forward(leaf, oc, batch["input_ids"], batch["visual_pos"], batch["attention_mask"],
                                              batch["cluster_ids"], wl)
            assert abs(loss.item() - out["total_loss"].item()) < 3e-5
        else:
            out = O.xlxmert_matched_forward(leaf, oc, batch["input_ids"], batch["visual_pos"], batch["attention_mask"],
                                            batch["cluster_ids"], ml)
            assert abs(loss.item() - out["total_loss"].item()) < 3e-5
        out["total_loss"].backward()
        names = sorted(k for k, v in leaf.items() if v.grad is not None)
        norm, clipped = O.clip_grad_norm([leaf[k].grad for k in names], 1.0)
        assert abs(tr.grad_norm() - norm.item()) < 1e-4 * max(1.0, norm.item())
        lr = 1e-2 * linear_schedule(t - 1, 2, 10)
        for k, g in zip(names, clipped):
            nstep[k] += 1
            wd = 0.0 if ("bias" in k or "LayerNorm.weight" in k) else 0.01
            ref[k], m[k], v2[k] = O.adamw_update(ref[k], g, m[k], v2[k], nstep[k], lr, weight_decay=wd)
            ref[k] = ref[k].detach()
        for k in ref:
            if k in store.index:
                d = (tr.store.view(k) - ref[k]).abs().max().item()
                assert d < 3e-5, (t, task, k, d)
    assert nstep["obj_predict_head.linear_feat.weight"] == 2 and nstep["cls.seq_relationship.weight"] == 1


def _free_port():
    """A listening port BELOW the kernel's ephemeral range (32768+): a port handed out by bind(0) can be taken by an outgoing
    connection of the previous test's gloo pairs before the store listens on it (EADDRINUSE seen once on the GPU box)."""
    import random
    rng = random.Random(os.getpid() * 7919 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        p = rng.randrange(15000, 30000)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", p))
        except OSError:
            continue
        finally:
            s.close()
        return p
    raise RuntimeError("no free port")


def _dp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = XLxmertConfig(**TINY)
    B, L, grid = 2, 8, 4
    tr, sd = make_step(cfg, B, L, grid, lr=1e-2, bucket_mb=0.05)       # several overlapped buckets
    batch = synthetic_batch(cfg, B, L, grid, seed=500 + rank)         # disjoint per-rank minibatch
    tr.step(batch)
    assert len(tr._works) > 3, len(tr._works)
    # two growing ranges: the language stream's block (language layers + embeddings) is exchanged while the visual
    # stack's prefix is still growing, and the slices tile the used range exactly
    lo, hi = tr.store.language_range()
    order = [("l" if lo <= a < hi else "v") for a, _ in tr._slices]
    assert "l" in order and order.index("l") < len(order) - 1 - order[::-1].index("v"), order
    assert sum(b - a for a, b in tr._slices) == tr.store.n_used
    assert tr._slices[-1][1] == tr.store.n_used                        # the visual feature encoder closes the step
    torch.save({k: tr.store.view(k).clone() for k in tr.store.names()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_world2_gloo(tmp_path):
    """2 ranks x different minibatches: replicas stay identical and equal one AdamW step on the MEAN of the per-rank
    gradients (DDP semantics: mean of per-shard mean losses, SURVEY 8e caveat)."""
    world, port = 2, _free_port()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k
    cfg = XLxmertConfig(**TINY)
    sd = O.make_state_dict(oracle_cfg(cfg), 3)
    gs = [oracle_grads(cfg, sd, synthetic_batch(cfg, 2, 8, 4, seed=500 + r))[0] for r in range(world)]
    names = sorted(gs[0])
    mean = [(gs[0][k] + gs[1][k]) / 2 for k in names]
    _, clipped = O.clip_grad_norm(mean, 1.0)
    lr = 1e-2 * linear_schedule(0, 0, 10)
    for k, g in zip(names, clipped):
        p, _, _ = O.adamw_update(sd[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, lr)
        assert (r0[k] - p).abs().max().item() < 2e-5, k


def _dp_worker_sharded(rank, world, port, out_dir):
    """the same three steps through both collectives: all-reduce + replicated AdamW, and reduce-scatter ->
    shard-local norm / AdamW -> all-gather of the master weights."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = XLxmertConfig(**TINY)
    B, L, grid = 2, 8, 4
    res = {}
    for name, kw in (("ar", dict(collective="allreduce")), ("rs", dict(collective="rs+ag")),
                     ("ar_noclip", dict(collective="allreduce", clip_grad_norm=0.0)), ("rs_noclip", dict(collective="rs+ag", clip_grad_norm=0.0)),
                     ("rs_bf16", dict(collective="rs+ag", grad_comm_dtype=torch.bfloat16)),
                     ("ar_bf16", dict(collective="allreduce", grad_comm_dtype=torch.bfloat16))):
        tr, sd = make_step(cfg, B, L, grid, lr=1e-2, weight_decay=0.01, bucket_mb=0.05, **kw)
        assert tr.collective == kw["collective"] and tr.sharded == (kw["collective"] == "rs+ag")
        for t in range(3):
            tr.step(synthetic_batch(cfg, B, L, grid, seed=500 + 10 * t + rank))
            if tr.sharded:
                # this rank's pass covers its piece of every scattered slice + the replicated tails: half of the buffer, and the
                # pieces are whole 256-element optimizer chunks
                own = tr.owned_ranges()
                assert all(a % 256 == 0 and b % 256 == 0 for a, b in own)
                n_rs = sum(hi - lo for k, lo, hi in tr._segments if k == "rs")
                n_ar = sum(hi - lo for k, lo, hi in tr._segments if k == "ar")
                assert n_rs + n_ar == tr.store.n_used and n_ar < 256 * world * len(tr._slices)
                assert sum(b - a for a, b in own) == n_rs // world + n_ar
        # master weights and compute copy are whole and identical on both ranks after every step; the Adam moments once gathered
        assert tr.verify_replicas() == [], (name, tr.verify_replicas()[:4])
        res[name] = {k: tr.store.view(k).clone() for k in tr.store.names()}
        res[name + ":m"] = tr.store.exp_avg[:tr.store.n_used].clone()
        res[name + ":norm"] = tr.grad_norm()
    torch.save(res, os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _dp_worker_bf16_gather(rank, world, port, out_dir):
    """collective="rs+ag" on a bf16 compute copy: the all-gather of the fp32 master slices against gather="bf16" (the compute copy
    on the wire + the sparse fp32 side car of the elements that are read in fp32)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = XLxmertConfig(**TINY)
    B, L, grid = 2, 8, 4
    res = {}
    for name, kw in (("g32", dict(gather="fp32")), ("g16", dict(gather="bf16"))):
        store = ParamStore(cfg, "cpu", torch.bfloat16, task="vis_mask")
        store.load_named(O.make_state_dict(oracle_cfg(cfg), 3))
        tr = PretrainStep(cfg, B, L, grid * grid, dtype=torch.bfloat16, device="cpu", store=store, ops=FakeOps(torch.bfloat16),
                          total_steps=10, lr=1e-2, weight_decay=0.01, bucket_mb=0.05, collective="rs+ag", visual_losses="obj,feat", **kw)
        assert tr.sharded and tr.gather_bf16 == (name == "g16")
        for t in range(3):
            tr.step(synthetic_batch(cfg, B, L, grid, seed=500 + 10 * t + rank))
        st = tr.store
        res[name + ":compute"] = st.compute[:st.n_used].clone()
        # the elements the kernels read in fp32 (biases, LayerNorm affines ...) are whole in the master buffer on every rank
        idx = st.fp32_read_index(0, st.n_used).long()
        res[name + ":fp32_read"] = st.master[idx].clone()
        if name == "g16":                      # ... the matrices' fp32 master is the owner's business until gather_state()
            n_rs = sum(hi - lo for k, lo, hi in tr._segments if k == "rs")
            assert 0 < idx.numel() < 0.2 * st.n_used and n_rs > 0.9 * st.n_used
        assert tr.verify_replicas() == [], (name, tr.verify_replicas()[:4])       # (gathers the state first)
        res[name + ":master"] = st.master[:st.n_used].clone()
    torch.save(res, os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_all_gather_equals_fp32_gather_world2_gloo(tmp_path):
    """VERDICT r04 item 8 / SURVEY 5.8 "bf16 on the wire": in rs+ag mode the updated parameters travel back as the bf16 compute copy
    (half the bytes) + a sparse fp32 side car for the elements read in fp32; compute copy, fp32-read elements and (after
    gather_state) the whole master buffer equal the fp32-gather path's bit for bit, on both ranks."""
    world, port = 2, _free_port()
    mp.spawn(_dp_worker_bf16_gather, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k                               # replicas identical
    for what in ("compute", "fp32_read", "master"):
        assert torch.equal(r0["g32:" + what], r0["g16:" + what]), what     # and equal to the fp32-gather path


def test_sharded_exchange_equals_allreduce_world2_gloo(tmp_path):
    """collective="rs+ag": replicas identical, and after three steps the parameters equal the all-reduce path's -- bit for bit
    without clipping (two ranks: a + b either way), to 1e-6 with it (the norm's summation order differs: shard-local partial
    sums + one scalar all-reduce), also with bf16 buckets; the first moments agree as well."""
    world, port = 2, _free_port()
    mp.spawn(_dp_worker_sharded, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    for name in ("ar", "rs", "ar_noclip", "rs_noclip", "rs_bf16", "ar_bf16"):
        for k in r0[name]:
            assert torch.equal(r0[name][k], r1[name][k]), (name, k)
        assert torch.equal(r0[name + ":m"], r1[name + ":m"]), name
    for k in r0["ar"]:
        assert torch.equal(r0["ar_noclip"][k], r0["rs_noclip"][k]), k
        assert (r0["ar"][k] - r0["rs"][k]).abs().max().item() < 1e-6, k
        assert (r0["ar_bf16"][k] - r0["rs_bf16"][k]).abs().max().item() < 1e-6, k
    assert torch.equal(r0["ar_noclip:m"], r0["rs_noclip:m"])
    assert abs(r0["ar:norm"] - r0["rs:norm"]) < 1e-5 * max(1.0, r0["ar:norm"])


def _dp_worker_modes(rank, world, port, out_dir):
    """bf16 gradient buckets; VQA step; task round-robin with the QA head; per-tensor replica verification."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = XLxmertConfig(**TINY)
    oc = oracle_cfg(cfg)
    B, L, grid = 2, 8, 4
    res = {}
    # (1) masked-visual-token step with bf16 buckets
    tr, sd = make_step(cfg, B, L, grid, lr=1e-2, bucket_mb=0.05, grad_comm_dtype=torch.bfloat16)
    assert tr.comm_buf is not None and tr.comm_buf.dtype == torch.bfloat16
    tr.step(synthetic_batch(cfg, B, L, grid, seed=500 + rank))
    assert tr.verify_replicas() == []
    res["bf16"] = {k: tr.store.view(k).clone() for k in tr.store.names()}
    # (2) VQA fine-tune step
    A = 29
    store = ParamStore(cfg, "cpu", torch.float32, task="vqa", num_answers=A)
    store.load_named(O.make_vqa_state_dict(oc, A, 5))
    trv = PretrainStep(cfg, B, L, grid * grid, dtype=torch.float32, device="cpu", store=store, ops=FakeOps(torch.float32),
                       total_steps=10, lr=1e-2, task="vqa", num_answers=A, bucket_mb=0.05)
    trv.step(O.make_vqa_inputs(oc, A, 600 + rank, B, L, grid))
    assert trv.verify_replicas() == [] and len(trv._works) >= 2       # (paired weight-gradient launches report every second layer)
    res["vqa"] = {k: trv.store.view(k).clone() for k in trv.store.names()}
    # (3) task round-robin on one parameter set with the QA head riding on every branch
    NQ = 11
    store = ParamStore(cfg, "cpu", torch.float32, task="all", num_answers=NQ)
    store.load_named(O.make_qa_state_dict(oc, NQ, 9))
    tra = PretrainStep(cfg, B, L, grid * grid, dtype=torch.float32, device="cpu", store=store, ops=FakeOps(torch.float32),
                       total_steps=10, lr=1e-2, task="all", num_answers=NQ, bucket_mb=0.05, visual_losses="obj,feat")
    for t, task in enumerate(["vis_mask", "word_mask", "matched", "qa"]):
        batch = synthetic_batch(cfg, B, L, grid, seed=700 + 10 * t + rank)
        wl, ml = O.make_lang_task_labels(oc, batch["input_ids"], 800 + 10 * t + rank)
        batch.update(word_labels=wl, matched_labels=ml, qa_labels=O.make_qa_labels(NQ, B, 900 + 10 * t + rank))
        tra.step(batch, task=task)
        assert tra.verify_replicas() == [], (task, tra.verify_replicas()[:4])
    res["all"] = {k: tra.store.view(k).clone() for k in tra.store.names()}
    # (4) a diverged replica is caught, tensor by tensor
    if rank == 1:
        tra.store.view("bert.pooler.dense.bias")[3] += 1.0
    bad = tra.verify_replicas()
    assert bad == ["param:bert.pooler.dense.bias"], bad
    torch.save(res, os.path.join(out_dir, f"m{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("collective", ["allreduce", "rs+ag"])
def test_data_parallel_world2_gloo_bf16_buckets_vqa_and_round_robin(tmp_path, monkeypatch, collective):
    """world 2 over gloo: (1) bf16 gradient buckets = one AdamW step on the mean of the per-rank gradients up to the bf16
    rounding of the exchanged values; (2) VQA step and (3) task round-robin with the QA head keep the replicas bit-identical
    and equal the oracle's mean-gradient update; (4) verify_replicas names a diverged tensor.  Both collectives: all-reduce with
    the replicated optimizer pass, and reduce-scatter -> shard-local AdamW -> all-gather (per-chunk skip flags and update counts
    sliced per shard)."""
    world, port = 2, _free_port()
    monkeypatch.setenv("XL_COLLECTIVE", collective)          # (inherited by the spawned ranks)
    mp.spawn(_dp_worker_modes, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "m0.pt"), torch.load(tmp_path / "m1.pt")
    for mode in ("bf16", "vqa"):
        for k in r0[mode]:
            assert torch.equal(r0[mode][k], r1[mode][k]), (mode, k)
    cfg = XLxmertConfig(**TINY)
    oc = oracle_cfg(cfg)
    # bf16 buckets against the exact mean-gradient update: first-step AdamW moves every element by ~lr * sign(g); a bf16-
    # rounded gradient changes that by < 1 % of lr except where the mean gradient is within rounding of zero
    sd = O.make_state_dict(oc, 3)
    gs = [oracle_grads(cfg, sd, synthetic_batch(cfg, 2, 8, 4, seed=500 + r))[0] for r in range(world)]
    names = sorted(gs[0])
    mean = [(gs[0][k] + gs[1][k]) / 2 for k in names]
    _, clipped = O.clip_grad_norm(mean, 1.0)
    lr = 1e-2 * linear_schedule(0, 0, 10)
    close = tot = 0
    for k, g in zip(names, clipped):
        p, _, _ = O.adamw_update(sd[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, lr)
        d = (r0["bf16"][k] - p).abs()
        close += (d < 2e-2 * lr).sum().item()
        tot += d.numel()
        assert d.max().item() <= 2.0 * lr + 1e-6, k
    assert close / tot > 0.97, close / tot
    # VQA: exact (fp32 exchange)
    A = 29
    sdv = O.make_vqa_state_dict(oc, A, 5)
    grads = []
    for r in range(world):
        leaf = {k: v.clone().requires_grad_(True) for k, v in sdv.items()}
        b = O.make_vqa_inputs(oc, A, 600 + r, 2, 8, 4)
        O.vqa_forward(leaf, oc, b["input_ids"], b["visual_feats"], b["visual_pos"], targets=b["targets"])["loss"].backward()
        grads.append({k: v.grad for k, v in leaf.items() if v.grad is not None})
    names = sorted(grads[0])
    _, clipped = O.clip_grad_norm([(grads[0][k] + grads[1][k]) / 2 for k in names], 1.0)
    for k, g in zip(names, clipped):
        p, _, _ = O.adamw_update(sdv[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, lr)
        assert (r0["vqa"][k] - p).abs().max().item() < 2e-5, k


def test_random_word_batch_statistics():
    from xlxmert_amd.trainer import random_word_batch
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(1000, 30000, (400, 20), generator=g)
    masked, labels = random_word_batch(ids, generator=g)
    chosen = labels != -100
    assert not chosen[:, 0].any() and not chosen[:, -1].any()
    assert torch.equal(labels[chosen], ids[chosen]) and torch.equal(masked[~chosen], ids[~chosen])
    frac = chosen[:, 1:-1].float().mean().item()
    assert 0.13 < frac < 0.17
    is_mask = (masked == 103) & chosen
    assert 0.75 < is_mask.sum().item() / chosen.sum().item() < 0.85
    kept = (masked == ids) & chosen
    assert 0.06 < kept.sum().item() / chosen.sum().item() < 0.14


@pytest.mark.parametrize("overwrite", [False, True])
def test_optimizer_pass_clears_the_gradients_when_asked(overwrite):
    """drop_grads: xl_adamw clears the gradient buffer in its own pass and the next step's backward starts without a clear --
    same losses and parameters as the step that clears before every backward; the gradients are gone after step().
    overwrite (PretrainStep overwrite_grads, the default): the weight gradients with one contribution per step are STORED by the
    next backward, so the pass leaves them uncleared (flag bit 2) -- the host restatement of xl_adamw poisons exactly those chunks
    with NaN, so a path that accumulated into one of them instead of overwriting it could not give the reference's numbers."""
    cfg = XLxmertConfig(**TINY)
    B, L, grid = 3, 8, 4
    keep, _ = make_step(cfg, B, L, grid, lr=1e-2, overwrite_grads=False)
    drop, _ = make_step(cfg, B, L, grid, lr=1e-2, drop_grads=True, overwrite_grads=overwrite)
    assert not keep.drop_grads and drop.drop_grads and drop.overwrite_grads == overwrite
    for t in range(3):
        batch = synthetic_batch(cfg, B, L, grid, seed=200 + t)
        lk, ld = keep.step(batch).clone(), drop.step(batch).clone()
        assert torch.equal(lk, ld)
        assert keep.grad_norm() == drop.grad_norm()
        assert keep.store.grad[:keep.store.n_used].abs().max().item() > 0
        g = drop.store.grad[:drop.store.n_used]
        kept = ((drop.store.decay_flags[:drop.store.n_used // 256] & 4) != 0).repeat_interleave(256)
        assert bool(kept.any()) == overwrite
        if overwrite:            # most of the buffer: every Linear weight of the encoder and the head
            assert kept.float().mean().item() > 0.5 and torch.isnan(g[kept]).all()
        assert g[~kept].abs().max().item() == 0 and drop.engine.grad_is_zero
        assert torch.equal(keep.store.master, drop.store.master)
    # a backward outside the trainer dirties the buffer again: the next step clears it
    drop.engine.set_inputs(batch["input_ids"], batch["attention_mask"], None, batch["visual_pos"], cluster_ids=batch["cluster_ids"],
                           vis_mask=batch["vis_mask"], obj_labels=batch["obj_labels"])
    drop.engine.vis_mask_forward_backward(True)
    assert not drop.engine.grad_is_zero
    batch = synthetic_batch(cfg, B, L, grid, seed=300)
    assert torch.equal(keep.step(batch), drop.step(batch)) and torch.equal(keep.store.master, drop.store.master)


def test_overwrite_mode_survives_a_k_split_launch_on_a_kept_range():
    """Round-4 advisor finding: the grouped weight-gradient launch decides per launch whether it has one writer per tile
    (xl_gemm_wgrad_group_splitk, a function of K = the packed language row count, which moves across the K / 512 threshold from
    batch to batch).  The engine used to DROP the overwrite mask for a K-split launch -- into a range the optimizer pass no longer
    clears: the launch then accumulated onto the previous step's gradient.  Here the one-writer answer alternates from step to
    step; the drop_grads + overwrite trainer must stay bit-identical to the clear-then-accumulate trainer (a range already kept
    keeps its mask bit: xl_gemm_wgrad_group clears C itself in front of a split launch)."""
    cfg = XLxmertConfig(**TINY)
    B, L, grid = 3, 8, 4
    ref, _ = make_step(cfg, B, L, grid, lr=1e-2, overwrite_grads=False)
    got, _ = make_step(cfg, B, L, grid, lr=1e-2, drop_grads=True, overwrite_grads=True)
    state = {"t": 0, "asked": 0}

    def one_writer(problems):
        state["asked"] += 1
        return state["t"] % 2 == 0          # even steps: one writer per tile; odd steps: "this launch splits K"
    got.ops.wgrad_group_one_writer = one_writer
    for t in range(5):
        state["t"] = t
        batch = synthetic_batch(cfg, B, L, grid, seed=900 + t)
        lr_, lg = ref.step(batch).clone(), got.step(batch).clone()
        assert torch.equal(lr_, lg), t
        assert torch.equal(ref.store.master, got.store.master), t
        assert torch.isfinite(got.store.master).all()
    assert state["asked"] > 0


def test_kept_range_that_a_step_does_not_write_sees_a_zero_gradient():
    """Round-4 advisor finding: mark_overwritten is permanent, so a step that skips a producer (here: a weight-gradient launch the
    test suppresses on one step) would re-apply the previous step's gradient.  optimizer_step zeroes every kept range no backward
    has stored since the last pass: the parameters must equal those of the clear-then-accumulate trainer, whose buffer was cleared."""
    cfg = XLxmertConfig(**TINY)
    B, L, grid = 3, 8, 4
    ref, _ = make_step(cfg, B, L, grid, lr=1e-2, overwrite_grads=False)
    got, _ = make_step(cfg, B, L, grid, lr=1e-2, drop_grads=True, overwrite_grads=True)
    skip = {"on": False}
    for tr in (ref, got):
        orig = tr.ops.gemm_wgrad_group

        def grouped(problems, overwrite_mask=0, _orig=orig, _eng=tr.engine):
            if skip["on"]:                               # the first problem of every launch "has no producer" on this step
                if overwrite_mask & 1:
                    _eng.written_now.discard(_eng._flat_range(problems[0][2], problems[0][3], problems[0][4], problems[0][8]))
                problems, overwrite_mask = problems[1:], overwrite_mask >> 1
                if not problems:
                    return
            _orig(problems, overwrite_mask=overwrite_mask)
        tr.ops.gemm_wgrad_group = grouped
    for t in range(4):
        skip["on"] = t == 2
        batch = synthetic_batch(cfg, B, L, grid, seed=950 + t)
        lr_, lg = ref.step(batch).clone(), got.step(batch).clone()
        assert torch.equal(lr_, lg), t
        assert torch.equal(ref.store.master, got.store.master), t


@pytest.mark.parametrize("task", ["vis_mask_accum", "word_mask", "matched", "vqa", "nlvr2"])
def test_overwritten_weight_gradients_every_task(task):
    """PretrainStep(overwrite_grads=True, drop_grads=True) against the clear-then-accumulate step, bit for bit, for every task and
    for gradient-accumulation windows (first micro-batch overwrites, the others add).  The host restatement of xl_adamw writes
    NaN into every chunk it is told not to clear: a weight gradient that some branch accumulated instead of storing -- or did not
    touch at all -- would poison the parameters (ParamStore.mark_overwritten is the static claim, this is its check)."""
    cfg = XLxmertConfig(**TINY)
    oc = oracle_cfg(cfg)
    B, L, grid = 2, 8, 4
    A = 29

    def make(overwrite):
        kw = dict(dtype=torch.float32, device="cpu", ops=FakeOps(torch.float32), total_steps=20, lr=1e-2, weight_decay=0.01,
                  drop_grads=True, overwrite_grads=overwrite)
        if task == "vis_mask_accum":
            tr, _ = make_step(cfg, B, L, grid, lr=1e-2, weight_decay=0.01, drop_grads=True, overwrite_grads=overwrite)
            return tr
        if task in ("vqa", "nlvr2"):
            store = ParamStore(cfg, "cpu", torch.float32, task=task, num_answers=A if task == "vqa" else 0)
            store.load_named(O.make_vqa_state_dict(oc, A, 5) if task == "vqa" else O.make_nlvr2_state_dict(oc, 5))
            return PretrainStep(cfg, B if task == "vqa" else 2 * B, L, grid * grid, store=store, task=task,
                                num_answers=A if task == "vqa" else 0, **kw)
        store = ParamStore(cfg, "cpu", torch.float32, task=task)
        store.load_named(O.make_cls_state_dict(oc, 5))
        return PretrainStep(cfg, B, L, grid * grid, store=store, task=task, **kw)

    def batch_of(t):
        if task == "vqa":
            return O.make_vqa_inputs(oc, A, 600 + t, B, L, grid)
        if task == "nlvr2":
            return O.make_nlvr2_inputs(oc, 650 + t, B, L, grid)
        b = synthetic_batch(cfg, B, L, grid, seed=700 + t)
        if task in ("word_mask", "matched"):
            wl, ml = O.make_lang_task_labels(oc, b["input_ids"], 800 + t)
            b.update(word_labels=wl, matched_labels=ml)
        return b

    ref, got = make(False), make(True)
    assert got.overwrite_grads and not ref.overwrite_grads and got.drop_grads
    for t in range(4):
        update = task != "vis_mask_accum" or t % 2 == 1          # accumulation: windows of two micro-batches
        b = batch_of(t)
        lr_, lg = ref.step(b, update=update), got.step(b, update=update)
        assert torch.equal(torch.as_tensor(lr_).float(), torch.as_tensor(lg).float()), t
        assert torch.equal(ref.store.master, got.store.master), t
        assert torch.isfinite(got.store.master).all()
    n = got.store.n_used
    kept = ((got.store.decay_flags[:n // 256] & 4) != 0).float().mean().item()
    assert kept > 0.4, kept                                        # the Linear weights: most of the buffer (tiny model: ~60 %)


@pytest.mark.parametrize("task,num_answers", [("vis_mask", 0), ("vqa", 29), ("nlvr2", 0), ("all", 0), ("word_mask", 13)])
def test_forward_groups_tile_the_optimizer_range(task, num_answers):
    """ParamStore.forward_groups (the order in which an optimizer pass queued behind the step updates the parameters): the ranges
    tile [0, n_used) exactly, every range is chunk-aligned and holds tensors of ONE group, and the groups come in the order the
    forward first reads them (feature encoder and embeddings, language / visual layers ascending, cross layers, heads last)."""
    cfg = XLxmertConfig(**TINY)
    st = ParamStore(cfg, "cpu", torch.float32, task=task, num_answers=num_answers)
    groups = st.forward_groups()
    assert sum(hi - lo for _, lo, hi in groups) == st.n_used
    covered = sorted((lo, hi) for _, lo, hi in groups)
    assert covered[0][0] == 0 and covered[-1][1] == st.n_used and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert all(lo % 256 == 0 and hi % 256 == 0 for _, lo, hi in groups)
    for key, lo, hi in groups:
        for u in st.units:
            if u.used and lo <= u.offset < hi:
                assert all(ParamStore.group_key(m.name) == key for m in u.members), (key, [m.name for m in u.members])
    keys = [k for k, _, _ in groups]
    assert keys[0] == "visn" and keys[1] == "emb" and keys[-1] == "heads"
    for kind, n in (("lang", cfg.l_layers), ("vis", cfg.r_layers), ("x", cfg.x_layers)):
        idx = [keys.index((kind, i)) for i in range(n)]
        assert idx == sorted(idx)
    assert max(keys.index(("lang", cfg.l_layers - 1)), keys.index(("vis", cfg.r_layers - 1))) < keys.index(("x", 0))


def test_gradient_accumulation_update_freq():
    """--update_freq (ref tasks/vqa.py:152-159, 189-198): backward of several minibatches accumulates, ONE clip + AdamW step on
    the summed gradient, gradients dropped afterwards.  step(update=False) twice + step(update=True) == one AdamW update of the
    oracle on the SUM of the three minibatches' gradients; the next window starts from a clean buffer."""
    cfg = XLxmertConfig(**TINY)
    B, L, grid = 3, 8, 4
    tr, sd = make_step(cfg, B, L, grid, lr=1e-2, warmup_ratio=0.0)
    batches = [synthetic_batch(cfg, B, L, grid, seed=300 + i) for i in range(4)]
    seeds = []
    for i in range(3):
        tr.step(batches[i], update=(i == 2))
        assert tr.t == (1 if i == 2 else 0)
        seeds.append(tr.engine._seed)
    # every micro-batch of the window draws its own dropout masks (the reference runs a fresh forward per minibatch): the step
    # part of the seeds follows the forward counter, not the update counter
    assert seeds == [0, 1, 2] and tr.micro == 3 and tr.state() == {"t": 1, "micro": 3}
    gs = [oracle_grads(cfg, sd, b)[0] for b in batches[:3]]
    names = sorted(gs[0])
    total = [gs[0][k] + gs[1][k] + gs[2][k] for k in names]
    norm, clipped = O.clip_grad_norm(total, 1.0)
    assert abs(tr.grad_norm() - norm.item()) < 1e-4 * max(1.0, norm.item())
    ref = {}
    for k, g in zip(names, clipped):
        ref[k], _, _ = O.adamw_update(sd[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, 1e-2)
        assert (tr.store.view(k) - ref[k]).abs().max().item() < 2e-5, k
    # the following plain step clears the accumulated gradients first (they are not added to)
    tr.step(batches[3])
    ref_sd = {k: (ref[k].detach() if k in ref else v) for k, v in sd.items()}
    g4 = oracle_grads(cfg, ref_sd, batches[3])[0]
    for k in names:
        assert (tr.store.gview(k) - g4[k]).abs().max().item() < 3e-5, k


def test_hyper_parameters_and_codebook_changes_drop_recorded_plans():
    """launch plans freeze the step scalars and the codebook pointer: assigning lr / weight decay / clip / schedule
    length, or set_centroids, drops them; set_centroids also keeps the store's device buffers (views stay valid)."""
    cfg = XLxmertConfig(**TINY)
    tr, sd = make_step(cfg, 3, 8, 4, lr=1e-2)
    ptr, ptr_c = tr.store.centroids.data_ptr(), tr.store.centroids_c.data_ptr()
    for attr, val in (("lr", 5e-3), ("wd", 0.1), ("clip", 0.5), ("total_steps", 77), ("warmup_steps", 3)):
        tr._plans[("fake",)] = object()
        setattr(tr, attr, getattr(tr, attr))          # same value: plans stay
        assert tr._plans
        setattr(tr, attr, val)
        assert not tr._plans and getattr(tr, attr) == val, attr
    tr._plans[("fake",)] = object()
    new = torch.rand(cfg.num_clusters, cfg.visual_feat_dim)
    tr.set_centroids(new)
    assert not tr._plans
    assert tr.store.centroids.data_ptr() == ptr and tr.store.centroids_c.data_ptr() == ptr_c and torch.equal(tr.store.centroids, new)


def test_segmented_plan_alternates_segments_and_host_operations():
    """_lib.SegmentedPlan (the data-parallel step's launch plan): host operations recorded between C-ABI calls cut the record
    into segments; replay runs segment, host operation, segment ... in order.  Host-only entry points, no GPU."""
    from xlxmert_amd._lib import SegmentedPlan, get_lib
    lib = get_lib()
    log = []
    with lib.record() as calls:
        lib.call("xl_set_deferred_reduce", 1)
        lib.record_host(lambda: log.append("a"))
        lib.call("xl_set_deferred_reduce", 0)
        lib.call("xl_set_step_seed_ptr", None)
        lib.record_host(lambda: log.append("b"))
        lib.record_host(lambda: log.append("c"))
    lib.record_host(lambda: log.append("never"))         # outside a recording: ignored
    plan = lib.make_plan(calls)
    assert isinstance(plan, SegmentedPlan) and plan.n_calls == 3 and plan.n_segments == 2 and plan.n_host_ops == 3
    plan.run()
    plan.run()
    assert log == ["a", "b", "c", "a", "b", "c"]


def _metrics_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = XLxmertConfig(**TINY)
    tr, _ = make_step(cfg, 2, 8, 4)
    got = tr.reduce_metrics({"obj_loss": torch.tensor(1.5 + rank), "n": 10 * (rank + 1), "feat_loss": 0.25})
    torch.save(got, os.path.join(out_dir, f"m{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_metrics_world2_gloo(tmp_path):
    """ref x-lxmert/src/utils.py:11-39 reduce_dict: per-rank epoch metrics are SUMMED onto rank 0 by one reduce of the key-sorted
    vector; the other ranks get None."""
    world, port = 2, _free_port()
    mp.spawn(_metrics_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert torch.load(tmp_path / "m0.pt") == {"feat_loss": 0.5, "n": 30.0, "obj_loss": 4.0}
    assert torch.load(tmp_path / "m1.pt") is None


def edge_batch(cfg, lens, masks, L, V, seed):
    """hand-made batch: text lengths `lens` ([CLS] ... [SEP] then PAD), visual masks `masks` (list of index lists)."""
    g = torch.Generator().manual_seed(seed)
    B = len(lens)
    ids = torch.randint(1, cfg.vocab_size - 2, (B, L), generator=g)
    ids[:, 0] = cfg.vocab_size - 2
    for b, n in enumerate(lens):
        ids[b, n - 1] = cfg.vocab_size - 1
        ids[b, n:] = 0
    cid = torch.randint(0, cfg.num_clusters, (B, V), generator=g)
    vm = torch.zeros(B, V, dtype=torch.bool)
    for b, idx in enumerate(masks):
        vm[b, idx] = True
    lab = cid.clone()
    lab[~vm] = -100
    grid = int(V ** 0.5)
    pos = torch.tensor([[j / grid, i / grid, (j + 1) / grid, (i + 1) / grid] for i in range(grid) for j in range(grid)])
    am = ids > 0
    return {"input_ids": ids, "attention_mask": am, "token_type_ids": torch.zeros_like(ids), "cluster_ids": cid, "vis_mask": vm,
            "obj_labels": lab, "visual_pos": pos[None].expand(B, -1, -1).contiguous()}


EDGE_CASES = {"one_example_two_tokens": ([2], [[5]]),
              "ragged_extremes": ([2, 8, 5, 3], [[], list(range(16)), [15], [0, 3, 7, 8]]),
              "all_full": ([8, 8], [list(range(16)), list(range(16))])}


@pytest.mark.parametrize("case", sorted(EDGE_CASES))
def test_edge_case_batches_match_oracle(case):
    """Corners of the input domain through the engine's host-side logic (packed language rows derived by the engine, masked-row
    head, loss normalisers), every gradient against the oracle: a [CLS][SEP]-only text, an example without any masked visual
    token next to one with all of them masked, no padding at all.  (GPU twin: test_engine_gpu.py, same cases on the HIP kernels.)"""
    cfg = XLxmertConfig(**TINY)
    lens, masks = EDGE_CASES[case]
    L, V = 8, 16
    batch = edge_batch(cfg, lens, masks, L, V, seed=len(lens) * 17 + 1)
    tr, sd = make_step(cfg, len(lens), L, 4, seed=11, lr=0.0)
    for _ in range(2):
        losses = tr.step(batch)
    want, out = oracle_grads(cfg, sd, batch)
    assert abs(losses[0].item() - out["obj_loss"].item()) < 1e-5 * max(1.0, abs(out["obj_loss"].item()))
    assert abs(losses[1].item() - out["feat_loss"].item()) < 1e-5 * max(1.0, abs(out["feat_loss"].item()))
    n = 0
    for k, g in want.items():
        if k == "obj_predict_head.out_cluster.weight":
            continue
        d = (tr.store.gview(k) - g).abs().max().item()
        assert d < 3e-5 * max(1.0, g.abs().max().item()), (case, k, d)
        n += 1
    assert n > 80


def test_new_kept_ranges_drop_recorded_plans_and_partial_master_is_guarded():
    """(i) a launch plan recorded before a gradient range became "kept" may accumulate into it on the strength of the
    optimizer pass clearing the range -- ParamStore.mark_overwritten reports new ranges and the trainer drops its plans; (ii) with the
    sharded exchange's bf16 gather the fp32 master matrices are whole on their owner only: the store refuses to export parameters and
    the engine keeps its compute copy until gather_state()."""
    cfg = XLxmertConfig(**TINY)
    tr, sd = make_step(cfg, 3, 8, 4, lr=1e-2)
    st = tr.store
    assert st.mark_overwritten([(0, 512)]) is True
    assert st.mark_overwritten([(0, 512)]) is False
    tr._plans[("fake",)] = object()
    tr.engine.overwritten = {(0, 512)}
    tr.engine.written_now = {(0, 512)}
    tr.optimizer_step()
    assert tr._plans                                # nothing new: plans stay
    tr._plans[("fake",)] = object()
    tr.engine.overwritten = {(0, 512), (1024, 2048)}
    tr.engine.written_now = {(0, 512), (1024, 2048)}
    tr.optimizer_step()
    assert not tr._plans                            # a range became kept: every recorded plan is dropped
    st.master_partial = True
    with pytest.raises(RuntimeError, match="gather_state"):
        st.named_state()
    before = st.master.clone()
    st.master.add_(1.0)                             # a stale master copy must NOT reach the compute copy
    tr.engine.sync_compute_weights()
    st.master_partial = False
    st.master.copy_(before)
    assert st.named_state()
