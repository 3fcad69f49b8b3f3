"""GPU parity of the whole path: engine on HipOps (C-ABI kernels on a real MI355X) against the committed golden
fixtures (outputs of the reference itself) and against the CPU oracle on seeded inputs."""
import os

import pytest
import torch

import lxmert_oracle as O
from _util import golden_cfg, golden_inputs, load_golden, maxdiff

pytestmark = pytest.mark.gpu

CFG_KEYS = ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size", "max_position_embeddings",
            "type_vocab_size", "l_layers", "x_layers", "r_layers", "visual_feat_dim", "visual_pos_dim", "num_clusters")


def build(g, dtype, need_lang):
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.ops import HipOps
    from xlxmert_amd.params import ParamStore
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in CFG_KEYS})
    sd = O.make_state_dict(oc, int(g["seed"]))
    inp = golden_inputs(g)
    B, L = inp["input_ids"].shape
    V = inp["cluster_ids"].shape[1]
    store = ParamStore(cfg, "cuda", dtype, task="vis_mask" if not need_lang else "all")
    store.load_named(sd)
    eng = Engine(cfg, store, HipOps(dtype), B, L, V, need_lang=need_lang)
    eng.sync_compute_weights()
    dev = {k: v.cuda() for k, v in inp.items()}
    eng.set_inputs(dev["input_ids"], dev["attention_mask"], dev["token_type_ids"], dev["visual_pos"],
                   cluster_ids=dev["cluster_ids"], vis_mask=dev["vis_mask"], obj_labels=dev["obj_labels"])
    return eng, oc, sd, inp


@pytest.mark.parametrize("name", ["tiny_222", "tiny_955"])
def test_forward_fp32_matches_reference_fixture(name):
    g = load_golden(name)
    eng, oc, sd, inp = build(g, torch.float32, True)
    lang, vis, pooled = eng.encoder_forward()
    feat, logits = eng.head_forward()
    losses = eng.losses_forward_backward(want_grad=False)
    torch.cuda.synchronize()
    B, L = inp["input_ids"].shape
    real = inp["attention_mask"].reshape(-1)
    assert maxdiff(lang.cpu().view(B * L, -1)[real], torch.from_numpy(g["lang"]).view(B * L, -1)[real]) < 1e-4
    assert maxdiff(vis.cpu().view(g["vis"].shape), g["vis"]) < 1e-4
    assert maxdiff(pooled.cpu(), g["pooled"]) < 1e-4
    assert maxdiff(feat.cpu().view(g["feat"].shape), g["feat"]) < 1e-4
    assert maxdiff(logits.cpu().view(g["obj"].shape), g["obj"]) < 1e-3          # north-star logit tolerance
    assert abs(losses[0].item() - g["obj_loss"].item()) < 1e-4
    assert abs(losses[1].item() - g["feat_loss"].item()) < 1e-4


def test_step_gradients_fp32_match_reference_fixture():
    g = load_golden("tiny_222")
    eng, oc, sd, inp = build(g, torch.float32, False)
    losses = eng.vis_mask_forward_backward()
    torch.cuda.synchronize()
    assert abs(losses[0].item() - g["obj_loss"].item()) < 1e-4
    for k in [str(n) for n in g["grad_names"]]:
        assert maxdiff(eng.store.gview(k).cpu(), g["grad:" + k]) < 1e-4, k


def test_step_gradients_bf16_close_to_reference_fixture():
    """bf16 operands / fp32 accumulate: stated tolerance = loss within 2e-2, every gradient tensor within 6% of its
    norm (relative L2) of the fp32 reference gradient (3x the measured worst tensor, 2.1 %)."""
    g = load_golden("tiny_222")
    eng, oc, sd, inp = build(g, torch.bfloat16, False)
    losses = eng.vis_mask_forward_backward()
    torch.cuda.synchronize()
    assert abs(losses[0].item() - g["obj_loss"].item()) < 2e-2
    assert abs(losses[1].item() - g["feat_loss"].item()) < 2e-2
    worst = 0.0
    for k in [str(n) for n in g["grad_names"]]:
        ref = torch.from_numpy(g["grad:" + k]).double()
        got = eng.store.gview(k).cpu().double()
        # key biases have an analytically ZERO gradient (softmax shift invariance): the fixture holds fp32 noise
        rel = (got - ref).norm().item() / max(ref.norm().item(), 1e-4)
        worst = max(worst, rel)
        assert rel < 6e-2, (k, rel)
    print("worst relative gradient error (bf16):", worst)


def test_config1_logits_fp32_within_1e3():
    """BASELINE config 1: single forward, 1+1+1 layers, d=768, 8x8 grid, seq=20, 10k codebook, random weights."""
    g = load_golden("config1")
    eng, oc, sd, inp = build(g, torch.float32, True)
    lang, vis, pooled = eng.encoder_forward()
    feat, logits = eng.head_forward()
    losses = eng.losses_forward_backward(want_grad=False)
    maxprob, argmax = eng.predict_codes()
    torch.cuda.synchronize()
    rows = g["obj_rows_idx"]
    lg = logits.cpu()
    err = maxdiff(lg[rows], g["obj_rows"])
    print("config1 fp32 logits max abs err:", err)
    assert err < 1e-3
    assert maxdiff(torch.logsumexp(lg.double(), 1), g["obj_lse"]) < 1e-3
    assert (lg.argmax(1).numpy() == g["obj_argmax"]).all()
    assert (argmax.cpu().numpy() == g["obj_argmax"]).all()
    assert maxdiff(vis.cpu().view(g["vis"].shape), g["vis"]) < 2e-4
    assert maxdiff(pooled.cpu(), g["pooled"]) < 2e-4
    assert maxdiff(feat.cpu()[rows], g["feat_rows"]) < 2e-4
    assert abs(losses[0].item() - g["obj_loss"].item()) < 1e-3
    assert abs(losses[1].item() - g["feat_loss"].item()) < 1e-4


def test_config1_logits_bf16_stated_tolerance():
    """bf16 throughput mode against the same fixture.  Stated tolerance (logit std ~ 16): max abs err < 1.5,
    mean abs err < 0.25, argmax agreement >= 90 % (measured 0.58 / 0.10 / 96.1 %; the reference's own bf16 autocast: 0.76 /
    0.115 / 97.7 %)."""
    g = load_golden("config1")
    eng, oc, sd, inp = build(g, torch.bfloat16, True)
    eng.encoder_forward()
    feat, logits = eng.head_forward()
    torch.cuda.synchronize()
    rows = g["obj_rows_idx"]
    lg = logits.cpu()
    d = (lg[rows].double() - torch.from_numpy(g["obj_rows"]).double()).abs()
    agree = (lg.argmax(1).numpy() == g["obj_argmax"]).mean()
    print(f"config1 bf16 logits: max abs err {d.max():.4f}, mean abs err {d.mean():.4f}, argmax agreement {agree:.3f}")
    assert d.max() < 1.5 and d.mean() < 0.25 and agree >= 0.9


# ---------------------------------------------------------------- the benchmarked architecture against the reference
def _grad_report(eng, g, slice_rel_floor=1e-3):
    """per-tensor (relative norm error, relative L2 error of the stored samples) against full_955.npz / config1.npz."""
    from _util import slice_idx
    worst_n, worst_s = ("", 0.0), ("", 0.0)
    all_s = []
    for k in [str(n) for n in g["grad_names"]]:
        got = eng.store.gview(k).detach().cpu().double().reshape(-1)
        ref_n = g["gnorm:" + k].item()
        en = abs(got.norm().item() - ref_n) / max(ref_n, 1e-6)
        if en > worst_n[1] and ref_n > 1e-5:
            worst_n = (k, en)
        if "gslice:" + k in g:
            ref = torch.from_numpy(g["gslice:" + k]).double()
            smp = got[torch.from_numpy(slice_idx(got.numel()))]
        elif "grad:" + k in g:
            ref = torch.from_numpy(g["grad:" + k]).double().reshape(-1)
            smp = got
        else:
            continue
        # samples of a tensor: error relative to the samples' own norm, floored by the tensor's rms (a handful of
        # near-zero samples must not decide)
        rms = ref_n / max(got.numel(), 1) ** 0.5
        es = (smp - ref).norm().item() / max(ref.norm().item(), slice_rel_floor * rms * ref.numel() ** 0.5, 1e-12)
        if ref_n > 1e-5:            # (key biases: analytically zero gradient, the fixture holds rounding noise)
            all_s.append(es)
            if es > worst_s[1]:
                worst_s = (k, es)
    all_s.sort()
    return worst_n, worst_s, all_s[len(all_s) // 2]


def test_full_width_fp32_matches_reference_fixture():
    """9/5/5 layers, d=768, 12 heads, dff=3072, 10k codebook, B=8 with ragged lengths (tests/golden/full_955.npz, outputs of
    the reference itself): outputs, logits within 1e-3, losses, and the norm + stored samples of all 439 gradients."""
    g = load_golden("full_955")
    eng, oc, sd, inp = build(g, torch.float32, False)
    eng.encoder_forward(want_pooled=False)
    feat, logits = eng.head_forward()
    torch.cuda.synchronize()
    rows = g["obj_rows_idx"]
    lg = logits.cpu()
    assert maxdiff(eng.vis_final.cpu().view(g["vis"].shape), g["vis"]) < 2e-4
    err = maxdiff(lg[rows], g["obj_rows"])
    print("full_955 fp32 logits max abs err:", err)
    assert err < 1e-3
    assert (lg.argmax(1).numpy() == g["obj_argmax"]).all()
    losses = eng.vis_mask_forward_backward()
    torch.cuda.synchronize()
    assert abs(losses[0].item() - g["obj_loss"].item()) < 1e-4 * g["obj_loss"].item()
    assert abs(losses[1].item() - g["feat_loss"].item()) < 1e-4
    wn, ws, med = _grad_report(eng, g)
    print("full_955 fp32 worst gradient-norm error", wn, "worst sample error", ws, "median", med)
    assert wn[1] < 1e-4 and ws[1] < 2e-4, (wn, ws)              # measured 3.5e-6 / 1.1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-2)])
def test_last_visual_ffn_on_masked_rows_only_changes_nothing(dtype, tol):
    """The masked-visual-token step runs the LAST cross layer's visual feed-forward block on the masked rows only (nothing but
    the codebook head reads its output: Engine.encoder_forward ffn_rows) -- against the same engine with every row computed
    (compact_last_ffn = False), full width (9/5/5, d = 768), 256 of the 512 visual rows masked, dropout off: losses and all
    gradients equal to fp32 rounding (bf16: to the rounding of the re-ordered weight-gradient contractions)."""
    g = load_golden("full_955")
    res = {}
    for compact in (False, True):
        eng, oc, sd, inp = build(g, dtype, False)
        eng.compact_last_ffn = compact
        assert 0 < eng.n_mrows < eng.MV
        losses = eng.vis_mask_forward_backward().clone()
        torch.cuda.synchronize()
        assert (eng._ffn_rows_run is not None) == compact and (eng.x_layers[-1]["ffn_v"].rows == eng.n_mrows) == compact
        res[compact] = (losses.cpu(), eng.store.grad[:eng.store.n_used].float().cpu().clone())
    assert maxdiff(res[False][0], res[True][0]) <= tol * max(1.0, res[False][0].abs().max().item())
    gn = res[False][1].norm().item()
    assert (res[False][1] - res[True][1]).norm().item() <= tol * gn, ((res[False][1] - res[True][1]).norm().item(), gn)


@pytest.mark.parametrize("pingpong", [2, 1])
def test_full_width_bf16_benchmark_configuration_close_to_reference(pingpong):
    """The configuration bench.py times -- bf16, 256x256 ping-pong GEMM forced (pingpong=2) or chosen by shape, grouped weight
    gradients, deferred column reductions, two streams, codebook head on the masked rows only -- against the same reference
    fixture.  Yardstick for the tolerance: the reference's own arithmetic under torch bf16 autocast (CPU, same fixture) is off
    by 4.5 % in the worst gradient norm, 15.4 % relative L2 in the worst tensor's samples (r_layers.0 key weight, 19 layers
    from the loss), 5.0 % in the median tensor.  This path keeps fp32 accumulators / LayerNorm / softmax and measures
    2.2 % / 9.2 % / ~2 %: asserted below the yardstick, the median at 3 %."""
    g = load_golden("full_955")
    eng, oc, sd, inp = build(g, torch.bfloat16, False)
    assert eng.compact_head and eng.side is not None
    eng.ops.set_gemm_pingpong(pingpong)
    try:
        eng.set_inputs(*[inp[k].cuda() for k in ("input_ids", "attention_mask", "token_type_ids", "visual_pos")],
                       cluster_ids=inp["cluster_ids"].cuda(), vis_mask=inp["vis_mask"].cuda(), obj_labels=inp["obj_labels"].cuda(),
                       masked_rows=inp["vis_mask"].reshape(-1).nonzero().reshape(-1))
        losses = eng.vis_mask_forward_backward()
        torch.cuda.synchronize()
    finally:
        eng.ops.set_gemm_pingpong(1)
    assert eng._deferred is False and 0 < eng.n_mrows < eng.MV
    rel_loss = abs(losses[0].item() - g["obj_loss"].item()) / g["obj_loss"].item()
    wn, ws, med = _grad_report(eng, g)
    print(f"full_955 bf16 (pingpong={pingpong}): loss rel err {rel_loss:.4f}, worst gradient-norm error {wn}, worst sample error {ws}, "
          f"median sample error {med:.4f}")
    assert rel_loss < 5e-3 and abs(losses[1].item() - g["feat_loss"].item()) < 2e-3
    assert wn[1] < 4.5e-2 and ws[1] < 0.154 and med < 4e-2, (wn, ws, med)      # measured 2.2 % / 9.2 % / 2.9 %


@pytest.mark.parametrize("dtype,tn,ts", [(torch.float32, 1e-4, 1e-4), (torch.bfloat16, 1e-2, 3e-2)])     # measured: 1e-6 / 3e-6; 0.3 % / 1.0 %
def test_config1_gradients_match_reference_fixture(dtype, tn, ts):
    """BASELINE config 1 (1+1+1 layers at d=768): the norm of all 79 reference gradients (`gnorm:*`) and the stored ones."""
    g = load_golden("config1")
    eng, oc, sd, inp = build(g, dtype, False)
    losses = eng.vis_mask_forward_backward()
    torch.cuda.synchronize()
    assert abs(losses[0].item() - g["obj_loss"].item()) < (1e-3 if dtype == torch.float32 else 1e-2 * g["obj_loss"].item())
    wn, ws, med = _grad_report(eng, g)
    print(f"config1 {dtype}: worst gradient-norm error {wn}, worst sample error {ws}, median {med}")
    assert wn[1] < tn and ws[1] < ts, (wn, ws)


def test_trainer_two_steps_fp32_vs_oracle():
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import PretrainStep, linear_schedule, synthetic_batch
    cfg = XLxmertConfig(vocab_size=100, hidden_size=64, num_attention_heads=4, intermediate_size=128,
                        max_position_embeddings=32, visual_feat_dim=32, num_clusters=56, l_layers=2, x_layers=2, r_layers=1)
    oc = O.OracleConfig(**{k: getattr(cfg, k) for k in CFG_KEYS})
    sd = O.make_state_dict(oc, 3)
    store = ParamStore(cfg, "cuda", torch.float32)
    store.load_named(sd)
    tr = PretrainStep(cfg, 3, 8, 16, dtype=torch.float32, device="cuda", store=store, lr=1e-2, weight_decay=0.01,
                      warmup_ratio=0.2, total_steps=10, visual_losses="obj,feat")
    ref = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in ref.items()}
    v2 = {k: torch.zeros_like(v) for k, v in ref.items()}
    for t in (1, 2):
        batch = synthetic_batch(cfg, 3, 8, 4, seed=100 + t)
        tr.step({k: v.cuda() for k, v in batch.items()})
        leaf = {k: v.clone().requires_grad_(v.is_floating_point() and k != "vis_emb.weight") for k, v in ref.items()}
        leaf["obj_predict_head.out_cluster.weight"] = leaf["vis_emb.weight"]
        out = O.xlxmert_vis_mask_forward(leaf, oc, batch["input_ids"], batch["visual_pos"], batch["attention_mask"],
                                         batch["cluster_ids"], batch["vis_mask"], batch["obj_labels"])
        out["total_loss"].backward()
        names = sorted(k for k, v in leaf.items() if v.grad is not None)
        norm, clipped = O.clip_grad_norm([leaf[k].grad for k in names], 1.0)
        assert abs(tr.grad_norm() - norm.item()) < 1e-3 * max(1.0, norm.item())
        lr = 1e-2 * linear_schedule(t - 1, 2, 10)
        for k, gk in zip(names, clipped):
            wd = 0.0 if ("bias" in k or "LayerNorm.weight" in k) else 0.01
            ref[k], m[k], v2[k] = O.adamw_update(ref[k], gk, m[k], v2[k], t, lr, weight_decay=wd)
        for k in names:
            d = (tr.store.view(k).cpu() - ref[k]).abs().max().item()
            assert d < 1e-4, (t, k, d)


def test_gradient_ranges_are_final_when_reported():
    """The data-parallel exchange queues an in-place all-reduce the moment the engine reports a range (trainer.py
    _on_grad_ready), on the reporting stream.  Check the report points on the real streams: a copy of every reported
    range, taken on the stream the callback runs on, equals the gradient at the end of the step bit for bit, the
    ranges tile the used buffer, and the language stream's block is reported before the visual stack is done."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig(l_layers=3, x_layers=2, r_layers=2)
    tr = PretrainStep(cfg, 64, 20, 64, dtype=torch.bfloat16, device="cuda", seed=1)
    g = torch.Generator().manual_seed(0)
    tr.store.view("mask_feat").copy_(torch.randn(cfg.visual_feat_dim, generator=g).relu())
    tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
    batch = {k: v.cuda() for k, v in synthetic_batch(cfg, 64, 20, 8, seed=7).items()}
    eng, st = tr.engine, tr.store
    snap = torch.full_like(st.grad, float("nan"))
    seen = []

    def on_ready(lo, hi, flush, lane):
        snap[lo:hi].copy_(st.grad[lo:hi])                      # current stream = the one a collective would follow
        seen.append((lo, hi, lane, torch.cuda.current_stream()))

    for _ in range(2):
        seen.clear()
        snap.fill_(float("nan"))
        eng.grad_ready = on_ready
        eng.set_step_seed(5)
        eng.set_inputs(batch["input_ids"], batch["attention_mask"], None, batch["visual_pos"],
                       cluster_ids=batch["cluster_ids"], vis_mask=batch["vis_mask"], obj_labels=batch["obj_labels"],
                       masked_rows=batch.get("masked_rows"))
        eng.vis_mask_forward_backward(True)
        main = torch.cuda.current_stream()
        for _, _, _, s in seen:
            main.wait_stream(s)
        torch.cuda.synchronize()
        eng.grad_ready = None
        pos = 0
        for lo, hi, _, _ in sorted((r for r in seen if r[1] > r[0]), key=lambda r: r[0]):
            assert lo == pos, (lo, pos)
            pos = hi
        assert pos == st.n_used
        assert torch.equal(snap[:st.n_used], st.grad[:st.n_used])
        lanes = [r[2] for r in seen if r[1] > r[0]]
        assert "l" in lanes and lanes.index("l") < len(lanes) - 1 - lanes[::-1].index("v"), lanes
        lo, hi = st.language_range()
        assert all((lane == "l") == (lo <= a < hi) for a, b, lane, _ in seen if b > a)
        if eng.side is not None:
            assert any(lane == "l" and s != main for _, _, lane, s in seen)


@pytest.mark.parametrize("via,plan,comm_dtype,collective", [
    ("torch", False, None, "allreduce"), ("torch", True, None, "allreduce"), ("rccl", False, None, "allreduce"),
    ("rccl", True, None, "allreduce"), ("rccl", True, torch.bfloat16, "allreduce"),
    ("rccl", False, None, "rs+ag"), ("rccl", True, None, "rs+ag"), ("rccl", True, torch.bfloat16, "rs+ag"), ("torch", True, None, "rs+ag")])
def test_exchange_on_one_rank_rccl_group_is_identity(via, plan, comm_dtype, collective):
    """The data-parallel exchange through a real RCCL process group (one rank: every all-reduce is the identity): the
    collectives are queued from three streams (main, language-range reports, end of step) exactly as on N GPUs, and the
    training steps must give the same losses, gradient norm and parameters as the same steps without the exchange.
    via = "torch": torch.distributed issues the collectives (in a launch plan: host operations between plan segments);
    via = "rccl": the library's own RCCL binding (xl_comm_*, XL_COMM=rccl) -- the collectives are entries of ONE plan.
    plan: the third step is a plan replay.
    collective = "rs+ag": reduce-scatter -> shard-local norm (+ scalar all-reduce) / AdamW -> all-gather of the master slices + cast
    on the collectives' stream, the next forward waiting slice by slice (xl_comm_reduce_scatter / _allgather, events)."""
    import os
    import socket
    import torch.distributed as dist
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig(l_layers=3, x_layers=2, r_layers=2)
    batch = {k: v.cuda() for k, v in synthetic_batch(cfg, 64, 20, 8, seed=7).items()}
    g = torch.Generator().manual_seed(0)
    mask_feat = torch.randn(cfg.visual_feat_dim, generator=g).relu()
    cent = torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu()

    def run(tr):
        tr.store.view("mask_feat").copy_(mask_feat)
        tr.set_centroids(cent)
        out = [tr.step(batch).clone() for _ in range(3)]
        torch.cuda.synchronize()
        return out, tr.store.master.clone(), tr.grad_norm()

    ref = run(PretrainStep(cfg, 64, 20, 64, dtype=torch.bfloat16, device="cuda", seed=1, bucket_mb=8, plan=plan, drop_grads=False))
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    os.environ["XL_FORCE_EXCHANGE"] = "1"
    os.environ["XL_COMM"] = via
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        tr = PretrainStep(cfg, 64, 20, 64, dtype=torch.bfloat16, device="cuda", seed=1, bucket_mb=8, plan=plan, drop_grads=False,
                          grad_comm_dtype=comm_dtype, collective=collective)
        assert tr.exchange and tr.world == 1 and (tr.xl_comm is not None) == (via == "rccl") and tr.collective == collective
        got = run(tr)
        assert len(tr._slices) > 4, len(tr._slices)
        if collective == "rs+ag":           # one rank: every slice is one whole shard, no replicated tail; the forward is hooked
            assert tr.sharded and [k for k, _, _ in tr._segments] == ["rs"] * len(tr._slices)
            assert tr.owned_ranges() == [(a, b) for _, a, b in tr._segments]
            assert (tr.engine.params_ready is not None) == (via == "rccl") and (via != "rccl" or len(tr._group_seg) > 5)
        lo, hi = tr.store.language_range()
        assert any(lo <= a < hi for a, _ in tr._slices)
        if plan:
            (p,) = tr._plans.values()
            if via == "rccl":               # one plan, the collectives inside it
                assert p.n_segments == 1 and p.n_host_ops == 0
            elif collective == "allreduce":  # segments around torch.distributed's collectives
                assert p.n_host_ops == len(tr._slices) + 1 and p.n_segments >= len(tr._slices)
            else:                           # ... + the norm's scalar all-reduce and one all-gather per slice
                assert p.n_host_ops == 2 * len(tr._slices) + 2
    finally:
        dist.destroy_process_group()
        os.environ.pop("XL_FORCE_EXCHANGE", None)
        os.environ.pop("XL_COMM", None)
    # (not bit-identical: split-K and loss sums use float atomics, whose order differs from run to run; bf16 buckets round
    #  the gradients once more)
    tol = 1e-5 if comm_dtype is None else 2e-3
    for a, b in zip(ref[0], got[0]):
        assert torch.allclose(a, b, rtol=tol, atol=1e-6), (a, b)
    assert (ref[1] - got[1]).abs().max().item() < (2e-5 if comm_dtype is None else 2e-3) and abs(ref[2] - got[2]) < (1e-4 if comm_dtype is None else 5e-3) * ref[2]


def test_full_size_step_properties_bf16():
    """BASELINE sizes (9/5/5, d=768, bs=256, 20x64 tokens, 10k codebook): size-independent properties.
    (1) pad isolation: changing token ids at padded positions changes no visual output bit;
    (2) masked-row exactness: rows with label -100 get exactly zero d(logits);
    (3) loss and every gradient are finite, loss ~ log-scale of a random 10k-way classifier."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig()
    tr = PretrainStep(cfg, 256, 20, 64, dtype=torch.bfloat16, device="cuda", seed=1)
    g = torch.Generator().manual_seed(0)
    # reference init has mask_feat = 0 and visn_fc.bias = 0: masked rows enter visn_layer_norm as the exact zero vector
    # (rstd = 1/sqrt(1e-12) = 1e6) and the first-step gradient norm is ~1e7 in the reference too.  Move off that point.
    tr.store.view("mask_feat").copy_(torch.randn(cfg.visual_feat_dim, generator=g).relu())
    tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
    batch = {k: v.cuda() for k, v in synthetic_batch(cfg, 256, 20, 8, seed=7).items()}
    eng = tr.engine
    eng.set_inputs(batch["input_ids"], batch["attention_mask"], None, batch["visual_pos"], cluster_ids=batch["cluster_ids"],
                   vis_mask=batch["vis_mask"], obj_labels=batch["obj_labels"])
    eng.encoder_forward(want_pooled=False)
    vis1 = eng.vis_final.clone()
    ids2 = batch["input_ids"].clone()
    pad = ~batch["attention_mask"]
    ids2[pad] = torch.randint(1000, 30000, (int(pad.sum().item()),), device="cuda")
    eng.set_inputs(ids2, batch["attention_mask"], None, batch["visual_pos"], cluster_ids=batch["cluster_ids"],
                   vis_mask=batch["vis_mask"], obj_labels=batch["obj_labels"])
    eng.encoder_forward(want_pooled=False)
    assert torch.equal(vis1, eng.vis_final), "padded language tokens leaked into the visual stream"
    losses = tr.step(batch)
    torch.cuda.synchronize()
    unmasked = (batch["obj_labels"].reshape(-1) == -100)
    if eng.compact_head:            # the training step runs the head on the masked rows only: unmasked rows never exist
        n = int((~unmasked).sum().item())
        assert eng.n_mrows == (n + 255) // 256 * 256            # row list padded to the GEMM row tile with -1 entries
        assert torch.equal(eng.mrows[:n].long().cpu(), (~unmasked).nonzero().reshape(-1).cpu())
        assert (eng.mrows[n:eng.n_mrows] == -1).all()
    else:
        assert eng.dlogits[unmasked].abs().max().item() == 0.0
    assert torch.isfinite(losses).all() and 5.0 < losses[0].item() < 200.0
    assert torch.isfinite(tr.store.grad[:tr.store.n_used]).all()
    assert torch.isfinite(tr.store.master).all()
    assert 0.0 < tr.grad_norm() < 1e5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 6e-2)])
def test_training_mode_dropout_step_matches_host_restatement(dtype, tol):
    """Dropout on (p=0.1 hidden and attention): the counter-based masks are reproducible on the host, so the whole
    training-mode step (loss + every gradient) is compared with the same engine driven by tests/fake_ops.FakeOps."""
    from fake_ops import FakeOps
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.ops import HipOps
    from xlxmert_amd.params import ParamStore
    g = load_golden("tiny_222")
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in CFG_KEYS})
    sd = O.make_state_dict(oc, int(g["seed"]))
    inp = golden_inputs(g)
    B, L = inp["input_ids"].shape
    V = inp["cluster_ids"].shape[1]
    res = {}
    for dev, ops in (("cpu", FakeOps(torch.float32)), ("cuda", HipOps(dtype))):
        store = ParamStore(cfg, dev, torch.float32 if dev == "cpu" else dtype)
        store.load_named(sd)
        eng = Engine(cfg, store, ops, B, L, V, need_lang=False, train_dropout=True)
        eng.sync_compute_weights()
        eng.set_step_seed(11)
        x = {k: v.to(dev) for k, v in inp.items()}
        eng.set_inputs(x["input_ids"], x["attention_mask"], x["token_type_ids"], x["visual_pos"],
                       cluster_ids=x["cluster_ids"], vis_mask=x["vis_mask"], obj_labels=x["obj_labels"])
        losses = eng.vis_mask_forward_backward()
        res[dev] = (losses.cpu().clone(), store.grad[:store.n_used].cpu().clone(), store)
    assert abs(res["cpu"][0][0] - res["cuda"][0][0]).item() < (1e-4 if dtype == torch.float32 else 3e-2)
    st = res["cpu"][2]
    for name in st.names():
        m = st.index[name]
        if m.offset >= st.n_used:
            continue
        a = res["cpu"][1][m.offset:m.offset + st.view(name).numel()].double()
        b = res["cuda"][1][m.offset:m.offset + st.view(name).numel()].double()
        rel = (a - b).norm().item() / max(a.norm().item(), 1e-4)
        assert rel < tol, (name, rel)


# ---------------------------------------------------------------- VQA / GQA fine-tune step
def test_nlvr2_step_fp32_matches_reference_fixture():
    """NLVR2 fine-tune step on the GPU (pair head over pooled_output viewed [P, 2d], CE) == the reference's NLVR2Model."""
    from test_engine_cpu import make_nlvr2_engine, check_nlvr2_grads
    from xlxmert_amd.ops import HipOps
    g = load_golden("nlvr2_tiny")
    eng, inp = make_nlvr2_engine(g, HipOps(torch.float32), device="cuda", dtype=torch.float32)
    loss = eng.nlvr2_forward_backward(inp["labels"].cuda())
    torch.cuda.synchronize()
    assert maxdiff(eng.answer.logit.cpu(), g["logit"]) < 1e-4
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    check_nlvr2_grads(eng, g, 1e-4)


def test_nlvr2_step_bf16_close_to_reference_fixture():
    from test_engine_cpu import make_nlvr2_engine
    from xlxmert_amd.ops import HipOps
    g = load_golden("nlvr2_tiny")
    eng, inp = make_nlvr2_engine(g, HipOps(torch.bfloat16), device="cuda", dtype=torch.bfloat16)
    loss = eng.nlvr2_forward_backward(inp["labels"].cuda())
    torch.cuda.synchronize()
    assert maxdiff(eng.answer.logit.cpu(), g["logit"]) < 5e-2
    assert abs(loss.item() - float(g["loss"])) < 2e-2
    for k in (str(n) for n in g["grad_names"]):
        ref = torch.from_numpy(g["grad:" + k]).double()
        got = eng.store.gview(k).cpu().double()
        if k.endswith("embeddings.weight"):
            ref = ref.clone(); ref[0] = got[0]
        assert (got - ref).norm().item() <= 0.05 * max(ref.norm().item(), 1e-3), k


def _vqa_engine(g, dtype):
    from test_engine_cpu import make_vqa_engine
    from xlxmert_amd.ops import HipOps
    return make_vqa_engine(g, HipOps(dtype), device="cuda", dtype=dtype)


def test_vqa_step_fp32_matches_reference_fixture():
    g = load_golden("vqa_tiny")
    eng, inp = _vqa_engine(g, torch.float32)
    loss = eng.vqa_forward_backward(inp["targets"].cuda())
    torch.cuda.synchronize()
    assert maxdiff(eng.answer.logit.cpu(), g["logit"]) < 1e-4
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    for k in [str(n) for n in g["grad_names"]]:
        ref = torch.from_numpy(g["grad:" + k])
        assert maxdiff(eng.store.gview(k).cpu(), ref) <= 1e-4 * max(1.0, ref.abs().max().item()), k


def test_vqa_step_bf16_close_to_reference_fixture():
    """bf16 operands / fp32 accumulate, same stated tolerance as the pretraining step: loss within 2e-2, every gradient
    tensor within 6 % relative L2 of the fp32 reference gradient."""
    g = load_golden("vqa_tiny")
    eng, inp = _vqa_engine(g, torch.bfloat16)
    loss = eng.vqa_forward_backward(inp["targets"].cuda())
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 2e-2
    logit = eng.answer.logit.cpu()
    assert maxdiff(logit, g["logit"]) < 5e-2
    for k in [str(n) for n in g["grad_names"]]:
        ref = torch.from_numpy(g["grad:" + k]).double()
        got = eng.store.gview(k).cpu().double()
        rel = (got - ref).norm().item() / max(ref.norm().item(), 1e-4)
        assert rel < 6e-2, (k, rel)


@pytest.mark.parametrize("B", [64, 512])
def test_vqa_full_size_step_properties_bf16(B):
    """BASELINE config 3 geometry (full encoder, 3129 answers, real 2048-d features) at bs 64 and at the configuration's own
    bs 512: finite loss near ln(2)-level for near-zero logits, finite gradients, d(visn_fc.weight) non-zero (real-feature
    input path), and batch consistency: the first 64 examples' logits do not depend on what else is in the batch."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.ops import HipOps
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import init_reference_weights
    cfg = XLxmertConfig()
    L, V, A = 20, 64, 3129
    store = ParamStore(cfg, "cuda", torch.bfloat16, task="vqa", num_answers=A)
    init_reference_weights(store, 7)
    eng = Engine(cfg, store, HipOps(torch.bfloat16), B, L, V, need_lang=True)
    eng.sync_compute_weights()
    oc = O.OracleConfig()
    inp = O.make_vqa_inputs(oc, A, 3, B, L, 8)
    eng.set_inputs(inp["input_ids"].cuda(), inp["attention_mask"].cuda(), None, inp["visual_pos"].cuda(),
                   visual_feats=inp["visual_feats"].cuda())
    loss = eng.vqa_forward_backward(inp["targets"].cuda())
    torch.cuda.synchronize()
    assert 0.5 < loss.item() < 0.9, loss.item()                 # logits ~ 0 at init -> BCE ~ ln 2
    gr = store.grad[:store.n_used]
    assert torch.isfinite(gr).all()
    assert store.gview("bert.encoder.visn_fc.visn_fc.weight").abs().max().item() > 0
    assert store.gview("answer_head.logit_fc.3.weight").abs().max().item() > 0
    if B > 64:
        # examples are independent (no cross-example op on the path): the first 64 examples alone give the same logits
        # (different tile shapes / kernel choices at M = 64*20 vs 512*20 rows: agreement to bf16 rounding, not bit-wise)
        logit_big = eng.answer.logit[:64].clone()
        small = Engine(cfg, store, HipOps(torch.bfloat16), 64, L, V, need_lang=True)
        small.set_inputs(inp["input_ids"][:64].cuda(), inp["attention_mask"][:64].cuda(), None, inp["visual_pos"][:64].cuda(),
                         visual_feats=inp["visual_feats"][:64].cuda())
        logit_small = small.vqa_forward().clone()
        torch.cuda.synchronize()
        d = (logit_big - logit_small).abs().max().item()
        print("vqa bs512 vs bs64 logits max abs diff:", d, "logit scale", logit_small.abs().max().item())
        # every default kernel sums a contraction in the same K order whatever the row count, so the two engines agree to the last
        # bits (measured 2.4e-7).  XL_GEMM_SPLIT_EPI=1 (opt-in) runs the small engine's deep contractions as K slices: another
        # bf16 re-association of the same sums -- both are 0.051 from the fp32 path, 0.043 from each other (tools/split_epi_engine.py)
        bound = 5e-2 if os.environ.get("XL_GEMM_SPLIT_EPI", "0") != "0" else 1e-3
        assert d < bound * max(1.0, logit_small.abs().max().item())


# ---------------------------------------------------------------- on-device iterative sampler
def test_sampler_loop_fp32_matches_reference_fixture():
    from test_engine_cpu import make_sampler_engine
    from xlxmert_amd.ops import HipOps
    g = load_golden("sampler_tiny")
    eng, sd = make_sampler_engine(g, HipOps(torch.float32), device="cuda")
    cid, code, prob = eng.sample_codes_nar(int(g["n_steps"]))
    torch.cuda.synchronize()
    assert maxdiff(code.cpu().view(g["code"].shape), g["code"]) == 0.0            # same codes chosen at every position
    assert maxdiff(prob.cpu().view(g["step_pred_prob"][-1].shape), g["step_pred_prob"][-1]) < 1e-4
    assert torch.equal(eng.vmask.long().cpu(), torch.from_numpy(g["step_masks"][-1]))


def test_sampler_loop_bf16_agrees_where_the_reference_is_decisive():
    """bf16 may flip an argmax between near-tied codes (and a flipped code changes later steps); stated tolerance:
    >= 85 % of the positions carry the reference's code after 4 steps."""
    from test_engine_cpu import make_sampler_engine
    from xlxmert_amd.ops import HipOps
    g = load_golden("sampler_tiny")
    eng, sd = make_sampler_engine(g, HipOps(torch.bfloat16), device="cuda", dtype=torch.bfloat16)
    cid, code, prob = eng.sample_codes_nar(int(g["n_steps"]))
    torch.cuda.synchronize()
    ref_ids = torch.from_numpy(g["step_pred_ids"][-1])
    # the reference's final ids of positions last predicted at earlier steps are not in the fixture's last row: compare codes
    same = (code.float().cpu().view(g["code"].shape) - torch.from_numpy(g["code"])).abs().amax(-1) < 2e-2
    assert same.float().mean().item() >= 0.85, same.float().mean().item()


def test_sampler_full_size_properties_bf16():
    """BASELINE config 5 geometry (bs 64 here, 4 steps, 8x8 grid, 10k codebook): ids in range, every position predicted,
    deterministic (two runs agree), no host synchronisation needed between steps."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.ops import HipOps
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import init_reference_weights
    cfg = XLxmertConfig()
    B, L, V = 64, 20, 64
    store = ParamStore(cfg, "cuda", torch.bfloat16, task="vis_mask")
    init_reference_weights(store, 11)
    gen = torch.Generator().manual_seed(5)
    store.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=gen).relu())
    store.view("mask_feat").copy_(torch.randn(cfg.visual_feat_dim, generator=gen).cuda() * 0.1)
    eng = Engine(cfg, store, HipOps(torch.bfloat16), B, L, V, need_lang=False)
    eng.sync_compute_weights()
    inp = O.make_inputs(O.OracleConfig(), 21, B, L, 8)
    eng.set_inputs(inp["input_ids"].cuda(), inp["attention_mask"].cuda(), None, inp["visual_pos"].cuda(),
                   cluster_ids=torch.zeros(B, V, dtype=torch.long, device="cuda"), vis_mask=torch.ones(B, V, dtype=torch.bool, device="cuda"))
    cid1 = eng.sample_codes_nar(4)[0].clone()
    code1 = eng.feats.clone()
    cid2 = eng.sample_codes_nar(4)[0].clone()
    torch.cuda.synchronize()
    assert torch.equal(cid1, cid2) and cid1.min().item() >= 0 and cid1.max().item() < cfg.num_clusters
    assert torch.equal(code1.view(B * V, -1), store.centroids_c[cid1.view(-1)])
    assert eng.vmask.sum(1).eq(16).all()            # last step re-masks int(1/4 * 64) positions per image
    # the runs above ended the codebook contraction in the row-max epilogue (no logits in memory); the path that writes the
    # fp32 logits and reduces them with xl_ce_fwd_bwd must give the same codes
    assert eng.fused_predict_available() and eng.ops.block is not None
    prob_fused = eng.row_maxprob.clone()
    os.environ["XL_FUSED_PREDICT"] = "0"
    try:
        assert not eng.fused_predict_available()
        cid3 = eng.sample_codes_nar(4)[0].clone()
    finally:
        del os.environ["XL_FUSED_PREDICT"]
    torch.cuda.synchronize()
    assert torch.equal(cid1, cid3)
    assert (prob_fused - eng.row_maxprob).abs().max().item() < 1e-5


# ---------------------------------------------------------------- language pretraining branches
@pytest.mark.parametrize("task", ["word_mask", "matched"])
def test_language_pretraining_steps_fp32_match_reference_fixture(task):
    from test_engine_cpu import check_lang_task, make_lang_task_engine
    from xlxmert_amd.ops import HipOps
    g = load_golden("lang_tasks_tiny")
    eng, inp = make_lang_task_engine(g, task, HipOps(torch.float32), device="cuda")
    check_lang_task(g, task, eng, inp, 2e-5, 1e-4, dev="cuda")


@pytest.mark.parametrize("task", ["word_mask", "matched"])
def test_language_pretraining_steps_bf16_stated_tolerance(task):
    """bf16: loss within 2e-2, every gradient tensor within 6 % relative L2 of the fp32 reference gradient."""
    from test_engine_cpu import make_lang_task_engine
    from xlxmert_amd.ops import HipOps
    g = load_golden("lang_tasks_tiny")
    eng, inp = make_lang_task_engine(g, task, HipOps(torch.bfloat16), device="cuda", dtype=torch.bfloat16)
    from test_engine_cpu import labelled_rows
    labels = inp["word_labels" if task == "word_mask" else "matched_labels"].cuda()
    for rows in ((False, True) if task == "word_mask" else (False,)):      # True: decoder + loss on the labelled rows only
        if task == "word_mask":
            loss = eng.word_mask_forward_backward(labels, labelled_rows(labels) if rows else None)
        else:
            loss = eng.matched_forward_backward(labels)
        torch.cuda.synchronize()
        assert abs(loss.item() - float(g[task + ":loss"])) < 2e-2
        for k in [str(n) for n in g[task + ":grad_names"]]:
            ref = torch.from_numpy(g[task + ":grad:" + k]).double()
            got = eng.store.gview(k).cpu().double()
            rel = (got - ref).norm().item() / max(ref.norm().item(), 1e-4)
            assert rel < 6e-2, (k, rel, rows)


def test_batch_uploader_slot_is_released_after_the_labels_are_copied():
    """the label tensors of the language / VQA branches are copied into the engine's buffers INSIDE the
    *_forward_backward calls, so the staging slot of a BatchUploader may only be handed back after them.  Four word_mask batches
    with distinct labels through a two-slot uploader, the main stream held busy in front of every label copy (so that a slot
    released early IS refilled by the upload of the batch two steps ahead before its labels are read): every step's loss must
    be the loss of ITS batch (learning rate 0: the reference value is the same batch stepped directly)."""
    from test_trainer_cpu import TINY, oracle_cfg
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import BatchUploader, PretrainStep, synthetic_batch, word_rows_of
    cfg = XLxmertConfig(**TINY)
    oc = oracle_cfg(cfg)
    B, L, grid = 4, 8, 4
    tr = PretrainStep(cfg, B, L, grid * grid, dtype=torch.float32, device="cuda", task="word_mask", seed=5, lr=0.0, total_steps=100)
    tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=torch.Generator().manual_seed(2)).relu())
    host = []
    for i in range(4):
        b = synthetic_batch(cfg, B, L, grid, seed=40 + i)
        wl, _ = O.make_lang_task_labels(oc, b["input_ids"], 60 + i)
        host.append({"input_ids": b["input_ids"], "visual_pos": b["visual_pos"], "cluster_ids": b["cluster_ids"], "word_labels": wl,
                     "word_rows": word_rows_of(wl), "lang_rows": b["lang_rows"], "lang_off": b["lang_off"]})
    ref = []
    for b in host:
        ref.append(tr.step({k: v.cuda() for k, v in b.items()}).clone())
    torch.cuda.synchronize()
    assert len({round(r.item(), 5) for r in ref}) == 4          # the four batches really have four different losses
    orig = tr.engine.word_mask_forward_backward

    def held(*a, **kw):
        torch.cuda._sleep(150_000_000)                          # ~70 ms of main-stream time in front of the label copies
        return orig(*a, **kw)
    tr.engine.word_mask_forward_backward = held
    up = BatchUploader("cuda")
    packed = [BatchUploader.pin(b) for b in host]
    nxt, got = up.upload(packed[0]), []
    for i in range(4):
        cur, nxt = nxt, up.upload(packed[(i + 1) % 4])
        got.append(tr.step(cur).clone())
    torch.cuda.synchronize()
    for i in range(4):
        assert abs(got[i].item() - ref[i].item()) < 1e-5, (i, got[i].item(), [r.item() for r in ref])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 6e-2)])
def test_text_longer_than_64_tokens_step_matches_oracle(dtype, tol):
    """--max_text_length above 64 (ref param.py:140): one step at L = 72 (language self-attention 72 x 72, cross-attention 72 x 16 and
    16 x 72 on the long-sequence attention kernels, packed rows) against the oracle: loss and every gradient (fp32 1e-4; bf16: relative
    L2 per tensor)."""
    from test_trainer_cpu import TINY, oracle_cfg, oracle_grads
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig(**dict(TINY, max_position_embeddings=128))
    B, L, grid = 2, 72, 4
    sd = O.make_state_dict(oracle_cfg(cfg), 3)
    store = ParamStore(cfg, "cuda", dtype, task="vis_mask")
    store.load_named(sd)
    tr = PretrainStep(cfg, B, L, grid * grid, dtype=dtype, device="cuda", store=store, total_steps=10, lr=1e-2, visual_losses="obj,feat")
    for seed, ragged in ((94, False), (95, True)):
        batch = synthetic_batch(cfg, B, L, grid, seed=seed, ragged=ragged)
        losses = tr.engine  # noqa: F841
        tr.engine.set_inputs(*(batch[k].cuda() for k in ("input_ids", "attention_mask", "token_type_ids", "visual_pos")),
                             cluster_ids=batch["cluster_ids"].cuda(), vis_mask=batch["vis_mask"].cuda(), obj_labels=batch["obj_labels"].cuda())
        out = tr.engine.vis_mask_forward_backward(True)
        torch.cuda.synchronize()
        grads, ref = oracle_grads(cfg, sd, batch)
        assert abs(out[0].item() - ref["obj_loss"].item()) < (3e-5 if dtype == torch.float32 else 3e-2)
        for k, g in grads.items():
            if k == "obj_predict_head.out_cluster.weight":
                continue
            got = store.gview(k).cpu().double()
            if dtype == torch.float32:
                assert (got - g.double()).abs().max().item() < tol * max(1.0, g.abs().max().item()), k
            else:
                assert (got - g.double()).norm().item() <= tol * max(g.double().norm().item(), 1e-3), k


def test_word_mask_full_size_step_properties_bf16():
    """full encoder, 30522-way tied decoder at bs 64: loss ~ ln(30522) at init, finite gradients, the word-embedding
    gradient has a non-zero row 0 (decoder side; the embedding scatter skips padding_idx 0)."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.ops import HipOps
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import init_reference_weights
    cfg = XLxmertConfig()
    B, L, V = 64, 20, 64
    store = ParamStore(cfg, "cuda", torch.bfloat16, task="word_mask")
    init_reference_weights(store, 3)
    gen = torch.Generator().manual_seed(5)
    store.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=gen).relu())
    eng = Engine(cfg, store, HipOps(torch.bfloat16), B, L, V, need_lang=True)
    eng.sync_compute_weights()
    oc = O.OracleConfig()
    inp = O.make_inputs(oc, 31, B, L, 8)
    wl, _ = O.make_lang_task_labels(oc, inp["input_ids"], 32)
    eng.set_inputs(inp["input_ids"].cuda(), inp["attention_mask"].cuda(), None, inp["visual_pos"].cuda(),
                   cluster_ids=inp["cluster_ids"].cuda())
    loss = eng.word_mask_forward_backward(wl.cuda())
    torch.cuda.synchronize()
    assert 9.5 < loss.item() < 11.5, loss.item()                # ln(30522) = 10.33
    assert torch.isfinite(store.grad[:store.n_used]).all()
    assert store.gview("bert.embeddings.word_embeddings.weight")[0].abs().max().item() > 0


@pytest.mark.parametrize("overlap", [False, True])
def test_task_round_robin_full_size_bf16(overlap):
    """one multi-task parameter set, full size: vis_mask / word_mask / matched steps in turn; tensors outside a step's branch
    are bit-identical after it, everything stays finite.  overlap: the optimizer pass queued behind the step, group by group
    (per-chunk skip flags and update counts sliced per group)."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch, word_rows_of
    cfg = XLxmertConfig()
    B = 64
    tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda", seed=2, task="all", train_dropout=True, warmup_ratio=0.0,
                      total_steps=100, overlap_optimizer=overlap)
    assert (tr.opt_stream is not None) == overlap
    g = torch.Generator().manual_seed(0)
    tr.store.view("mask_feat").copy_(torch.randn(cfg.visual_feat_dim, generator=g).relu())
    tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
    oc = O.OracleConfig()
    for t, task in enumerate(["vis_mask", "word_mask", "matched"]):
        batch = synthetic_batch(cfg, B, 20, 8, seed=40 + t)
        wl, ml = O.make_lang_task_labels(oc, batch["input_ids"], 50 + t)
        batch["word_labels"], batch["matched_labels"] = wl, ml
        dev_batch = {k: v.cuda() for k, v in batch.items()}
        dev_batch["word_rows"] = word_rows_of(wl)                  # host-side row list: masked-row decoder of the word_mask step
        before = {k: tr.store.view(k).clone() for k in ("obj_predict_head.linear_feat.weight", "cls.seq_relationship.weight",
                                                        "cls.predictions.transform.dense.weight", "bert.encoder.layer.0.output.dense.weight")}
        loss = tr.step(dev_batch, task=task)
        torch.cuda.synchronize()
        assert torch.isfinite(loss).all() and torch.isfinite(tr.store.master).all()
        if task == "word_mask":             # labelled rows, the list padded to the GEMM row tile with -1 entries
            n_lab = int((wl >= 0).sum().item())
            assert tr.engine.lang_heads.n_rows == min(tr.engine.ML, (n_lab + 255) // 256 * 256)
            assert (tr.engine.lang_heads.rows[n_lab:tr.engine.lang_heads.n_rows] == -1).all()
        same = {k: torch.equal(before[k], tr.store.view(k)) for k in before}
        assert same["obj_predict_head.linear_feat.weight"] == (task != "vis_mask")
        assert same["cls.seq_relationship.weight"] == (task != "matched")
        assert same["cls.predictions.transform.dense.weight"] == (task != "word_mask")
        assert not same["bert.encoder.layer.0.output.dense.weight"]


@pytest.mark.parametrize("mode", ["confidence", "tlbr", "random"])
def test_ar_sampler_fp32_matches_reference_fixture(mode):
    from test_engine_cpu import check_ar_sampler, make_sampler_engine
    from xlxmert_amd.ops import HipOps
    g = load_golden("sampler_ar_tiny")
    eng, sd = make_sampler_engine(g, HipOps(torch.float32), device="cuda")
    check_ar_sampler(g, eng, mode)


@pytest.mark.parametrize("overlap", [False, True])
def test_training_reduces_the_loss_full_size_bf16(overlap):
    """end-to-end sanity at the full architecture (overlap: with the optimizer pass behind the step): 40 optimisation steps on ONE fixed batch (bf16, dropout on, clip, AdamW,
    warm-up) must drive the masked-token loss down from ~ln(10000) -- every kernel, the hand-derived backward and the optimizer
    have to agree for that."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig()
    B = 32
    tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda", seed=3, lr=2e-4, warmup_ratio=0.1, total_steps=60,
                      train_dropout=True, overlap_optimizer=overlap)
    g = torch.Generator().manual_seed(1)
    tr.store.view("mask_feat").copy_(torch.randn(cfg.visual_feat_dim, generator=g).relu() * 0.1)
    tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
    batch = {k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=5).items()}
    first = last = None
    for t in range(40):
        losses = tr.step(batch)
        if t == 1:
            first = losses[0].item()
        last = losses[0].item()
    torch.cuda.synchronize()
    assert first > 8.0, first                        # ~ ln(10000) = 9.2 at initialisation
    assert last < 0.6 * first, (first, last)
    assert torch.isfinite(tr.store.master).all()


@pytest.mark.parametrize("drop", [False, True], ids=["keep-grads", "adamw-clears-grads"])
def test_plan_replay_equals_eager_steps_bf16(drop):
    """The planned step (PretrainStep(plan=True): one recorded launch plan per masked-row geometry, replayed by one C call;
    dropout step seed / schedule scalars / inputs in device memory) against the same step enqueued from Python, full
    architecture, bf16, dropout on: 9 steps over 3 batches, so plans are recorded (steps 1-3), replayed (4-8) and re-used with
    fresh dropout masks.  Every step starts from the SAME state in both trainers (the eager trainer's parameters and Adam
    moments are copied over after each step: bf16 training amplifies the fp32-atomic summation order of the weight
    gradients chaotically over steps, which is not what is under test) and must give the same loss, gradient norm and
    gradients up to that summation order."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig()
    B = 64
    g = torch.Generator().manual_seed(1)
    mask_feat = torch.randn(cfg.visual_feat_dim, generator=g).relu() * 0.1
    cent = torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu()
    batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=50 + i).items()} for i in range(3)]
    trs = []
    for plan in (False, True):
        tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda", seed=3, lr=1e-4, total_steps=100,
                          train_dropout=True, plan=plan, drop_grads=drop)
        assert tr.plan_mode == plan and tr.drop_grads == drop
        tr.store.view("mask_feat").copy_(mask_feat)
        tr.set_centroids(cent)
        trs.append(tr)
    te, tp = trs
    replayed, losses_p = 0, []
    for t in range(9):
        n_plans = len(tp._plans)
        before = te.store.master[:te.store.n_used].clone()
        le = te.step(batches[t % 3])[0:1].clone()
        lp = tp.step(batches[t % 3])[0:1].clone()
        torch.cuda.synchronize()
        replayed += int(len(tp._plans) == n_plans and t > 0)
        losses_p.append(lp.item())
        assert abs(le.item() - lp.item()) <= 2e-5 * abs(le.item()), (t, le.item(), lp.item())
        assert abs(te.grad_norm() - tp.grad_norm()) <= 2e-4 * te.grad_norm(), (t, te.grad_norm(), tp.grad_norm())
        n = te.store.n_used
        if drop:            # the optimizer pass cleared the gradients (all but the chunks the next backward overwrites, flag bit 2:
            for tr_ in (te, tp):        # PretrainStep overwrite_grads): compare what it did with them
                cleared = ((tr_.store.decay_flags[:n // 256] & 4) == 0).repeat_interleave(256)
                assert tr_.overwrite_grads and 0.05 < cleared.float().mean().item() < 0.3      # (embeddings, biases, LayerNorm affines)
                assert tr_.store.grad[:n][cleared].abs().max().item() == 0
            pe, pp = te.store.master[:n], tp.store.master[:n]
            assert (pe - pp).norm().item() <= 2e-3 * (pe - before).norm().item(), (t, (pe - pp).norm().item())
        else:
            ge, gp = te.store.grad[:n], tp.store.grad[:n]
            assert (ge - gp).norm().item() <= 1e-3 * ge.norm().item(), (t, (ge - gp).norm().item(), ge.norm().item())
        for name in ("master", "exp_avg", "exp_avg_sq", "compute"):        # same starting point for the next step
            getattr(tp.store, name).copy_(getattr(te.store, name))
    assert replayed >= 5 and len(tp._plans) == 3 and tp.t == 9 and int(tp.step_dev.item()) == 9
    # replays of one plan draw fresh dropout masks: steps 4 and 7 replay the same plan on the same batch
    assert losses_p[4] != losses_p[7]


@pytest.mark.parametrize("plan", [False, True])
def test_overlapped_optimizer_equals_plain_steps_bf16(plan):
    """PretrainStep(overlap_optimizer=True): AdamW runs behind the step on a CU-masked stream, group by group in forward order,
    and the next forward waits group by group.  Against the plain trainer, full architecture, bf16, dropout on, full learning
    rate from the first update (a forward that read a not-yet-updated group would show in the loss): 8 steps, both trainers
    started from the same state each step (see test_plan_replay_equals_eager_steps_bf16 for why) -- same loss, gradient norm
    and updated parameters; and the state the overlapped trainer leaves behind is complete (every group updated)."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig()
    B = 64
    g = torch.Generator().manual_seed(2)
    mask_feat = torch.randn(cfg.visual_feat_dim, generator=g).relu() * 0.1
    cent = torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu()
    batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=70 + i).items()} for i in range(2)]
    trs = []
    for overlap in (False, True):
        tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda", seed=3, lr=5e-4, warmup_ratio=0.0, total_steps=100,
                          weight_decay=0.01, train_dropout=True, plan=(plan and overlap), drop_grads=True, overlap_optimizer=overlap)
        assert (tr.opt_stream is not None) == overlap
        tr.store.view("mask_feat").copy_(mask_feat)
        tr.set_centroids(cent)
        trs.append(tr)
    te, to = trs
    assert sum(hi - lo for _, lo, hi in to._opt_groups) == to.store.n_used
    n = te.store.n_used
    for t in range(8):
        before = te.store.master[:n].clone()
        le = te.step(batches[t % 2])[0:1].clone()
        lo_ = to.step(batches[t % 2])[0:1].clone()
        to.sync()
        assert abs(le.item() - lo_.item()) <= 2e-5 * abs(le.item()), (t, le.item(), lo_.item())
        assert abs(te.grad_norm() - to.grad_norm()) <= 2e-4 * te.grad_norm(), (t, te.grad_norm(), to.grad_norm())
        cleared = ((to.store.decay_flags[:n // 256] & 4) == 0).repeat_interleave(256)      # (all but the chunks the next backward overwrites)
        assert to.store.grad[:n][cleared].abs().max().item() == 0              # every group's pass ran (it clears the gradients)
        pe, po = te.store.master[:n], to.store.master[:n]
        moved = (pe - before).norm().item()
        assert moved > 0 and (pe - po).norm().item() <= 2e-3 * moved, (t, (pe - po).norm().item(), moved)
        assert torch.equal(to.store.compute[:n], to.store.master[:n].to(torch.bfloat16))      # the compute copy follows the master
        for name in ("master", "exp_avg", "exp_avg_sq", "compute"):
            getattr(to.store, name).copy_(getattr(te.store, name))
        torch.cuda.synchronize()
    assert int(to.step_dev.item()) == 8
    if plan:
        assert len(to._plans) >= 1


def test_visual_attention_mask_hidden_states_and_pooled_gradient_fp32():
    from test_engine_cpu import check_vismask, make_vismask_engine
    from xlxmert_amd.ops import HipOps
    g = load_golden("vismask_tiny")
    check_vismask(g, make_vismask_engine(g, HipOps(torch.float32), device="cuda"), 1e-4, 1e-4)


# ---------------------------------------------------------------- QA branch (task_qa model)
@pytest.mark.parametrize("task", ["qa", "vis_mask", "word_mask", "matched"])
def test_qa_branch_steps_fp32_match_reference_fixture(task):
    from test_engine_cpu import check_qa_task, make_qa_engine
    from xlxmert_amd.ops import HipOps
    g = load_golden("qa_tasks_tiny")
    eng, inp = make_qa_engine(g, task, HipOps(torch.float32), device="cuda")
    check_qa_task(g, task, eng, inp, 2e-5, 1e-4, dev="cuda")


@pytest.mark.parametrize("task", ["qa", "vis_mask"])
def test_qa_branch_steps_bf16_stated_tolerance(task):
    """bf16, multi-task ("all") store: total loss within 1 %, every stored gradient within 5 % of the tensor's scale."""
    from test_engine_cpu import make_qa_engine, run_qa_task
    from xlxmert_amd.ops import HipOps
    g = load_golden("qa_tasks_tiny")
    eng, inp = make_qa_engine(g, task, HipOps(torch.bfloat16), device="cuda", dtype=torch.bfloat16, store_task="all")
    total = run_qa_task(eng, inp, task, "cuda")
    torch.cuda.synchronize()
    assert abs(total.item() - float(g[task + ":total_loss"])) < 1e-2 * float(g[task + ":total_loss"])
    worst = 0.0
    for k in [str(n) for n in g[task + ":grad_names"]]:
        ref_n = g[f"{task}:gnorm:{k}"].item()
        got_n = eng.store.gview(k).double().norm().item()
        if ref_n > 1e-4:
            worst = max(worst, abs(got_n - ref_n) / ref_n)
    print(f"qa fixture bf16 {task}: worst gradient-norm error {worst:.4f}")
    assert worst < 3e-2                 # measured 0.9 %


def test_inputs_embeds_fp32_matches_reference_fixture():
    from test_engine_cpu import check_inputs_embeds
    from xlxmert_amd.ops import HipOps
    check_inputs_embeds(load_golden("embeds_tiny"), HipOps(torch.float32), device="cuda", tol=1e-4, gtol=1e-4)


# ---------------------------------------------------------------- two ranks, real kernels, one GPU
def _gpu_dp_worker(rank, world, port, out_dir, overlap):
    import torch.distributed as dist
    from test_trainer_cpu import TINY, oracle_cfg
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import reserve_streams
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    reserve_streams("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = XLxmertConfig(**TINY)
    store = ParamStore(cfg, "cuda:0", torch.float32, task="vis_mask")
    store.load_named(O.make_state_dict(oracle_cfg(cfg), 3))
    tr = PretrainStep(cfg, 2, 8, 16, dtype=torch.float32, device="cuda:0", store=store, total_steps=10, lr=1e-2, bucket_mb=0.05,
                      visual_losses="obj,feat", overlap_optimizer=overlap)
    assert tr.exchange and tr.world == 2 and (tr.opt_stream is not None) == overlap
    batch = {k: v.cuda() for k, v in synthetic_batch(cfg, 2, 8, 4, seed=500 + rank).items()}       # disjoint per-rank minibatch
    tr.step(batch)
    tr.sync()
    assert len(tr._works) > 3 and sum(b - a for a, b in tr._slices) == tr.store.n_used
    bad = tr.verify_replicas()
    assert bad == [], (len(bad), bad[:12])
    torch.save({k: tr.store.view(k).cpu().clone() for k in tr.store.names()}, os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_data_parallel_two_ranks_real_kernels_one_gpu(tmp_path, overlap):
    """Two processes x disjoint minibatches on ONE MI355X (gloo carries the collectives: RCCL refuses two ranks on a device):
    the HIP kernels, the four engine streams, the bucketed exchange issued from the streams that finish the gradient slices
    and (overlap) the optimizer pass behind the step -- with a real second rank, i.e. collectives that are not identities.
    Replicas identical, and equal to one AdamW step on the MEAN of the per-rank gradients of the oracle (DDP semantics)."""
    import torch.multiprocessing as mp
    from test_trainer_cpu import TINY, _free_port, oracle_cfg, oracle_grads
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import linear_schedule, synthetic_batch
    world, port = 2, _free_port()
    mp.spawn(_gpu_dp_worker, args=(world, port, str(tmp_path), overlap), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k
    cfg = XLxmertConfig(**TINY)
    sd = O.make_state_dict(oracle_cfg(cfg), 3)
    gs = [oracle_grads(cfg, sd, synthetic_batch(cfg, 2, 8, 4, seed=500 + r))[0] for r in range(world)]
    names = sorted(gs[0])
    _, clipped = O.clip_grad_norm([(gs[0][k] + gs[1][k]) / 2 for k in names], 1.0)
    lr = 1e-2 * linear_schedule(0, 0, 10)
    for k, g in zip(names, clipped):
        p, _, _ = O.adamw_update(sd[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, lr)
        assert (r0[k] - p).abs().max().item() < 5e-5, (k, (r0[k] - p).abs().max().item())


def _gpu_dp_worker_full(rank, world, port, out_dir, comm):
    import torch.distributed as dist
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import reserve_streams
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    reserve_streams("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = XLxmertConfig()
    B = 32
    tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda:0", seed=3, lr=2e-4, warmup_ratio=0.0, total_steps=100,
                      train_dropout=True, overlap_optimizer=True, grad_comm_dtype=comm)
    g = torch.Generator().manual_seed(1)
    tr.store.view("mask_feat").copy_(torch.randn(cfg.visual_feat_dim, generator=g).relu() * 0.1)
    tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
    assert tr.exchange and (tr.comm_buf is not None) == (comm == torch.bfloat16)
    losses = []
    for t in range(3):
        batch = {k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=900 + 10 * t + rank).items()}
        losses.append(tr.step(batch)[0:1].clone())
        bad = tr.verify_replicas()
        assert bad == [], (t, len(bad), bad[:8])
    tr.sync()
    assert all(torch.isfinite(l).all() for l in losses) and torch.isfinite(tr.store.master).all()
    torch.save({"norm": tr.grad_norm(), "p": tr.store.master[:4096].cpu().clone()}, os.path.join(out_dir, f"f{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("comm", [torch.float32, torch.bfloat16])
def test_data_parallel_two_ranks_full_size_bf16_replicas_stay_identical(tmp_path, comm):
    """the benchmarked architecture (bf16, ping-pong GEMMs, grouped weight gradients, deferred reductions, four streams, dropout,
    optimizer pass behind the step) on two ranks sharing one GPU, fp32 and bf16 gradient buckets: after each of three steps every
    parameter and Adam moment is bit-identical on both ranks (per-tensor checksums), i.e. nothing after the exchange depends on
    the rank -- each rank's dropout masks and minibatch do differ."""
    import torch.multiprocessing as mp
    from test_trainer_cpu import _free_port
    world, port = 2, _free_port()
    mp.spawn(_gpu_dp_worker_full, args=(world, port, str(tmp_path), comm), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "f0.pt"), torch.load(tmp_path / "f1.pt")
    assert r0["norm"] == r1["norm"] and torch.equal(r0["p"], r1["p"])


def _gpu_dp_worker_tasks(rank, world, port, out_dir):
    import torch.distributed as dist
    from test_trainer_cpu import TINY, oracle_cfg
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import reserve_streams
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    reserve_streams("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = XLxmertConfig(**TINY)
    oc = oracle_cfg(cfg)
    B, L, grid = 2, 8, 4
    cuda = lambda d: {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}
    # VQA fine-tune step and NLVR2 step
    A = 29
    store = ParamStore(cfg, "cuda:0", torch.float32, task="vqa", num_answers=A)
    store.load_named(O.make_vqa_state_dict(oc, A, 5))
    trv = PretrainStep(cfg, B, L, grid * grid, dtype=torch.float32, device="cuda:0", store=store, total_steps=10, lr=1e-2,
                       task="vqa", num_answers=A, bucket_mb=0.05, overlap_optimizer=True)
    trv.step(cuda(O.make_vqa_inputs(oc, A, 600 + rank, B, L, grid)))
    assert trv.verify_replicas() == []
    trn = PretrainStep(cfg, 2 * B, L, grid * grid, dtype=torch.float32, device="cuda:0", total_steps=10, lr=1e-2, task="nlvr2",
                       bucket_mb=0.05, overlap_optimizer=True)
    trn.step(cuda(O.make_nlvr2_inputs(oc, 650 + rank, B, L, grid)))
    assert trn.verify_replicas() == []
    # task round-robin on one parameter set with the QA head riding on every branch
    NQ = 11
    store = ParamStore(cfg, "cuda:0", torch.float32, task="all", num_answers=NQ)
    store.load_named(O.make_qa_state_dict(oc, NQ, 9))
    tra = PretrainStep(cfg, B, L, grid * grid, dtype=torch.float32, device="cuda:0", store=store, total_steps=10, lr=1e-2,
                       task="all", num_answers=NQ, bucket_mb=0.05, visual_losses="obj,feat", overlap_optimizer=True)
    for t, task in enumerate(["vis_mask", "word_mask", "matched", "qa"]):
        batch = synthetic_batch(cfg, B, L, grid, seed=700 + 10 * t + rank)
        wl, ml = O.make_lang_task_labels(oc, batch["input_ids"], 800 + 10 * t + rank)
        batch.update(word_labels=wl, matched_labels=ml, qa_labels=O.make_qa_labels(NQ, B, 900 + 10 * t + rank))
        tra.step(cuda(batch), task=task)
        bad = tra.verify_replicas()
        assert bad == [], (task, bad[:4])
    if rank == 1:                                   # a diverged replica is caught, tensor by tensor
        tra.store.view("bert.pooler.dense.bias")[3] += 1.0
    assert tra.verify_replicas() == ["param:bert.pooler.dense.bias"]
    torch.save({"ok": True}, os.path.join(out_dir, f"t{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_other_tasks_one_gpu(tmp_path):
    """VQA and NLVR2 fine-tune steps and the four-task round-robin (per-tensor update counts, skipped tensors) on two ranks sharing
    one GPU, optimizer pass behind the step: replicas bit-identical after every step; a deliberately diverged tensor is named."""
    import torch.multiprocessing as mp
    from test_trainer_cpu import _free_port
    world, port = 2, _free_port()
    mp.spawn(_gpu_dp_worker_tasks, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "t0.pt").exists() and (tmp_path / "t1.pt").exists()


def test_benchmark_geometry_bs256_step_matches_oracle():
    """The geometry bench.py times, against the oracle AT THAT SIZE: bs 256, 9/5/5, d = 768, 20 x 64 tokens, 10k codebook, bf16,
    launch plan, codebook head on the masked rows, four streams, tail split, two-layer grouped weight gradients, optimizer pass
    behind the step (64 row tiles per GEMM, 3-5 tile rounds: what the B = 8 reference fixture cannot reach).  Dropout off and
    lr = 0, so the third step -- a plan REPLAY -- sees the initial parameters; its loss and every gradient are compared with the
    CPU oracle's fp32 step on the same batch (~20 s of host time).  Tolerances: the yardstick of the full_955 fixture test -- the
    reference's own arithmetic under torch bf16 autocast is off by 4.5 % in its worst gradient norm, 15.4 % relative L2 in its
    worst tensor and 5.0 % in the median tensor (B = 8); measured here at bs 256 over WHOLE tensors: loss 5e-4, 4.5 % / 11.0 %
    (r_layers.0 query bias, 19 layers from the loss) / 4.1 %."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig()
    oc = O.OracleConfig(**{k: getattr(cfg, k) for k in CFG_KEYS})
    sd = O.make_state_dict(oc, 2718)
    B = 256
    store = ParamStore(cfg, "cuda", torch.bfloat16)
    store.load_named(sd)
    tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda", store=store, lr=0.0, total_steps=1000, plan=True,
                      drop_grads=False, overlap_optimizer=True, train_dropout=False)
    assert tr.plan_mode and tr.engine.compact_head and tr.engine.side is not None
    batch = synthetic_batch(cfg, B, 20, 8, seed=31)
    dev = {k: v.cuda() for k, v in batch.items()}
    for _ in range(3):                              # eager (allocates), eager + record, replay
        losses = tr.step(dev)
    tr.sync()
    assert len(tr._plans) == 1 and 0 < tr.engine.n_mrows < tr.engine.MV
    assert tr.t == 3
    # oracle step (fp32, CPU) on the same parameters and batch; canonical recipe: obj loss only (scripts/pretrain.bash:15)
    from bench import usable_cores                  # min(affinity mask, cgroup quota): more threads than that thrash (347 s vs ~20 s)
    torch.set_num_threads(usable_cores())
    leaf = {k: v.clone().requires_grad_(v.is_floating_point() and k != "vis_emb.weight") for k, v in sd.items()}
    leaf["obj_predict_head.out_cluster.weight"] = leaf["vis_emb.weight"]
    ref = O.xlxmert_vis_mask_forward(leaf, oc, batch["input_ids"], batch["visual_pos"], batch["attention_mask"],
                                     batch["cluster_ids"], batch["vis_mask"], batch["obj_labels"])
    ref["obj_loss"].backward()
    rel_loss = abs(losses[0].item() - ref["obj_loss"].item()) / ref["obj_loss"].item()
    worst_n, worst_t, rels = ("", 0.0), ("", 0.0), []
    n_checked = 0
    for k, v in leaf.items():
        if v.grad is None or k == "obj_predict_head.out_cluster.weight":
            continue
        assert store.index[k].offset < store.n_used, k
        got, want = store.gview(k).cpu().double(), v.grad.double()
        wn = want.norm().item()
        if wn < 1e-7:                               # key biases: analytically zero gradient
            assert got.norm().item() < 1e-4, k
            continue
        n_checked += 1
        en = abs(got.norm().item() - wn) / wn
        et = (got - want).norm().item() / wn
        rels.append(et)
        if en > worst_n[1]:
            worst_n = (k, en)
        if et > worst_t[1]:
            worst_t = (k, et)
    rels.sort()
    med = rels[len(rels) // 2]
    print(f"bs-256 bench geometry vs oracle: loss rel err {rel_loss:.5f}; {n_checked} gradients: worst norm error {worst_n}, "
          f"worst tensor relative L2 {worst_t}, median {med:.4f}")
    assert n_checked > 400
    assert rel_loss < 5e-3
    assert worst_n[1] < 6e-2 and worst_t[1] < 0.154 and med < 5e-2, (worst_n, worst_t, med)


@pytest.mark.parametrize("sweep_deadline", [None, "0.2"], ids=["sweep", "sweep_past_its_deadline"])
def test_bench_two_ranks_sharing_the_gpu_end_to_end(tmp_path, sweep_deadline):
    """`python bench.py --gpus 2` started as a plain script (no WORLD_SIZE: it launches its own two ranks, as the reference's
    entry point does with mp.spawn, ref lxmert_pretrain.py:865) with both ranks on this GPU and gloo carrying the exchange: ONE
    JSON line, the exchange filled in, the step replayed from a SEGMENTED launch plan (collectives between the segments)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XL_BENCH_SHARE_GPU="1", XL_BENCH_FAULT_TIMEOUT="380", XL_BENCH_SWEEP_SHARED="1")
    if sweep_deadline is not None:                  # the sweep cannot finish in time: the headline line must still come out, once
        env["XL_BENCH_SWEEP_TIMEOUT"] = sweep_deadline
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    # (output into files, not pipes: a helper process of the launcher that outlives it would keep a pipe open and the read
    #  would wait for it -- the launcher's own exit is what ends the run)
    with open(tmp_path / "out", "w") as fo, open(tmp_path / "err", "w") as fe:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "4",
                            "--warmup", "3", "--batch", "64", "--no-extra", "--no-cpu-baseline", "--prewarm-steps", "0"], env=env, stdout=fo, stderr=fe,
                           stdin=subprocess.DEVNULL, timeout=600)
    stdout, stderr = (tmp_path / "out").read_text(), (tmp_path / "err").read_text()
    assert r.returncode == 0, stderr[-6000:]
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout[-2000:]
    out = json.loads(lines[0])
    assert out["shared_gpu"] is True and out["n_ranks"] == 2 and out["n_gpus"] == 1
    ex = out["config"]["gradient_exchange"]
    assert ex["bytes_per_step"] > 5e8 and ex["backend"].startswith("gloo") and ex["exposed_comm_ms_per_step"] >= 0.0
    assert out["config"]["step_launch"].startswith("launch plan") and "host operations" in out["config"]["step_launch"]
    assert out["value"] > 0 and out["roofline"]["frac"] > 0.0
    # the other exchange variants, each with a trainer and an exchange of its own, measured after the headline in the same run
    sw = out["exchanges"]
    assert sw["headline"] == "allreduce_fp32" and ex["rccl_nranks"] == 2
    if sweep_deadline is not None:
        assert "not finished" in sw["error"] and sw["headline_ms_per_step"] == out["ms_per_step"]
        return
    for name in ("allreduce_bf16_buckets", "rs+ag_fp32_gather", "rs+ag_bf16_gather"):
        assert "error" not in sw[name], (name, sw[name])
        assert sw[name]["ms_per_step"] > 0 and sw[name]["exposed_comm_ms_per_step"] >= 0.0
    assert sw["allreduce_bf16_buckets"]["bytes_per_step"] * 2 == ex["bytes_per_step"]
    print("2 ranks on one GPU:", out["ms_per_step"], "ms/step, host", out["host_enqueue_ms_per_step"], ex, sw)


def test_two_library_contexts_interleaved_in_one_process():
    """SURVEY 8b 'no global mutable state except the handle': a trainer (dropout step-seed pointer, deferred column reductions,
    launch plan) and a sampler engine (no dropout, other GEMM kernel choice) in ONE process, their calls interleaved: each
    object's settings live in its own library context (xl_ctx_*), so neither sees the other's -- the trainer's losses, gradient
    norm and parameters equal those of the same trainer running alone, the sampler's codes those of the sampler alone."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.ops import HipOps
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import PretrainStep, init_reference_weights, synthetic_batch
    cfg = XLxmertConfig(vocab_size=200, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                        max_position_embeddings=32, visual_feat_dim=64, num_clusters=96, l_layers=2, x_layers=2, r_layers=2)
    g = torch.Generator().manual_seed(3)
    cents = torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu()
    B = 8
    batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=40 + i).items()} for i in range(4)]

    def trainer():
        tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda", seed=5, lr=1e-3, total_steps=100, train_dropout=True,
                          plan=True, drop_grads=False)
        tr.set_centroids(cents)
        tr.ops.set_gemm_wgrad_slabs(1)              # K-split weight gradients through slabs, summed in slice order: no atomics noise there
        return tr

    def sampler():
        store = ParamStore(cfg, "cuda", torch.bfloat16)
        init_reference_weights(store, 9)
        store.set_centroids(cents)
        ops = HipOps(torch.bfloat16)
        ops.set_gemm_pingpong(0)                    # a setting of ITS context only
        eng = Engine(cfg, store, ops, B, 20, 64, need_lang=False)
        eng.sync_compute_weights()
        b = batches[0]
        eng.set_inputs(b["input_ids"], b["attention_mask"], None, b["visual_pos"],
                       cluster_ids=torch.zeros(B, 64, dtype=torch.long, device="cuda"), vis_mask=torch.ones(B, 64, dtype=torch.bool, device="cuda"))
        return eng

    tr = trainer()
    alone = []
    for i in range(4):
        alone.append(tr.step(batches[i]).clone())
    tr.sync()
    alone_p, alone_n = tr.store.master.clone(), tr.grad_norm()
    sm = sampler()
    codes_alone = sm.sample_codes_nar(3)[0].clone()
    del tr, sm
    tr, sm = trainer(), sampler()
    assert tr.ops.ctx != sm.ops.ctx
    mixed = []
    for i in range(4):
        mixed.append(tr.step(batches[i]).clone())
        codes = sm.sample_codes_nar(3)[0].clone()           # between two trainer steps (eager, recording and replaying ones)
        assert torch.equal(codes, codes_alone)
    tr.sync()
    for a, b in zip(alone, mixed):
        assert torch.allclose(a, b, rtol=2e-5, atol=1e-6), (a, b)        # (the loss scalars are summed by one fp32 atomic per block)
    # Round 4 (tools/race_probe.py) traced a BIMODAL run-to-run deviation of this comparison -- 1e-8, or 3.3e-4 on the norm -- to fp32
    # summation order in two column sums of the feature encoder's backward: the second stage of every two-stage column reduction
    # added its G-slices with atomics.  Round 5: that combine sums the slices in a fixed order (csrc/rowops.hip reduce_partials_entry),
    # the embedding scatter has one writer per table row, and the K-split weight gradients go through slabs here -- every gradient
    # is reproducible, so the two runs agree to the last bit (test_training_step_is_bit_reproducible is the direct statement).
    assert abs(tr.grad_norm() - alone_n) <= 2e-5 * alone_n
    assert torch.equal(tr.store.master, alone_p)


def test_paired_blocks_equal_the_two_stream_engine_on_the_gpu():
    """Engine.run_pair on the real kernels and streams (fp32: every contraction of a pair falls back to its two xl_gemm launches
    inside xl_gemm_pair; the lanes, scratch generations, deferred column sums and the language stream hand-overs are the real ones)"""
    from test_engine_cpu import check_paired_blocks
    from xlxmert_amd.ops import HipOps
    check_paired_blocks(lambda: HipOps(torch.float32), "cuda", 2e-5)


@pytest.mark.parametrize("side", [False, True])
def test_paired_blocks_full_width_bf16_step(side):
    """the benchmarked architecture (9/5/5, d = 768) in bf16 with the pairs going through ONE launch each (xl_gemm_pair's paired
    instances, ping-pong kernel forced): loss and every gradient norm against the reference fixture full_955, as the two-stream
    engine is checked -- and against the two-stream engine itself on the same inputs."""
    from test_engine_cpu import make_engine
    from xlxmert_amd.ops import HipOps
    g = load_golden("full_955")
    res = {}
    for mode in ("two_streams", "paired"):
        ops = HipOps(torch.bfloat16)
        ops.set_gemm_pingpong(2)
        eng, oc, sd, inp = make_engine_on(g, ops, torch.bfloat16, pair=(mode == "paired"), side=side)
        losses = eng.vis_mask_forward_backward()
        torch.cuda.synchronize()
        res[mode] = (losses.clone(), eng.store.grad[:eng.store.n_used].clone())
    (l0, g0), (l1, g1) = res["two_streams"], res["paired"]
    assert torch.allclose(l0, l1, rtol=2e-3, atol=1e-4), (l0, l1)
    rel = (g0 - g1).norm().item() / g0.norm().item()
    assert rel < 2e-2, rel                         # (bf16: the paired launches are bit-identical, fp32 atomics / dropout-free noise only)


@pytest.mark.parametrize("side", [False, True])
def test_vqa_step_paired_blocks_full_depth(side):
    """A fine-tune step (language-side answer head: its backward leaves column-sum partials filed under the LANGUAGE stream) with
    the blocks paired onto the visual chain, at the full 9 / 5 / 5 depth: the language lane's scratch generations wrap around (8
    sets) while those partials are still pending, so the lane must combine them on their own stream before it reuses the set
    (round 5: an assertion fired here).  Same gradients as the two-stream schedule."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig(vocab_size=300, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                        max_position_embeddings=32, visual_feat_dim=64, num_clusters=96, l_layers=9, x_layers=5, r_layers=5)
    B, A = 16, 40
    g = torch.Generator().manual_seed(11)
    b = synthetic_batch(cfg, B, 20, 8, seed=21)
    tgt = torch.zeros(B, A)
    tgt[torch.arange(B), torch.randint(0, A, (B,), generator=g)] = 1.0
    batch = {"input_ids": b["input_ids"].cuda(), "visual_pos": b["visual_pos"].cuda(),
             "visual_feats": torch.randn(B, 64, cfg.visual_feat_dim, generator=g).relu().cuda(), "targets": tgt.cuda()}
    res = {}
    for mode in ("two_streams", "paired"):
        tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda", seed=5, task="vqa", num_answers=A, train_dropout=False,
                          total_steps=100, plan=False, drop_grads=False)
        tr.engine.pair_blocks, tr.engine.pair_side = mode == "paired", side
        losses = [tr.step(batch).clone() for _ in range(2)]
        tr.sync()
        res[mode] = (losses, tr.store.master[:tr.store.n_used].clone())
    (l0, p0), (l1, p1) = res["two_streams"], res["paired"]
    for a, c in zip(l0, l1):
        assert torch.allclose(a, c, rtol=2e-3, atol=1e-4), (a, c)
    assert torch.isfinite(p1).all()
    rel = (p0 - p1).norm().item() / p0.norm().item()
    assert rel < 1e-4, rel                      # parameters after two steps (paired launches are bit-identical; fp32 atomics noise only)


def test_two_scratch_generations_give_the_same_gradients(monkeypatch):
    """XL_SCRATCH_GENS=2 (the minimum): every backward scratch set is reused two closes later.  A cross-modality layer closes two
    sets per layer (self-attention + cross-attention) while its FFN / self-attention weight gradients are HELD for a grouped launch
    with the next layer, so the set they read comes up for reuse before they are launched: the engine has to launch them first
    (Engine._advance_gen).  Round 5: with two sets the next layer overwrote dpre / dqkv under them -- 36 fixture tests off by
    1e-2 -- which the default eight sets hid.  Same step, 2 against 8 sets: identical parameters after two steps."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig(vocab_size=300, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                        max_position_embeddings=32, visual_feat_dim=64, num_clusters=96, l_layers=3, x_layers=3, r_layers=3)
    g = torch.Generator().manual_seed(3)
    cents = torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu()
    B = 16
    batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=50 + i).items()} for i in range(2)]
    res = {}
    for ngen in (8, 2):
        monkeypatch.setattr(Engine, "NGEN", ngen)
        tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda", seed=5, lr=1e-3, total_steps=100, train_dropout=True,
                          plan=False, drop_grads=True, overlap_optimizer=True)
        tr.set_centroids(cents)
        tr.ops.set_gemm_wgrad_slabs(1)
        for b in batches:
            tr.step(b)
        tr.sync()
        res[ngen] = tr.store.master[:tr.store.n_used].clone()
    assert torch.equal(res[8], res[2]), ((res[8] - res[2]).abs().max().item(), (res[8] != res[2]).float().mean().item())


def make_engine_on(g, ops, dtype, pair, side):
    from _util import golden_cfg, golden_inputs
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.params import ParamStore
    oc = golden_cfg(g)
    cfg = XLxmertConfig(**{k: getattr(oc, k) for k in ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size",
                                                      "max_position_embeddings", "type_vocab_size", "l_layers", "x_layers",
                                                      "r_layers", "visual_feat_dim", "visual_pos_dim", "num_clusters")})
    sd = O.make_state_dict(oc, int(g["seed"]))
    inp = {k: v.cuda() for k, v in golden_inputs(g).items()}
    B, L = inp["input_ids"].shape
    V = inp["cluster_ids"].shape[1]
    store = ParamStore(cfg, "cuda", dtype, task="vis_mask")
    store.load_named(sd)
    eng = Engine(cfg, store, ops, B, L, V, need_lang=False)
    eng.pair_blocks, eng.pair_side = pair, side
    eng.sync_compute_weights()
    eng.set_inputs(inp["input_ids"], inp["attention_mask"], inp["token_type_ids"], inp["visual_pos"],
                   cluster_ids=inp["cluster_ids"], vis_mask=inp["vis_mask"], obj_labels=inp["obj_labels"])
    return eng, oc, sd, inp


@pytest.mark.parametrize("plan", [False, True])
def test_training_step_is_bit_reproducible(plan):
    """The same three training steps (bf16, dropout on, four streams, deferred reductions, grouped weight gradients, overwrite
    mode, AdamW behind the step) twice in one process from the same state: parameters, both Adam moments and the gradient norm must
    be BIT-identical.  What makes that true: column sums combine their partial slabs in a fixed order, the embedding scatter has
    one writer per table row, the position table one writer per element, xl_sumsq adds its block partials in index order, and
    K-split weight gradients meet in slabs summed in slice order (xl_set_gemm_wgrad_slabs(1): opt-in, 1-3 % slower than fp32
    atomics, which is why the DEFAULT keeps atomics for the few K-split launches).  The two loss scalars are the exception: one fp32
    atomic per block -- compared to 1e-5."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig(vocab_size=200, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                        max_position_embeddings=32, visual_feat_dim=64, num_clusters=96, l_layers=3, x_layers=2, r_layers=2)
    g = torch.Generator().manual_seed(3)
    cents = torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu()
    B = 16
    batches = [{k: v.cuda() for k, v in synthetic_batch(cfg, B, 20, 8, seed=70 + i).items()} for i in range(3)]

    def run():
        tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device="cuda", seed=5, lr=1e-3, total_steps=100, train_dropout=True,
                          plan=plan, drop_grads=True, overlap_optimizer=True)
        tr.set_centroids(cents)
        tr.ops.set_gemm_wgrad_slabs(1)
        losses, params = [], []
        for i in range(6):
            losses.append(tr.step(batches[i % 3]).clone())
            tr.sync()
            params.append(tr.store.master.clone())
        st = tr.store
        return losses, st.master.clone(), st.exp_avg.clone(), st.exp_avg_sq.clone(), tr.grad_norm(), params

    a, b = run(), run()
    for i, (x, y) in enumerate(zip(a[5], b[5])):
        assert torch.equal(x, y), ("parameters after step", i, (x - y).abs().max().item(), (x != y).float().mean().item())
    for i, (x, y) in enumerate(zip(a[0], b[0])):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-7), (i, (x - y).abs().max().item(), x, y)
    assert a[4] == b[4], (a[4], b[4])
    for name, x, y in (("master", a[1], b[1]), ("exp_avg", a[2], b[2]), ("exp_avg_sq", a[3], b[3])):
        assert torch.equal(x, y), (name, (x - y).abs().max().item(), (x != y).float().mean().item())


@pytest.mark.parametrize("plan", [False, True])
@pytest.mark.parametrize("case", ["one_example_two_tokens", "ragged_extremes", "all_full"])
def test_edge_case_batches_match_oracle_fp32(case, plan):
    """Batches at the corners of the input domain, every gradient against the CPU oracle (fp32, dropout off):
    one example whose text is [CLS][SEP] only; lengths 2 / L / 5 / 3 next to an example with NO masked visual token (its feature
    loss term is 0 / max(0, 1), its rows carry label -100), one with ALL tokens masked and one with a single masked token; and a
    batch without any padding or any unmasked token.  The loader's optional row lists are left out, so the engine derives the
    packed language rows itself.  With plan = True the third step is a replay of the recorded launch plan."""
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import PretrainStep
    cfg = XLxmertConfig(vocab_size=100, hidden_size=64, num_attention_heads=4, intermediate_size=128,
                        max_position_embeddings=32, visual_feat_dim=32, num_clusters=56, l_layers=2, x_layers=2, r_layers=1)
    oc = O.OracleConfig(**{k: getattr(cfg, k) for k in CFG_KEYS})
    sd = O.make_state_dict(oc, 11)
    from test_trainer_cpu import EDGE_CASES, edge_batch
    L, V = 8, 16
    lens, masks = EDGE_CASES[case]
    batch = edge_batch(cfg, lens, masks, L, V, seed=len(lens) * 17 + 1)
    B = len(lens)
    store = ParamStore(cfg, "cuda", torch.float32)
    store.load_named(sd)
    tr = PretrainStep(cfg, B, L, V, dtype=torch.float32, device="cuda", store=store, lr=0.0, total_steps=10, plan=plan,
                      drop_grads=False, visual_losses="obj,feat", train_dropout=False)
    dev = {k: v.cuda() for k, v in batch.items()}
    for _ in range(3):
        losses = tr.step(dev)
    tr.sync()
    leaf = {k: v.clone().requires_grad_(v.is_floating_point() and k != "vis_emb.weight") for k, v in sd.items()}
    leaf["obj_predict_head.out_cluster.weight"] = leaf["vis_emb.weight"]
    ref = O.xlxmert_vis_mask_forward(leaf, oc, batch["input_ids"], batch["visual_pos"], batch["attention_mask"],
                                     batch["cluster_ids"], batch["vis_mask"], batch["obj_labels"])
    ref["total_loss"].backward()
    assert abs(losses[0].item() - ref["obj_loss"].item()) < 1e-4 * max(1.0, abs(ref["obj_loss"].item()))
    assert abs(losses[1].item() - ref["feat_loss"].item()) < 1e-4 * max(1.0, abs(ref["feat_loss"].item()))
    n = 0
    for k, v in leaf.items():
        if v.grad is None or k == "obj_predict_head.out_cluster.weight":
            continue
        got, want = store.gview(k).cpu(), v.grad
        assert torch.isfinite(got).all(), k
        d = (got - want).abs().max().item()
        assert d < 1e-4 * max(1.0, want.abs().max().item()), (case, k, d)
        n += 1
    assert n > 80


def _compare_grads_with_oracle(store, leaf, skip=()):
    """worst gradient-norm error, worst tensor relative L2, median tensor relative L2 of the engine's gradients against the oracle's"""
    worst_n, worst_t, rels, n = ("", 0.0), ("", 0.0), [], 0
    for k, v in leaf.items():
        if v.grad is None or k in skip or k not in store.index:
            continue
        got, want = store.gview(k).cpu().double(), v.grad.double()
        if k.endswith("word_embeddings.weight"):
            want = want.clone(); want[0] = got[0]          # padding_idx row: the decoder-side part only (see the fixture tests)
        wn = want.norm().item()
        if wn < 1e-7:
            assert got.norm().item() < 1e-4, k
            continue
        n += 1
        en, et = abs(got.norm().item() - wn) / wn, (got - want).norm().item() / wn
        rels.append(et)
        if en > worst_n[1]:
            worst_n = (k, en)
        if et > worst_t[1]:
            worst_t = (k, et)
    rels.sort()
    return n, worst_n, worst_t, rels[len(rels) // 2]


@pytest.mark.parametrize("task", ["word_mask", "matched", "vqa", "nlvr2", "vqa512"])
def test_next_rows_at_bench_geometry_match_oracle(task):
    """SURVEY 8f rows N1 / N3 at the sizes bench.py's `other_workloads` times them (full 9/5/5 encoder, bf16, bs 256 for the
    language pretraining branches, bs 128 x 3129 answers for the VQA step, 128 statements = 64 image pairs for NLVR2), dropout off,
    against the CPU oracle's fp32 step on the
    same parameters and batch: the loss and EVERY gradient tensor (these branches' gradients enter through the pooled output / the
    language rows and come out far closer than the vis_mask step's, whose worst tensors sit 19 layers below 8448 masked rows)."""
    from bench import usable_cores
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.ops import HipOps
    from xlxmert_amd.params import ParamStore
    torch.set_num_threads(usable_cores())
    cfg = XLxmertConfig()
    oc = O.OracleConfig()
    L, V = 20, 64
    big = task == "vqa512"               # BASELINE config 4 at its OWN batch (VQA fine-tune, bs 512: ref tasks/vqa.py:166-198) -- ~40 s of host time
    if big:
        task = "vqa"
    if task == "vqa":
        B, A = (512 if big else 128), 3129
        sd = O.make_vqa_state_dict(oc, A, 41)
        inp = O.make_vqa_inputs(oc, A, 43, B, L, 8)
        store = ParamStore(cfg, "cuda", torch.bfloat16, task="vqa", num_answers=A)
    elif task == "nlvr2":
        P = 64
        B = 2 * P
        sd = O.make_nlvr2_state_dict(oc, 41)
        inp = O.make_nlvr2_inputs(oc, 43, P, L, 8)
        store = ParamStore(cfg, "cuda", torch.bfloat16, task="nlvr2")
    else:
        B = 256
        sd = O.make_cls_state_dict(oc, 41)
        inp = O.make_inputs(oc, 43, B, L, 8)
        wl, ml = O.make_lang_task_labels(oc, inp["input_ids"], 44)
        store = ParamStore(cfg, "cuda", torch.bfloat16, task=task)
    store.load_named(sd)
    eng = Engine(cfg, store, HipOps(torch.bfloat16), B, L, V, need_lang=True)
    eng.sync_compute_weights()
    leaf = {k: v.clone().requires_grad_(v.is_floating_point() and k != "vis_emb.weight") for k, v in sd.items()}
    if "vis_emb.weight" in leaf and "obj_predict_head.out_cluster.weight" in leaf:
        leaf["obj_predict_head.out_cluster.weight"] = leaf["vis_emb.weight"]
    if "cls.predictions.decoder.weight" in leaf:
        leaf["cls.predictions.decoder.weight"] = leaf["bert.embeddings.word_embeddings.weight"]       # tied (HF:589-599)
    if task == "vqa":
        eng.set_inputs(inp["input_ids"].cuda(), inp["attention_mask"].cuda(), None, inp["visual_pos"].cuda(),
                       visual_feats=inp["visual_feats"].cuda())
        loss = eng.vqa_forward_backward(inp["targets"].cuda())
        ref = O.vqa_forward(leaf, oc, inp["input_ids"], inp["visual_feats"], inp["visual_pos"], inp["attention_mask"], inp["targets"])
        ref_loss = ref["loss"]
    elif task == "nlvr2":
        Fd = inp["visual_feats"].shape[-1]
        eng.set_inputs(inp["input_ids"].cuda(), inp["attention_mask"].cuda(), None, inp["visual_pos"].reshape(B, V, -1).cuda(),
                       visual_feats=inp["visual_feats"].reshape(B, V, Fd).cuda())
        loss = eng.nlvr2_forward_backward(inp["labels"].cuda())
        ref = O.nlvr2_forward(leaf, oc, inp["input_ids"], inp["visual_feats"], inp["visual_pos"], inp["attention_mask"], inp["labels"])
        ref_loss = ref["loss"]
    else:
        eng.set_inputs(inp["input_ids"].cuda(), inp["attention_mask"].cuda(), None, inp["visual_pos"].cuda(),
                       cluster_ids=inp["cluster_ids"].cuda())
        if task == "word_mask":
            from test_engine_cpu import labelled_rows
            loss = eng.word_mask_forward_backward(wl.cuda(), labelled_rows(wl))
            ref = O.xlxmert_word_mask_forward(leaf, oc, inp["input_ids"], inp["visual_pos"], inp["attention_mask"], inp["cluster_ids"], wl)
        else:
            loss = eng.matched_forward_backward(ml.cuda())
            ref = O.xlxmert_matched_forward(leaf, oc, inp["input_ids"], inp["visual_pos"], inp["attention_mask"], inp["cluster_ids"], ml)
        ref_loss = ref["total_loss"]
    torch.cuda.synchronize()
    ref_loss.backward()
    rel_loss = abs(loss.item() - ref_loss.item()) / max(abs(ref_loss.item()), 1e-6)
    n, worst_n, worst_t, med = _compare_grads_with_oracle(store, leaf, skip=("obj_predict_head.out_cluster.weight",))
    print(f"{task} at bench geometry vs oracle: loss rel err {rel_loss:.5f}; {n} gradients: worst norm error {worst_n}, "
          f"worst tensor relative L2 {worst_t}, median {med:.4f}")
    assert n > 300
    # measured: loss 2e-5 / 1.4e-4 / 1e-6 / 1.7e-3, worst norm 0.24 / 0.26 / 0.28 / 0.84 %, worst tensor 3.3 / 1.8 / 1.1 / 2.5 %, median
    # 1.3 / 0.6 / 0.6 / 1.0 % (word_mask / matched / vqa / nlvr2: a 2-way CE over 64 pairs, 1.2e-3 absolute on a loss of 0.7) -- well
    # inside the vis_mask yardstick; the bounds are ~2x what was measured
    assert rel_loss < 4e-3
    assert worst_n[1] < 2e-2 and worst_t[1] < 7e-2 and med < 3e-2, (worst_n, worst_t, med)


def test_sampler_first_step_at_bench_geometry_matches_oracle_where_decisive():
    """SURVEY 8f row N2 at the size bench.py times (bs 64, 8x8 grid, 10k codebook, bf16, fused row-max head: no logits in memory):
    the first Mask-Predict step -- every position masked -- against the CPU oracle's logits for the same parameters.  bf16 may
    flip an argmax between near-tied codes (the logits are O(15-60): one bf16 ulp of the operands is worth ~0.1-0.2 of a logit), so the
    stated comparison is: wherever the oracle's best logit leads the runner-up by more than 1/64 of its size (69 % of the positions)
    the engine picks the oracle's code, the two agree at >= 92 % of ALL positions, and the confidence -- max softmax, the quantity
    the later steps rank by -- agrees to 4 % in the median where the codes agree."""
    from bench import usable_cores
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.ops import HipOps
    from xlxmert_amd.params import ParamStore
    torch.set_num_threads(usable_cores())
    cfg = XLxmertConfig()
    oc = O.OracleConfig()
    B, L, V = 64, 20, 64
    sd = O.make_state_dict(oc, 19)
    inp = O.make_inputs(oc, 23, B, L, 8)
    ids = inp["input_ids"]
    store = ParamStore(cfg, "cuda", torch.bfloat16, task="vis_mask")
    store.load_named(sd)
    eng = Engine(cfg, store, HipOps(torch.bfloat16), B, L, V, need_lang=False)
    eng.sync_compute_weights()
    pos = torch.from_numpy(O.box_position(8)).unsqueeze(0).expand(B, -1, -1).float()
    eng.set_inputs(ids.cuda(), (ids > 0).cuda(), None, pos.cuda(), cluster_ids=torch.zeros(B, V, dtype=torch.long, device="cuda"),
                   vis_mask=torch.ones(B, V, dtype=torch.bool, device="cuda"))
    cid, code, prob = eng.sample_codes_nar(1)
    torch.cuda.synchronize()
    assert eng.fused_predict_available()
    with torch.no_grad():
        feats = sd["mask_feat"].view(1, 1, -1).expand(B, V, -1)
        _, vis, _ = O.lxmert_model(sd, oc, ids, feats, pos, ids > 0)
        _, obj = O.visual_obj_head(sd, oc, vis)
        top2 = obj.topk(2, dim=2).values
        margin = top2[..., 0] - top2[..., 1]
        ref_prob, ref_id = torch.softmax(obj, dim=2).max(dim=2)
    got_id, got_prob = cid.view(B, V).cpu(), prob.view(B, V).float().cpu()
    scale = top2[..., 0].abs().clamp_min(1.0)
    decisive = margin > scale * 2.0 ** -6            # 4 bf16 ulps of the best logit (the logits here are O(15-60)); measured: 69 % of the positions
    same = got_id == ref_id
    for t in (2.0 ** -7, 2.0 ** -6, 2.0 ** -5, 2.0 ** -4):
        d = margin > scale * t
        print(f"  margin > {t:.4f} x |best logit|: {d.float().mean().item():.3f} of the positions, same code among them {same[d].float().mean().item():.4f}")
    rel = ((got_prob - ref_prob).abs() / ref_prob)[same]
    print(f"sampler step 1 at bs 64: logits std {obj.std().item():.3f}, best logit mean {top2[..., 0].mean().item():.2f}, same code overall "
          f"{same.float().mean().item():.4f}, confidence rel err among agreeing positions: max {rel.max().item():.4f} median {rel.median().item():.4f}")
    assert decisive.float().mean().item() > 0.5
    if os.environ.get("XL_GEMM_SPLIT_EPI", "0") != "0":           # (opt-in K slices: another bf16 re-association -- one or two of the
        assert same[decisive].float().mean().item() > 0.999       #  ~2800 decisive positions flip; both are one realisation of bf16 rounding)
    else:
        assert same[decisive].all()                   # measured: 100 % down to 1/64, 99.1 % at 1/128
    assert same.float().mean().item() >= 0.92         # measured 0.945
    assert rel.median().item() < 4e-2                 # measured 0.023 (max 0.37: a probability is exp of a logit difference)


def test_sampler_four_steps_at_bs256_match_oracle_step_by_step():
    """BASELINE config 5 at the timed geometry (bs 256, T = 4, 8x8 grid, 10k codebook, bf16, fused row-max head): EVERY refinement step
    against the CPU oracle, teacher-forced -- step t's input state (the engine's own code ids after step t-1 and the mask it re-drew
    from its own confidences) goes through the oracle's encoder + codebook head, and the engine's argmax / confidence of step t are
    compared with the oracle's wherever the oracle's best logit is decisive (leads the runner-up by more than 1/64 of its size).
    Teacher forcing is what makes steps 2-4 comparable at all: one bf16 argmax flip between near-tied codes at step 1 changes the masks
    and inputs of every later step of a free-running pair (steps 2-4 at bs 256 rested on the tiny fixture).  Also
    checked per step: the re-masked set is the n_mask lowest-confidence positions of the previous step (ties aside) and codes change
    only where masked (ref tasks/imggen_model.py:199-243)."""
    from bench import usable_cores
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.ops import HipOps
    from xlxmert_amd.params import ParamStore
    torch.set_num_threads(usable_cores())
    cfg = XLxmertConfig()
    oc = O.OracleConfig()
    B, L, V, T = 256, 20, 64, 4
    sd = O.make_state_dict(oc, 19)
    inp = O.make_inputs(oc, 23, B, L, 8)
    ids = inp["input_ids"]
    store = ParamStore(cfg, "cuda", torch.bfloat16, task="vis_mask")
    store.load_named(sd)
    eng = Engine(cfg, store, HipOps(torch.bfloat16), B, L, V, need_lang=False)
    eng.sync_compute_weights()
    pos = torch.from_numpy(O.box_position(8)).unsqueeze(0).expand(B, -1, -1).float()
    eng.set_inputs(ids.cuda(), (ids > 0).cuda(), None, pos.cuda(), cluster_ids=torch.zeros(B, V, dtype=torch.long, device="cuda"),
                   vis_mask=torch.ones(B, V, dtype=torch.bool, device="cuda"))
    snaps = []

    def grab(i):
        snaps.append({"mask": eng.vmask.view(B, V).clone().cpu().bool(), "argmax": eng.row_argmax.view(B, V).clone().cpu().long(),
                      "prob": eng.row_maxprob.view(B, V).float().clone().cpu(), "cid": eng.cid.view(B, V).clone().cpu().long()})

    eng.sample_codes_nar(T, on_step=grab)
    torch.cuda.synchronize()
    assert eng.fused_predict_available() and len(snaps) == T
    cent = sd["vis_emb.weight"].float()
    prev_cid = torch.zeros(B, V, dtype=torch.long)
    for t, s in enumerate(snaps):
        n_mask = int((T - t) / T * V)
        assert (s["mask"].sum(1) == n_mask).all(), (t, s["mask"].sum(1).unique())
        if t > 0:
            # the re-masked positions are the n_mask least confident of the previous step: every masked position's confidence is <= every
            # kept position's (ties at the boundary aside)
            pp = snaps[t - 1]["prob"]
            worst_kept = torch.where(s["mask"], torch.full_like(pp, 2.0), pp).min(1).values
            best_masked = torch.where(s["mask"], pp, torch.full_like(pp, -1.0)).max(1).values
            assert (best_masked <= worst_kept + 1e-6).all(), t
        assert torch.equal(torch.where(s["mask"], s["argmax"], prev_cid), s["cid"]), t          # codes change only where masked
        with torch.no_grad():
            feats = torch.where(s["mask"].unsqueeze(-1), sd["mask_feat"].view(1, 1, -1).float(), cent[prev_cid])
            _, vis, _ = O.lxmert_model(sd, oc, ids, feats, pos, ids > 0)
            _, obj = O.visual_obj_head(sd, oc, vis)
            top2 = obj.topk(2, dim=2).values
            ref_prob, ref_id = torch.softmax(obj, dim=2).max(dim=2)
        margin, scale = top2[..., 0] - top2[..., 1], top2[..., 0].abs().clamp_min(1.0)
        decisive = margin > scale * 2.0 ** -6
        same = s["argmax"] == ref_id
        rel = ((s["prob"] - ref_prob).abs() / ref_prob)[same]
        print(f"sampler step {t + 1}/{T} at bs 256 ({n_mask} masked): decisive {decisive.float().mean().item():.3f} of the positions, same code among "
              f"them {same[decisive].float().mean().item():.5f}, overall {same.float().mean().item():.4f}, confidence rel err median {rel.median().item():.4f}")
        assert decisive.float().mean().item() > 0.4
        assert same[decisive].float().mean().item() >= 0.9995, t          # step 1 at bs 64 measured 100 %; later steps see bf16 centroid rows as inputs too
        assert same.float().mean().item() >= 0.90
        assert rel.median().item() < 5e-2
        prev_cid = s["cid"]
