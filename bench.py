"""Headline benchmark: masked-visual-token pretraining step throughput (BASELINE.json metric).

    python bench.py --gpus 1 --steps 100 --warmup 20        (the defaults: SURVEY 8d)
    python bench.py --gpus N ...                     (no WORLD_SIZE in the environment: starts its own N ranks, as the reference's
                                                      entry point does with mp.spawn, ref pretrain/lxmert_pretrain.py:865)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward + backward + gradient all-reduce (N>1) + clip + AdamW of the full X-LXMERT encoder
(9 language / 5 visual / 5 cross layers, d=768) with the 10k-codebook head, per-GPU batch 256 of
(20 text tokens x 64 visual tokens), bf16 operands / fp32 accumulate, synthetic inputs resident in HBM.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the hosts' driver shares device memory between processes through dmabuf only: without this RCCL's IPC set-up fails with
# `hipIpcGetMemHandle: invalid argument` (read when the HSA runtime starts, so before torch is imported; launched ranks inherit it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

GFLOP_PER_EXAMPLE = 50.782      # necessary fwd+bwd, SURVEY.md section 8d / Appendix D (contract figure)
PEAK_BF16_TFLOPS = 2500.0       # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (measured 2495)


def usable_cores():
    """cores this process may actually use: min(affinity mask, cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(cfg, bs, budget_s=20.0, big_bs=256):
    """The oracle (CPU restatement of the reference path, fp32) timed on this box's host cores: fwd + bwd + clip + AdamW."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lxmert_oracle as O
    from xlxmert_amd.trainer import synthetic_batch
    keys = ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size", "max_position_embeddings",
            "type_vocab_size", "l_layers", "x_layers", "r_layers", "visual_feat_dim", "visual_pos_dim", "num_clusters")
    oc = O.OracleConfig(**{k: getattr(cfg, k) for k in keys})
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = O.make_state_dict(oc, 9595, perturb=False)
    leaf = {k: v.clone().requires_grad_(v.is_floating_point() and k != "vis_emb.weight") for k, v in sd.items()}
    leaf["obj_predict_head.out_cluster.weight"] = leaf["vis_emb.weight"]
    params = [v for k, v in leaf.items() if v.requires_grad and k != "obj_predict_head.out_cluster.weight"]
    m = [torch.zeros_like(p) for p in params]
    v2 = [torch.zeros_like(p) for p in params]
    batch = synthetic_batch(cfg, bs, 20, 8, seed=9595)

    def one(t):
        nonlocal batch
        for p in params:
            p.grad = None
        out = O.xlxmert_vis_mask_forward(leaf, oc, batch["input_ids"], batch["visual_pos"], batch["attention_mask"],
                                         batch["cluster_ids"], batch["vis_mask"], batch["obj_labels"])
        out["total_loss"].backward()
        live = [i for i, p in enumerate(params) if p.grad is not None]
        _, clipped = O.clip_grad_norm([params[i].grad for i in live], 1.0)
        with torch.no_grad():
            for i, g in zip(live, clipped):
                p, m[i], v2[i] = O.adamw_update(params[i].data, g, m[i], v2[i], t, 1e-4)
                params[i].data.copy_(p)

    one(1)                      # warm-up
    times = []
    t0 = time.time()
    while len(times) < 5 and (time.time() - t0 < budget_s or not times):
        s = time.time()
        one(len(times) + 2)
        times.append(time.time() - s)
    times.sort()
    med = times[len(times) // 2]
    res = {"value": round(bs / med, 3), "unit": "examples/s", "cores": cores, "kind": "port",
           "sample": f"{len(times)} full steps (fwd+bwd+clip+AdamW, fp32 torch-CPU oracle) at bs={bs}, median {med:.2f} s/step"}
    if big_bs and time.time() - t0 < 2.5 * budget_s:      # SURVEY 8d also asks for the metric's own batch size: ONE step
        batch = synthetic_batch(cfg, big_bs, 20, 8, seed=9595)
        s = time.time()
        one(len(times) + 2)
        dt = time.time() - s
        res["value_bs%d" % big_bs] = round(big_bs / dt, 3)
        res["sample"] += f"; 1 step at bs={big_bs}: {dt:.1f} s"
    return res


class GemmTimer:
    """Wraps HipOps.gemm / gemm_wgrad_group with HIP events on the launch stream (torch's current stream) for ONE instrumented step."""

    def __init__(self, ops):
        self.ops, self.orig, self.orig_group, self.orig_pair, self.rec = ops, ops.gemm, ops.gemm_wgrad_group, ops.gemm_pair, []

    def __enter__(self):
        def timed(A, B, C, bias, residual, aux, M, N, K, *a, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self.orig(A, B, C, bias, residual, aux, M, N, K, *a, **kw)
            e.record()
            eb = 2 if A.dtype == torch.bfloat16 else 4
            nbytes = eb * (M * K + N * K) + C.element_size() * M * N + (eb * M * N if residual is not None else 0) + \
                (eb * M * N if aux is not None else 0)
            self.rec.append((s, e, 2.0 * M * N * K, (M, N, K, kw.get("a_kmajor", 1), kw.get("b_kmajor", 1), kw.get("epilogue", 0)),
                             nbytes, getattr(self.ops, "block", "")))
        def timed_group(problems, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self.orig_group(problems, **kw)
            e.record()
            eb = problems[0][0].element_size()
            nbytes = sum(eb * pr[5] * (pr[3] + pr[4]) + 2 * 4 * pr[3] * pr[4] for pr in problems)      # operands + fp32 read-modify-write
            self.rec.append((s, e, sum(2.0 * pr[3] * pr[4] * pr[5] for pr in problems),
                             ("group", len(problems), problems[0][5], sum(pr[3] * pr[4] for pr in problems) // 65536, 0, 0),
                             nbytes, getattr(self.ops, "block", "")))
        def timed_pair(c0, c1):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self.orig_pair(c0, c1)
            e.record()
            fl, nb, Ms = 0.0, 0, []
            for c in (c0, c1):
                A, B, C, bias, residual, aux, M, N, K = c.a[:9]
                eb = 2 if A.dtype == torch.bfloat16 else 4
                fl += 2.0 * M * N * K
                nb += eb * (M * K + N * K) + C.element_size() * M * N + (eb * M * N if residual is not None else 0) + \
                    (eb * M * N if aux is not None else 0)
                Ms.append(M)
            kw = c0.kw
            self.rec.append((s, e, fl, ("pair", Ms[0], Ms[1], c0.a[7], c0.a[8], kw.get("b_kmajor", 1) * 10 + kw.get("epilogue", 0)), nb,
                             getattr(self.ops, "block", "")))
        self.ops.gemm = timed
        self.ops.gemm_wgrad_group = timed_group
        self.ops.gemm_pair = timed_pair
        return self

    def __exit__(self, *exc):
        self.ops.gemm = self.orig
        self.ops.gemm_wgrad_group = self.orig_group
        self.ops.gemm_pair = self.orig_pair
        torch.cuda.synchronize()
        self.total_ms = sum(r[0].elapsed_time(r[1]) for r in self.rec)
        self.flops = sum(r[2] for r in self.rec)
        self.bytes = sum(r[4] for r in self.rec)
        self.launches = len(self.rec)

    def by_block(self, peak_tflops):
        """GEMM time / FLOPs per group of model blocks (labels set by the engine): language stack, visual stack, the five
        cross-modality layers (their Q/KV/out projections + the self-attention and FFN sub-blocks = row A7), head."""
        groups = {"cross_modality_layers": lambda t: t.startswith("x"), "cross_attention_blocks_only": lambda t: t.startswith("x") and t[1:].isdigit(),
                  "language_layers": lambda t: t.startswith("l"), "visual_layers": lambda t: t.startswith("r"),
                  "codebook_head_and_feature_encoder": lambda t: t in ("head", "visn_fc")}
        out = {}
        for name, pred in groups.items():
            rs = [r for r in self.rec if pred(r[5])]
            if rs:
                ms = sum(r[0].elapsed_time(r[1]) for r in rs)
                tf = sum(r[2] for r in rs) / (ms * 1e-3) / 1e12
                out[name] = {"launches": len(rs), "gemm_ms": round(ms, 3), "achieved": round(tf, 1), "frac": round(tf / peak_tflops, 4)}
        return out


def other_workloads(cfg, dev, steps=20, warm=3):
    """The SURVEY 8f rows next to the hot path, timed in the same run (one GPU, bf16, dropout on, synthetic data; after the
    headline measurement, never part of `value`): VQA fine-tune step at BASELINE config 3's per-GPU batch, NLVR2 step,
    word_mask / matched pretraining steps, T=4 Mask-Predict sampling (config 4).  Each entry: ms per step and rate."""
    from xlxmert_amd.engine import Engine
    from xlxmert_amd.ops import HipOps
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import PretrainStep, init_reference_weights, random_word_batch, synthetic_batch, word_rows_of

    def timed(fn):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / steps

    out = {}
    g = torch.Generator().manual_seed(4242)
    cents = torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu()

    def feats(*shape):
        return torch.randn(*shape, cfg.visual_feat_dim, generator=g).relu().to(dev)

    # N1: VQA / GQA fine-tune step (3129 answers, real 2048-d grid features in; config 3 = bs 512 over 4 GPUs -> 128 per GPU,
    # and the full 512 on one GPU)
    for B in (128, 512):
        tr = PretrainStep(cfg, B, 20, 64, device=dev, task="vqa", num_answers=3129, train_dropout=True, total_steps=1000, overlap_optimizer=True)
        b = {k: v.to(dev) for k, v in synthetic_batch(cfg, B, 20, 8, seed=7).items()}
        tgt = torch.zeros(B, 3129)
        tgt[torch.arange(B), torch.randint(0, 3129, (B,), generator=g)] = 1.0
        batch = {"input_ids": b["input_ids"], "visual_pos": b["visual_pos"], "visual_feats": feats(B, 64), "targets": tgt.to(dev)}
        dt = timed(lambda: tr.step(batch))
        out[f"vqa_step_bs{B}"] = {"ms": round(dt * 1e3, 2), "examples_per_s": round(B / dt, 1)}
        del tr, batch
    # N1: NLVR2 step (128 statements x 2 images = 256 encoder rows, ref tasks/nlvr2_model.py:50-86)
    P = 128
    tr = PretrainStep(cfg, 2 * P, 20, 64, device=dev, task="nlvr2", train_dropout=True, total_steps=1000, overlap_optimizer=True)
    b = synthetic_batch(cfg, P, 20, 8, seed=8)
    batch = {"input_ids": b["input_ids"].repeat_interleave(2, 0).to(dev), "visual_pos": b["visual_pos"][:, None].expand(-1, 2, -1, -1).contiguous().to(dev),
             "visual_feats": feats(P, 2, 64), "labels": torch.randint(0, 2, (P,), generator=g).to(dev)}
    dt = timed(lambda: tr.step(batch))
    out["nlvr2_step_128_statements"] = {"ms": round(dt * 1e3, 2), "statements_per_s": round(P / dt, 1)}
    del tr, batch
    # N3: language pretraining branches (30522-way tied decoder on the labelled rows / matched head), bs 256
    B = 256
    for task in ("word_mask", "matched"):
        tr = PretrainStep(cfg, B, 20, 64, device=dev, task=task, train_dropout=True, total_steps=1000, overlap_optimizer=True)
        tr.set_centroids(cents)
        b = synthetic_batch(cfg, B, 20, 8, seed=9)
        ids, wl = random_word_batch(b["input_ids"], generator=g)
        batch = {"input_ids": ids.to(dev), "visual_pos": b["visual_pos"].to(dev), "cluster_ids": b["cluster_ids"].to(dev),
                 "word_labels": wl.to(dev), "matched_labels": torch.randint(0, 2, (B,), generator=g).to(dev), "word_rows": word_rows_of(wl)}
        dt = timed(lambda: tr.step(batch))
        out[f"{task}_step_bs{B}"] = {"ms": round(dt * 1e3, 2), "examples_per_s": round(B / dt, 1)}
        del tr, batch
    # N2: Mask-Predict sampling, T = 4 refinement steps over the 8x8 grid, codes for the frozen GAN decoder (config 4)
    for B in (64, 256):
        store = ParamStore(cfg, dev, torch.bfloat16, task="vis_mask")
        init_reference_weights(store, 1)
        store.set_centroids(cents)
        eng = Engine(cfg, store, HipOps(torch.bfloat16), B, 20, 64, need_lang=False)
        eng.sync_compute_weights()
        b = {k: v.to(dev) for k, v in synthetic_batch(cfg, B, 20, 8, seed=10).items()}
        eng.set_inputs(b["input_ids"], b["attention_mask"], None, b["visual_pos"],
                       cluster_ids=torch.zeros(B, 64, dtype=torch.long, device=dev), vis_mask=torch.ones(B, 64, dtype=torch.bool, device=dev))
        dt = timed(lambda: eng.sample_codes_nar(4))
        out[f"sampler_T4_bs{B}"] = {"ms": round(dt * 1e3, 2), "images_per_s": round(B / dt, 1)}
        if B == 64:         # the autoregressive variant (ref tasks/imggen_model.py:49-153): 64 steps, one grid position per step
            for _ in range(2):
                eng.sample_codes_ar(mode="confidence")
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                eng.sample_codes_ar(mode="confidence")
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 5
            out["sampler_AR64_bs64"] = {"ms": round(dt * 1e3, 2), "images_per_s": round(B / dt, 1)}
        del eng, store
    return out



def exchange_sweep(cfg, B, local, rank, world, args, host_batches, headline):
    """N > 1: the other gradient-exchange variants, 10 timed steps each after their own warm-up, in the SAME launch as the headline
    (a multi-GPU node is the driver's to lease: one run has to tell the whole story -- VERDICT r05 item 4).  Every variant builds
    its own trainer and its own RCCL communicator; a variant that fails is reported with its error and does not touch the headline."""
    import torch.distributed as dist
    from xlxmert_amd.trainer import PretrainStep
    out = {}
    variants = [("allreduce_fp32", "allreduce", torch.float32, None), ("allreduce_bf16_buckets", "allreduce", torch.bfloat16, None),
                ("rs+ag_fp32_gather", "rs+ag", torch.float32, "fp32"), ("rs+ag_bf16_gather", "rs+ag", torch.float32, "bf16")]
    for name, coll, gdt, gather in variants:
        if name == headline:
            continue
        tr = None
        try:
            tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device=f"cuda:{local}", seed=9595, total_steps=1000, train_dropout=not args.no_dropout,
                              plan=not args.eager, drop_grads=True, overlap_optimizer=not args.no_opt_overlap, collective=coll, gather=gather,
                              grad_comm_dtype=gdt)
            g = torch.Generator().manual_seed(9595)
            tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
            dev = [{k: v.to(f"cuda:{local}") for k, v in b.items()} for b in host_batches]
            for i in range(12):
                tr.step(dev[i % 4])
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(10):
                tr.step(dev[i % 4])
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            dt = torch.tensor([time.perf_counter() - t0], device="cuda")
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            out[name] = {"ms_per_step": round(dt.item() / 10 * 1e3, 3), "examples_per_s": round(B * world * 10 / dt.item(), 1),
                         "exposed_comm_ms_per_step": round(tr.exposed_comm(), 3),
                         "bytes_per_step": int(tr.store.n_used * (2 if tr.comm_buf is not None else 4)),
                         "issued_by": "xl_comm" if tr.xl_comm is not None else "torch.distributed",
                         "resident_inputs": True, "steps": 10, "warmup": 12}
        except Exception as e:                      # (the headline has been measured already; say what happened and go on)
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            try:
                if tr is not None:
                    tr.close()
            except Exception:
                pass
            del tr
            torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--prewarm-steps", type=int, default=32, help="free-running untimed training steps before the W warm-up steps (runtime pools grow to their high-water mark)")
    ap.add_argument("--steps", type=int, default=100)          # >= 100 timed steps after >= 20 warm-up steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (reference --batchSize, param.py:70)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the SURVEY 8f workloads (VQA / NLVR2 / word_mask / matched / sampler)")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--no-dropout", action="store_true", help="eval-parity mode (the reference trains with p=0.1)")
    ap.add_argument("--single-stream", action="store_true", help="no language/visual stream overlap (profiling)")
    ap.add_argument("--gemm-table", action="store_true", help="print the instrumented step's GEMM time by shape (stderr)")
    ap.add_argument("--gemm-list", default=None, help="write the instrumented step's GEMM launches in launch order (shape key, algorithmic "
                                                      "bytes, block, us) as JSON: joined with per-dispatch PMC counters by tools/gemm_overfetch.py")
    ap.add_argument("--no-opt-overlap", action="store_true", help="AdamW on the main stream, in front of the next forward (default: behind the step on a side stream)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo: test rig for N ranks SHARING one GPU, "
                    "which RCCL refuses -- with XL_BENCH_SHARE_GPU=1 every rank uses cuda:0)")
    ap.add_argument("--eager", action="store_true", help="enqueue every step from Python instead of replaying the recorded launch plan")
    ap.add_argument("--collective", default=None, choices=["allreduce", "rs+ag"], help="gradient exchange at N > 1: all-reduce + "
                    "replicated AdamW (default) or reduce-scatter -> shard-local AdamW -> all-gather (trainer.PretrainStep collective)")
    ap.add_argument("--gather", default=None, choices=["fp32", "bf16"], help="rs+ag: what the all-gather moves -- the fp32 master slices "
                    "(default) or the bf16 compute copy + a sparse fp32 side car (half the bytes; trainer.PretrainStep gather)")
    ap.add_argument("--no-exchange-sweep", action="store_true", help="N > 1: skip the 10-step timings of the other gradient-exchange variants")
    ap.add_argument("--resident-inputs", action="store_true", help="minibatches resident in HBM before the timed region (default: "
                    "pinned host memory, uploaded inside the timed step on a copy stream, one step ahead)")
    args = ap.parse_args()
    if os.environ.get("XL_BENCH_FAULT_TIMEOUT"):        # debugging aid: dump every thread's stack and exit if the run takes longer
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["XL_BENCH_FAULT_TIMEOUT"]), exit=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain script: launch the N ranks ourselves (one process per GPU through torch.distributed.run, the
        # same launcher the driver uses) and pass rank 0's single JSON line through on the inherited stdout
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        raise SystemExit(subprocess.call(cmd))
    # stdout carries exactly ONE line, the result: everything else that writes to file descriptor 1 during the run (RCCL's
    # version banner at communicator creation, library warnings) is sent to stderr, the JSON goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch_issues = os.environ.get("XL_COMM", "rccl") == "torch" or args.backend != "nccl"
    if (world > 1 or os.environ.get("XL_FORCE_EXCHANGE", "0") == "1") and torch_issues:
        # torch.distributed's RCCL stream needs a hardware queue of its own: on HIP's default 4 it shares one with a compute
        # stream, whose kernels then queue behind the collective's barrier packets (22.5 ms per step against 21.3 with 8
        # queues, measured with the exchange on a one-rank group).  The library's own binding (xl_comm_*, the default with the
        # nccl backend) runs the collectives on a stream that owns one of the four default queues already, and more queues
        # than streams that need them only cost (every further active queue: +3 ms per step measured): default 4 there.
        # Read at HIP init.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share_gpu = os.environ.get("XL_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    from xlxmert_amd.engine import reserve_streams
    torch.zeros(8, device=f"cuda:{local}").add_(1.0)          # main stream first, then the engine's three: one hardware queue each,
    grouped = world > 1 or os.environ.get("XL_FORCE_EXCHANGE", "0") == "1"    # (one-rank group: exercises the exchange on 1 GPU)
    reserve_streams(f"cuda:{local}", comm=grouped and os.environ.get("XL_COMM_STREAM") == "own")   # before RCCL's stream can take one
                                                                                                   # of the four (engine.reserve_streams)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        if os.environ.get("XL_GROUP_ONLY") == "1":          # (diagnostic: group initialised, exchange not used)
            os.environ["XL_FORCE_EXCHANGE"] = "0"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    cfg = XLxmertConfig()
    B = args.batch
    tr = PretrainStep(cfg, B, 20, 64, dtype=torch.bfloat16, device=f"cuda:{local}", seed=9595,
                      total_steps=max(1000, args.steps + args.warmup), train_dropout=not args.no_dropout,
                      bucket_mb=float(os.environ.get("XL_BUCKET_MB", "64")),
                      plan=(not args.eager and not args.single_stream), drop_grads=True,
                      overlap_optimizer=not (args.no_opt_overlap or args.single_stream), collective=args.collective, gather=args.gather)
    if args.single_stream:
        tr.engine.side = None
    g = torch.Generator().manual_seed(9595)
    tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
    from xlxmert_amd.trainer import BatchUploader
    host = [synthetic_batch(cfg, B, 20, 8, seed=9595 + 17 * rank + i) for i in range(4)]     # per-rank disjoint synthetic minibatches
    if args.resident_inputs:
        batches = [{k: v.cuda() for k, v in b.items()} for b in host]
        up, nxt = None, None
        get = lambda i: batches[i % 4]
    else:
        # SURVEY 8d counts the upload of the batch: minibatches wait in pinned host memory and every step's batch is copied to
        # the device INSIDE the timed region (copy stream, one step ahead of the compute, trainer.BatchUploader)
        batches = [BatchUploader.pin(b) for b in host]
        up = BatchUploader(f"cuda:{local}")
        get = lambda i: up.upload(batches[i % 4])

    def sync():
        # device first: the step's collectives run on the library's own RCCL communicator, the barrier on torch's -- two communicators
        # must not have kernels in flight at the same time (no order between them: a known deadlock hazard)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    nxt = get(0)
    # Runtime pre-warm (reported as config.prewarm).  The host queues a step in ~2 ms and the GPU runs it in ~16.5, so a loop of steps
    # lets the host run ahead until the queues' back-pressure stops it -- about 10 steps deep.  The FIRST time it gets that deep the
    # runtime grows its pools (completion signals, kernel-argument segments, event records) under the running GPU, and the steps in
    # flight lose 0.7-2 ms each: --steps 20 --warmup 5 put that transient inside the timed region (steps 3, 5, 6 and 8 of every run at
    # 17.1-18.4 ms, the rest at 16.3-16.5: mean 16.68 against a median of 16.46; profiles/r06b/bench_series.txt), --warmup 25 did not
    # (the pools had grown during the warm-up).  A pre-warm paced by synchronize() does NOT remove it (measured: the host never gets
    # deep), so these are free-running steps, the same training step on the same batches, untimed like the W warm-up steps that
    # follow them; --prewarm-steps 0 switches them off.
    n_pre = 0
    for _ in range(max(0, args.prewarm_steps)):
        cur, nxt = nxt, get(n_pre + 1)
        tr.step(cur)
        n_pre += 1
    for i in range(args.warmup):
        cur, nxt = nxt, get(n_pre + i + 1)
        tr.step(cur)
    sync()
    # one timing event per step boundary on the main stream (a marker packet, no wait): the spread of the step time inside this
    # run (ms_per_step_p10 / p50 / p90); `value` and ms_per_step come from the wall clock around the whole region
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        cur, nxt = nxt, get(n_pre + args.warmup + i + 1)
        losses = tr.step(cur)
        marks[i + 1].record()
    t_enqueue = time.perf_counter() - t0          # host time to queue the work (GPU-bound if << wall time)
    sync()
    dt = time.perf_counter() - t0
    series = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    per_step = sorted(series)
    pct = lambda q: round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 3)
    tmax = torch.tensor([dt], device="cuda")
    if world > 1:
        torch.cuda.synchronize()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    loss_val = losses[0].item()

    # one extra, instrumented step (not in the timed region).  It runs single-stream so that every GEMM launch is
    # timed alone (in the timed region language-stream kernels overlap visual-stream kernels on a second stream,
    # which lengthens individual kernels while shortening the step).
    # host time to enqueue ONE step on an empty queue (the timed loop above runs the host into the runtime's queue
    # back-pressure: its per-step enqueue time says how far ahead of the GPU the host got, not what a step costs it)
    host_ms = []
    for i in range(6):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        tr.step(get(i))
        host_ms.append((time.perf_counter() - h0) * 1e3)
    torch.cuda.synchronize()
    host_ms = sorted(host_ms)[len(host_ms) // 2]
    side, tr.engine.side = tr.engine.side, None
    planned, tr.plan_mode = tr.plan_mode, False
    with GemmTimer(tr.ops) as gt:
        tr.step(get(0))
    tr.engine.side, tr.plan_mode = side, planned
    # what the event pair itself adds to every timed launch: the same start / end markers around a one-wave kernel (a 64-element cast);
    # reported next to the roofline figures, never subtracted from them
    probe, probe_o = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda", dtype=torch.bfloat16)
    pairs = []
    if hasattr(torch.cuda, "_sleep"):
        torch.cuda._sleep(40000000)           # (the stream stays busy while the 200 triplets are queued: no host latency in the intervals)
    for _ in range(200):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record(); tr.ops.cast_from_f32(probe, probe_o, 64); e_.record()
        pairs.append((s_, e_))
    torch.cuda.synchronize()
    ev_floor = sorted(a.elapsed_time(b) for a, b in pairs)[100] * 1e3
    traffic, traffic_src = None, None
    try:        # HBM bytes per GEMM launch from the committed PMC profile of this build (cannot be collected live)
        tj = json.load(open(os.path.join(ROOT, "profiles", "gemm_traffic.json")))
        traffic, traffic_src = round(tj["bytes_per_launch"]), tj.get("source")
    except Exception:
        pass
    if rank == 0 and args.gemm_list:
        json.dump([{"key": list(key), "bytes": _b, "block": _t, "us": s_.elapsed_time(e_) * 1e3, "flops": f_}
                   for s_, e_, f_, key, _b, _t in gt.rec], open(args.gemm_list, "w"))
    if rank == 0 and args.gemm_table:
        agg = {}
        for s_, e_, f_, key, _b, _t in gt.rec:
            a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += s_.elapsed_time(e_); a[2] += f_
        for key, (c, t, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"  {str(key):44s} x{c:3d} {t:8.3f} ms  {t / c * 1e3:8.1f} us  {f / t / 1e9:7.1f} TF/s", file=sys.stderr)
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = B * world * args.steps / dt
        plans = list(tr._plans.values())
        gemm_tflops = gt.flops / (gt.total_ms * 1e-3) / 1e12
        out = {
            "metric": "pretrain examples/sec (20 text tok x 64 vis tok, bs=256)", "value": round(value, 1),
            "unit": "examples/s", "n_gpus": 1 if share_gpu else world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "ms_per_step_p10": pct(0.10), "ms_per_step_p50": pct(0.50), "ms_per_step_p90": pct(0.90),
            # SURVEY 8d defines the metric on the MEDIAN step; `value` is the contract's wall-clock mean over the K timed steps
            # (barrier + synchronize on both sides), which the first few steps after the warm-up drag down: both are reported
            "value_p50": round(B * world / (pct(0.50) * 1e-3), 1),
            "ms_per_step_series": [round(x, 2) for x in series[:32]],          # in issue order: what the first steps after the sync() cost
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]+[2]: full X-LXMERT encoder 9L/5R/5X d=768 + obj_predict_head over 10k codebook, "
                                   "masked-visual-token step fwd+bwd+clip+AdamW", "per_gpu_batch": B, "global_batch": B * world,
                       "text_len": 20, "visual_tokens": 64, "parallelism": f"dp{world}",
                       "prewarm": (f"{n_pre} free-running untimed training steps before the {args.warmup} warm-up steps: the first time the host "
                                   "runs ~10 steps ahead of the GPU the runtime grows its pools under the running kernels (DESIGN.md "
                                   "section 6, round 6)") if n_pre else "none",
                       "value_definition": "value = global_batch x steps / wall time of the timed region (mean step, max over ranks); "
                                           "value_p50 = global_batch / median step (timing events at the step boundaries of rank 0) -- "
                                           "the quantity SURVEY 8d's 'median' refers to",
                       "dropout": "off (eval-parity mode)" if args.no_dropout else "0.1 hidden + 0.1 attention (training mode, 94 sites)",
                       "loss": round(loss_val, 4),
                       "step_launch": (f"launch plan: the step's recorded C-ABI calls replayed by xl_plan_run ({len(plans)} masked-row "
                                       f"geometries, {max(p.n_calls for p in plans)} calls in {max(p.n_segments for p in plans)} segment(s) each"
                                       + (f", {max(p.n_host_ops for p in plans)} host operations = the gradient exchange's all-reduce issue "
                                          "points + one wait, between segments" if max(p.n_host_ops for p in plans) else "") + ")"
                                       if tr.plan_mode and plans else "eager (every launch enqueued from Python)"),
                       "visual_losses": "obj,feat" if tr.feat_loss else "obj (scripts/pretrain.bash:15)",
                       "optimizer_pass": ("AdamW behind the step on a side stream, group by group in forward order; the next step's forward "
                                          "waits per group (all of it inside the timed region)") if tr.opt_stream is not None
                                         else "AdamW in front of the next forward",
                       "inputs": ("4 synthetic minibatches per rank resident in HBM before the timed region: no H2D inside it (--resident-inputs)"
                                  if args.resident_inputs else
                                  "4 synthetic minibatches per rank in pinned host memory; every step's batch (~0.6 MB of ids / masks / labels / "
                                  "row list) is uploaded INSIDE the timed region, on a copy stream one step ahead of the compute (SURVEY 8d)"),
                       "vis_mask": "--vis_mask_predict masks, n ~ U{1..64} per image (ref lxmert_data.py:414-419)",
                       "head_rows": "codebook head + both losses on the masked rows only (exact: the reference's losses read "
                                    "nothing else); all rows with XL_COMPACT_HEAD=0"},
            "host_enqueue_ms_per_step": round(host_ms, 3),
            "host_enqueue_note": "median host time of tr.step() on an empty queue (6 steps after the timed region, device "
                                 "synchronised before each); inside the timed loop the host ran "
                                 f"{round(t_enqueue / args.steps * 1e3, 2)} ms per step, i.e. up to the runtime's queue back-pressure",
            # whole-step MFMA utilisation from the dense-contraction FLOPs actually EXECUTED in a step (the head runs on the
            # masked rows only, so this is below the contract figure 50.782 GFLOP/example x batch)
            "step_mfma_frac": round(gt.flops / (ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12), 4),
            "roofline": {"bound": "mfma", "kernel": "gemm_bf16_pp_kernel + gemm_bf16_mfma_kernel (all dense contractions of one step, grouped weight gradients included)",
                         "achieved": round(gemm_tflops, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4),
                         # the WHOLE STEP's executed dense-contraction FLOPs over the step time and the peak (= step_mfma_frac): the
                         # contract figure 50.782 GFLOP/example credits head rows the step does not compute (masked-row head), so
                         # value x 50.782 GFLOP / peak would overstate it -- this one takes no such credit
                         "frac_contract_uncredited": round(gt.flops / (ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12), 4),
                         # informational, NOT the contract's peak: what a register-only bf16 MFMA loop sustains on all 256 CUs with
                         # non-trivial operand bits (the 2.5 PFLOP/s figure is reached on zero operands only: the chip clocks to its
                         # power limit) -- tools/probes/clock_probe.hip, profiles/r05c/clock_probe.txt
                         "peak_real_data_measured": {"value": 1850.0, "unit": "TFLOP/s", "frac": round(gemm_tflops / 1850.0, 4),
                                                     "source": "profiles/r05c/clock_probe.txt"},
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(gt.bytes / gt.launches),
                         "algorithmic_bytes_per_step": int(gt.bytes),
                         "blocks": gt.by_block(PEAK_BF16_TFLOPS),
                         "launches_per_step": gt.launches, "avg_launch_us": round(gt.total_ms * 1e3 / gt.launches, 2),
                         "gemm_ms_per_step": round(gt.total_ms, 3),
                         "event_pair_floor_us": round(ev_floor, 2),     # start/end markers around a one-wave kernel: part of every avg_launch_us

                         "algorithmic_gflop_per_step": round(gt.flops / 1e9, 1),
                         "contract_gflop_per_step": round(GFLOP_PER_EXAMPLE * B, 1)},
        }
        if share_gpu:
            # N ranks on ONE device (test rig for the exchange with a real second rank; RCCL refuses it, gloo does not): this is
            # NOT a scaling result -- every rank competes for the same GPU
            out["shared_gpu"] = True
            out["n_ranks"] = world
            out["config"]["parallelism"] = f"dp{world} ranks sharing ONE GPU (exchange test rig, not a scaling measurement)"
        if world > 1 or grouped:
            out["config"]["gradient_exchange"] = {
                "backend": args.backend + (" (= RCCL over xGMI)" if args.backend == "nccl" else " (host-staged: test rig)"),
                "collective": tr.collective,
                "all_gather_payload": ("bf16 compute copy + sparse fp32 side car" if tr.gather_bf16 else "fp32 master slices") if tr.sharded else None,
                "collective_note": ("reduce-scatter (sum) per finished slice of the flat gradient buffer -> shard-local norm + one scalar "
                                    "all-reduce -> AdamW over this rank's shards -> all-gather of the fp32 master slices, first-needed "
                                    "first, overlapping the next forward" if tr.sharded else
                                    "all-reduce (sum; 1/N folded into AdamW) per finished slice of the flat gradient buffer; every rank "
                                    "runs the whole AdamW pass"),
                "optimizer_elements_per_rank": int(sum(b - a for a, b in tr.owned_ranges())),
                "optimizer_elements_total": int(tr.store.n_used),
                "element_type": "bf16" if tr.comm_buf is not None else "fp32",
                "bucket_mb": round(tr.bucket_elems * (2 if tr.comm_buf is not None else 4) / (1 << 20), 1),
                "bytes_per_step": int(tr.store.n_used * (2 if tr.comm_buf is not None else 4)),
                "issued_by": ("xl_comm_* (the library's own RCCL binding: the collectives are entries of the launch plan, XL_COMM=rccl)"
                              if tr.xl_comm is not None else "torch.distributed (host operations between the segments of the launch plan)"),
                "exposed_comm_ms_per_step": round(tr.exposed_comm(), 3),
                "rccl_nranks": int(tr.comm_nranks()) if hasattr(tr, "comm_nranks") else world}
            out["config"]["gradient_exchange"]["default_rationale"] = (
                "fp32 all-reduce = what DDP moves for the reference's fp32 parameters (lxmert_pretrain.py:694-700); predicted step time per "
                "variant and N in DESIGN.md section 7 -- the `exchanges` key of this line holds the other variants measured in this run")
    if world > 1 and not args.no_exchange_sweep and (not share_gpu or os.environ.get("XL_BENCH_SWEEP_SHARED") == "1"):
        # every rank takes part (collectives); rank 0 reports
        hn = ("rs+ag_" + ("bf16" if tr.gather_bf16 else "fp32") + "_gather") if tr.sharded else \
             ("allreduce_bf16_buckets" if tr.comm_buf is not None else "allreduce_fp32")
        torch.cuda.synchronize(); dist.barrier()
        tr.close()
        # The headline is measured; a sweep that hangs (a variant's collective waiting for a rank that failed) must not lose it:
        # past the deadline rank 0 prints the line without the sweep and every rank leaves.
        import threading
        deadline = float(os.environ.get("XL_BENCH_SWEEP_TIMEOUT", "420"))
        swept, once = threading.Event(), threading.Lock()

        def give_up():
            if swept.wait(deadline) or not once.acquire(blocking=False):
                return
            if rank == 0:
                late = dict(out)
                late["exchanges"] = {"headline": hn, "headline_ms_per_step": out["ms_per_step"],
                                     "error": f"sweep not finished after {deadline:.0f} s; headline unaffected"}
                os.write(result_fd, (json.dumps(late) + "\n").encode())
            os._exit(0)
        threading.Thread(target=give_up, daemon=True).start()
        sweep = exchange_sweep(cfg, B, local, rank, world, args, host, hn)
        swept.set()
        if not once.acquire(blocking=False):
            time.sleep(3600)              # (the guard is printing and leaving)
        if rank == 0:
            out["exchanges"] = {"headline": hn, "headline_ms_per_step": out["ms_per_step"], **sweep}
    if rank == 0:
        if not args.no_extra and world == 1 and not args.single_stream:
            del tr, batches
            torch.cuda.empty_cache()
            out["other_workloads"] = other_workloads(cfg, f"cuda:{local}")
        if not args.no_cpu_baseline and world == 1:           # reported at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_batch)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if grouped:
        torch.cuda.synchronize()          # (the library's collectives are done before torch's communicator runs: see sync())
        dist.barrier()
        try:
            tr.close()                    # xl_comm_destroy, while the process group that carried its id is still up
        except NameError:
            pass                          # (rank 0 at N = 1 with the extra workloads: the trainer is gone already)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
