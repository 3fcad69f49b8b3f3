"""Masked-visual-token pretraining step on one GPU per process (data parallel over RCCL).

Reproduces the INTENDED step of the reference trainer (ref x-lxmert/src/pretrain/lxmert_pretrain.py:143-225,
295-366): labels `obj_labels[~vis_mask] = -100`, `attention_mask = word_id > 0`,
one forward, one backward, gradient all-reduce (DDP mean), `clip_grad_norm_(1.0)`, transformers==4.1.1 AdamW
(betas .9/.999, eps 1e-6, bias correction, decoupled decay on tensors whose name contains neither "bias" nor
"LayerNorm.weight"), linear warm-up then linear decay, gradients dropped after the step.

`weight_decay` (default 0.0 = the 4.1.1 class default) and `warmup_ratio` (default 0.05, what
scripts/pretrain.bash suggests) have no value anywhere in the reference: assumptions.
"""
import contextlib
import math
import os

import torch
import torch.distributed as dist

from .config import XLxmertConfig
from .engine import Engine
from .ops import HipOps
from .params import ParamStore


def init_reference_weights(store, seed):
    """HF `_init_weights`: Linear/Embedding weights ~ N(0, initializer_range), biases 0, LayerNorm (1, 0),
    embedding rows at padding_idx=0 zeroed; mask_feat zeros (ref lxrt/modeling.py:92)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    std = store.cfg.initializer_range
    with torch.no_grad():
        for name, m in store.index.items():
            v = store.view(name)
            if name == "mask_feat" or name.endswith(".bias"):
                v.zero_()
            elif name.endswith("LayerNorm.weight") or name.endswith("layer_norm.weight") or \
                    (len(m.shape) == 1 and name.endswith(".weight")):      # answer_head.logit_fc.2 is a LayerNorm too
                v.fill_(1.0)
            else:
                v.copy_((torch.randn(m.shape, generator=g) * std).to(v.device))
                if "embeddings.weight" in name:
                    v[0].zero_()


def linear_schedule(step, warmup_steps, total_steps):
    """transformers.get_linear_schedule_with_warmup (ref lxmert_pretrain.py:138-139); `step` counts completed updates."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))


class PretrainStep:
    def __init__(self, cfg: XLxmertConfig, batch_size, text_len=20, n_grids=64, dtype=torch.bfloat16, device=None,
                 lr=1e-4, weight_decay=0.0, warmup_ratio=0.05, total_steps=100000, clip_grad_norm=1.0,
                 betas=(0.9, 0.999), eps=1e-6, seed=9595, feat_loss=None, train_dropout=False, store=None,
                 bucket_mb=64, ops=None, task="vis_mask", num_answers=0, visual_losses="obj", grad_comm_dtype=None,
                 plan=None, drop_grads=None, overlap_optimizer=None, collective=None, overwrite_grads=None, gather=None):
        """`ops` is injected only by the CPU test-suite (tests/fake_ops.py); the product always runs HipOps.
        task: "vis_mask" (masked-visual-token pretraining step, ref lxmert_pretrain.py), "word_mask" / "matched" (the
        language pretraining branches) or "vqa" (VQA/GQA fine-tune step on real grid features with `num_answers` answers,
        ref tasks/vqa.py:166-198) -- same clip / AdamW / schedule.  task="all": the three pretraining branches on ONE
        parameter set, `step(batch, task=...)` per call (the reference's round-robin, lxmert_pretrain.py:296-298): tensors
        without a gradient in the step's branch are skipped by the optimizer and keep their own update count.
        visual_losses: the reference's --visualLosses ("obj" in scripts/pretrain.bash:15 = the canonical recipe and the
        default here; "obj,feat" adds the masked SmoothL1 feature regression onto batch["feat_labels"] -- the real grid
        features, ref lxmert_pretrain.py:177-179 -- or, when the batch carries none, onto the centroid of each position's
        cluster id).  feat_loss=True/False overrides it (older call sites).
        grad_comm_dtype: element type of the gradient exchange, torch.float32 (default; what DDP moves for the reference's
        fp32 parameters) or torch.bfloat16 (half the bytes over xGMI: 405 MB instead of 810 MB per step; every finished
        slice is cast into a bf16 bucket on the stream that produced it, summed by RCCL, and cast back into the fp32
        gradient buffer before the norm / AdamW, which keep accumulating in fp32).  Env XL_GRAD_COMM=bf16|fp32 overrides.
        plan: replay the masked-visual-token step (forward, backward, norm, AdamW: ~560 launches and ~150 stream hand-offs
        on four streams) as ONE C call per step -- a launch plan recorded from the step's own C-ABI calls (csrc/plan.hip,
        _lib.LaunchPlan), one per launch geometry -- instead of enqueueing it from Python (13 ms of host time per step).
        Everything a step varies lives in device memory: inputs in the engine's static buffers, the dropout step seed
        (engine.seed_dev), the schedule scalars (xl_schedule_step), the masked-row list padded to the GEMM row tile.
        With a gradient exchange (world > 1) the plan is SEGMENTED: torch.distributed's all-reduce is not an entry point of the
        library, so every issue point of a bucket cuts the record and the replay alternates xl_plan_run(segment) with the
        collective (_lib.SegmentedPlan: ~15 segments per step).  Plans freeze the step's scalars (lr, weight decay, clip, schedule
        length) and the codebook pointer: changing one of them drops the recorded plans.  Default off, env XL_PLAN=1|0 overrides.
        (A hipGraph of the same step was measured and rejected: see csrc/plan.hip.)
        drop_grads: the optimizer pass clears the gradient buffer itself (the reference's optim.zero_grad(),
        lxmert_pretrain.py:241, folded into xl_adamw: 4 bytes per element more in that pass instead of a separate 0.8 GB
        clear before the next backward); store.grad is then zero after step().  Default: on with `plan`, else off (the
        gradients of the last step stay readable).  Not with task="all" (skipped tensors keep their buffers).
        overlap_optimizer: the AdamW pass (HBM-bound: 34 bytes per parameter, no matrix work) runs BEHIND the step on the language
        stream's weight-gradient companion stream (idle until the next backward), group by group in the order the forward reads
        the parameters, and the next step's forward waits group by group (engine.params_ready): the matrix units start the next
        forward ~0.2 ms after the gradient norm instead of after the whole pass (-0.15...-0.4 ms per step measured; a CU-masked
        stream for the pass, hipExtStreamCreateWithCUMask, was measured too: +14 ms per step, whatever the number of hardware
        queues).  Same arithmetic per element.  Parameters / optimizer state read from another stream need `sync()` first.  Default off
        (env XL_OPT_OVERLAP=1|0 overrides); HIP path only.
        collective: how the gradients meet.  "allreduce" (default): every finished slice of the flat gradient buffer is all-reduced
        and every rank runs the whole clip + AdamW pass (what DDP + a replicated optimizer do, ref lxmert_pretrain.py:102-106,
        343-359).  "rs+ag" (SURVEY 5.8): every finished slice is REDUCE-SCATTERED in place (rank r keeps the sum of the r-th
        1/N of the slice), the squared norm is the all-reduced scalar of the shard-local sums, AdamW runs over this rank's
        shards only (1/N of the pass), and the updated fp32 master weights are ALL-GATHERED slice by slice, first-needed
        first, on the collectives' stream -- followed there by the cast into the compute-dtype copy -- while the next forward starts
        (it waits slice by slice: engine.params_ready).  Same bytes on the wire as the all-reduce.  Same parameters as "allreduce"
        up to the summation order of the norm; a rank's Adam moments are current on its shards only (gather_state() makes them
        whole: checkpoints, verify_replicas).
        Env XL_COLLECTIVE=allreduce|rs+ag overrides.
        gather ("fp32" default | "bf16"; env XL_GATHER; collective="rs+ag" with a bf16 compute copy): what the all-gather moves.
        "bf16": every slice's COMPUTE copy (written by the owner's AdamW pass) -- half the bytes -- plus a sparse fp32 side car for
        the ~0.1 % of a slice that is read in fp32 (biases, LayerNorm affines, box_fc, mask_feat: ParamStore.fp32_read_index), made
        whole by one small sum all-reduce per slice (owner's value + zeros).  Same parameters as the fp32 gather, bit for bit;
        the fp32 master copy of the MATRICES is then current on the owning rank only (gather_state() makes it whole: checkpoints,
        verify_replicas).
        overwrite_grads: weight gradients with exactly one contribution per step (every Linear weight but the MLM decoder tied to
        the word embeddings) are STORED by the first backward after an optimizer pass instead of accumulated into a cleared buffer
        (engine.dw_overwrite -> xl_gemm_wgrad_group overwrite_mask): the optimizer pass does not clear them (decay_flags bit 2,
        ParamStore.mark_overwritten) and the weight-gradient epilogue does not read them back -- 8 bytes per parameter and step
        less HBM traffic.  Gradient accumulation (step(update=False)) overwrites with the window's first micro-batch and
        accumulates the others.  Same gradients either way.  Default on; env XL_GRAD_OVERWRITE=0|1 overrides."""
        self.cfg = cfg
        self.task = task
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        # the exchange also runs on a one-rank group when asked to (XL_FORCE_EXCHANGE=1): the collectives are identities then,
        # which is how the RCCL stream plumbing is tested on a single GPU
        self.exchange = self.world > 1 or (dist.is_available() and dist.is_initialized()
                                           and os.environ.get("XL_FORCE_EXCHANGE", "0") == "1")
        self.store = store if store is not None else ParamStore(cfg, self.device, dtype, task=task, num_answers=num_answers)
        self.ops = ops if ops is not None else HipOps(dtype)
        if store is None:
            init_reference_weights(self.store, seed)          # same seed on every rank == DDP's rank-0 broadcast
        self.engine = Engine(cfg, self.store, self.ops, batch_size, text_len, n_grids, need_lang=(task != "vis_mask"),
                             train_dropout=train_dropout)
        self.engine.sync_compute_weights()
        self.store.ensure_adam_state()
        self._plans, self._plan_warm = {}, False
        self.lr, self.wd, self.clip = lr, weight_decay, clip_grad_norm
        self.betas, self.eps = betas, eps
        self.total_steps, self.warmup_steps = total_steps, int(total_steps * warmup_ratio)
        self.feat_loss = ("feat" in visual_losses.split(",")) if feat_loss is None else bool(feat_loss)
        self.t = 0
        # forward passes run so far (every step() call, also the micro-batches of a gradient-accumulation window): the dropout
        # step seed derives from it, so that each forward draws fresh masks as the reference does (tasks/vqa.py update_freq loop);
        # equal to `t` while update_freq = 1.  Checkpoint it next to `t` (state() / load_state()).
        self.micro = 0
        # task round-robin on one parameter set (ref lxmert_pretrain.py:296-298): per-tensor update counts, as transformers'
        # AdamW keeps them (a tensor skipped in a step does not advance its bias correction)
        self.chunk_steps = (torch.zeros(self.store.n_total // 256, dtype=torch.int32, device=self.device)
                            if task == "all" else None)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.sumsq_scratch = (self.ops.sumsq_scratch(self.device) if hasattr(self.ops, "sumsq_scratch")
                              else torch.zeros(1, dtype=torch.float32, device=self.device))     # (xl_sumsq block partials + ticket)
        # update counter, learning rate and bias corrections live on the device and are advanced by a kernel queued in front
        # of AdamW (xl_schedule_step): the host, which runs several steps ahead of the GPU, never writes step scalars into
        # memory a queued step still has to read
        self.lrs = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        env = os.environ.get("XL_GRAD_COMM")
        if env:
            grad_comm_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[env]
        self.grad_comm_dtype = grad_comm_dtype or torch.float32
        self.comm_buf, self._comm_ops = None, self.ops
        if self.exchange and self.grad_comm_dtype == torch.bfloat16:
            self.comm_buf = torch.zeros(self.store.n_used, dtype=torch.bfloat16, device=self.device)
            if isinstance(self.ops, HipOps) and self.ops.dtype != torch.bfloat16:
                self._comm_ops = HipOps(torch.bfloat16)
        self.bucket_elems = max(1, int(bucket_mb * (1 << 20)) // (2 if self.comm_buf is not None else 4))
        # Who issues the collectives.  "rccl" (default with the nccl backend): the library's own RCCL binding (xl_comm_*,
        # csrc/comm.hip) -- the collectives are ordinary entries of the launch plan (one xl_plan_run per step) and run on a stream
        # that already owns a hardware queue.  "torch": torch.distributed (any backend; host operations between plan segments;
        # its NCCL stream is a fifth hardware queue).  XL_COMM overrides; a binding that fails its start-up self-test falls back.
        self.xl_comm = None
        want = os.environ.get("XL_COMM", "rccl")
        if (self.exchange and want == "rccl" and isinstance(self.ops, HipOps) and self.device.type == "cuda"
                and dist.get_backend() == "nccl"):
            try:
                self.xl_comm = self._init_xl_comm()
            except Exception as e:                   # (no librccl.so.1, an RCCL error at init, a wrong sum: keep torch's path)
                import warnings
                warnings.warn(f"xl_comm_* unavailable ({e}); the gradient exchange goes through torch.distributed")
                self.xl_comm = None
        want_c = os.environ.get("XL_COLLECTIVE") or collective or "allreduce"
        assert want_c in ("allreduce", "rs+ag"), want_c
        self.sharded = self.exchange and want_c == "rs+ag"
        self.collective = "rs+ag" if self.sharded else "allreduce"
        want_g = os.environ.get("XL_GATHER") or gather or "fp32"
        assert want_g in ("fp32", "bf16"), want_g
        self.gather_bf16 = self.sharded and want_g == "bf16" and self.store.compute_dtype == torch.bfloat16
        self._vec_idx = {}                     # slice (lo, hi) -> (int32 positions of its fp32-read elements, pack buffer)
        self._segments, self._seg_key, self._seg_events, self._group_seg = [], None, [], {}
        self.exposed_comm_ms = []              # per step: time the main stream waited for collectives after backward
        self._comm_t = None                    # (xl_comm path: one re-recorded event pair = the last step's wait)
        env = os.environ.get("XL_PLAN")
        self.plan_mode = bool(int(env)) if env else bool(plan)
        self.plan_mode = self.plan_mode and isinstance(self.ops, HipOps) and task == "vis_mask"
        env = os.environ.get("XL_GRAD_OVERWRITE")
        self.overwrite_grads = bool(int(env)) if env else (True if overwrite_grads is None else bool(overwrite_grads))
        self._accum_pending = False            # gradient accumulation (step(update=False)): the buffer holds earlier micro-batches
        self.drop_grads = (self.plan_mode if drop_grads is None else bool(drop_grads)) and task != "all"
        env = os.environ.get("XL_OPT_OVERLAP")
        overlap = bool(int(env)) if env else bool(overlap_optimizer)
        self.opt_stream, self._opt_groups = None, None
        if self.sharded:
            # the sharded pass has its own order (slices, first-needed first) and its own hand-over to the next forward (the
            # all-gathers' events): the side-stream pass of the replicated optimizer does not apply
            overlap = False
            if self.xl_comm is not None and self.engine._dw is not None:
                self.engine.params_ready = self._wait_gathered
        if overlap:
            # the pass is issued group by group in forward order; on the HIP path it also moves to the side stream, with an
            # event per group (injected host ops -- the CPU test-suite -- run the same grouped pass in place)
            self._opt_groups = self.store.forward_groups()
            self._opt_last = {k: i for i, (k, _, _) in enumerate(self._opt_groups)}      # a group's last range closes it
            if isinstance(self.ops, HipOps) and self.device.type == "cuda" and self.engine._dw is not None:
                self.opt_stream = self.engine._dw["l"]
                self._opt_events = {k: self.ops.new_event() for k, _, _ in self._opt_groups}
                self.engine.params_ready = self._wait_params
        if self.world > 1:
            self.sync_replicas()

    # Step scalars a recorded launch plan has frozen into its argument words: assigning one drops the plans (they are re-recorded
    # on the next step of each geometry).  Eager steps read them at every call anyway.
    def _hyper(name):
        attr = "_" + name

        def get(self):
            return getattr(self, attr)

        def set_(self, value):
            if getattr(self, attr, None) != value and getattr(self, "_plans", None):
                self._plans.clear()
            setattr(self, attr, value)
        return property(get, set_)

    lr, wd, clip = _hyper("lr"), _hyper("wd"), _hyper("clip")
    total_steps, warmup_steps = _hyper("total_steps"), _hyper("warmup_steps")
    betas, eps = _hyper("betas"), _hyper("eps")
    del _hyper

    def _init_xl_comm(self):
        """one RCCL communicator for this process behind the C ABI: rank 0 draws the 128-byte id, torch.distributed carries it
        to the other ranks (the only use of the process group by this path besides the start-up broadcast)."""
        import ctypes
        lib = self.ops.lib
        idbuf = (ctypes.c_uint8 * 128)()
        if self.rank == 0:
            lib.call("xl_comm_unique_id", ctypes.addressof(idbuf))
        t = torch.tensor(list(idbuf), dtype=torch.uint8, device=self.device)
        dist.broadcast(t, src=0)
        idbuf = (ctypes.c_uint8 * 128)(*t.cpu().tolist())
        # the collectives' stream.  Default: the language stream's weight-gradient companion -- one of the four streams that own a
        # hardware queue already (a FIFTH active queue costs the step 3-6 ms whatever sits in it, measured with the RCCL launch
        # itself skipped), idle most of the step, and never on the visual critical path.  XL_COMM_STREAM=own: a stream of its own.
        from .engine import comm_stream
        if os.environ.get("XL_COMM_STREAM", "dw") == "own" or self.engine._dw is None:
            self._comm_stream = comm_stream(self.device)       # (kept alive: the library holds its raw handle)
        else:
            self._comm_stream = self.engine._dw["l"]
        h = int(lib.raw("xl_comm_init")(ctypes.addressof(idbuf), self.rank, self.world, self._comm_stream.cuda_stream))
        if h <= 0:
            raise RuntimeError("xl_comm_init failed: " + lib.raw("xl_last_error")().decode())
        # self-test: every rank contributes rank + 1; the sum must be world (world + 1) / 2 on every rank
        probe = torch.full((1024,), float(self.rank + 1), dtype=torch.float32, device=self.device)
        self._comm_ops.comm_allreduce(h, probe, probe.numel())
        self._comm_ops.comm_wait(h)
        torch.cuda.synchronize(self.device)
        want = self.world * (self.world + 1) / 2
        if not bool((probe == want).all()):
            raise RuntimeError(f"xl_comm self-test: all-reduce gave {probe[0].item()} instead of {want}")
        return h

    def _wait_params(self, key):
        """engine hook: the current stream is about to read the parameters of group `key` -- wait for the optimizer pass of
        the previous step to have updated them (no-op before the first update: the event has never been recorded)."""
        ev = self._opt_events.get(key)
        if ev is not None:
            self.ops.stream_wait(ev, torch.cuda.current_stream())

    def comm_nranks(self):
        """ranks of the gradient exchange's communicator as RCCL itself reports them (xl_comm_nranks), or the process group's size"""
        if self.xl_comm is not None:
            return int(self.ops.lib.raw("xl_comm_nranks")(int(self.xl_comm)))
        return self.world

    def close(self):
        """give back the library-side communicator (xl_comm_destroy waits for its stream first); the trainer is done"""
        if self.xl_comm is not None:
            self.sync()
            self.ops.lib.raw("xl_comm_destroy")(int(self.xl_comm))
            self.xl_comm = None

    def sync(self):
        """everything this trainer has queued (the optimizer stream included) is done."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def sync_replicas(self):
        """DDP's constructor broadcast (ref lxmert_pretrain.py:102-106): rank 0's parameters AND optimizer state (Adam
        moments, update counters -- a resumed store carries them) go to every rank, unconditionally: a one-time cost."""
        st = self.store
        bufs = [st.master, st.exp_avg, st.exp_avg_sq, self.step_dev]
        if self.chunk_steps is not None:
            bufs.append(self.chunk_steps)
        if st.centroids is not None:
            bufs.append(st.centroids)
        for b in bufs:
            dist.broadcast(b, src=0)
        if st.centroids is not None and st.centroids_c is not st.centroids:
            st.centroids_c.copy_(st.centroids)
        self.t = int(self.step_dev.item())
        self.micro = max(self.micro, self.t)
        self.engine.sync_compute_weights()

    def verify_replicas(self):
        """names of the tensors (parameters, Adam moments) whose per-tensor checksum differs between ranks -- [] when the
        replicas agree.  One float64 sum per tensor and buffer, MIN / MAX all-reduced: cheap enough for every N-th step."""
        self.sync()                              # (an optimizer pass running behind the step: finish it first)
        self.gather_state()                      # (sharded exchange: every rank's fp32 state is current on its shards only)
        st = self.store
        names = [n for n in st.index if st.index[n].offset < st.n_used]
        sums = []
        for buf in (st.master, st.exp_avg, st.exp_avg_sq):
            for n in names:
                m = st.index[n]
                k = 1
                for d in m.shape:
                    k *= d
                sums.append(buf[m.offset:m.offset + k].double().sum())
        v = torch.stack(sums)
        lo, hi = v.clone(), v.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        bad = (lo != hi).nonzero().reshape(-1).tolist()
        kinds = ("param", "exp_avg", "exp_avg_sq")
        return [f"{kinds[i // len(names)]}:{names[i % len(names)]}" for i in bad]

    def set_centroids(self, centroids):
        """new codebook (ref lxrt/modeling.py:140-151 set_visual_embedding).  The store keeps the device buffers it already has
        (same shape: copied in place), and the recorded launch plans -- which hold raw pointers -- are dropped either way."""
        self.store.set_centroids(centroids)
        self._plans.clear()

    def reduce_metrics(self, results, dst=0):
        """ref x-lxmert/src/utils.py:11-39 `reduce_dict`: the per-rank values of an epoch's metric dict (losses, counts: python
        numbers or 0-d tensors) are summed onto rank `dst` with ONE reduce of the stacked, key-sorted vector; returns
        {name: float} on `dst` and None elsewhere (a single process gets its own values back)."""
        names = sorted(results.keys())
        vals = torch.stack([torch.as_tensor(results[k]).detach().float().reshape(()).to(self.device) for k in names])
        if self.world > 1:
            if self.xl_comm is not None:
                # torch's communicator and the library's are two RCCL communicators: a collective of one must not run while
                # collectives of the other are in flight (concurrent communicators without an order between them can deadlock).
                # Every xl_comm collective issued so far is finished before torch's reduce starts.
                self.ops.comm_wait(self.xl_comm)
                torch.cuda.current_stream().synchronize()
            dist.reduce(vals, dst=dst)
            if self.rank != dst:
                return None
        return {k: v for k, v in zip(names, vals.tolist())}

    # ---- gradient exchange (DDP semantics: SUM over ranks here, the 1/world factor is folded into the optimizer kernel).
    # The flat gradient buffer is laid out in backward-completion order, so the engine reports growing finished ranges
    # (one per stream: params.ParamStore.language_range); every time >= bucket_elems new elements of a range are final an
    # asynchronous all-reduce of that contiguous slice is queued (RCCL runs it on its own stream behind an event recorded
    # on the reporting stream), overlapping with the rest of backward.  Every rank queues the same slices in the same
    # order (the order is a function of the model layout only).
    def _begin_exchange(self, replay=False):
        """replay: a recorded plan re-issues the collectives of the step it was recorded from (same slices, same order)"""
        self._works = []
        if not replay:
            self._lanes, self._slices, self._segments = {}, [], []

    def _host_op(self, fn):
        """run `fn` now and, while a launch plan is being recorded, make it a host operation of the plan (replayed between
        two segments: _lib.Lib.record_host)."""
        fn()
        lib = getattr(self.ops, "lib", None)
        if lib is not None:
            lib.record_host(fn)

    def _send(self, lo, hi):
        """the finished slice [lo, hi) of the gradient buffer goes out.  Sharded exchange: its largest prefix that splits into
        `world` equal pieces of whole 256-element optimizer chunks is reduce-scattered, the remainder (< 256 * world elements)
        all-reduced -- a replicated tail every rank updates itself."""
        if self.sharded:
            gran = 256 * self.world
            n = (hi - lo) // gran * gran
            if n:
                self._issue("rs", lo, lo + n)
                self._segments.append(("rs", lo, lo + n))
            if lo + n < hi:
                self._issue("ar", lo + n, hi)
                self._segments.append(("ar", lo + n, hi))
        else:
            self._issue("ar", lo, hi)
        self._slices.append((lo, hi))

    class _ScatterWork:
        """reduce-scatter through a backend that wants separate buffers: wait, then put the piece where the in-place form leaves it"""

        def __init__(self, work, mine, tmp):
            self.work, self.mine, self.tmp = work, mine, tmp

        def wait(self):
            self.work.wait()
            self.mine.copy_(self.tmp)

    def _issue(self, kind, lo, hi):
        buf = self.store.grad
        if self.comm_buf is not None:           # bf16 bucket, filled on the stream that just finished the slice
            self._comm_ops.cast_from_f32(self.store.grad[lo:hi], self.comm_buf[lo:hi], hi - lo)
            buf = self.comm_buf
        piece = buf[lo:hi]
        if self.xl_comm is not None:            # a C-ABI call like any other: recorded into the plan as such
            if kind == "ar":
                self._comm_ops.comm_allreduce(self.xl_comm, piece, hi - lo)
            else:
                self._comm_ops.comm_reduce_scatter(self.xl_comm, piece, hi - lo, self.rank, self.world)
            self._works.append(None)
            return
        stream = torch.cuda.current_stream() if self.device.type == "cuda" else None      # the stream that finished the slice
        per = (hi - lo) // self.world
        mine = piece[self.rank * per:(self.rank + 1) * per]
        in_place = dist.get_backend() == "nccl"          # (RCCL's in-place layout: recv = send + rank * count)

        def issue():
            with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                if kind == "ar":
                    self._works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True))
                elif in_place:
                    self._works.append(dist.reduce_scatter_tensor(mine, piece, op=dist.ReduceOp.SUM, async_op=True))
                else:
                    tmp = torch.empty_like(mine)
                    self._works.append(PretrainStep._ScatterWork(
                        dist.reduce_scatter_tensor(tmp, piece, op=dist.ReduceOp.SUM, async_op=True), mine, tmp))
        self._host_op(issue)

    def owned_ranges(self):
        """[(lo, hi)] of the flat buffers this rank's optimizer pass covers, in issue order of their slices: its piece of every
        reduce-scattered slice and all of every replicated tail (everything, in "allreduce" mode)."""
        if not self.sharded:
            return [(0, self.store.n_used)]
        out = []
        for kind, lo, hi in self._segments:
            if kind == "rs":
                per = (hi - lo) // self.world
                out.append((lo + self.rank * per, lo + (self.rank + 1) * per))
            else:
                out.append((lo, hi))
        return out

    def _on_grad_ready(self, lo, hi, flush=False, lane="v"):
        ln = self._lanes.get(lane)
        if ln is None or ln[1] != lo:                          # a range grows at its end; a jump starts a new one
            if ln is not None and ln[1] > ln[0]:
                self._send(ln[0], ln[1])
            ln = self._lanes[lane] = [lo, lo]                  # [first unsent element, end]
        ln[1] = hi
        if hi > ln[0] and (flush or hi - ln[0] >= self.bucket_elems):
            self._send(ln[0], hi)
            ln[0] = hi

    def _finish_exchange(self):
        for lane in self._lanes.values():
            if lane[1] > lane[0]:
                self._send(lane[0], lane[1])
        pos = 0
        for lo, hi in sorted(self._slices):                    # the slices tile [0, n_used): nothing twice, nothing missed
            assert lo == pos, (lo, pos)
            pos = hi
        assert pos == self.store.n_used, (pos, self.store.n_used)
        if self.xl_comm is not None:
            # the compute stream continues after every collective of this step; the wait is bracketed by two timing events recorded
            # THROUGH the C ABI (plan-able: a replayed step re-records them), so exposed_comm() has a figure on this path too
            if self._comm_t is None and self.device.type == "cuda":
                self._comm_t = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                for e in self._comm_t:
                    e.record()                      # (creates the HIP event behind .cuda_event)
            cur = torch.cuda.current_stream()
            if self._comm_t is not None:
                self.ops.event_record(self._comm_t[0].cuda_event, cur)
            self.ops.comm_wait(self.xl_comm)
            if self._comm_t is not None:
                self.ops.event_record(self._comm_t[1].cuda_event, cur)
            self._buckets_to_grad()
            return
        timed = self.device.type == "cuda"

        def wait_all():                         # the compute stream waits for every bucket (a host operation of a recorded plan)
            if timed:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
            for w in self._works:
                w.wait()
            if timed:
                t1.record()
                self.exposed_comm_ms.append((t0, t1))
                del self.exposed_comm_ms[:-64]
        self._host_op(wait_all)
        self._buckets_to_grad()

    def _buckets_to_grad(self):
        """summed bf16 buckets back into the fp32 gradient buffer (norm + AdamW read it): the ranges this rank's pass covers"""
        if self.comm_buf is not None:
            for a, b in self.owned_ranges():
                self._comm_ops.cast_to_f32(self.comm_buf[a:b], self.store.grad[a:b], b - a)

    def exposed_comm(self):
        """mean time (ms) the compute stream spent waiting for gradient collectives at the end of backward (last <= 64 steps;
        call after a synchronize)."""
        if self._comm_t is not None:            # xl_comm_* issuer: the pair brackets the LAST step's wait
            return self._comm_t[0].elapsed_time(self._comm_t[1])
        ts = [a.elapsed_time(b) for a, b in self.exposed_comm_ms]
        return sum(ts) / len(ts) if ts else 0.0

    def step(self, batch, task=None, update=True):
        """batch: dict with input_ids, attention_mask (optional), token_type_ids (optional), visual_pos,
        cluster_ids, vis_mask, obj_labels (optional: derived from cluster_ids/vis_mask as the reference does).
        update=False: gradient accumulation (the reference's --update_freq, tasks/vqa.py:152-159,189-198): forward + backward
        only, the gradients stay in the buffer and the next call adds to them; the call with update=True exchanges the SUM of
        the micro-batches' gradients (DDP reduces every backward: the same sum), clips, steps and drops them."""
        eng, st, ops = self.engine, self.store, self.ops
        if batch.get("_ready") is not None:           # staged by a BatchUploader: the copies must have landed
            torch.cuda.current_stream().wait_event(batch["_ready"])
        eng.accumulate = self._accum_pending          # earlier micro-batches are in the gradient buffer: do not clear it
        eng.dw_overwrite = self.overwrite_grads and not eng.accumulate      # ... and add to it; otherwise one-writer gradients are stored
        accumulating = self._accum_pending or not update
        self._accum_pending = not update
        exchange = self.exchange and update
        run = self.task if self.task != "all" else task           # a multi-task step object is told which branch to run
        assert run in ("vis_mask", "word_mask", "matched", "qa", "vqa", "nlvr2"), "PretrainStep(task='all').step(batch, task=...)"
        # a model built with task_qa (ParamStore num_answers > 0 on a pretraining task): batch["qa_labels"] [B] (-100 = no
        # answer; the caller applies the reference's matched-task flip mask, lxmert_pretrain.py:186-190) joins every branch;
        # the qa loss of the step is self.engine.answer.loss
        self._step_task = run
        ids = batch["input_ids"]
        am = batch.get("attention_mask")
        if am is None:
            am = ids > 0                                       # ref lxmert_pretrain.py:206 / tasks/vqa.py:178
        qa_labels = batch.get("qa_labels") if eng.task_qa else None
        if run in ("word_mask", "matched", "qa"):
            # language pretraining branches (ref lxmert_pretrain.py:159-160,180-182,192-195): un-masked codebook features;
            # batch: input_ids (masked_word_id / other_word_id), visual_pos, cluster_ids, word_labels (+ optional word_rows:
            # word_rows_of(word_labels), the masked-row decoder) | matched_labels
            eng.set_step_seed(self._next_seed())
            eng.set_inputs(ids, am, batch.get("token_type_ids"), batch["visual_pos"], cluster_ids=batch["cluster_ids"],
                           lang_rows=batch.get("lang_rows"), lang_off=batch.get("lang_off"),
                           word_order=batch.get("word_order"))
            eng.grad_ready = None
            if exchange:
                self._begin_exchange()
                eng.grad_ready = self._on_grad_ready
            if run == "word_mask":
                loss = eng.word_mask_forward_backward(batch["word_labels"], batch.get("word_rows"), qa_labels=qa_labels)
            elif run == "matched":
                loss = eng.matched_forward_backward(batch["matched_labels"], qa_labels=qa_labels)
            else:
                loss = eng.qa_forward_backward(qa_labels)
            self._mark_consumed(batch)          # the label tensors are copied inside *_forward_backward: only now is the slot free
            if exchange:
                self._finish_exchange()
            if update:
                self.optimizer_step()
            return loss
        if run in ("vqa", "nlvr2"):
            # vqa: input_ids (word_ids), visual_feats [B,V,F] (vis_feats), visual_pos (boxes), targets [B,A] soft scores
            # nlvr2 (ref tasks/nlvr2_model.py:35-66): input_ids [2P,L] (every statement twice), visual_feats [P,2,V,F],
            #        visual_pos [P,2,V,4], labels [P]
            feats, pos = batch["visual_feats"], batch["visual_pos"]
            if run == "nlvr2":
                feats, pos = feats.reshape(-1, *feats.shape[2:]), pos.reshape(-1, *pos.shape[2:])
            eng.set_step_seed(self._next_seed())
            eng.set_inputs(ids, am, batch.get("token_type_ids"), pos, visual_feats=feats,
                           lang_rows=batch.get("lang_rows"), lang_off=batch.get("lang_off"),
                           word_order=batch.get("word_order"))
            eng.grad_ready = None
            if exchange:
                self._begin_exchange()
                eng.grad_ready = self._on_grad_ready
            loss = eng.vqa_forward_backward(batch["targets"]) if run == "vqa" else eng.nlvr2_forward_backward(batch["labels"])
            self._mark_consumed(batch)          # (targets / labels are copied inside *_forward_backward)
            if exchange:
                self._finish_exchange()
            if update:
                self.optimizer_step()
            return loss
        labels = batch.get("obj_labels")
        if labels is None:
            labels = batch["cluster_ids"].clone()
            labels[~batch["vis_mask"].bool()] = -100           # ref lxmert_pretrain.py:163-166
        eng.set_step_seed(self._next_seed())
        eng.set_inputs(ids, am, batch.get("token_type_ids"), batch["visual_pos"], cluster_ids=batch["cluster_ids"],
                       vis_mask=batch["vis_mask"], obj_labels=labels, masked_rows=batch.get("masked_rows"),
                       feat_labels=batch.get("feat_labels") if self.feat_loss else None,
                       lang_rows=batch.get("lang_rows"), lang_off=batch.get("lang_off"),
                           word_order=batch.get("word_order"))
        if qa_labels is None:                   # every tensor of the batch went through set_inputs: the staging slot is free
            self._mark_consumed(batch)
        if self.plan_mode and qa_labels is None and not accumulating:
            return self._planned_step()
        losses = self._eager_vis_mask(exchange, update, qa_labels)
        if qa_labels is not None:               # (qa_labels are copied after the encoder forward: Engine._qa_forward)
            self._mark_consumed(batch)
        return losses

    def _next_seed(self):
        """step part of the dropout seeds of the forward about to run: one value per (forward pass, rank)"""
        s = self.micro * self.world + self.rank
        self.micro += 1
        return s

    def state(self):
        """host-side counters a resumed run needs next to the ParamStore buffers (parameters, Adam moments, step_dev)"""
        return {"t": self.t, "micro": self.micro}

    def load_state(self, st):
        self.t, self.micro = int(st["t"]), int(st.get("micro", st["t"]))

    @staticmethod
    def _mark_consumed(batch):
        """every tensor of the batch has been copied into the engine's static buffers (the copies are queued on the current
        stream): a BatchUploader may refill the staging slot.  Called only AFTER the last such copy is queued -- the label
        tensors of the language / VQA / QA branches are copied inside the *_forward_backward calls."""
        if batch.get("_consumed") is not None:
            batch["_consumed"].record()

    def _eager_vis_mask(self, exchange, update=True, qa_labels=None):
        eng = self.engine
        eng.grad_ready = None
        if exchange:
            self._begin_exchange()
            eng.grad_ready = self._on_grad_ready
        losses = eng.vis_mask_forward_backward(self.feat_loss, qa_labels=qa_labels)
        if exchange:
            self._finish_exchange()
        if update:
            self.optimizer_step()
        return losses

    def _planned_step(self):
        """forward + backward + optimizer of the batch set_inputs has just staged, replayed from a launch plan.  One plan per
        launch geometry = (padded number of masked rows, losses): with n ~ U{1..64} masks per image the padded count takes a
        handful of values at bs 256.  The very first step only runs (it allocates the backward scratch); a geometry's first
        use runs AND records; from then on the step is one xl_plan_run."""
        eng = self.engine
        use_rows = eng.compact_head and eng.has_vmask and 0 < eng.n_mrows < eng.MV
        key = (eng.n_mrows if use_rows else -1, bool(self.feat_loss), eng.feat_tgt is not None, eng.has_vmask, eng.use_codebook,
               eng.ML if eng.packed else -1)                  # (padded count of packed language rows: a launch size too)
        plan = self._plans.get(key)
        if plan is not None:
            if self.exchange:
                self._begin_exchange(replay=True)       # (the recorded host operations append to this replay's work list)
            plan.run()
            HipOps.forget_binding()             # (the replay ends in whichever context its last recorded bind named)
            self.t += 1
            return eng.losses
        if not self._plan_warm:
            self._plan_warm = True
            return self._eager_vis_mask(self.exchange)
        with self.ops.lib.record() as calls:
            self.ops.rebind()                   # first entry of the plan: this trainer's library context
            self._eager_vis_mask(self.exchange)
        self._plans[key] = self.ops.lib.make_plan(calls)
        return eng.losses

    def optimizer_step(self):
        self.t += 1                      # host mirror of step_dev (seeds, logging): never read by a kernel
        eng = self.engine
        if eng.overwritten:              # ranges the backward stores rather than accumulates: the pass leaves them uncleared
            if self.store.mark_overwritten(sorted(eng.overwritten)) and self._plans:
                # a plan recorded BEFORE a range became "kept" may hold an accumulating (mask 0) K-split launch into it that relied
                # on the optimizer pass clearing the range; the pass has just stopped doing so.  Geometries re-record on next use.
                self._plans.clear()
            # "every backward rewrites a kept range" is a claim about the step's composition: a step that skipped a producer (the
            # answer head when qa_labels is None, ...) left the PREVIOUS step's gradient in a range nobody clears any more.  The
            # reference's tensor would have .grad None there: zero it, so that the pass sees a zero gradient like before overwrite mode
            for lo, hi in sorted(eng.overwritten - eng.written_now):
                self.ops.zero(self.store.grad[lo:hi])
            eng.written_now = set()
        self._optimizer_launches()

    # ---- sharded optimizer (collective="rs+ag"): shard-local norm + one scalar all-reduce, AdamW over this rank's shards,
    # all-gather of the updated compute weights slice by slice in the order the next forward needs them
    def _allreduce_scalar(self, t):
        if self.xl_comm is not None:
            self._comm_ops.comm_allreduce(self.xl_comm, t, t.numel())
            self.ops.comm_wait(self.xl_comm)
        else:
            self._host_op(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM))

    def _map_groups_to_segments(self):
        """for every parameter group of the forward (ParamStore.forward_groups): the all-gather it has to wait for = the one
        issued LAST among the slices that hold any of its parameters (the collectives' stream runs them in issue order).
        Slices are issued last-finished first, so that is the intersecting slice with the lowest index."""
        key = tuple(self._segments)
        if key == self._seg_key:
            return
        self._seg_key = key
        while len(self._seg_events) < len(self._segments):
            self._seg_events.append(self.ops.new_event())
        self._group_seg = {}
        for gkey, lo, hi in self.store.forward_groups():
            for i, (kind, a, b) in enumerate(self._segments):
                if kind == "rs" and a < hi and lo < b:
                    self._group_seg[gkey] = min(i, self._group_seg.get(gkey, i))

    def _wait_gathered(self, key):
        """engine hook (sharded exchange through xl_comm_*): the current stream is about to read the parameters of group `key`"""
        i = self._group_seg.get(key)
        if i is not None:
            self.ops.stream_wait(self._seg_events[i], torch.cuda.current_stream())

    def _sharded_optimizer(self):
        st, ops = self.store, self.ops
        b1, b2 = self.betas
        W, r = self.world, self.rank
        ops.schedule_step(self.step_dev, self.lr, self.warmup_steps, self.total_steps, b1, b2, self.lrs)
        owned = self.owned_ranges()
        if self.clip > 0:
            ops.zero(self.sumsq)
            for (kind, _, _), (a, b) in zip(self._segments, owned):
                if kind == "rs":
                    ops.sumsq(st.grad[a:b], self.sumsq, b - a, self.sumsq_scratch)
            self._allreduce_scalar(self.sumsq)                  # the shards tile the scattered slices: sum over ranks = their norm^2
            for (kind, _, _), (a, b) in zip(self._segments, owned):
                if kind == "ar":                                # replicated tails: the same addends on every rank, after the all-reduce
                    ops.sumsq(st.grad[a:b], self.sumsq, b - a, self.sumsq_scratch)
        flags = st.decay_flags
        cs = self.chunk_steps
        if cs is not None:
            flags = st.task_flags(self._step_task)
            cs.add_(((flags & 2) == 0).to(torch.int32))
        fp32 = st.compute_dtype == torch.float32
        hooked = self.engine.params_ready is not None and self.xl_comm is not None
        if hooked:
            self._map_groups_to_segments()
        # What travels back is the fp32 MASTER slice (biases and LayerNorm affines are read from it in fp32, the matrices through
        # the compute-dtype copy, which every rank re-derives from the gathered slice with one cast on the collectives' stream):
        # the same bytes on the wire as the all-reduce's second half, and every rank's master weights stay whole.
        bf16_wire = self.gather_bf16
        if bf16_wire and W > 1:
            # from here on the fp32 master copy of the MATRICES is current on the owning rank only: the store refuses to export
            # parameters (named_state) and the engine keeps its compute copy (sync_compute_weights) until gather_state() -- ADVICE r5
            st.master_partial = True
        for i in reversed(range(len(self._segments))):          # last-finished slice first: feature encoder, embeddings, layer 0 ...
            kind, lo, hi = self._segments[i]
            a, b = owned[i]
            c0, c1 = a // 256, b // 256
            ops.adamw(st.master[a:b], st.grad[a:b], st.exp_avg[a:b], st.exp_avg_sq[a:b],
                      st.compute[a:b] if ((kind == "ar" or bf16_wire) and not fp32) else None,
                      flags[c0:c1], self.sumsq if self.clip > 0 else None, self.lrs, b - a, b1, b2, self.eps, self.wd, self.clip,
                      grad_scale=1.0 / W, chunk_steps=cs[c0:c1] if cs is not None else None, zero_grad=False)
            if kind != "rs":
                continue
            piece = st.compute[lo:hi] if bf16_wire else st.master[lo:hi]
            side = None
            if bf16_wire:                   # the slice's fp32-read elements: this rank's (zeros elsewhere), to be summed over the ranks
                idx, pack = self._vector_pack(lo, hi)
                if idx is not None:
                    ops.take_f32(st.master[lo:hi], idx, a - lo, b - lo, pack)
                    side = (idx, pack)
            if self.xl_comm is not None:
                self.ops.comm_allgather(self.xl_comm, piece, hi - lo, r, W)
                if side is not None:
                    self.ops.comm_allreduce(self.xl_comm, side[1], side[0].numel())
                with torch.cuda.stream(self._comm_stream):
                    if side is not None:
                        ops.put_f32(st.master[lo:hi], side[0], side[1])
                    if not fp32 and not bf16_wire:
                        ops.cast_from_f32(piece, st.compute[lo:hi], hi - lo)
                    if hooked:
                        ops.event_record(self._seg_events[i], self._comm_stream)
            else:
                per = (hi - lo) // W
                self._host_op(lambda piece=piece, per=per: dist.all_gather_into_tensor(piece, piece[r * per:(r + 1) * per].clone()))
                if side is not None:
                    self._host_op(lambda t=side[1]: dist.all_reduce(t, op=dist.ReduceOp.SUM))
                    ops.put_f32(st.master[lo:hi], side[0], side[1])
                if not fp32 and not bf16_wire:
                    ops.cast_from_f32(piece, st.compute[lo:hi], hi - lo)
        if self.xl_comm is not None and not hooked:
            ops.comm_wait(self.xl_comm)
        # the non-owned parts of the gradient buffer hold partial sums: the next backward clears the whole buffer itself
        self.engine.grad_is_zero = False

    def _vector_pack(self, lo, hi):
        """(int32 positions inside [lo, hi) of the elements read in fp32, fp32 pack buffer) of a scattered slice -- (None, None)
        when it has none; built once per slice (the segment list repeats from step to step)"""
        if (lo, hi) not in self._vec_idx:
            idx = self.store.fp32_read_index(lo, hi)
            self._vec_idx[(lo, hi)] = (idx, torch.zeros(idx.numel(), dtype=torch.float32, device=self.device)) if idx.numel() else (None, None)
        return self._vec_idx[(lo, hi)]

    def gather_state(self):
        """sharded exchange: make the Adam moments whole on every rank (each rank's are current on its shards only) -- before a
        checkpoint is written or replicas are compared.  Two all-gathers per slice (three with gather="bf16": the fp32 master
        copy of the matrices is current on the owning rank only); off the step."""
        if not self.sharded or not self._segments:
            return
        self.sync()
        st, W, r = self.store, self.world, self.rank
        for kind, lo, hi in self._segments:
            if kind != "rs":
                continue
            per = (hi - lo) // W
            for buf in (st.exp_avg, st.exp_avg_sq) + ((st.master,) if self.gather_bf16 else ()):      # (fp32 gather: the master weights are gathered by every step)
                piece = buf[lo:hi]
                dist.all_gather_into_tensor(piece, piece[r * per:(r + 1) * per].clone())
        self.sync()
        st.master_partial = False

    def _optimizer_launches(self):
        if self.sharded and self._segments:
            return self._sharded_optimizer()
        st, ops = self.store, self.ops
        b1, b2 = self.betas
        ops.schedule_step(self.step_dev, self.lr, self.warmup_steps, self.total_steps, b1, b2, self.lrs)
        n = st.n_used
        if self.clip > 0:
            ops.zero(self.sumsq)
            ops.sumsq(st.grad, self.sumsq, n, self.sumsq_scratch)
        flags = st.decay_flags
        if self.chunk_steps is not None:
            flags = st.task_flags(self._step_task)
            self.chunk_steps.add_(((flags & 2) == 0).to(torch.int32))
        if self._opt_groups is not None:
            # behind the step, on a stream that is idle until the next backward: one launch per parameter group in forward order,
            # an event per group
            side = self.opt_stream
            if side is not None:
                ops.stream_fork(torch.cuda.current_stream(), side)
            cs = self.chunk_steps
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                for i, (key, lo, hi) in enumerate(self._opt_groups):
                    c0, c1 = lo // 256, hi // 256
                    ops.adamw(st.master[lo:hi], st.grad[lo:hi], st.exp_avg[lo:hi], st.exp_avg_sq[lo:hi],
                              st.compute[lo:hi] if st.compute_dtype != torch.float32 else None, flags[c0:c1],
                              self.sumsq if self.clip > 0 else None, self.lrs, hi - lo, b1, b2, self.eps, self.wd, self.clip,
                              grad_scale=1.0 / self.world, chunk_steps=cs[c0:c1] if cs is not None else None,
                              zero_grad=self.drop_grads)
                    if side is not None and self._opt_last[key] == i:
                        ops.event_record(self._opt_events[key], side)
            if self.drop_grads:
                self.engine.grad_is_zero = True
            return
        ops.adamw(st.master, st.grad, st.exp_avg, st.exp_avg_sq,
                  st.compute if st.compute_dtype != torch.float32 else None, flags,
                  self.sumsq if self.clip > 0 else None, self.lrs, n, b1, b2, self.eps, self.wd, self.clip,
                  grad_scale=1.0 / self.world, chunk_steps=self.chunk_steps, zero_grad=self.drop_grads)
        if self.drop_grads:
            self.engine.grad_is_zero = True          # the next backward starts from a clean buffer without clearing it

    def grad_norm(self):
        return math.sqrt(float(self.sumsq.item())) / self.world


class PackedBatch(dict):
    """a host minibatch whose tensors are views of ONE page-locked byte buffer (`buf`): a single H2D copy moves all of it.
    `layout`: [(name, dtype, shape, byte offset)], offsets 16-byte aligned; non-tensor entries are plain dict items."""
    buf, layout = None, ()


class BatchUploader:
    """Host -> device hand-over of minibatches (the reference's `.cuda()` calls at the top of every step, ref
    lxmert_pretrain.py:145-160), one step ahead of the compute: `upload(packed_batch)` queues ONE copy of the NEXT batch from
    pinned host memory into one of two device staging slots on a copy stream of its own and returns the staged batch (views of
    the slot); PretrainStep.step() makes the compute stream wait for that upload (`_ready`) before it moves the tensors into the
    engine's static input buffers, and marks the slot free afterwards (`_consumed`).  One copy per step and no wait on anything
    younger than two steps: the copy stream shares a hardware queue with a compute stream (engine.reserve_streams: four queues,
    four compute streams), and every packet it queues sits in front of that stream's kernels (nine separate copies per step cost
    0.3 ms of step time, measured)."""

    def __init__(self, device, slots=2):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.slots = [None] * slots
        self.ready = [torch.cuda.Event() for _ in range(slots)]
        self.consumed = [None] * slots
        self.n = 0

    @staticmethod
    def pin(batch):
        """pack a host batch into one page-locked buffer (what a collate function with pin_memory would hand over, ref
        lxmert_data.py:666-671)."""
        layout, off = [], 0
        for name, v in batch.items():
            if torch.is_tensor(v):
                v = v.contiguous()
                layout.append((name, v.dtype, tuple(v.shape), off))
                off = (off + v.numel() * v.element_size() + 15) // 16 * 16
        out = PackedBatch()
        out.buf = torch.empty(max(off, 16), dtype=torch.uint8).pin_memory()
        out.layout = layout
        for name, dtype, shape, o in layout:
            view = BatchUploader._view(out.buf, dtype, shape, o)
            view.copy_(batch[name])
            out[name] = view
        for name, v in batch.items():
            if not torch.is_tensor(v):
                out[name] = v
        return out

    @staticmethod
    def _view(buf, dtype, shape, off):
        n = 1
        for d in shape:
            n *= d
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        return buf[off:off + nbytes].view(dtype).view(shape)

    def upload(self, batch):
        assert isinstance(batch, PackedBatch), "BatchUploader.upload takes BatchUploader.pin(batch)"
        k = self.n % len(self.slots)
        self.n += 1
        if self.slots[k] is None or self.slots[k].numel() < batch.buf.numel():
            self.slots[k] = torch.empty(batch.buf.numel(), dtype=torch.uint8, device=self.device)
        dev = self.slots[k]
        with torch.cuda.stream(self.stream):
            if self.consumed[k] is not None:
                self.stream.wait_event(self.consumed[k])       # the step that read this slot (two uploads ago) has taken its copy
            dev[:batch.buf.numel()].copy_(batch.buf, non_blocking=True)
            self.ready[k].record(self.stream)
        out = {name: v for name, v in batch.items() if not torch.is_tensor(v)}
        for name, dtype, shape, o in batch.layout:
            out[name] = self._view(dev, dtype, shape, o)
        out["_ready"] = self.ready[k]
        self.consumed[k] = out["_consumed"] = torch.cuda.Event()
        return out


def synthetic_batch(cfg, B, L=20, grid=8, seed=9595, device="cpu", ragged=True):
    """SURVEY.md section 8d generators (torch RNG): ids with [CLS]=101 / [SEP]=102 / PAD=0, lengths U{6..L};
    cluster ids U{0..K-1}; `--vis_mask_predict` masks (n ~ U{1..V} random positions per example)."""
    g = torch.Generator().manual_seed(seed)
    V = grid * grid
    ids = torch.randint(1000 if cfg.vocab_size > 2000 else 1, cfg.vocab_size, (B, L), generator=g)
    ids[:, 0] = min(101, cfg.vocab_size - 2)
    lens = torch.randint(min(6, L), L + 1, (B,), generator=g) if ragged else torch.full((B,), L)
    ar = torch.arange(L)[None, :]
    ids[ar == (lens[:, None] - 1)] = min(102, cfg.vocab_size - 1)
    ids[ar >= lens[:, None]] = 0
    cid = torch.randint(0, cfg.num_clusters, (B, V), generator=g)
    n_mask = torch.randint(1, V + 1, (B,), generator=g)
    rank = torch.rand(B, V, generator=g).argsort(1).argsort(1)
    vm = rank < n_mask[:, None]
    lab = cid.clone()
    lab[~vm] = -100
    pos = torch.zeros(grid * grid, 4)
    for i in range(grid):
        for j in range(grid):
            pos[i * grid + j] = torch.tensor([j / grid, i / grid, (j + 1) / grid, (i + 1) / grid])
    am = ids > 0
    batch = {"input_ids": ids, "attention_mask": am, "token_type_ids": torch.zeros_like(ids),
             "cluster_ids": cid, "vis_mask": vm, "obj_labels": lab, "visual_pos": pos[None].expand(B, -1, -1).contiguous(),
             "masked_rows": vm.reshape(-1).nonzero().reshape(-1),      # computed where the mask is drawn: on the host
             # ... and likewise the real tokens' row list + per-example offsets (packed language rows, engine pack_lang)
             "lang_rows": am.reshape(-1).nonzero().reshape(-1),
             "lang_off": torch.cat([torch.zeros(1, dtype=torch.int64), am.sum(1).cumsum(0)]).to(torch.int32),
             # ... and the token positions sorted by (id, position): the embedding backward's one-writer-per-row scatter
             "word_order": word_order_of(ids)}
    return {k: v.to(device) for k, v in batch.items()}


def word_order_of(input_ids):
    """flat token positions b*L+l sorted by (input id, position) -- a stable argsort of the ids, int32 [B*L] -- what the data
    loader hands over next to `input_ids` (batch["word_order"]): xl_embed_bwd gives every word-embedding row one writer that adds
    the row's occurrences in this order (ref: nn.Embedding's backward, HF:184-186, is an index_add over the same rows)."""
    return torch.sort(input_ids.reshape(-1), stable=True).indices.to(torch.int32)


def word_rows_of(word_labels):
    """flat indices b*L+l of the positions that carry an MLM label - what the data loader hands over next to `word_labels`
    (batch["word_rows"], kept on the host: its length sizes the masked-row decoder launches without a device round trip)."""
    return (word_labels.reshape(-1) >= 0).nonzero().reshape(-1).to(torch.int32).cpu()


def random_word_batch(input_ids, mask_token_id=103, vocab_size=30522, mlm_probability=0.15, generator=None):
    """Host-side BERT masking of the `word_mask` task (ref pretrain/lxmert_data.py:697-724): each non-special position is chosen
    with probability `mlm_probability`; of the chosen ones 80 % become [MASK], 10 % a random token, 10 % stay.  Returns
    (masked_input_ids, labels) with labels = the original id at chosen positions and -100 elsewhere (the reference writes -1
    there, which its own CrossEntropyLoss(ignore_index=-100) would reject)."""
    labels = input_ids.clone()
    masked = input_ids.clone()
    shape = labels.shape

    def bern(p):
        return torch.rand(shape, generator=generator) < p

    chosen = bern(mlm_probability)
    chosen[:, 0] = False                    # "do not mask special tokens": first and last position (ref :706-707)
    chosen[:, -1] = False
    labels[~chosen] = -100
    replaced = bern(0.8) & chosen
    masked[replaced] = mask_token_id
    rnd = bern(0.5) & chosen & ~replaced
    words = torch.randint(vocab_size, shape, generator=generator, dtype=torch.long)
    masked[rnd] = words[rnd]
    return masked, labels
