"""The data-parallel step on SEVERAL ranks with the real kernels (ref pretrain/lxmert_pretrain.py:694-700, 102-106:
one process per GPU, NCCL backend, DistributedDataParallel): every rank draws its own minibatch, the gradients meet through the
bucketed exchange, the replicas must stay bit-identical and equal ONE AdamW step on the mean of the oracle's per-rank gradients.

Two rigs, one worker:
  * two GPUs, RCCL over xGMI (backend nccl) -- through both issuers (the library's own binding xl_comm_*, and torch.distributed),
    fp32 and bf16 buckets, eager and plan replay, both collectives (all-reduce; reduce-scatter -> shard AdamW -> all-gather).
    Skipped where fewer than two GPUs are visible: green wherever a multi-GPU lease runs the suite.
  * two ranks SHARING one GPU with gloo carrying the collectives (RCCL refuses two ranks on a device): the same worker, so the
    code the two-GPU tests run is exercised on every one-GPU box too.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

import lxmert_oracle as O

pytestmark = pytest.mark.gpu

N_GPUS = torch.cuda.device_count() if torch.cuda.is_available() else 0
two_gpus = pytest.mark.skipif(N_GPUS < 2, reason="needs two visible GPUs (RCCL refuses two ranks on one device)")


def _worker(rank, world, port, out_dir, backend, issuer, comm_bf16, collective):
    import torch.distributed as dist
    from test_trainer_cpu import TINY, oracle_cfg
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import reserve_streams
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), XL_COMM=issuer)
    dev = f"cuda:{rank}" if backend == "nccl" else "cuda:0"
    torch.cuda.set_device(dev)
    reserve_streams(dev)                                     # before RCCL's stream can take a hardware queue
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    cfg = XLxmertConfig(**TINY)
    sd = O.make_state_dict(oracle_cfg(cfg), 3)
    out = {}
    for plan in (False, True):
        store = ParamStore(cfg, dev, torch.float32, task="vis_mask")
        store.load_named(sd)
        tr = PretrainStep(cfg, 2, 8, 16, dtype=torch.float32, device=dev, store=store, total_steps=10, lr=1e-2, bucket_mb=0.05,
                          visual_losses="obj,feat", plan=plan, drop_grads=False, collective=collective,
                          grad_comm_dtype=torch.bfloat16 if comm_bf16 else None)
        assert tr.exchange and tr.world == world and tr.collective == collective
        assert (tr.xl_comm is not None) == (issuer == "rccl" and backend == "nccl"), "the library's RCCL binding failed its self-test"
        for t in range(3):                                   # (plan: warm-up, record, replay)
            batch = {k: v.to(dev) for k, v in synthetic_batch(cfg, 2, 8, 4, seed=500 + 10 * t + rank).items()}   # disjoint per rank
            tr.step(batch)
            if t == 0 and not plan:
                tr.sync()
                out["step1"] = {k: tr.store.view(k).cpu().clone() for k in tr.store.names()}
        tr.sync()
        assert len(tr._slices) > 3 and sum(b - a for a, b in tr._slices) == tr.store.n_used
        if plan:
            (p,) = tr._plans.values()
            assert (p.n_host_ops == 0) == (tr.xl_comm is not None)
        if tr.sharded:
            n_rs = sum(hi - lo for k, lo, hi in tr._segments if k == "rs")
            assert sum(b - a for a, b in tr.owned_ranges()) == n_rs // world + (tr.store.n_used - n_rs)
        bad = tr.verify_replicas()
        assert bad == [], (plan, len(bad), bad[:8])
        assert tr.exposed_comm() >= 0.0
        out["plan" if plan else "eager"] = {k: tr.store.view(k).cpu().clone() for k in tr.store.names()}
        tr.close()
    torch.save(out, os.path.join(out_dir, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _worker_gather(rank, world, port, out_dir, backend, issuer):
    """rs+ag on a bf16 model: fp32 master slices on the wire against the bf16 compute copy + fp32 side car (trainer gather=)"""
    import torch.distributed as dist
    from test_trainer_cpu import TINY, oracle_cfg
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.engine import reserve_streams
    from xlxmert_amd.params import ParamStore
    from xlxmert_amd.trainer import PretrainStep, synthetic_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), XL_COMM=issuer)
    dev = f"cuda:{rank}" if backend == "nccl" else "cuda:0"
    torch.cuda.set_device(dev)
    reserve_streams(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    cfg = XLxmertConfig(**TINY)
    sd = O.make_state_dict(oracle_cfg(cfg), 3)
    out = {}
    for name in ("fp32", "bf16"):
        for plan in (False, True):
            store = ParamStore(cfg, dev, torch.bfloat16, task="vis_mask")
            store.load_named(sd)
            tr = PretrainStep(cfg, 2, 8, 16, dtype=torch.bfloat16, device=dev, store=store, total_steps=10, lr=1e-2, bucket_mb=0.05,
                              visual_losses="obj,feat", plan=plan, drop_grads=False, collective="rs+ag", gather=name)
            tr.ops.set_gemm_wgrad_slabs(1)                   # (reproducible K-split weight gradients: the two runs are compared bit for bit)
            assert tr.sharded and tr.gather_bf16 == (name == "bf16")
            for t in range(3):
                tr.step({k: v.to(dev) for k, v in synthetic_batch(cfg, 2, 8, 4, seed=500 + 10 * t + rank).items()})
            tr.sync()
            st = tr.store
            idx = st.fp32_read_index(0, st.n_used).long()
            out[f"{name}:{plan}:compute"] = st.compute[:st.n_used].cpu().clone()
            out[f"{name}:{plan}:fp32_read"] = st.master[idx].cpu().clone()
            assert tr.verify_replicas() == []
            out[f"{name}:{plan}:master"] = st.master[:st.n_used].cpu().clone()
            tr.close()
    torch.save(out, os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _run_gather(tmp_path, backend, issuer):
    import torch.multiprocessing as mp
    from test_trainer_cpu import _free_port
    world, port = 2, _free_port()
    mp.spawn(_worker_gather, args=(world, port, str(tmp_path), backend, issuer), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k
    for plan in (False, True):
        for what in ("compute", "fp32_read", "master"):
            assert torch.equal(r0[f"fp32:{plan}:{what}"], r0[f"bf16:{plan}:{what}"]), (plan, what)


def test_bf16_all_gather_two_ranks_sharing_one_gpu_gloo(tmp_path):
    """gather="bf16" (the compute copy + the sparse fp32 side car on the wire) against the fp32 master gather, real kernels, eager
    and plan: the same compute copy, fp32-read elements and (gathered) master weights, bit for bit, on both ranks"""
    _run_gather(tmp_path, "gloo", "torch")


@two_gpus
@pytest.mark.parametrize("issuer", ["rccl", "torch"])
def test_bf16_all_gather_two_gpus_rccl(tmp_path, issuer):
    _run_gather(tmp_path, "nccl", issuer)


def _run_and_check(tmp_path, backend, issuer, comm_bf16, collective):
    import torch.multiprocessing as mp
    from test_trainer_cpu import TINY, _free_port, oracle_cfg, oracle_grads
    from xlxmert_amd.config import XLxmertConfig
    from xlxmert_amd.trainer import linear_schedule, synthetic_batch
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), backend, issuer, comm_bf16, collective), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "w0.pt"), torch.load(tmp_path / "w1.pt")
    for part in ("step1", "eager", "plan"):
        for k in r0[part]:
            assert torch.equal(r0[part][k], r1[part][k]), (part, k)
    # a replayed plan runs the step the eager trainer runs (fp32 atomics: not bit for bit)
    for k in r0["eager"]:
        assert (r0["eager"][k] - r0["plan"][k]).abs().max().item() < (5e-5 if not comm_bf16 else 2e-2), k
    if comm_bf16:
        return
    cfg = XLxmertConfig(**TINY)
    sd = O.make_state_dict(oracle_cfg(cfg), 3)
    gs = [oracle_grads(cfg, sd, synthetic_batch(cfg, 2, 8, 4, seed=500 + r))[0] for r in range(world)]
    names = sorted(gs[0])
    _, clipped = O.clip_grad_norm([(gs[0][k] + gs[1][k]) / 2 for k in names], 1.0)
    lr = 1e-2 * linear_schedule(0, 0, 10)
    for k, g in zip(names, clipped):
        p, _, _ = O.adamw_update(sd[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, lr)
        assert (r0["step1"][k] - p).abs().max().item() < 5e-5, (k, (r0["step1"][k] - p).abs().max().item())


@pytest.mark.parametrize("collective", ["allreduce", "rs+ag"])
@pytest.mark.parametrize("comm_bf16", [False, True], ids=["fp32-buckets", "bf16-buckets"])
def test_two_ranks_sharing_one_gpu_gloo(tmp_path, comm_bf16, collective):
    """the worker of the two-GPU tests on the one-GPU rig (gloo; torch.distributed issues the collectives)"""
    _run_and_check(tmp_path, "gloo", "torch", comm_bf16, collective)


@two_gpus
@pytest.mark.parametrize("collective", ["allreduce", "rs+ag"])
@pytest.mark.parametrize("comm_bf16", [False, True], ids=["fp32-buckets", "bf16-buckets"])
@pytest.mark.parametrize("issuer", ["rccl", "torch"])
def test_two_gpus_rccl_data_parallel_equals_mean_gradient_step(tmp_path, issuer, comm_bf16, collective):
    """first contact with RCCL over xGMI: two ranks on two GPUs (ref lxmert_pretrain.py:694-700, 102-106)"""
    _run_and_check(tmp_path, "nccl", issuer, comm_bf16, collective)


@two_gpus
@pytest.mark.parametrize("collective", ["allreduce", "rs+ag"])
def test_bench_two_gpus_end_to_end(tmp_path, collective):
    """`python bench.py --gpus 2` as the driver starts it for the scaling curve: one JSON line, n_gpus 2, the exchange filled in
    and issued by the library's own RCCL binding from ONE launch plan"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XL_BENCH_FAULT_TIMEOUT="300")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "XL_BENCH_SHARE_GPU", "XL_COMM"):
        env.pop(k, None)
    with open(tmp_path / "out", "w") as fo, open(tmp_path / "err", "w") as fe:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "4", "--no-extra",
                            "--no-cpu-baseline", "--collective", collective], env=env, stdout=fo, stderr=fe,
                           stdin=subprocess.DEVNULL, timeout=500)
    stdout, stderr = (tmp_path / "out").read_text(), (tmp_path / "err").read_text()
    assert r.returncode == 0, stderr[-6000:]
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and "shared_gpu" not in out and out["config"]["global_batch"] == 512
    ex = out["config"]["gradient_exchange"]
    assert ex["collective"] == collective and ex["bytes_per_step"] > 5e8 and ex["issued_by"].startswith("xl_comm")
    assert ex["exposed_comm_ms_per_step"] >= 0.0
    if collective == "rs+ag":
        assert ex["optimizer_elements_per_rank"] < 0.51 * ex["optimizer_elements_total"]
    assert out["config"]["step_launch"].startswith("launch plan") and "host operations" not in out["config"]["step_launch"]
    assert out["value"] > 0 and out["roofline"]["frac"] > 0.0
