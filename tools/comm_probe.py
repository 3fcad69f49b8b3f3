"""one-rank probe of the two exchange paths (torch.distributed vs xl_comm_*): host time per collective call, step time.
Usage (GPU box): XL_FORCE_EXCHANGE=1 [XL_COMM=rccl] python tools/comm_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import torch.distributed as dist
from xlxmert_amd.engine import reserve_streams
torch.cuda.set_device(0)
torch.zeros(8, device="cuda").add_(1.0)
reserve_streams("cuda:0", comm=os.environ.get("XL_COMM_STREAM") == "own")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from xlxmert_amd.config import XLxmertConfig
from xlxmert_amd.trainer import PretrainStep, synthetic_batch
cfg = XLxmertConfig()
tr = PretrainStep(cfg, 256, 20, 64, dtype=torch.bfloat16, device="cuda", seed=1, plan=not os.environ.get("EAGER"), drop_grads=True,
                  overlap_optimizer=True, train_dropout=True)
g = torch.Generator().manual_seed(0)
tr.set_centroids(torch.randn(cfg.num_clusters, cfg.visual_feat_dim, generator=g).relu())
b = {k: v.cuda() for k, v in synthetic_batch(cfg, 256, 20, 8, seed=3).items()}
for _ in range(5):
    tr.step(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.step(b)
th = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"xl_comm={tr.xl_comm} exchange={tr.exchange}: {dt / 20 * 1e3:.2f} ms/step, host enqueue {th / 20 * 1e3:.2f} ms/step, slices {len(tr._slices)}")
if tr.xl_comm is not None:
    buf = tr.store.grad[:1 << 24]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        tr.ops.comm_allreduce(tr.xl_comm, buf, buf.numel())
    t1 = time.perf_counter()
    tr.ops.comm_wait(tr.xl_comm)
    torch.cuda.synchronize()
    print(f"host time per xl_comm_allreduce call on an idle GPU: {(t1 - t0) / 50 * 1e6:.1f} us; drained after {(time.perf_counter() - t1) * 1e3:.2f} ms")
else:
    buf = tr.store.grad[:1 << 24]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ws = [dist.all_reduce(buf, async_op=True) for _ in range(50)]
    t1 = time.perf_counter()
    for w in ws:
        w.wait()
    torch.cuda.synchronize()
    print(f"host time per dist.all_reduce call on an idle GPU: {(t1 - t0) / 50 * 1e6:.1f} us; drained after {(time.perf_counter() - t1) * 1e3:.2f} ms")
dist.destroy_process_group()
