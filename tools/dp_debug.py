import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
def w(rank, world, port, mode):
    import lxmert_oracle as O
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
                      visual_losses="obj,feat")
    if mode == "single":
        tr.engine.side = None
    if mode != "full":
        tr.optimizer_step = lambda: None
    batch = {k: v.cuda() for k, v in synthetic_batch(cfg, 2, 8, 4, seed=500 + (rank if mode != "same" else 0)).items()}
    tr.step(batch)
    torch.cuda.synchronize()
    g = (tr.store.master if mode == "full" else tr.store.grad)[:tr.store.n_used].cpu().clone()
    if mode == "full":
        v = torch.arange(6, device="cuda", dtype=torch.float64) * (rank + 1)
        lo, hi = v.clone(), v.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        print(rank, "min/max probe", lo.tolist(), hi.tolist(), "verify:", len(tr.verify_replicas()), "gnorm", tr.grad_norm(), flush=True)
    torch.save((g, tr._slices), f"/tmp/dpdbg_{rank}.pt")
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "dp"
    mp.spawn(w, args=(2, 29631, mode), nprocs=2, join=True)
    (g0, s0), (g1, s1) = torch.load("/tmp/dpdbg_0.pt"), torch.load("/tmp/dpdbg_1.pt")
    print(mode, "slices equal:", s0 == s1, "n slices", len(s0))
    d = (g0 - g1).abs()
    print("max |g0-g1|", d.max().item(), "nonzero", int((d > 0).sum()), "of", d.numel(), "norm", g0.norm().item())
    for lo, hi in s0:
        dd = d[lo:hi]
        if dd.max().item() > 0:
            print("  slice", lo, hi, "max diff", dd.max().item(), "first idx", int((dd > 0).nonzero()[0]) + lo, "count", int((dd > 0).sum()))
