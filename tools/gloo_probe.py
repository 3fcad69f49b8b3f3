import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    big = torch.full((1 << 20,), float(rank + 1), device="cuda")
    s = torch.cuda.Stream()
    works = []
    with torch.cuda.stream(s):
        big.mul_(2.0)                                   # producer kernel on a side stream
        works.append(dist.all_reduce(big[1000:200000], op=dist.ReduceOp.SUM, async_op=True))
        works.append(dist.all_reduce(big[200000:], op=dist.ReduceOp.SUM, async_op=True))
    for x in works: x.wait()
    torch.cuda.synchronize()
    print(rank, big[0].item(), big[1000].item(), big[199999].item(), big[200000].item(), big[-1].item(), flush=True)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(2, 29611), nprocs=2, join=True)
