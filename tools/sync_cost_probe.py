"""What does a cross-stream hand-over cost the stream that issues it?  A chain of N dependent small kernels (a LayerNorm forward over
4096 x 768 rows) on one stream, timed (a) plain, (b) with an event recorded behind every kernel, (c) with a wait for an
ALREADY-SIGNALLED event in front of every kernel, (d) with a record + a wait by an otherwise idle second stream (the weight-gradient
fork of the training step), (e) with a full fork -> kernel on the second stream -> join per link (run_pair XL_PAIR_SIDE=1).
All calls go through the C ABI, queued behind a long kernel so that the host is never the limit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xlxmert_amd.ops import HipOps

ops = HipOps(torch.bfloat16)
M, N = 4096, 768
x = torch.randn(M, N, device="cuda").bfloat16()
y = torch.zeros_like(x)
g, b = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
mean, rstd = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
y2, mean2, rstd2 = torch.zeros_like(x), torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
with torch.cuda.stream(side):
    torch.zeros(8, device="cuda").add_(1.0)
torch.cuda.synchronize()
evs = [ops.new_event() for _ in range(512)]
done = ops.new_event()
ops.event_record(done, main)
torch.cuda.synchronize()
big = torch.zeros(1 << 28, device="cuda")
NK = 200


def kern():
    ops.layernorm_fwd(x, g, b, y, mean, rstd, M, N, 1e-12)


def chain(mode):
    for _ in range(4):
        big.add_(1.0)                       # ~2 ms of backlog: everything below is queued before it runs
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(NK):
        if mode == "wait_signalled":
            ops.stream_wait(done, main)
        if mode == "fork_join":
            ops.event_record(evs[2 * i], main)
            ops.stream_wait(evs[2 * i], side)
            with torch.cuda.stream(side):
                ops.layernorm_fwd(x, g, b, y2, mean2, rstd2, 512, N, 1e-12)
            ops.event_record(evs[2 * i + 1], side)
            ops.stream_wait(evs[2 * i + 1], main)
        kern()
        if mode == "record":
            ops.event_record(evs[i], main)
        if mode == "record_and_side_wait":
            ops.event_record(evs[i], main)
            ops.stream_wait(evs[i], side)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / NK


for mode in ("plain", "record", "wait_signalled", "record_and_side_wait", "fork_join", "plain"):
    chain(mode)
    t = min(chain(mode) for _ in range(3))
    print(f"{mode:22s} {t:7.2f} us per link")
