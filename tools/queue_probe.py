"""Does a kernel on stream A start while stream B has a long backlog of ready kernels?  (the step's visual stack starts ~1.5 ms late,
when the language stack enqueued BEFORE it is almost through: tools/timeline.py)
side: N kernels back to back; main: ONE kernel enqueued after them (no dependency).  Reports when main's kernel finished relative
to the side stream's first kernel, for main = the default stream / a torch side stream, and small / chip-filling side kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = "cuda"
def run(main_stream, side_elems, main_elems, n=60, order="side_first"):
    side = torch.cuda.Stream()
    xs = torch.zeros(side_elems, device=dev)
    xm = torch.zeros(main_elems, device=dev)
    for _ in range(2):
        with torch.cuda.stream(side): xs.add_(1.0)
        with torch.cuda.stream(main_stream): xm.add_(1.0)
    torch.cuda.synchronize()
    e0, e1, m1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    gate = torch.cuda.Event()
    # hold both streams behind a long kernel so that everything below is enqueued before anything runs
    big = torch.zeros(1 << 28, device=dev)
    hold = torch.cuda.Stream()
    with torch.cuda.stream(hold):
        for _ in range(6): big.add_(1.0)
        gate.record()
    side.wait_event(gate); main_stream.wait_event(gate)
    def enq_side():
        with torch.cuda.stream(side):
            e0.record()
            for _ in range(n): xs.add_(1.0)
            e1.record()
    def enq_main():
        with torch.cuda.stream(main_stream):
            xm.add_(1.0)
            m1.record()
    if order == "side_first": enq_side(); enq_main()
    else: enq_main(); enq_side()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3, e0.elapsed_time(m1) * 1e3
for label, ms in (("default stream", torch.cuda.default_stream()), ("torch side stream", torch.cuda.Stream())):
    for se, me in ((1 << 18, 1 << 18), (1 << 24, 1 << 18), (1 << 24, 1 << 24), (1 << 18, 1 << 24)):
        for order in ("side_first", "main_first"):
            a, b = run(ms, se, me, order=order)
            print(f"main = {label:18s} side kernels {se * 4 >> 10:6d} KiB x60, main kernel {me * 4 >> 10:6d} KiB, {order}: side backlog takes {a:8.1f} us, "
                  f"main kernel done at {b:8.1f} us")
