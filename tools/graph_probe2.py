"""hipGraph capture of nested stream forks (debug probe)."""
import faulthandler, sys
faulthandler.enable()
import torch
case = sys.argv[1]
x = torch.zeros(1 << 20, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for s in (s1, s2):
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()


def body():
    main = torch.cuda.current_stream()
    x.add_(1.0)
    if case == "v7":                         # two siblings of the origin stream, one waits for the other
        s1.wait_event(main.record_event()); s2.wait_event(main.record_event())
        with torch.cuda.stream(s2):
            x.mul_(1.0)
        with torch.cuda.stream(s1):
            x.add_(1.0)
            s1.wait_event(s2.record_event())
            x.add_(1.0)
        main.wait_event(s1.record_event())
        x.add_(1.0)
        return
    if case in ("v5", "v6"):                 # s2 joins the capture from the ORIGIN stream first
        s2.wait_event(main.record_event())
        if case == "v6":
            with torch.cuda.stream(s2):
                x.mul_(1.0)
    s1.wait_event(main.record_event())
    with torch.cuda.stream(s1):
        x.add_(1.0)
        s2.wait_event(s1.record_event())
        with torch.cuda.stream(s2):
            x.mul_(1.0)
        if case in ("v3", "v4"):
            x.add_(1.0)                      # more work on s1 after the inner fork
        if case in ("v1", "v3", "v5", "v6"):
            s1.wait_event(s2.record_event())
        if case == "v4":                     # fork s2 a second time from s1 before joining
            s1.wait_event(s2.record_event())
            x.add_(1.0)
            s2.wait_event(s1.record_event())
            with torch.cuda.stream(s2):
                x.mul_(1.0)
            s1.wait_event(s2.record_event())
    if case == "v2":
        main.wait_event(s2.record_event())
    main.wait_event(s1.record_event())
    x.add_(1.0)


g = torch.cuda.CUDAGraph()
print(case, "capture", flush=True)
with torch.cuda.graph(g):
    body()
print(case, "replay", flush=True)
g.replay()
torch.cuda.synchronize()
print(case, "OK", x[0].item(), flush=True)
