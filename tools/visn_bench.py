"""visual-feature-encoder LayerNorm pair (xl_visn_ln_fwd / xl_visn_ln_bwd) timed alone."""
import sys
import torch
sys.path.insert(0, ".")
from xlxmert_amd.ops import HipOps
ops = HipOps(torch.bfloat16)
N, P = 768, 4
for M in (2048, 16384, 65536):
    g = torch.Generator().manual_seed(0)
    xv = torch.randn(M, N, generator=g).bfloat16().cuda(); dy = torch.randn(M, N, generator=g).bfloat16().cuda()
    pos = torch.rand(M, P, generator=g).cuda()
    wbox, bbox = (torch.randn(N, P, generator=g) * 0.5).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
    gv, bv, gb, bb = [(torch.randn(N, generator=g) * 0.1 + o).cuda() for o in (1, 0, 1, 0)]
    y = torch.zeros_like(xv); st = [torch.zeros(M, device="cuda") for _ in range(4)]
    outs = [torch.zeros_like(xv)] + [torch.zeros(N, device="cuda") for _ in range(4)] + [torch.zeros(N, P, device="cuda"),
            torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")]
    ws = torch.zeros(ops.workspace_floats(N), device="cuda")
    def fwd(): ops.visn_ln_fwd(xv, pos, wbox, bbox, gv, bv, gb, bb, y, *st, M, N, P, 1e-12)
    def bwd(): ops.visn_ln_bwd(dy, xv, pos, wbox, bbox, gv, gb, *st, *outs, M, N, P, ws=ws)
    for name, f in (("visn_ln_fwd", fwd), ("visn_ln_bwd", bwd)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        print(f"{name} M={M}: {s.elapsed_time(e) / 20 * 1e3:.1f} us")
