import torch, sys
sys.path.insert(0, ".")
from xlxmert_amd.ops import HipOps
ops = HipOps(torch.bfloat16)
for M in (16384, 5120, 21504):
    N = 768
    dy = torch.randn(M, N, device="cuda").bfloat16(); x = torch.randn(M, N, device="cuda").bfloat16()
    g = torch.randn(N, device="cuda"); mean = torch.zeros(M, device="cuda"); rstd = torch.ones(M, device="cuda")
    dx = torch.zeros_like(x); dxd = torch.zeros_like(x)
    dg = torch.zeros(N, device="cuda"); db = torch.zeros(N, device="cuda"); dbp = torch.zeros(N, device="cuda")
    ws = torch.zeros(ops.workspace_floats(N), device="cuda")
    y = torch.zeros_like(x)
    import os
    if os.environ.get("NODROP"):
        def bwd(): ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, dbp, M, N, ws=ws)
    else:
        def bwd(): ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, dbp, M, N, ws=ws, dx_dropped=dxd, p_drop=0.1, seed=5)
    def fwd(): ops.layernorm_fwd(x, g, g, y, mean, rstd, M, N, 1e-12)
    for name, f in (("ln_bwd", bwd), ("ln_fwd", fwd)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        print(f"{name} M={M}: {s.elapsed_time(e)/20*1e3:.1f} us")
