import sys, torch
sys.path.insert(0, ".")
from xlxmert_amd.ops import HipOps
ops = HipOps(torch.bfloat16)
for M, N in ((16384, 768), (8192, 512), (8192, 264)):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, N, generator=g) * 2 + 0.5).bfloat16().cuda()
    dy = torch.randn(M, N, generator=g).bfloat16().cuda()
    gamma = (torch.randn(N, generator=g) * 0.2 + 1).cuda()
    xf = x.float(); mean = xf.mean(1); rstd = 1.0 / torch.sqrt(xf.var(1, unbiased=False) + 1e-12)
    ws = torch.zeros(ops.workspace_floats(N), device="cuda")
    def run(lo, hi):
        m = hi - lo
        dx = torch.full((m, N), 7.0, dtype=torch.bfloat16, device="cuda")
        dg, db, dbp = (torch.zeros(N, device="cuda") for _ in range(3))
        ops.layernorm_bwd(dy[lo:hi], x[lo:hi], gamma, mean[lo:hi].contiguous(), rstd[lo:hi].contiguous(), dx, dg, db, dbp, m, N, ws=ws)
        torch.cuda.synchronize()
        return dx
    full = run(0, M)
    ref = torch.cat([run(lo, min(M, lo + 4096)) for lo in range(0, M, 4096)])
    bad = (full != ref)
    rows = bad.any(1).nonzero().reshape(-1)
    print(f"M={M} N={N}: bad rows {rows.numel()} of {M}; first {rows[:12].tolist()}; bad cols of first bad row:",
          bad[rows[0]].nonzero().reshape(-1)[:10].tolist() if rows.numel() else None, "n bad cols", int(bad[rows[0]].sum()) if rows.numel() else 0)
    if rows.numel():
        r = int(rows[0]); print(" got", full[r, :8].tolist(), "\n ref", ref[r, :8].tolist())
        good = (~bad.any(1)).nonzero().reshape(-1); print(" good rows first", good[:12].tolist(), "count", good.numel())
