"""Is the GEMM CU-bound or power-bound, and can the optimizer pass hide under the next step's forward?  Kernels on CU-masked
streams (hipExtStreamCreateWithCUMask): (a) a chip-filling GEMM restricted to fewer CUs, (b) AdamW's HBM rate as a function of
the CUs it may use, (c) both at once.  Usage: python tools/cumask_probe.py"""
import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from xlxmert_amd.ops import HipOps, EPI_GELU_DG, EPI_GELU, EPI_DGELU, EPI_MULAUX

hip = ctypes.CDLL("libamdhip64.so")
ops = HipOps(torch.bfloat16)
n = 202_400_000 // 256 * 256
p, g, m, v = (torch.randn(n, device="cuda") * 0.02 for _ in range(4))
v.abs_()
pc = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
flags = torch.ones(n // 256, dtype=torch.uint8, device="cuda")
ss = torch.zeros(1, device="cuda"); lrs = torch.tensor([1e-4, 0.1, 0.001, 0.0], device="cuda")
def adam(): ops.adamw(p, g, m, v, pc, flags, ss, lrs, n, 0.9, 0.999, 1e-6, 0.01, 1.0)

M, N, K = 16384, 3072, 768
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
Bn = torch.randn(K, N, device="cuda").to(torch.bfloat16)
C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); X = torch.zeros_like(C); bias = torch.randn(N, device="cuda")
def gemm(epi=EPI_GELU_DG): ops.gemm(A, B, C, bias, None, X, M, N, K, K, K, N, ldx=N, epilogue=epi)
def gemm_nn(epi): ops.gemm(A, Bn, C, None, None, X, M, N, K, K, N, N, ldx=N, b_kmajor=0, epilogue=epi)

def masked(ncu, lo=0):
    bits = [0] * 8
    for i in range(lo, lo + ncu):
        bits[i // 32] |= 1 << (i % 32)
    words = (ctypes.c_uint32 * 8)(*bits)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)

def timed(fn, reps, stream=None):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps): fn()
        e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

print("epilogue pairs, same shape, isolated: GELU %.1f + DGELU %.1f us | GELU_DG %.1f + MULAUX %.1f us" % (
    timed(lambda: gemm(EPI_GELU), 20) * 1e3, timed(lambda: gemm_nn(EPI_DGELU), 20) * 1e3,
    timed(lambda: gemm(EPI_GELU_DG), 20) * 1e3, timed(lambda: gemm_nn(EPI_MULAUX), 20) * 1e3))
print(f"adamw, all CUs: {timed(adam, 5) * 1e3:.0f} us;  gemm 16384x3072x768 (768 tiles) alone, all CUs: {timed(gemm, 20) * 1e3:.1f} us")
for ncu in (224, 192, 160, 128, 64):
    print(f"gemm on a {ncu}-CU stream: {timed(gemm, 20, masked(ncu)) * 1e3:.1f} us")
for ncu, lo in ((32, 0), (64, 0), (64, 192), (96, 0), (128, 0)):
    ms = masked(ncu, lo)
    t = timed(adam, 5, ms)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ms):
        a0.record()
        for _ in range(3): adam()
        a1.record()
    time.sleep(0.0005)
    s.record()
    for _ in range(30): gemm()
    e.record()
    torch.cuda.synchronize()
    print(f"adamw on CUs [{lo},{lo + ncu}): alone {t * 1e3:.0f} us ({34 * n / t / 1e9:.2f} TB/s); next to 30 GEMMs on the default stream: "
          f"adamw {a0.elapsed_time(a1) / 3 * 1e3:.0f} us, gemm {s.elapsed_time(e) / 30 * 1e3:.1f} us")
