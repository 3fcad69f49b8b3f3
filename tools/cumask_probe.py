"""Can the optimizer pass hide under the next step's forward?  AdamW on a CU-masked stream (hipExtStreamCreateWithCUMask): its
HBM rate as a function of the number of CUs it may use, and what a chip-filling GEMM loses while it runs next to it.
Usage: python tools/cumask_probe.py"""
import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from xlxmert_amd.ops import HipOps, EPI_GELU_DG

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
C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); X = torch.zeros_like(C); bias = torch.randn(N, device="cuda")
def gemm(): ops.gemm(A, B, C, bias, None, X, M, N, K, K, K, N, ldx=N, epilogue=EPI_GELU_DG)

def masked_stream(pattern_bits):
    words = (ctypes.c_uint32 * 8)(*([pattern_bits] * 8))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)

def timed(fn, reps, stream=None):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(2): fn()
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps): fn()
        e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

print(f"adamw, all CUs: {timed(adam, 5) * 1e3:.0f} us ({34 * n / timed(adam, 5) / 1e9:.2f} TB/s at 34 B/element)")
print(f"gemm 16384x3072x768 GELU alone: {timed(gemm, 20) * 1e3:.1f} us")
for name, bits in (("32 CUs (1 of 8)", 0x01010101), ("64 CUs (1 of 4)", 0x11111111), ("96 CUs (3 of 8)", 0x49494949 & 0xffffffff),
                   ("128 CUs (1 of 2)", 0x55555555), ("low 64 bits = 64 CUs", None)):
    if bits is None:
        words = (ctypes.c_uint32 * 8)(0xffffffff, 0xffffffff, 0, 0, 0, 0, 0, 0)
        st = ctypes.c_void_p(); assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words) == 0
        ms = torch.cuda.ExternalStream(st.value)
    else:
        ms = masked_stream(bits)
    t = timed(adam, 5, ms)
    # GEMM on the default stream while AdamW runs on the masked one
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ms):
        a0.record()
        for _ in range(3): adam()
        a1.record()
    time.sleep(0.0005)
    s.record()
    for _ in range(20): gemm()
    e.record()
    torch.cuda.synchronize()
    print(f"{name}: adamw alone {t * 1e3:.0f} us ({34 * n / t / 1e9:.2f} TB/s); next to 20 GEMMs: adamw {a0.elapsed_time(a1) / 3 * 1e3:.0f} us, "
          f"gemm {s.elapsed_time(e) / 20 * 1e3:.1f} us")
