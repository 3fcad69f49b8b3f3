"""profiles/<tag>/pmc_fetch.txt + pmc_write.txt -> profiles/gemm_traffic.json (HBM bytes per GEMM launch, read by bench.py).
Usage: python tools/make_traffic_json.py r02b"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]


def grab(fn, key):
    txt = open(os.path.join(ROOT, "profiles", tag, fn)).read()
    calls = int(re.search(r"ALL gemm_bf16_\* launches.*?calls (\d+)", txt).group(1))
    val = float(re.search(rf"^\s+{key}: ([0-9.e+]+)", txt, re.M).group(1))
    return calls, val


calls, fetch = grab("pmc_fetch.txt", "FETCH_SIZE")
_, write = grab("pmc_write.txt", "WRITE_SIZE")
rd, wr = fetch * 1024 * 2, write * 1024
out = {"source": f"profiles/{tag}/pmc_fetch.txt + pmc_write.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                 "bench.py --steps 3 --warmup 2 --single-stream)",
       "gemm_launches": calls, "fetch_kib": fetch, "write_kib": write, "hbm_read_bytes": rd, "hbm_write_bytes": wr,
       "note": "all gemm_bf16_* kernels (ping-pong, grouped weight gradients, 128x128); FETCH_SIZE doubled (gfx950 half-count of wide "
               "coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported (x1) -- both factors calibrated on this build's own adamw_kernel rows: 202.4 M elements x 16 B read = 3238 MB against 3248 MB from FETCH_SIZE x 2, x 14.5 B written (master, two moments, bf16 copy, 12 % gradient clear) = 2935 MB against 3021 MB from WRITE_SIZE x 1",
       "bytes_per_launch": round((rd + wr) / calls)}
json.dump(out, open(os.path.join(ROOT, "profiles", "gemm_traffic.json"), "w"), indent=1)
print(out["bytes_per_launch"], "bytes per launch over", calls, "launches")
