"""profiles/<tag>/pmc_fetch.txt + pmc_write.txt -> per-kernel HBM traffic per launch and achieved rate (FETCH_SIZE doubled: gfx950
half-count of wide coalesced reads, MI355X_MICROARCH.md; WRITE_SIZE as reported -- FETCH x 2 / WRITE x 1 are calibrated on the adamw_kernel rows below, whose byte count is known: 16 B read and ~14.5 B written per element; durations of the same passes).
Usage: python tools/hbm_rates.py r02e > profiles/r02e/hbm_rates.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]


def table(fn):
    rows = {}
    for ln in open(os.path.join(ROOT, "profiles", tag, fn)):
        p = [x.strip() for x in ln.split("|")]
        if len(p) == 4 and p[1].isdigit():
            rows[p[0]] = (int(p[1]), float(p[2]), float(p[3]))
    return rows


rd, wr = table("pmc_fetch.txt"), table("pmc_write.txt")
print(f"{'kernel':70s} {'calls':>6s} {'avg us':>8s} {'read MB':>9s} {'write MB':>9s} {'TB/s':>6s}")
for k, (calls, ms, fetch) in sorted(rd.items(), key=lambda kv: -kv[1][1]):
    if k not in wr:
        continue
    r = fetch * 1024 * 2 / calls / 1e6
    w = wr[k][2] * 1024 / calls / 1e6
    us = ms * 1e3 / calls
    print(f"{k[:70]:70s} {calls:6d} {us:8.1f} {r:9.1f} {w:9.1f} {(r + w) / us:6.2f}")
