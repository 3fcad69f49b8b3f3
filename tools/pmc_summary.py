"""Per-kernel sums of rocprofv3 --pmc counters from a rocpd sqlite database (pmc_events view)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("xl::", "")[:100]


def main(path, top=14):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events").fetchall()
    agg, seen = {}, set()
    for name, disp, dur, cn, cv in rows:
        k = short(name)
        a = agg.setdefault(k, {"calls": 0, "ns": 0})
        if (disp,) not in seen:
            seen.add((disp,))
            a["calls"] += 1
            a["ns"] += dur
        a[cn] = a.get(cn, 0.0) + cv
    counters = sorted({c for a in agg.values() for c in a if c not in ("calls", "ns")})
    print("kernel | calls | total_ms | " + " | ".join(counters))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"])[:top]:
        print(f"{k} | {a['calls']} | {a['ns'] / 1e6:.3f} | " + " | ".join(f"{a.get(c, 0):.4g}" for c in counters))
    g = {c: sum(a.get(c, 0) for k, a in agg.items() if "gemm_bf16" in k) for c in counters}
    ns = sum(a["ns"] for k, a in agg.items() if "gemm_bf16" in k)
    calls = sum(a["calls"] for k, a in agg.items() if "gemm_bf16" in k)
    print(f"\nALL gemm_bf16_* launches (ping-pong, grouped, 128x128): calls {calls}, total {ns / 1e6:.3f} ms")
    for c in counters:
        print(f"  {c}: {g[c]:.6g}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g and g["GRBM_GUI_ACTIVE"]:
        # MFMA busy cycles are summed over the 1024 SIMDs of the chip; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        # (it reads 8 x wall cycles: 3.46e9 for 176 ms), so wall cycles = GRBM_GUI_ACTIVE / 8
        print(f"  MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs) = "
              f"{g['SQ_VALU_MFMA_BUSY_CYCLES'] / (g['GRBM_GUI_ACTIVE'] / 8 * 1024):.4f}")
        print(f"  effective clock = GRBM_GUI_ACTIVE/8 / time = {g['GRBM_GUI_ACTIVE'] / 8 / ns:.3f} GHz")
    if "FETCH_SIZE" in g:
        print(f"  HBM read bytes (FETCH_SIZE KiB x 1024 x 2, gfx950 half-count correction): {g['FETCH_SIZE'] * 1024 * 2:.6g}")
    if "WRITE_SIZE" in g:
        print(f"  HBM write bytes (WRITE_SIZE KiB x 1024, uncalibrated): {g['WRITE_SIZE'] * 1024:.6g}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14)
