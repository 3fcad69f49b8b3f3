"""Summarise a rocprofv3 --kernel-trace sqlite (rocpd) database: per-kernel count / total / average duration."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("xl::", "")
    return name[:110]


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, (end - start) from kernels").fetchall()
    agg = {}
    for n, d in rows:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += d
    tot = sum(v[1] for v in agg.values())
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:110s} {c:7d} {t / 1e6:10.3f} {t / c / 1e3:9.2f} {100 * t / tot:6.2f}")
    print(f"{'TOTAL':110s} {sum(v[0] for v in agg.values()):7d} {tot / 1e6:10.3f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
