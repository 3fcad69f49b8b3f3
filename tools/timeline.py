"""Timeline analysis of one training step from a rocprofv3 --kernel-trace database (multi-stream run):
wall time, GPU-busy union, concurrency histogram, per-kernel totals, largest idle gaps.
Usage: python tools/timeline.py <results.db> [n_gaps]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("xl::", "")
    name = re.sub(r"<.*", "", name)
    return name[:60]


def main(path, ngaps=15):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("stream_id", "queue_id", "queue") if c in cols), None)
    sel = f"select name, start, end, {qcol if qcol else 0} from kernels order by start"
    rows = cur.execute(sel).fetchall()
    marks = [i for i, r in enumerate(rows) if "adamw" in r[0]]
    if len(marks) < 2:
        print("need at least two optimizer steps in the trace")
        return
    lo, hi = marks[-2] + 1, marks[-1] + 1
    step = rows[lo:hi]
    t0, t1 = min(r[1] for r in step), max(r[2] for r in step)
    print(f"columns: {cols}")
    print(f"step: {len(step)} kernels, wall {(t1 - t0) / 1e6:.3f} ms, queues {sorted(set(r[3] for r in step))}")
    ev = []
    for n, s, e, q in step:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    level, last, hist = 0, t0, {}
    for t, d in ev:
        hist[level] = hist.get(level, 0) + (t - last)
        level += d; last = t
    for k in sorted(hist):
        print(f"  {k} kernels in flight: {hist[k] / 1e6:8.3f} ms")
    agg = {}
    for n, s, e, q in step:
        a = agg.setdefault(short(n), [0, 0]); a[0] += 1; a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    print(f"sum of kernel durations {tot / 1e6:.3f} ms")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"  {k:60s} {c:5d} {t / 1e6:8.3f} ms {t / c / 1e3:8.1f} us")
    for q in sorted(set(r[3] for r in step)):
        rs = [r for r in step if r[3] == q]
        print(f"  queue {q}: {len(rs)} kernels, busy {sum(r[2] - r[1] for r in rs) / 1e6:.3f} ms")
    # idle gaps of the whole GPU
    gaps, cur_end, prev = [], t0, None
    for n, s, e, q in step:
        if s > cur_end:
            gaps.append((s - cur_end, prev, short(n)))
        if e > cur_end:
            cur_end, prev = e, short(n)
    print(f"idle (no kernel running): {sum(g[0] for g in gaps) / 1e6:.3f} ms in {len(gaps)} gaps")
    for g, a, b in sorted(gaps, reverse=True)[:ngaps]:
        print(f"  {g / 1e3:7.1f} us between {a} -> {b}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 15)
