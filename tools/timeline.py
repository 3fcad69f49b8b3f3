"""Timeline analysis of one training step from a rocprofv3 --kernel-trace database (multi-stream run).
CAVEAT (measured, round 3): with the tracer attached a stream whose work was enqueued AFTER another stream's backlog starts late --
the trace shows the visual stack beginning 1.5 ms after the language stack and the relational-stack backward waiting 2 ms for the
language-stack backward, in plan, eager, resident-input and 8-queue runs alike.  HIP events recorded through the C ABI inside the
replayed plan (tools/overlap_probe.py, no tracer) show both pairs starting within 0.2 ms / 7 us of each other.  Per-kernel
durations and per-queue totals from the trace are sound; cross-queue ORDER is not.
wall time, GPU-busy union, concurrency histogram, per-kernel totals, largest idle gaps.
Usage: python tools/timeline.py <results.db> [n_gaps] [step]   (step: index of the optimizer step to analyse, counted over the whole
trace by its sumsq kernel; default -2 = the last complete one -- with bench.py that is one of the host-enqueue probe steps, which
run against an EMPTY queue; pick one inside the timed loop, e.g. warmup + 3)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("xl::", "")
    name = re.sub(r"<.*", "", name)
    return name[:60]


def main(path, ngaps=15, which=-2):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("stream_id", "queue_id", "queue") if c in cols), None)
    sel = f"select name, start, end, {qcol if qcol else 0} from kernels order by start"
    rows = cur.execute(sel).fetchall()
    marks = [i for i, r in enumerate(rows) if "sumsq" in r[0]]   # one per step (AdamW runs group by group)
    if len(marks) < 2:
        print("need at least two optimizer steps in the trace")
        return
    lo, hi = marks[which] + 1, marks[which + 1] + 1
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
    # how long is at least one dense contraction in flight (the chip near its power limit), how long only row kernels, how long nothing
    gev = []
    for n, s, e, q in step:
        g = 1 if "gemm" in n else 0
        gev.append((s, 1, g)); gev.append((e, -1, -g))
    gev.sort()
    lv = gl = 0
    last, t_gemm, t_rows = t0, 0, 0
    for t, d, g in gev:
        if gl > 0: t_gemm += t - last
        elif lv > 0: t_rows += t - last
        lv += d; gl += g; last = t
    print(f"  >= 1 contraction in flight {t_gemm / 1e6:.3f} ms, only other kernels {t_rows / 1e6:.3f} ms, nothing {(t1 - t0 - t_gemm - t_rows) / 1e6:.3f} ms")
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
    # coarse Gantt chart: per 0.25 ms bin and queue, the busy share (0-9, '.' = idle) and the kernel that owns most of the bin
    qs = sorted(set(r[3] for r in step))
    binw = 250000
    nb = int((t1 - t0 + binw - 1) // binw)
    print("\nGantt (0.25 ms bins; digit = tenths of the bin busy on that queue; then the dominant kernel per queue)")
    for b in range(nb):
        lo_t, hi_t = t0 + b * binw, t0 + (b + 1) * binw
        cells, names = [], []
        for q in qs:
            busy, own = 0, {}
            for n, s_, e_, qq in step:
                if qq != q or e_ <= lo_t or s_ >= hi_t:
                    continue
                d = min(e_, hi_t) - max(s_, lo_t)
                busy += d
                own[short(n)] = own.get(short(n), 0) + d
            cells.append("." if busy == 0 else str(min(9, int(10 * busy / binw))))
            names.append(max(own, key=own.get)[:26] if own else "-")
        print(f"  {b * 0.25:6.2f} ms  {' '.join(cells)}   " + " | ".join(f"{n:26s}" for n in names))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 15, int(sys.argv[3]) if len(sys.argv) > 3 else -2)
