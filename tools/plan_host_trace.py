"""Host-side timing of one launch-plan replay inside bench.py's timed loop (XL_PLAN_TRACE, csrc/plan.hip): which calls make the
enqueueing thread wait, and when each stream receives its work.
Usage: XL_PLAN_TRACE=/tmp/plan.txt XL_PLAN_TRACE_RUN=12 python bench.py --steps 12 --warmup 4 --no-extra --no-cpu-baseline
       python tools/plan_host_trace.py /tmp/plan.txt"""
import sys
rows = [l.split() for l in open(sys.argv[1])]
rows = [(int(r[0]), r[1], float(r[2]), float(r[3]), r[4]) for r in rows]
tot = rows[-1][2] + rows[-1][3]
print(f"{len(rows)} calls, host {tot / 1e3:.3f} ms")
streams = {}
for i, fn, t, d, last in rows:
    if fn in ("xl_stream_fork", "xl_ctx_bind", "xl_set_deferred_reduce", "xl_set_step_seed_ptr"):
        continue
    streams.setdefault(last, []).append((t, d, fn))
for s, v in streams.items():
    print(f"stream {s}: {len(v)} launches, first at {v[0][0] / 1e3:.3f} ms, last at {v[-1][0] / 1e3:.3f} ms, host time in them {sum(x[1] for x in v) / 1e3:.3f} ms")
slow = sorted(rows, key=lambda r: -r[3])[:25]
print("slowest calls:")
for i, fn, t, d, last in sorted(slow):
    print(f"  #{i:4d} at {t / 1e3:8.3f} ms  {d:9.1f} us  {fn}  (last arg {last})")
by = {}
for i, fn, t, d, last in rows:
    a = by.setdefault(fn, [0, 0.0]); a[0] += 1; a[1] += d
for fn, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {fn:28s} {c:4d} calls {d / 1e3:8.3f} ms  {d / c:7.1f} us each")
