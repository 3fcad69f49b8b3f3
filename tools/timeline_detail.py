"""Kernel-by-kernel listing of a window of one step (rocprofv3 --kernel-trace database): start / end relative to the step, stream
id, hardware queue id, name.  Usage: python tools/timeline_detail.py <results.db> <from_ms> <to_ms> [step]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
lo_ms, hi_ms = float(sys.argv[2]), float(sys.argv[3])
rows = db.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "sumsq" in r[0]]
w = int(sys.argv[4]) if len(sys.argv) > 4 else -2
step = rows[marks[w] + 1:marks[w + 1] + 1]
t0 = min(r[1] for r in step)
pairs = sorted(set((r[3], r[4]) for r in step))
print("stream_id -> queue_id pairs in this step:", pairs)
for n, s, e, st, q in step:
    a, b = (s - t0) / 1e6, (e - t0) / 1e6
    if b < lo_ms or a > hi_ms:
        continue
    n = re.sub(r"\(.*$", "", n).replace("void ", "").replace("xl::", "")
    print(f"{a:8.3f} {b:8.3f}  s{st} q{q}  {n[:90]}")
