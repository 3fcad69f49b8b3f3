"""HBM bytes per GEMM launch against its algorithmic bytes, shape by shape, inside the real step.
Inputs: the rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE databases of `bench.py --single-stream --gemm-list L.json` and L.json.
The last len(L) gemm_bf16_* dispatches of each database are the instrumented step's launches, in launch order.
Usage: python tools/gemm_overfetch.py <fetch.db> <write.db> <list.json>"""
import json, sqlite3, sys


def per_dispatch(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, dispatch_id, counter_name, counter_value from pmc_events order by dispatch_id").fetchall()
    out, idx = [], {}
    for name, disp, cn, cv in rows:
        if "gemm_bf16" not in name or cn != counter:
            continue
        if disp not in idx:
            idx[disp] = len(out)
            out.append([name, 0.0])
        out[idx[disp]][1] += cv
    return out


lst = json.load(open(sys.argv[3]))
n = len(lst)
rd = per_dispatch(sys.argv[1], "FETCH_SIZE")[-n:]
wr = per_dispatch(sys.argv[2], "WRITE_SIZE")[-n:]
assert len(rd) == n and len(wr) == n, (len(rd), len(wr), n)
agg = {}
for rec, (kn, f), (_, w) in zip(lst, rd, wr):
    key = tuple(rec["key"])
    a = agg.setdefault(key, dict(n=0, alg=0.0, rd=0.0, wr=0.0, us=0.0, blocks=set(), kern=kn.split("<")[0].split("::")[-1][:28]))
    a["n"] += 1; a["alg"] += rec["bytes"]; a["rd"] += f * 1024 * 2; a["wr"] += w * 1024; a["us"] += rec["us"]; a["blocks"].add(rec["block"][:1])
tot_alg = sum(a["alg"] for a in agg.values()); tot_hbm = sum(a["rd"] + a["wr"] for a in agg.values())
print(f"{n} launches: algorithmic {tot_alg / 1e9:.2f} GB, HBM {tot_hbm / 1e9:.2f} GB (reads x2-corrected {sum(a['rd'] for a in agg.values()) / 1e9:.2f} + writes "
      f"{sum(a['wr'] for a in agg.values()) / 1e9:.2f}), ratio {tot_hbm / tot_alg:.2f}")
print(f"{'shape (M, N, K, ak, bk, epi)':44s} {'n':>3s} {'alg MB':>8s} {'read MB':>8s} {'write MB':>8s} {'ratio':>6s} {'excess GB':>9s} {'us':>7s}  blocks")
for key, a in sorted(agg.items(), key=lambda kv: -(kv[1]["rd"] + kv[1]["wr"] - kv[1]["alg"])):
    c = a["n"]
    print(f"{str(key):44s} {c:3d} {a['alg'] / c / 1e6:8.1f} {a['rd'] / c / 1e6:8.1f} {a['wr'] / c / 1e6:8.1f} {(a['rd'] + a['wr']) / a['alg']:6.2f} "
          f"{(a['rd'] + a['wr'] - a['alg']) / 1e9:9.2f} {a['us'] / c:7.1f}  {''.join(sorted(a['blocks']))}")
