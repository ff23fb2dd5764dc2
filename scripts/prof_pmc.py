"""Sum the PMC counters of a rocprofv3 --pmc run (rocpd database) per kernel name."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
rows = db.execute("select * from counters_collection")
ix = {c: i for i, c in enumerate(cols)}
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    name = r[ix["kernel_name"]] if "kernel_name" in ix else r[ix["name"]]
    if filt and filt not in name: continue
    acc[name][r[ix["counter_name"]]] += r[ix["value"]]
    cnt[name].add(r[ix["dispatch_id"]])
for name, d in acc.items():
    print(name[:100], "dispatches", len(cnt[name]))
    for k, v in sorted(d.items()):
        print(f"    {k:32s} {v:.4e}  per-dispatch {v / max(len(cnt[name]), 1):.4e}")
