"""Turn the rocprofv3 (rocpd sqlite) outputs of `bench.py` into the committed summaries under profiles/.

  python scripts/collect_profiles.py <tag> <kernel_trace.db> <pmc_sq.db> <pmc_fetch.db> <pmc_write.db> [steps of the PMC runs] [what ran]

  profiles/<tag>_bench_kernel_stats.csv   per-kernel calls / total / average (== rocprofv3 --kernel-trace --stats)
  profiles/<tag>_hbm_traffic.json         FETCH_SIZE (x2, gfx950 note) + WRITE_SIZE per launch, separate passes
  profiles/<tag>_pmc_sq.txt               SQ counters per kernel, per launch
"""
import collections, json, os, re, sqlite3, sys

tag, trace_db, sq_db, fetch_db, write_db = sys.argv[1:6]
pmc_steps = int(sys.argv[6]) if len(sys.argv) > 6 else 0           # training steps the profiled command ran (warm-up + timed + instrumented)
what = sys.argv[7] if len(sys.argv) > 7 else "bench.py --no-cpu-baseline --steps 3 --warmup 2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].replace("gfs3d::", "")


def counters(db_path):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in db.execute("select * from counters_collection"):
        k = short(r[ix["kernel_name"]])
        acc[k][r[ix["counter_name"]]] += r[ix["value"]]
        cnt[k].add(r[ix["dispatch_id"]])
    return acc, {k: len(v) for k, v in cnt.items()}


db = sqlite3.connect(trace_db)
with open(os.path.join(out, f"{tag}_bench_kernel_stats.csv"), "w") as f:
    f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage"\n')
    for name, calls, tot, avg, pct in db.execute("select * from top_kernels"):
        f.write(f'"{short(name)}",{calls},{int(tot * 1e3)},{avg * 1e3:.1f},{pct:.3f}\n')

fa, fn = counters(fetch_db)
wa, wn = counters(write_db)
traffic = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `" + what + "`, "
                    "per-launch averages over ALL launches of a kernel name; `_steps` = training steps of the profiled run (bench.py divides a "
                    "kernel group's bytes by it). Units KB in the tool; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of "
                    "coalesced reads). WRITE_SIZE is exact for stores and float atomics."}
if pmc_steps:
    traffic["_steps"] = pmc_steps
for k in fa:
    n = max(fn.get(k, 1), 1)
    fetch_kb = fa[k].get("FETCH_SIZE", 0.0) / n
    write_kb = wa.get(k, {}).get("WRITE_SIZE", 0.0) / max(wn.get(k, 1), 1)
    traffic[k] = {"launches": n, "fetch_kb_reported": round(fetch_kb, 1), "write_kb": round(write_kb, 1),
                  "traffic_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024)}
json.dump(traffic, open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w"), indent=1)

sa, sn = counters(sq_db)
with open(os.path.join(out, f"{tag}_pmc_sq.txt"), "w") as f:
    names = sorted({c for k in sa for c in sa[k]})
    f.write("rocprofv3 --pmc " + " ".join(names) + "\non `" + what + "`; per-launch averages\n")
    f.write("kernel".ljust(46) + "".join(n.replace("SQ_", "")[:15].rjust(16) for n in names) + "\n")
    for k in sorted(sa, key=lambda k: -sa[k].get("SQ_WAVE_CYCLES", 0)):
        if sa[k].get("SQ_WAVE_CYCLES", 0) / max(sn[k], 1) < 1e6:
            continue
        f.write(k[:45].ljust(46) + "".join(f"{sa[k].get(n, 0) / sn[k]:16.4g}" for n in names) + "\n")
print("wrote", tag)
