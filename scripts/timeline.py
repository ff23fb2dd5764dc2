"""GPU timeline of a rocprofv3 --kernel-trace run (rocpd database): per step window, the union of the busy intervals of
all streams, the sum of the kernel durations and the idle gaps - tells a launch-bound step from a device-bound one.
  python scripts/timeline.py <r_results.db> [marker kernel substring] [skip first N marker hits] [dump]
With a 4th argument the launches of one step are listed in start order (offset, duration, gap to the previous end, queue).
Steps are delimited by consecutive launches of the marker kernel (default: the Adam multi_tensor_apply kernel)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "multi_tensor_apply"
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[0]]
# the Adam step is two launches per step: keep the first of each close pair
steps = []
for i in marks:
    if not steps or rows[i][1] - rows[steps[-1]][1] > 500_000:
        steps.append(i)
steps = steps[skip:]
out = []
for a, b in zip(steps[:-1], steps[1:]):
    seg = rows[a:b]
    t0, t1 = seg[0][1], rows[b][1]
    iv = sorted((r[1], r[2]) for r in seg)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s_, e_ in iv[1:]:
        if s_ > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    busy += cur_e - cur_s
    out.append(((t1 - t0) / 1e6, busy / 1e6, sum(r[2] - r[1] for r in seg) / 1e6, len(seg), len({r[3] for r in seg})))
for o in out[:12]:
    print("step %.3f ms  busy(union) %.3f ms  sum %.3f ms  launches %d  queues %d" % o)
if out:
    n = len(out)
    print("mean: step %.3f  busy %.3f  sum %.3f  idle %.3f" % (sum(o[0] for o in out) / n, sum(o[1] for o in out) / n, sum(o[2] for o in out) / n,
                                                          sum(o[0] - o[1] for o in out) / n))

if len(sys.argv) > 4 and len(steps) > 2:
    a, b = steps[1], steps[3] if len(steps) > 3 else steps[2]
    t0, last_end = rows[a][1], rows[a][1]
    import re
    for name, st, en, q in rows[a:b]:
        short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:58]
        print("%9.1f us  %7.1f us  gap %6.1f  q%d  %s" % ((st - t0) / 1e3, (en - st) / 1e3, (st - last_end) / 1e3, q, short))
        last_end = max(last_end, en)
