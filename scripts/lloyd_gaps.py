"""Distribution of the Lloyd iteration periods in a rocprofv3 kernel trace of scripts/c3_thrice.py: per c3 flow (separated by gaps > 20 ms
between consecutive screened-assignment launches) the number of iterations, median / mean period and where the long ones are.
  python scripts/lloyd_gaps.py <r_results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
marks = [r[1] for r in rows if "kmeans_screen_res" in r[0] or "kmeans_assign_x6_kernelILi7ELi3ELb0" in r[0]]
calls, cur = [], [marks[0]]
for a, b in zip(marks[:-1], marks[1:]):
    if b - a > 20e6:
        calls.append(cur); cur = []
    cur.append(b)
calls.append(cur)
for i, c in enumerate(calls):
    if len(c) < 20:
        continue
    d = sorted((b - a) / 1e3 for a, b in zip(c[:-1], c[1:]))
    seq = [(b - a) / 1e3 for a, b in zip(c[:-1], c[1:])]
    big = [(k, round(v)) for k, v in enumerate(seq) if v > 1.5 * d[len(d) // 2]]
    print(f"flow {i}: {len(c)} assignments, period median {d[len(d)//2]:.0f} us, mean {sum(d)/len(d):.0f} us, total {sum(d)/1e3:.1f} ms; "
          f"{len(big)} periods > 1.5 x median, their sum {sum(v for _, v in big)/1e3:.1f} ms; first of them {big[:12]}")
