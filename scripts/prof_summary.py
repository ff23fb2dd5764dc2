"""Print the top kernels of a rocprofv3 rocpd database (calls, total ms, avg us)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
filt = sys.argv[3] if len(sys.argv) > 3 else ""
for name, calls, tot, avg, pct in db.execute("select * from top_kernels"):
    if filt and filt not in name: continue
    print(f"{name[:110]:110s} calls={calls:5d} tot_ms={tot/1e3:9.3f} avg_us={avg:9.1f} {pct:5.1f}%")
    n -= 1
    if n <= 0: break
