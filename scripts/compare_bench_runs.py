"""Two bench.py JSON lines from ONE box: every `ms`-like figure side by side and their relative difference (verdict r5 item 4:
the side measurements of two consecutive runs agree within 3 %).  python scripts/compare_bench_runs.py a.json b.json"""
import json, sys
a, b = (json.loads(open(p).read().strip().splitlines()[-1]) for p in sys.argv[1:3])
rows = []
def walk(x, y, path):
    if isinstance(x, dict) and isinstance(y, dict):
        for k in x:
            if k in y:
                walk(x[k], y[k], path + [k])
    elif isinstance(x, (int, float)) and isinstance(y, (int, float)) and not isinstance(x, bool):
        key = path[-1]
        if key in ("ms", "ms_per_step", "total_ms", "lloyd_ms_per_iteration", "avg_launch_ms") or key.endswith("_ms") and "stage" in "".join(path):
            if path[0] != "cpu_baseline" and "min_max" not in key:
                rows.append((".".join(path), x, y, abs(x - y) / max(abs(x), abs(y), 1e-12)))
walk(a, b, [])
worst = 0.0
for p, x, y, d in rows:
    worst = max(worst, d)
    print(f"{p:70s} {x:10.4f} {y:10.4f}  {100 * d:5.2f} %")
print(f"{len(rows)} figures, largest difference {100 * worst:.2f} %")
