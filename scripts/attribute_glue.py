"""Which Python lines launch the small tensor-library kernels of a training step?  torch.profiler with stacks over a few steps of a bench
workload; prints, per (operator, source line), launches per step and device time.  python scripts/attribute_glue.py [c2|c4|c5]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    sys.path.insert(0, _p)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from gfs3d import dist as gdist
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda:0")
cfg = dict(bench.WORKLOADS[wl])
step, net, bucket, x = bench.build_step(cfg, dev, 0, 1, False, gdist)
for _ in range(5): step()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N): step()
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_stack_n=12):
    t = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if not ev.key.startswith("aten::") or t <= 0:
        continue
    frame = next((f for f in ev.stack if ("gfs-3dseg_gws_amd" in f or "bench.py" in f) and "attribute_glue" not in f), ev.stack[0] if ev.stack else "?")
    rows.append((t / N, ev.count / N, ev.key, frame.split("/")[-1][:80]))
rows.sort(reverse=True)
tot = 0.0
for t, n, name, frame in rows[:45]:
    print(f"{n:6.2f}/step {t:8.1f} us/step  {name:28s} {frame}")
    tot += t
print(f"listed: {tot:.1f} us per step")
