import os, sys, time
ROOT = "/root/repo"
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    sys.path.insert(0, _p)
import torch, bench
dev = torch.device("cuda:0")
t0 = time.time()
for i in range(3):
    d = bench.c3_end_to_end(dev)
    print(i, round(time.time() - t0, 1), d["total_ms"], d["stage_ms"]["lloyd"], d["lloyd_ms_per_iteration"], flush=True)
