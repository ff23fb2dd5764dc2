"""A/B inside one process: bench.py's configs[2] flow (extra.c3_end_to_end: 2.1 M x 192 EdgeConv feature points, 200 centroids, 300
Lloyd iterations) with the host one iteration ahead of the Lloyd loop's decisions (get_basis._RUN_AHEAD) and with the loop that reads
its three scalars every iteration, alternating."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    sys.path.insert(0, _p)
import torch
import bench
import get_basis
dev = torch.device("cuda:0")
bench.c3_end_to_end(dev)            # warm-up
for rep in range(3):
    for mode in (True, False):
        get_basis._RUN_AHEAD = mode
        d = bench.c3_end_to_end(dev)
        print(f"run-ahead {mode}: total {d['total_ms']} ms, lloyd {d['stage_ms']['lloyd']} ms, {d['lloyd_ms_per_iteration']} ms per iteration", flush=True)
