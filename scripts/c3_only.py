import os, sys
ROOT = "/root/repo"
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    sys.path.insert(0, _p)
import torch, bench
print(os.environ.get("GFS3D_KM_IMG"), bench.c3_end_to_end(torch.device("cuda:0")))
