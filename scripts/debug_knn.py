import sys, numpy as np, torch
sys.path.insert(0, "gfs-3dseg_gws_amd"); sys.path.insert(0, ".")
from gfs3d import ops
from oracle import c_oracle
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (B, C, N, k) in ((1, 64, 256, 20), (1, 9, 256, 20), (1, 64, 64, 5)):
    x = torch.randn(B, C, N, generator=g)
    i32 = ops.knn_topk(x.to(dev), k, want_i64=False, want_i32=True).cpu().numpy()
    ref = c_oracle.knn(x.numpy(), k)
    bad = (i32 != ref).any(-1)
    print(B, C, N, k, "bad rows", int(bad.sum()), "of", bad.size)
    if bad.any():
        r = np.argwhere(bad)[0]
        print(" row", r, "mine", i32[r[0], r[1]], "ref", ref[r[0], r[1]])
        print(" bad row ids", np.argwhere(bad)[:40, 1])
