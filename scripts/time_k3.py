"""K3 measurements (BASELINE config 4 shapes): geometric-word assignment (C=192, K_w=200) and
get_pred (C=128, K=14) at B=32 x 2048, device time from the library's hipEvent brackets."""
import sys, os, ctypes, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch
from gfs3d import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
B, N = 32, 2048
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); L.gfs3d_profile_enable(1)
    for _ in range(iters): fn()
    torch.cuda.synchronize(); L.gfs3d_profile_enable(0)
    ms, n = ctypes.c_double(0), ctypes.c_longlong(0)
    L.gfs3d_profile_read(3, ctypes.byref(ms), ctypes.byref(n))
    return ms.value / n.value
feat = torch.randn(B, 192, N, device=dev); gp = torch.randn(200, 192, device=dev)
t = timed(lambda: ops.word_assign_raw(feat, gp, 10.0))
flops = 2.0 * B * N * 192 * 200; byts = 4.0 * B * N * (192 + 200 + 200) + 4 * B * N
print(json.dumps({"kernel": "proto_mfma_kernel<8,softmax> word assign C=192 Kw=200", "ms": round(t, 4),
                  "TFLOPps": round(flops / t / 1e9, 2), "algorithmic_GBps": round(byts / t / 1e6, 1)}))
f2 = torch.randn(B, 128, N, device=dev); pr = torch.randn(14, 128, device=dev)
t = timed(lambda: ops.proto_cosine_raw(f2, pr, 10.0))
flops = 2.0 * B * N * 128 * 14; byts = 4.0 * B * N * (128 + 14)
print(json.dumps({"kernel": "proto_mfma_kernel<1,logits> get_pred C=128 K=14", "ms": round(t, 4),
                  "TFLOPps": round(flops / t / 1e9, 2), "algorithmic_GBps": round(byts / t / 1e6, 1)}))
pts = torch.randn(300000, 192, device=dev); cent = torch.randn(200, 192, device=dev)
for _ in range(2): ops.kmeans_assign(pts, cent)
torch.cuda.synchronize(); L.gfs3d_profile_enable(1)
for _ in range(5): ops.kmeans_assign(pts, cent)
torch.cuda.synchronize(); L.gfs3d_profile_enable(0)
ms, n = ctypes.c_double(0), ctypes.c_longlong(0); L.gfs3d_profile_read(5, ctypes.byref(ms), ctypes.byref(n))
t = ms.value / n.value
print(json.dumps({"kernel": "kmeans_assign_kernel M=300000 C=192 K=200", "ms": round(t, 4),
                  "TFLOPps": round(2.0 * 300000 * 192 * 200 / t / 1e9, 2)}))
