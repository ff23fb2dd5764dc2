"""Could the kNN screen of EdgeConv layer l >= 2 skip its threshold sweep by taking the threshold from the PREVIOUS layer's graph?
A lower bound on the K-th best score of a query is the worst score, under the new features, of its K previous neighbours (K distinct
candidates reach it); every candidate above that bound (minus the screen's error band) would have to be listed.  Measured here: how
long those lists get (the kernel's lists hold 64; longer = exhaustive fallback), against the lists of the shipped rule (K-th best).
Synthetic recipe of the bench and a structured cloud, seeded DGCNN in train() mode (batch statistics, like the step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd"), os.path.join(ROOT, "scripts")):
    sys.path.insert(0, _p)
import torch
from gfs3d import ops
from gfs3d.synthetic import synthetic_blocks
from model.dgcnn import DGCNN

dev = torch.device("cuda:0")
B, N, K = 8, 2048, 20
torch.manual_seed(0)
net = DGCNN([[64, 64]] * 3, [512, 256], 9, k=K, return_edgeconvs=True).to(dev).train()


def scores(f):                       # (B,C,N) -> (B,N,N) negative squared distances (the kNN's ranking value)
    inner = torch.matmul(f.transpose(1, 2), f)
    sq = (f * f).sum(1)
    return 2 * inner - sq[:, :, None] - sq[:, None, :]


def run(label, x):
    x = x.to(dev)
    with torch.no_grad():
        feats = net.edgeconv_features(x)
    ins = [x] + list(feats[:-1])
    prev = None
    print(label)
    for li, inp in enumerate(ins):
        s = scores(inp.double())
        idx = ops.knn_topk(inp, K)
        kth = torch.gather(s, 2, idx)[:, :, -1]                                   # the K-th best score
        band = 2 * (1.25e-4 * ((inp.double() ** 2).sum(1).sqrt()[:, :, None] * (inp.double() ** 2).sum(1).sqrt().amax(1)[:, None, None]))
        own = (s >= (kth[:, :, None] - band)).sum(2).float()
        line = f"  layer {li + 1}: shipped rule list length mean {own.mean():.1f} p99 {own.flatten().quantile(0.99):.0f} max {own.max():.0f}"
        if prev is not None:
            lb = torch.gather(s, 2, prev).amin(2)                                  # worst of the previous neighbours under the new features
            ln = (s >= (lb[:, :, None] - band)).sum(2).float()
            ov = (prev[:, :, :, None] == idx[:, :, None, :]).any(3).float().mean()
            line += f" | from the previous graph: mean {ln.mean():.1f} median {ln.flatten().median():.0f} p90 {ln.flatten().quantile(0.9):.0f} share > 64: {(ln > 64).float().mean():.3f}; neighbour overlap {ov:.2f}"
        print(line)
        prev = idx


from _clouds import structured          # noqa: E402
run("synthetic (bench recipe)", synthetic_blocks(B, N, seed=0))
run("structured (box faces, colour per face)", structured(B, N, 1))
