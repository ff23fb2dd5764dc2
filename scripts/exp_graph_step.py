"""Would replaying the c2 training step as ONE hipGraph (forward + backward + fused Adam) beat the eager step of bench.py?
Same model, input and optimiser as bench.py's c2 workload; prints eager and graph ms per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch
from gfs3d.synthetic import synthetic_blocks
from model.dgcnn import DGCNN
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = synthetic_blocks(B, 2048, seed=0).to(dev)
net = DGCNN([[64, 64]] * 3, [512, 256], 9, k=20, return_edgeconvs=True).to(dev).train()
params = list(net.parameters())
opt = torch.optim.Adam(params, lr=1e-3, fused=True, capturable=True)

def step():
    for p in params: p.grad = None
    _, out = net(x)
    out.mean().backward()
    opt.step()

def timeit(fn, n=60):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print(f"eager (capturable Adam): {timeit(step):.4f} ms")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
for p in params: p.grad = None
with torch.cuda.graph(g):
    _, out = net(x)
    out.mean().backward()
    opt.step()
print(f"graph replay: {timeit(g.replay):.4f} ms")
print(f"eager again: {timeit(step):.4f} ms")
