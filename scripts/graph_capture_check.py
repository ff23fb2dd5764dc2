import sys, os
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch, time
from gfs3d.synthetic import synthetic_blocks
from model.dgcnn import DGCNN
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = DGCNN([[64, 64]] * 3, [512, 256], 9, k=20, return_edgeconvs=True).to(dev)
"""Capture-safety check of the C ABI (no allocation / host sync inside the entry points; the side-stream fork/join of the
training forward is event-based): the DGCNN eval forward and a whole training forward + backward are captured into
hipGraphs through torch.cuda.CUDAGraph and replayed.  Also shows where graphs pay: B = 1 is launch-bound."""
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = synthetic_blocks(BS, 2048, seed=0).to(dev)
# eval forward
net.eval()
with torch.no_grad():
    for _ in range(3): ref = net(x)[1].clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): net(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = net(x)[1]
    g.replay(); torch.cuda.synchronize()
    print("eval graph replay max|diff|", float((out - ref).abs().max()))
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50): net(x)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 50
    print(f"eval B={BS}: eager {te*1e3:.3f} ms, graph {tg*1e3:.3f} ms")
# training forward + backward with the side stream, captured
net.train()
xs = x.clone().requires_grad_(True)
opt_params = list(net.parameters())
def step():
    for p in opt_params: p.grad = None
    _, y = net(xs)
    y.mean().backward()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
ref_g = opt_params[0].grad.clone()
g2 = torch.cuda.CUDAGraph()
for p in opt_params: p.grad = None
with torch.cuda.graph(g2):
    _, y = net(xs)
    y.mean().backward()
g2.replay(); torch.cuda.synchronize()
print("train graph replay grad max|diff|", float((opt_params[0].grad - ref_g).abs().max()), "scale", float(ref_g.abs().max()))
