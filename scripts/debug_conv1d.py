import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch
from model.dgcnn import conv1d
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, N) in [(4, 256), (32, 2048)]:
    m = conv1d(192, [512, 256]).to(dev).train()
    x = torch.randn(B, 192, N, device=dev)
    cot = torch.randn(B, 256, N, device=dev)
    def run(path):
        for p in m.parameters(): p.grad = None
        xx = x.clone().requires_grad_(True)
        y = m(xx) if path == "new" else m.layer(xx)
        (y * cot).sum().backward()
        return y.detach(), xx.grad, [p.grad.clone() for p in m.parameters()]
    # double reference on GPU
    m64 = conv1d(192, [512, 256]).to(dev).double().train(); m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    y64 = m64.layer(x64); (y64 * cot.double()).sum().backward()
    g64 = [p.grad for p in m64.parameters()]
    for path in ("old", "new"):
        y, gx, gp = run(path)
        rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
        print(B, N, path, "y", rel(y, y64.detach()), "dx", rel(gx, x64.grad), "dparams", [f"{rel(a, b):.1e}" for a, b in zip(gp, g64)])
        torch.cuda.synchronize(); t = time.time()
        for _ in range(5): run(path)
        torch.cuda.synchronize(); print("   time per fwd+bwd ms:", (time.time() - t) / 5 * 1e3)
