import sys, copy, torch
sys.path.insert(0, "gfs-3dseg_gws_amd")
from gfs3d import ops
from model.dgcnn import conv1d
dev = torch.device("cuda:0")
torch.manual_seed(11)
conv = conv1d(192, [512, 256]).to(dev).train()
ref = copy.deepcopy(conv)
ref64 = copy.deepcopy(conv).double()
B = 32
xs = [torch.randn(B, 2048, 64, device=dev).requires_grad_(True) for _ in range(3)]
xr = [t.detach().clone().requires_grad_(True) for t in xs]
x64 = [t.detach().double().requires_grad_(True) for t in xs]
cot = torch.randn(B, 256, 2048, device=dev)
y = ops.pointwise_stack(xs, conv.fused_parts(), True)
y_ref = ref(torch.cat([t.transpose(1, 2) for t in xr], dim=1))
y64 = ref64(torch.cat([t.transpose(1, 2) for t in x64], dim=1))
print("y err ours/torch vs fp64", float((y - y64).abs().max()), float((y_ref - y64).abs().max()))
y.backward(cot); y_ref.backward(cot); y64.backward(cot.double())
for name, a in (("ours", xs), ("torch32", xr)):
    g = torch.cat([t.grad for t in a], -1).double()
    g64 = torch.cat([t.grad for t in x64], -1)
    e = (g - g64).abs()
    mx = float(g64.abs().max())
    bad_pts = (e.max(-1).values > 1e-4 * mx)
    print(name, "max|g|", mx, "maxerr", float(e.max()), "bad elems", int((e > 1e-4 * mx).sum()), "bad points", int(bad_pts.sum()),
          "median err", float(e.median()))
for (n1, p1), (_, p2), (_, p3) in zip(conv.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
    mx = float(p3.grad.abs().max())
    print(n1, "ours", float((p1.grad - p3.grad).abs().max()) / mx, "torch32", float((p2.grad - p3.grad).abs().max()) / mx)
