"""CAPL train step (config c4: S3DIS shapes, 13+1 classes, K_w = 200, B = 32 x 2048): time and kernel breakdown."""
import sys, time, types, random, torch
sys.path.insert(0, "gfs-3dseg_gws_amd"); sys.path.insert(0, ".")
from model.capl import mpti_net_Point_GeoAsWeight_v2
from gfs3d.synthetic import synthetic_blocks
dev = torch.device("cuda:0")
a = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20,
                          base_widths=[128, 64], output_dim=64, eval_weight=1.2)
torch.manual_seed(0)
gp = torch.randn(200, 192, generator=torch.Generator().manual_seed(2))
m = mpti_net_Point_GeoAsWeight_v2(classes=13, args=a, base_num=7, gp=gp.to(dev), energy=0.9).to(dev).train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = synthetic_blocks(B, 2048, seed=0).to(dev)
y = torch.randint(0, 8, (B, 2048), generator=torch.Generator().manual_seed(1)).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
def step():
    for p in m.parameters(): p.grad = None
    random.seed(0)
    pred, loss = m(x, y=y)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("CAPL train step B=%d: %.2f ms  (%.0f blocks/s)" % (B, dt * 1e3, B / dt))
