"""Wall-clock of the CAPL eval forward (capl.forward(eval_model=True): backbone + heads + refinement + word re-weighting), B = 32 x 2048."""
import sys, os, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch
from gfs3d.synthetic import synthetic_blocks
from model.capl import mpti_net_Point_GeoAsWeight_v2
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
a = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20, base_widths=[128, 64],
                          output_dim=64, eval_weight=1.2)
torch.manual_seed(0)
gp = torch.randn(200, 192, generator=torch.Generator().manual_seed(2)).to(dev)
m = mpti_net_Point_GeoAsWeight_v2(classes=13, criterion=torch.nn.CrossEntropyLoss(ignore_index=255), args=a, base_num=7, gp=gp, energy=0.9).to(dev).eval()
x = synthetic_blocks(B, 2048, seed=1).to(dev)
y = torch.randint(0, 13, (B, 2048), generator=torch.Generator().manual_seed(3)).to(dev)
gened = torch.randn(13, 128, device=dev)
bc, nc = (torch.rand(7, 200, device=dev) > 0.5).float(), (torch.rand(6, 200, device=dev) > 0.5).float()
def fwd():
    with torch.no_grad():
        return m(x, y=y, gened_proto=gened, eval_model=True, base_class_coding=bc, novel_class_coding=nc)
for _ in range(50): fwd()                  # (a short run measures the clock ramp of a cold device, not the forward: 6 ms instead of 1.4)
torch.cuda.synchronize(); t = time.time()
for _ in range(100): fwd()
torch.cuda.synchronize(); dt = (time.time() - t) / 100
print(f"CAPL eval forward B={B}: {dt * 1e3:.3f} ms = {B / dt:.0f} blocks/s")
