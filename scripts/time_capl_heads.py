"""Device time of the torch-side tail of capl.forward (training): generate_fake_proto, get_pred x2, post_refine_proto_v2, the two
cross-entropies - forward and backward of each piece on a fixed point_feat, B = 32 x 2048."""
import sys, os, random, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch
from model.capl import mpti_net_Point_GeoAsWeight_v2
dev = torch.device("cuda:0")
B, N = 32, 2048
a = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20, base_widths=[128, 64],
                          output_dim=64, eval_weight=1.2)
torch.manual_seed(0)
gp = torch.randn(200, 192, device=dev)
m = mpti_net_Point_GeoAsWeight_v2(classes=13, criterion=torch.nn.CrossEntropyLoss(ignore_index=255), args=a, base_num=7, gp=gp, energy=0.9).to(dev).train()
pf0 = torch.randn(B, 128, N, device=dev)
y = torch.randint(0, 8, (B, N), device=dev)
nb = 7

def timed(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def piece(name, build):
    def f():
        pf = pf0.clone().requires_grad_(True)
        out = build(pf)
        out.backward()
    def ff():
        with torch.no_grad():
            build(pf0)
    print(f"{name:28s} fwd {timed(ff):7.3f} ms   fwd+bwd {timed(f):7.3f} ms")

def fake(pf):
    random.seed(0)
    p, _ = m.generate_fake_proto(x=pf[B // 2:], y=y[B // 2:], main_proto=m.main_proto.clone())
    return p.sum()
def loss1(pf):
    random.seed(0)
    p, _ = m.generate_fake_proto(x=pf[B // 2:], y=y[B // 2:], main_proto=m.main_proto.clone())
    return m.criterion(m.get_pred(pf, p, use_bg_proto=True), y)
def refine(pf):
    return m.post_refine_proto_v2(proto=m.main_proto.clone(), x=pf, point_feat=pf, use_bg_proto=True).sum()
def tail(pf):
    random.seed(0)
    ori, _ = m.generate_fake_proto(x=pf[B // 2:], y=y[B // 2:], main_proto=m.main_proto.clone())
    l1 = m.criterion(m.get_pred(pf, ori, use_bg_proto=True), y)
    r = m.post_refine_proto_v2(proto=m.main_proto.clone(), x=pf, point_feat=pf, use_bg_proto=True)
    r = torch.cat([r[:, :nb] + ori[:nb].unsqueeze(0), r[:, nb:] * 0 + ori[nb:].unsqueeze(0)], dim=1)
    l2 = m.criterion(m.get_pred(pf, r, use_bg_proto=True), y)
    return 0.5 * l2 + 0.5 * l1
piece("generate_fake_proto", fake)
piece("fake_proto + get_pred + CE", loss1)
piece("post_refine_proto_v2", refine)
piece("whole tail", tail)
