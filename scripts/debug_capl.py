import sys, os, random, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from conftest import load_golden, state_dict_from
from test_gpu_parity import _capl, _force_routing, relerr
dev = torch.device("cuda:0")
g = load_golden("capl")
m = _capl(g, dev).train()
mask = np.unpackbits(g["ft_mask"])[: int(np.prod(g["ft_mask_shape"]))].reshape(g["ft_mask_shape"])
m.att_learner._forced_keep_mask = torch.from_numpy(mask.astype(bool)).to(dev)
x = torch.from_numpy(g["x"]).to(dev).requires_grad_(True)
random.seed(0)
captured = []
orig = m.encoder.edgeconv_features
def spy(inp):
    feats = orig(inp); captured.extend(feats); return feats
m.encoder.edgeconv_features = spy
pred, loss = m(x, y=torch.from_numpy(g["y"]).to(dev))
for i in range(3):
    print("flips layer", i, _force_routing(captured[i], g[f"ft_route{i}"]))
loss.backward()
print("loss", float(loss), float(g["ft_loss"]))
print("dx", relerr(x.grad, g["ft_grad_x"]))
for n, p in m.named_parameters():
    if p.grad is not None:
        print(f"{n:45s} {relerr(p.grad, g['ft_grad.' + n]):.3e}")
