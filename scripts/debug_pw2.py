import sys, random, numpy as np, torch
sys.path.insert(0, "gfs-3dseg_gws_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import load_golden
import test_gpu_parity as T
dev = torch.device("cuda:0")
g = load_golden("capl")
res = {}
for mode in ("hip", "torch"):
    m = T._capl(g, dev).train()
    mask = np.unpackbits(g["ft_mask"])[: int(np.prod(g["ft_mask_shape"]))].reshape(g["ft_mask_shape"])
    m.att_learner._forced_keep_mask = torch.from_numpy(mask.astype(bool)).to(dev)
    if mode == "torch":
        m.encoder.conv.fused_parts = lambda: None
    x = torch.from_numpy(g["x"]).to(dev).requires_grad_(True)
    random.seed(0)
    cap = {}
    orig = m.encoder.forward
    def fwd(inp, orig=orig, cap=cap):
        l1, out = orig(inp)
        out.retain_grad(); [t.retain_grad() for t in l1]
        cap["out"] = out; cap["l1"] = l1
        return l1, out
    m.encoder.forward = fwd
    captured = []
    oe = m.encoder.edgeconv_features
    def spy(inp, oe=oe, captured=captured, **kw):
        r = oe(inp, **kw); captured.extend(r[0] if kw.get("with_point_major") else r); return r
    m.encoder.edgeconv_features = spy
    ogf = m.getFeatures
    def gf(*a, ogf=ogf, cap=cap, **kw):
        r = ogf(*a, **kw); cap["pf"] = r[0].detach(); cap["word"] = r[2].detach(); return r
    m.getFeatures = gf
    pred, loss = m(x, y=torch.from_numpy(g["y"]).to(dev))
    for i in range(3):
        T._force_routing(captured[i], g[f"ft_route{i}"])
    loss.backward()
    res[mode] = dict(out=cap["out"].detach(), dout=cap["out"].grad, dl1=torch.stack([t.grad if t.grad is not None else torch.zeros_like(t) for t in cap["l1"]]), f=torch.stack([t.detach() for t in cap["l1"]]), gx=x.grad, pf=cap["pf"], word=cap["word"].argmax(1),
                     gw={n: p.grad for n, p in m.named_parameters() if p.grad is not None})
a, b = res["hip"], res["torch"]
for k in ("f", "out", "dout", "dl1", "gx"):
    e = (a[k] - b[k]).abs()
    print(k, "max", float(b[k].abs().max()), "err", float(e.max()), "n>1e-4max", int((e > 1e-4 * b[k].abs().max()).sum()))
print("point_feat err", float((a["pf"] - b["pf"]).abs().max()), "word flips", int((a["word"] != b["word"]).sum()), "of", a["word"].numel())
gref = torch.from_numpy(g["ft_grad_x"]).to(dev)
for mode in res:
    e = (res[mode]["gx"] - gref).abs()
    print(mode, "vs golden gx err", float(e.max()) / float(gref.abs().max()), "bad pts", int((e.amax(1) > 1e-4 * gref.abs().max()).sum()))
for n in a["gw"]:
    mx = float(b["gw"][n].abs().max())
    er = float((a["gw"][n] - b["gw"][n]).abs().max())
    if er > 1e-4 * mx: print(n, er / max(mx, 1e-30))
print("---- L2-relative errors vs golden (hip | torch)")
def l2(a, b):
    b = torch.as_tensor(b).to(a.device).double(); a = a.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
print("gx", l2(res["hip"]["gx"], g["ft_grad_x"]), l2(res["torch"]["gx"], g["ft_grad_x"]))
for n in res["hip"]["gw"]:
    key = "ft_grad." + n
    if key in g.files:
        print(n, "%.2e %.2e" % (l2(res["hip"]["gw"][n], g[key]), l2(res["torch"]["gw"][n], g[key])), "max-rel %.2e" % (float((res["hip"]["gw"][n].cpu() - torch.from_numpy(g[key])).abs().max()) / max(float(np.abs(g[key]).max()), 1e-30)))
