"""Reduced-precision modes: error of K3 / K4 / model-level outputs against the fp32-accurate mode and goldens."""
import sys, os, json, types, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gfs3d import ops
from model.dgcnn import conv1d
from conftest import load_golden, state_dict_from
dev = torch.device("cuda:0")


def stats(a, ref):
    d = (a.double() - ref.double())
    return {"max/max": float(d.abs().max() / ref.abs().max()), "rms/rms": float(d.pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt())}


# K3 word assignment + gradient
g = torch.Generator().manual_seed(1)
feat = torch.randn(4, 192, 2048, generator=g).to(dev); gp = torch.randn(200, 192, generator=g).to(dev)
cot = torch.randn(4, 200, 2048, generator=g).to(dev)
out = {}
for mode in ("f32", "bf16", "f16"):
    with ops.compute(mode):
        f = feat.clone().requires_grad_(True)
        soft, oh, asg = ops.WordAssignFunction.apply(f, gp, 10.0)
        (soft * cot).sum().backward()
        out[mode] = (soft.detach(), asg, f.grad)
for mode in ("bf16", "f16"):
    print("word_assign", mode, "soft", stats(out[mode][0], out["f32"][0]), "abs max", float((out[mode][0] - out["f32"][0]).abs().max()),
          "argmax agree", float((out[mode][1] == out["f32"][1]).float().mean()), "dfeat", stats(out[mode][2], out["f32"][2]))

# K4 stack
torch.manual_seed(3)
conv = conv1d(192, [512, 256]).to(dev).train()
xs = [torch.randn(8, 2048, 64, device=dev) for _ in range(3)]
cot = torch.randn(8, 256, 2048, device=dev)
res = {}
st0 = {k: v.clone() for k, v in conv.state_dict().items()}
for mode in ("f32", "bf16", "f16"):
    conv.load_state_dict(st0)
    with ops.compute(mode):
        for p in conv.parameters(): p.grad = None
        xg = [t.clone().requires_grad_(True) for t in xs]
        y = ops.pointwise_stack(xg, conv.fused_parts(), True)
        y.backward(cot)
        res[mode] = (y.detach(), torch.cat([t.grad for t in xg], -1), conv.layer[0].weight.grad.clone(), conv.layer[3].weight.grad.clone())
for mode in ("bf16", "f16"):
    print("pointwise", mode, "y", stats(res[mode][0], res["f32"][0]), "dx", stats(res[mode][1], res["f32"][1]),
          "dw0", stats(res[mode][2], res["f32"][2]), "dw1", stats(res[mode][3], res["f32"][3]))

# model level: ScanNet golden
from model.capl import mpti_net_Point_GeoAsWeight_v2
def capl(gg, classes, base):
    a = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20,
                              base_widths=[128, 64], output_dim=64, eval_weight=1.2)
    m = mpti_net_Point_GeoAsWeight_v2(classes=classes, criterion=torch.nn.CrossEntropyLoss(ignore_index=255), args=a,
                                      base_num=base, gp=torch.from_numpy(gg["gp"]).to(dev), energy=0.9)
    m.load_state_dict(state_dict_from(gg))
    return m.to(dev)
gs = load_golden("capl_scannet")
m = capl(gs, 21, 15).eval()
coding = torch.from_numpy(gs["gw_coding"]).to(dev)
for mode in ("f32", "bf16", "f16"):
    with ops.compute(mode), torch.no_grad():
        pf, sem, oh = m.getFeatures(torch.from_numpy(gs["x"]).to(dev))
        lg, acc, nacc = m(torch.from_numpy(gs["x"]).to(dev), y=torch.from_numpy(gs["gw_lab21"]).to(dev),
                          gened_proto=torch.from_numpy(gs["fe_gened"]).to(dev).unsqueeze(0), eval_model=True,
                          base_class_coding=coding[:15], novel_class_coding=coding[15:])
    ref_pf, ref_lg = torch.from_numpy(gs["gf_point"]), torch.from_numpy(gs["fe_logits"])
    print("scannet", mode, "point_feat", stats(pf.cpu(), ref_pf), "sem", stats(sem.cpu(), torch.from_numpy(gs["gf_sem"])),
          "word agree", float((oh.argmax(1).cpu().numpy() == gs["gf_word"]).mean()),
          "logits", stats(lg.cpu(), ref_lg), "abs max", float((lg.cpu() - ref_lg).abs().max()),
          "pred agree", float((lg.argmax(1).cpu() == ref_lg.argmax(1)).float().mean()))

# CAPL train step vs golden
gc = load_golden("capl")
for mode in ("f32", "bf16", "f16"):
    m = capl(gc, 13, 7).train()
    mask = np.unpackbits(gc["ft_mask"])[: int(np.prod(gc["ft_mask_shape"]))].reshape(gc["ft_mask_shape"])
    m.att_learner._forced_keep_mask = torch.from_numpy(mask.astype(bool)).to(dev)
    random.seed(0)
    with ops.compute(mode):
        x = torch.from_numpy(gc["x"]).to(dev).requires_grad_(True)
        pred, loss = m(x, y=torch.from_numpy(gc["y"]).to(dev))
        loss.backward()
    cos = {}
    for n, p in m.named_parameters():
        if p.grad is not None and ("ft_grad." + n) in gc.files and np.abs(gc["ft_grad." + n]).max() > 1e-6:
            a, b = p.grad.detach().cpu().double().flatten(), torch.from_numpy(gc["ft_grad." + n]).double().flatten()
            cos[n] = float((a @ b) / (a.norm() * b.norm()))
    print("capl train", mode, "loss", float(loss), "golden", float(gc["ft_loss"]), "pred agree", float((pred.cpu().numpy() == gc["ft_pred"]).mean()),
          "min grad cosine", min(cos.values()), min(cos, key=cos.get), "dx", stats(x.grad.cpu(), torch.from_numpy(gc["ft_grad_x"])))
