"""Per-parameter gradient cosines of one CAPL training step in the reduced modes against the reference golden (the numbers behind
the floors of tests/test_gpu_reduced.py::test_capl_train_step_reduced_vs_reference_golden)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, _p)
import numpy as np
import torch
from conftest import load_golden
from gfs3d import ops
from test_gpu_reduced import _capl, cosine

dev = torch.device("cuda:0")
g = load_golden("capl")
for mode in ("f32", "f16", "bf16"):
    m = _capl(g, dev, 13, 7).train()
    mask = np.unpackbits(g["ft_mask"])[: int(np.prod(g["ft_mask_shape"]))].reshape(g["ft_mask_shape"])
    m.att_learner._forced_keep_mask = torch.from_numpy(mask.astype(bool)).to(dev)
    random.seed(0)
    with ops.compute(mode):
        pred, loss = m(torch.from_numpy(g["x"]).to(dev), y=torch.from_numpy(g["y"]).to(dev))
        loss.backward()
    cs = []
    for n, p in m.named_parameters():
        key = "ft_grad." + n
        if p.grad is None or key not in g.files or float(np.abs(g[key]).max()) < 1e-6:
            continue
        cs.append((cosine(p.grad, g[key]), n))
    cs.sort()
    print(mode, "loss rel err", abs(float(loss) - float(g["ft_loss"])) / float(g["ft_loss"]), "pred agree", float((pred.cpu().numpy() == g["ft_pred"]).mean()))
    print("  lowest cosines:", [(round(c, 4), n) for c, n in cs[:6]])
