"""GPU parity of the fused prototype head of the CAPL training step (csrc/head.hip, gfs3d_capl_head_*: reference
model/capl.py:194-242 - generate_fake_proto :364-411, three get_pred calls :290-322, post_refine_proto_v2 :245-287, the two
CrossEntropyLoss(ignore_index=255) of train.py:317, `x_pre_2.max(1)[1]`) - through the C ABI, against

  * the oracle's restatement of that chain (oracle/ref_ops.py, torch on the CPU) evaluated in float64: loss within 1e-5
    relative, every gradient within 1e-4 of the tensor's largest magnitude, arg-max labels equal wherever the top-two
    logit gap is clear;
  * the reference's own training forward (tests/golden/capl.npz) through `capl.forward`, in test_gpu_parity.py (unchanged:
    the model takes the fused head whenever its criterion is the reference's).
"""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from gfs3d import _lib
    _lib.lib()          # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def relmax(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def head_reference(feat, main, bg, y, fake, base_num, ignore_index):
    """The oracle's restatement of the training head (oracle/ref_ops.py::capl_train_head - the function the golden-pinned
    capl_forward_train is built from), here evaluated in float64.  Returns (logits_2, loss)."""
    from oracle import ref_ops as R
    logits_2, loss, _ = R.capl_train_head(feat, y, main, bg, base_num, fake_novel=list(fake), ignore_index=ignore_index)
    return logits_2, loss


CASES = [
    # B, N, classes, base_num, labels in [0, hi], fake-novel labels, ignore fraction
    (4, 300, 13, 7, 7, [2, 5, 7], 0.05),          # S3DIS shapes (13 + 1 rows -> the 16-row tables), ragged N, ignored points
    (2, 2048, 13, 7, 7, [1, 4], 0.0),             # full-size blocks
    (3, 100, 13, 7, 7, [], 0.0),                  # no fake-novel class drawn (fewer than two classes present), odd B
    (4, 257, 21, 15, 15, [3, 8, 9, 11, 14, 15], 0.1),      # ScanNet shapes (21 + 1 rows -> the 24-row tables, two row tiles in the backward)
    (2, 64, 5, 5, 5, [5], 0.0),                   # every class a base class
    (1, 130, 13, 7, 7, [3], 0.0),                 # B = 1: the support half is the whole batch (capl.py:198: fake_num = 0)
]


@pytest.mark.parametrize("B,N,classes,base_num,hi,fake,ignore_frac", CASES)
def test_fused_train_head_matches_the_reference_chain_in_float64(dev, B, N, classes, base_num, hi, fake, ignore_frac):
    from gfs3d import ops
    g = torch.Generator().manual_seed(1000 * B + N + classes)
    feat = torch.randn(B, 128, N, generator=g) * torch.rand(B, 1, N, generator=g).mul(3).add(0.2)      # norms spread over a decade
    main = torch.randn(classes, 128, generator=g)
    bg = torch.randn(1, 128, generator=g)
    y = torch.randint(0, hi + 1, (B, N), generator=g)
    if ignore_frac > 0:
        y[torch.rand(B, N, generator=g) < ignore_frac] = 255
    # pull the features towards their class prototype so that the heads are not at chance (peaked softmax, w > 0 rows)
    proto_all = torch.cat([bg, main], 0)
    feat = feat + 0.2 * proto_all[y.clamp_max(classes)].permute(0, 2, 1) * (feat.norm(dim=1, keepdim=True) / 11.3)
    f64, m64, b64 = (t.double().requires_grad_(True) for t in (feat, main, bg))
    logits_ref, loss_ref = head_reference(f64, m64, b64, y, fake, base_num, 255)
    loss_ref.backward()
    fd, md, bd = (t.to(dev).requires_grad_(True) for t in (feat, main, bg))
    pred, loss = ops.capl_train_head(fd, md, bd, y.to(dev), fake, base_num, 255)
    (loss * 1.0).backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-5 * abs(float(loss_ref.detach()))
    top2 = logits_ref.detach().topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert bool((pred.cpu() == logits_ref.argmax(1))[clear].all())
    assert float(clear.float().mean()) > 0.99
    assert relmax(fd.grad, f64.grad) < 1e-4
    assert relmax(md.grad, m64.grad) < 1e-4
    assert relmax(bd.grad, b64.grad) < 1e-4


def test_fused_train_head_is_bit_reproducible_and_scales_with_the_incoming_gradient(dev):
    from gfs3d import ops
    B, N, classes = 4, 513, 13
    g = torch.Generator().manual_seed(5)
    feat, main, bg = torch.randn(B, 128, N, generator=g), torch.randn(classes, 128, generator=g), torch.randn(1, 128, generator=g)
    y = torch.randint(0, 8, (B, N), generator=g).to(dev)

    def run(scale):
        fd, md, bd = (t.to(dev).requires_grad_(True) for t in (feat, main, bg))
        pred, loss = ops.capl_train_head(fd, md, bd, y, [2, 6], 7, 255)
        (loss * scale).backward()
        return pred, loss.detach(), fd.grad, md.grad, bd.grad
    a, b, c = run(1.0), run(1.0), run(-3.0)
    for u, v in zip(a, b):
        assert torch.equal(u, v)                       # no atomics: two runs agree bit for bit
    for u, v in zip(a[2:], c[2:]):
        assert relmax(v, -3.0 * u) < 1e-6


def test_fused_train_head_on_a_misaligned_feature_tensor_equals_the_aligned_run(dev):
    """The matrix-core passes read the features with 16-byte loads when every channel row is 16-byte aligned (N % 4 == 0, aligned
    base) and fall back to scalar loads otherwise: a contiguous (B,128,N) view that starts 4 bytes into its storage takes the
    fallback with the SAME N - same arithmetic in the same order, so every result equals the aligned run bit for bit."""
    from gfs3d import ops
    B, N, classes = 4, 512, 13
    g = torch.Generator().manual_seed(11)
    feat, main, bg = torch.randn(B, 128, N, generator=g), torch.randn(classes, 128, generator=g), torch.randn(1, 128, generator=g)
    y = torch.randint(0, 8, (B, N), generator=g).to(dev)

    def run(shifted):
        if shifted:
            store = torch.empty(B * 128 * N + 1, device=dev)
            fd = store[1:].view(B, 128, N)
            fd.copy_(feat)
            assert fd.data_ptr() % 16 == 4 and fd.is_contiguous()
            fd.requires_grad_(True)
        else:
            fd = feat.to(dev).requires_grad_(True)
        md, bd = main.to(dev).requires_grad_(True), bg.to(dev).requires_grad_(True)
        pred, loss = ops.capl_train_head(fd, md, bd, y, [2, 6], 7, 255)
        loss.backward()
        return pred, loss.detach(), fd.grad, md.grad, bd.grad
    for u, v in zip(run(False), run(True)):
        assert torch.equal(u, v)


def test_fused_train_head_full_size_matches_the_unfused_form(dev):
    """BASELINE configs[3] size (B = 32 x 2048, 13 + 1 classes): the fused head against the tensor-library form of the same model
    code on the device (loss 1e-5 relative, gradients 1e-4 of their max), and a second run bit for bit."""
    from gfs3d import ops
    from oracle import ref_ops as R
    B, N, classes, nb, fake = 32, 2048, 13, 7, [2, 4, 7]
    g = torch.Generator().manual_seed(123)
    feat = (torch.randn(B, 128, N, generator=g) * 2).to(dev)
    main, bg = torch.randn(classes, 128, generator=g).to(dev), torch.randn(1, 128, generator=g).to(dev)
    y = torch.randint(0, nb + 1, (B, N), generator=g).to(dev)
    y[0, :100] = 255
    runs = []
    for _ in range(2):
        fd, md, bd = (t.clone().requires_grad_(True) for t in (feat, main, bg))
        pred, loss = ops.capl_train_head(fd, md, bd, y, fake, nb, 255)
        loss.backward()
        runs.append((pred, loss.detach(), fd.grad, md.grad, bd.grad))
    for u, v in zip(*runs):
        assert torch.equal(u, v)
    fr, mr, br = (t.clone().requires_grad_(True) for t in (feat, main, bg))
    logits_ref, loss_ref, _ = R.capl_train_head(fr, y, mr, br, nb, fake_novel=fake, ignore_index=255)      # the oracle chain, fp32 on the device
    loss_ref.backward()
    pred, loss, gf, gm, gb = runs[0]
    assert abs(float(loss) - float(loss_ref.detach())) < 1e-5 * abs(float(loss_ref.detach()))
    assert float((pred == logits_ref.argmax(1)).float().mean()) > 0.9995
    assert relmax(gf, fr.grad) < 1e-4 and relmax(gm, mr.grad) < 1e-4 and relmax(gb, br.grad) < 1e-4
    assert bool(torch.isfinite(gf).all())


def test_fused_train_head_properties_at_full_size(dev):
    """Size-independent properties at BASELINE size (B = 32 x 2048): (i) relabelling the points of every block (one
    permutation of the point axis applied to features and labels) leaves the loss unchanged up to summation order and permutes
    predictions and feature gradients with it; (ii) the cosine heads do not see a positive rescaling of the PROTOTYPES: loss and
    predictions unchanged, prototype gradients scaled by 1 / s (the refinement blends with the raw main_proto, so this holds for
    the bg prototype and - with no base class refined, base_num = 0 - for main_proto as well)."""
    from gfs3d import ops
    B, N, classes = 32, 2048, 13
    g = torch.Generator().manual_seed(8)
    feat = torch.randn(B, 128, N, generator=g).to(dev)
    main, bg = torch.randn(classes, 128, generator=g).to(dev), torch.randn(1, 128, generator=g).to(dev)
    y = torch.randint(0, 8, (B, N), generator=g).to(dev)

    def run(f, m, b_, yy, base_num):
        fd, md, bd = (t.clone().requires_grad_(True) for t in (f, m, b_))
        pred, loss = ops.capl_train_head(fd, md, bd, yy, [1, 5, 6], base_num, 255)
        loss.backward()
        return pred, loss.detach(), fd.grad, md.grad, bd.grad
    p0, l0, gf0, gm0, gb0 = run(feat, main, bg, y, 7)
    perm = torch.randperm(N, generator=g).to(dev)
    p1, l1, gf1, gm1, gb1 = run(feat[:, :, perm].contiguous(), main, bg, y[:, perm].contiguous(), 7)
    assert abs(float(l1) - float(l0)) < 1e-6 * abs(float(l0))
    assert float((p1 == p0[:, perm]).float().mean()) > 0.9999
    assert relmax(gf1, gf0[:, :, perm]) < 1e-5 and relmax(gm1, gm0) < 1e-5 and relmax(gb1, gb0) < 1e-5
    p2, l2, gf2, gm2, gb2 = run(feat, main, bg, y, 0)
    p3, l3, gf3, gm3, gb3 = run(feat, main * 4.0, bg * 0.5, y, 0)
    assert abs(float(l3) - float(l2)) < 1e-6 * abs(float(l2)) and float((p3 == p2).float().mean()) > 0.9999
    assert relmax(gf3, gf2) < 1e-5 and relmax(gm3 * 4.0, gm2) < 1e-5 and relmax(gb3 * 0.5, gb2) < 1e-5


def test_model_takes_the_fused_head_and_matches_its_unfused_form(dev):
    """mpti_net_Point_GeoAsWeight_v2.train_head: the fused pipeline against the model's own tensor-library form on the same
    features (the form the golden tests pinned in rounds 1-2), gradients through to the features and the prototypes."""
    from model.capl import mpti_net_Point_GeoAsWeight_v2
    args = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20,
                                 base_widths=[128, 64], output_dim=64, eval_weight=1.2)
    torch.manual_seed(3)
    m = mpti_net_Point_GeoAsWeight_v2(classes=13, criterion=torch.nn.CrossEntropyLoss(ignore_index=255), args=args, base_num=7,
                                      gp=torch.randn(200, 192), energy=0.9).to(dev)
    g = torch.Generator().manual_seed(9)
    feat = torch.randn(4, 128, 700, generator=g).to(dev)
    y = torch.randint(0, 8, (4, 700), generator=g).to(dev)
    out = {}
    for fused in (True, False):
        m.fuse_head = fused
        m.zero_grad()
        f = feat.clone().requires_grad_(True)
        pred, loss = m.train_head(f, y, [1, 3, 6])
        loss.backward()
        out[fused] = (pred, loss.detach(), f.grad, m.main_proto.grad.clone(), m.bg_proto.grad.clone())
    assert float((out[True][0] == out[False][0]).float().mean()) > 0.999
    assert abs(float(out[True][1]) - float(out[False][1])) < 1e-5 * abs(float(out[False][1]))
    for u, v in zip(out[True][2:], out[False][2:]):
        assert relmax(u, v) < 1e-4
    # a criterion the fused head does not reproduce keeps the unfused path
    m.fuse_head = True
    m.criterion = torch.nn.CrossEntropyLoss(ignore_index=255, label_smoothing=0.1)
    assert not m._head_fusable(feat, y, [1])


@pytest.mark.parametrize("B,N,classes,base_num,Kw,with_labels", [(3, 300, 13, 7, 200, True), (2, 2048, 13, 7, 200, True),
                                                                 (2, 130, 21, 15, 180, True), (1, 64, 13, 7, 150, False)])
def test_fused_eval_head_matches_the_reference_chain_in_float64(dev, B, N, classes, base_num, Kw, with_labels):
    """capl.forward(eval_model=True) after getFeatures (capl.py:170-192): refinement without the bg row, gened_proto, get_pred,
    get_gp_weight - the oracle's pieces in float64 against gfs3d_capl_head_eval."""
    from gfs3d import ops
    from oracle import ref_ops as R
    g = torch.Generator().manual_seed(B * 31 + N)
    feat = torch.randn(B, 128, N, generator=g) * torch.rand(B, 1, N, generator=g).mul(3).add(0.2)
    main = torch.randn(classes, 128, generator=g)
    gened = torch.nn.functional.normalize(torch.randn(classes, 128, generator=g), dim=1)
    word = torch.randint(0, Kw, (B, N), generator=g, dtype=torch.int32)
    coding = (torch.rand(classes, Kw, generator=g) > 0.6).float()
    y = torch.randint(0, classes, (B, N), generator=g)
    f64, m64, g64 = feat.double(), main.double(), gened.double()
    rp = R.post_refine_proto_v2(m64, f64, f64)
    rp[:, :base_num] = rp[:, :base_num] + g64[:base_num].unsqueeze(0)
    rp[:, base_num:] = rp[:, base_num:] * 0 + g64[base_num:].unsqueeze(0)
    onehot = torch.nn.functional.one_hot(word.long(), Kw).permute(0, 2, 1).double()
    wref, acc_ref, nacc_ref = R.get_gp_weight(coding.double(), onehot, base_num, 1.2, gt_label=y if with_labels else None)
    ref = R.get_pred(f64, rp) * wref
    out, acc, nacc = ops.capl_eval_head(feat.to(dev), main.to(dev), gened.to(dev), word.to(dev), coding.to(dev), 1.2, base_num,
                                        y.to(dev) if with_labels else None)
    assert float((out.cpu().double() - ref).abs().max()) < 1e-4           # logits: 1e-4 absolute (values within [-12, 12])
    assert abs(float(acc) - float(acc_ref)) < 1e-6 and abs(float(nacc) - float(nacc_ref)) < 1e-6


def test_label_presence_equals_unique(dev):
    from gfs3d import ops
    g = torch.Generator().manual_seed(2)
    for n, hi in ((1, 3), (1000, 8), (70000, 22), (5, 64)):
        y = torch.randint(0, hi, (n,), generator=g)
        y[::7] = 255                                  # an ignore label: outside [0, 64), skipped
        assert ops.label_presence(y.to(dev)) == [int(v) for v in y.unique().tolist() if v < 64]


def test_capl_head_rejects_unsupported_shapes(dev):
    from gfs3d import ops
    feat = torch.randn(2, 64, 100, device=dev)
    with pytest.raises(NotImplementedError):
        ops.capl_train_head(feat, torch.randn(13, 64, device=dev), torch.randn(1, 64, device=dev),
                            torch.zeros(2, 100, dtype=torch.int64, device=dev), [1], 7, 255)


def _head_dp_rank(rank, world, port, out):
    import os, sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "gfs-3dseg_gws_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gfs3d import ops
    dev = torch.device("cuda:0")
    feat, main, bg, y = _dp_case()
    sel = [rank, rank + 2]                                 # items (0, 2) / (1, 3): the local support halves are items 2 and 3
    fd = feat[sel].to(dev).requires_grad_(True)
    md, bd = main.to(dev).requires_grad_(True), bg.to(dev).requires_grad_(True)
    pred, loss = ops.capl_train_head(fd, md, bd, y[sel].to(dev), [2, 5], 7, 255)
    loss.backward()
    for g_ in (md.grad, bd.grad):                          # the step's gradient all-reduce (mean over ranks)
        dist.all_reduce(g_)
        g_ /= world
    torch.cuda.synchronize()
    out[rank] = {"loss": float(loss.detach()), "pred": pred.cpu().numpy(), "dfeat": fd.grad.cpu().numpy(),
                 "dmain": md.grad.cpu().numpy(), "dbg": bd.grad.cpu().numpy()}
    dist.destroy_process_group()


def _dp_case():
    g = torch.Generator().manual_seed(77)
    feat = torch.randn(4, 128, 384, generator=g)
    main, bg = torch.randn(13, 128, generator=g), torch.randn(1, 128, generator=g)
    y = torch.randint(0, 8, (4, 384), generator=g)
    return feat, main, bg, y


def test_fused_train_head_two_block_shards_equal_the_whole_batch(dev):
    """Block-sharded data parallelism (two gloo ranks on this one GPU): the fake-novel sums (forward) and their gradients
    (backward) are all-reduced between the phases of the fused head, so two shards whose support halves together are the
    whole batch's support half reproduce the single-process loss (mean of the two), predictions and gradients."""
    import socket
    import numpy as np
    import torch.multiprocessing as mp
    from gfs3d import ops
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_head_dp_rank, args=(r, 2, port, out)) for r in range(2)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(300)
        assert p_.exitcode == 0
    feat, main, bg, y = _dp_case()
    fd = feat.to(dev).requires_grad_(True)
    md, bd = main.to(dev).requires_grad_(True), bg.to(dev).requires_grad_(True)
    pred, loss = ops.capl_train_head(fd, md, bd, y.to(dev), [2, 5], 7, 255)
    loss.backward()
    assert abs(0.5 * (out[0]["loss"] + out[1]["loss"]) - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))
    for r in range(2):
        sel = [r, r + 2]
        assert float((torch.from_numpy(out[r]["pred"]) == pred.cpu()[sel]).float().mean()) > 0.999
        # a rank's loss is the mean over ITS points: its feature gradients are twice the whole-batch ones (half as many points)
        assert relmax(0.5 * torch.from_numpy(out[r]["dfeat"]), fd.grad.cpu()[sel]) < 1e-4
        assert relmax(out[r]["dmain"], md.grad) < 1e-4 and relmax(out[r]["dbg"], bd.grad) < 1e-4
    assert np.array_equal(out[0]["dmain"], out[1]["dmain"])
