"""GPU parity of the CAPL heads on the HIP point-wise kernels (K4 generalised: conv bias, per-layer activation slopes,
wide input segments, point-major outputs; the q/k/v GEMM; the channel packing) against torch in fp64 - through the C ABI.

Reference code these replace: BaseLearner (model/dgcnn.py:130-152), q_map / k_map / v_map (model/attention.py:39-41), the
fusion layer and its torch.cat (model/capl.py:64-65, 357-359).  Tolerance: 1e-4 of the tensor's max magnitude (fp32 contract).
"""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from gfs3d import _lib
    _lib.lib()          # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def maxerr(a, b):
    return float((torch.as_tensor(a).detach().cpu().double() - torch.as_tensor(b).detach().cpu().double()).abs().max())


def relerr(a, b):
    b = torch.as_tensor(b).detach().cpu().double()
    return maxerr(a, b) / max(float(b.abs().max()), 1e-12)


# ------------------------------------------------------------------ pack / unpack
@pytest.mark.parametrize("B,N,widths,cf,ld", [
    (2, 100, [200, 64, 64, 64], [True, False, True, False], 448),      # the fusion input of capl.py:357-359, ragged N
    (1, 64, [64], [True], 64),
    (3, 257, [7, 130, 64], [False, True, False], 256),                 # odd widths, padding columns
    (2, 2048, [180, 64, 64, 64], [True, False, True, False], 384),     # ScanNet words
])
def test_pack_rows_and_adjoint_match_torch(dev, B, N, widths, cf, ld):
    from gfs3d import ops
    torch.manual_seed(sum(widths) + N)
    srcs = [(torch.randn(B, w, N, device=dev) if c else torch.randn(B, N, w, device=dev)).requires_grad_(True) for w, c in zip(widths, cf)]
    out = ops.pack_rows(srcs, cf, ld=ld)
    ref = torch.cat([t.transpose(1, 2) if c else t for t, c in zip(srcs, cf)], dim=2)
    ref = F.pad(ref, (0, ld - sum(widths)))
    assert out.shape == (B, N, ld)
    assert torch.equal(out, ref)                                       # a copy: bit-exact
    cot = torch.randn(B, N, ld, device=dev)
    out.backward(cot)
    g = [t.grad.clone() for t in srcs]
    for t in srcs:
        t.grad = None
    ref.backward(cot)
    for a, t in zip(g, srcs):
        assert torch.equal(a, t.grad)


# ------------------------------------------------------------------ q / k / v GEMM
@pytest.mark.parametrize("B,N,K,n_out", [(2, 300, 256, 3), (1, 64, 64, 1), (3, 1000, 128, 4)])
def test_linear_cf_forward_backward_matches_torch(dev, B, N, K, n_out):
    from gfs3d import ops
    torch.manual_seed(K + N)
    x = torch.randn(B, N, K, device=dev).requires_grad_(True)
    w = (torch.randn(64 * n_out, K, 1, device=dev) / K ** 0.5).requires_grad_(True)
    outs = ops.linear_cf(x, w)
    assert len(outs) == n_out
    x64, w64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    ref = F.conv1d(x64.transpose(1, 2), w64)                           # (B, 64 n, N)
    cots = [torch.randn(B, 64, N, device=dev) for _ in range(n_out)]
    for s, o in enumerate(outs):
        assert o.shape == (B, 64, N)
        assert relerr(o, ref[:, 64 * s:64 * s + 64]) < TOL
    torch.autograd.backward(outs, cots)
    ref.backward(torch.cat(cots, dim=1).double())
    assert relerr(x.grad, x64.grad) < TOL
    assert relerr(w.grad, w64.grad) < TOL


# ------------------------------------------------------------------ BaseLearner on K4
def _randomise_bn(mod):
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.5, 0.5)
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)


@pytest.mark.parametrize("B,N,cin,widths", [(2, 500, 256, [128, 64]), (1, 77, 64, [64]), (2, 1024, 128, [192, 128, 64])])
def test_base_learner_train_forward_backward_matches_torch(dev, B, N, cin, widths):
    """Conv1d WITH bias + BatchNorm1d, ReLU between the layers, none after the last (reference dgcnn.py:130-152): outputs
    in both layouts, running statistics (the bias moves the running mean) and every gradient (bias gradients are exactly
    zero in front of a training-mode BatchNorm: torch's own are rounding noise)."""
    from model.dgcnn import BaseLearner
    torch.manual_seed(cin + N)
    bl = BaseLearner(cin, widths).to(dev).train()
    _randomise_bn(bl)
    for blk in bl.convs:
        blk[0].bias.data.normal_(0, 0.5)
    ref = copy.deepcopy(bl).double()
    x_pm = torch.randn(B, N, cin, device=dev).requires_grad_(True)
    x64 = x_pm.detach().double().requires_grad_(True)
    y_cf, y_pm = bl.forward_pm(x_pm, want_cf=True)
    # fp64 reference
    h = x64.transpose(1, 2)
    for i, blk in enumerate(ref.convs):
        h = blk(h)
        if i != len(ref.convs) - 1:
            h = F.relu(h)
    assert y_cf.shape == (B, widths[-1], N) and y_pm.shape == (B, N, widths[-1])
    assert relerr(y_cf, h) < TOL
    assert torch.equal(y_pm, y_cf.transpose(1, 2).contiguous())
    for (n1, b1), (_, b2) in zip(bl.named_buffers(), ref.named_buffers()):
        assert maxerr(b1, b2) < 1e-5, n1
    cot = torch.randn(B, widths[-1], N, device=dev)
    # half of the gradient arrives through each view
    torch.autograd.backward([y_cf, y_pm], [0.5 * cot, 0.5 * cot.transpose(1, 2).contiguous()])
    h.backward(cot.double())
    bad = ((x_pm.grad.double() - x64.grad).abs().amax(-1) > TOL * float(x64.grad.abs().max())).float().mean()
    assert float(bad) <= 2e-3, float(bad)                           # points whose ReLU input sits within rounding of 0
    for (n1, p1), (_, p2) in zip(bl.named_parameters(), ref.named_parameters()):
        if n1.endswith("0.bias"):
            assert float(p1.grad.abs().max()) == 0.0 and float(p2.grad.abs().max()) < 1e-6 * B * N
        else:
            assert relerr(p1.grad, p2.grad) < 2e-3, n1


def test_base_learner_eval_matches_torch(dev):
    from model.dgcnn import BaseLearner
    torch.manual_seed(5)
    bl = BaseLearner(256, [128, 64]).to(dev)
    _randomise_bn(bl)
    for blk in bl.convs:
        blk[0].bias.data.normal_(0, 0.5)
    bl.eval()
    ref = copy.deepcopy(bl).double()
    x_pm = torch.randn(3, 333, 256, device=dev)
    with torch.no_grad():
        y_pm = bl.forward_pm(x_pm)
        y_cf, y_pm2 = bl.forward_pm(x_pm, want_cf=True)
        y_ref = ref(x_pm.double().transpose(1, 2))
    assert relerr(y_pm.transpose(1, 2), y_ref) < TOL
    assert torch.equal(y_pm, y_pm2) and torch.equal(y_cf, y_pm.transpose(1, 2).contiguous())


def test_attention_forward_pm_matches_module_forward(dev):
    """SelfAttention.forward_pm (one q|k|v GEMM + K5) == SelfAttention.forward (three torch convolutions + K5), outputs and
    gradients, with a forced dropout mask."""
    from model.attention import SelfAttention
    torch.manual_seed(2)
    att = SelfAttention(256, 64).to(dev).train()
    B, N = 2, 200
    att._forced_keep_mask = torch.rand(B, N, N, device=dev) > 0.1
    x = torch.randn(B, 256, N, device=dev).requires_grad_(True)
    x_pm = x.detach().transpose(1, 2).contiguous().requires_grad_(True)
    cot = torch.randn(B, 64, N, device=dev)
    y1 = att(x)
    y1.backward(cot)
    g1 = [p.grad.clone() for p in att.parameters()]
    for p in att.parameters():
        p.grad = None
    y2 = att.forward_pm(x_pm)
    y2.backward(cot)
    assert relerr(y2, y1) < TOL
    assert relerr(x_pm.grad.transpose(1, 2), x.grad) < TOL
    for a, p in zip(g1, att.parameters()):
        assert relerr(p.grad, a) < TOL


@pytest.mark.parametrize("training", [True, False])
def test_fusion_layer_on_packed_input_matches_torch(dev, training):
    """capl.py:357-359: Conv1d(392 -> 128, bias) + BatchNorm1d + LeakyReLU(0.2) on cat([soft, edge, att, base]) as one
    packed point-major matrix with K padded to 448."""
    from gfs3d import ops
    torch.manual_seed(9)
    B, N = 2, 300
    fusion = torch.nn.Sequential(torch.nn.Conv1d(392, 128, 1), torch.nn.BatchNorm1d(128), torch.nn.LeakyReLU(0.2)).to(dev)
    _randomise_bn(fusion)
    fusion.train(training)
    ref = copy.deepcopy(fusion).double()
    soft = torch.softmax(torch.randn(B, 200, N, device=dev), 1).requires_grad_(True)
    e0 = torch.randn(B, N, 64, device=dev).requires_grad_(True)
    att = torch.randn(B, 64, N, device=dev).requires_grad_(True)
    base = torch.randn(B, N, 64, device=dev).requires_grad_(True)
    srcs = [soft, e0, att, base]
    packed = ops.pack_rows(srcs, [True, False, True, False], ld=448)
    w = F.pad(fusion[0].weight.squeeze(-1), (0, 56))
    y = ops.pointwise_stack([packed], [((w, fusion[0].bias), fusion[1])], training, 0.2)
    s64 = [t.detach().double().requires_grad_(True) for t in srcs]
    y_ref = ref(torch.cat([s64[0], s64[1].transpose(1, 2), s64[2], s64[3].transpose(1, 2)], dim=1))
    assert relerr(y, y_ref) < TOL
    for (n1, b1), (_, b2) in zip(fusion.named_buffers(), ref.named_buffers()):
        assert maxerr(b1, b2) < 1e-5, n1
    if not training:
        return
    cot = torch.randn(B, 128, N, device=dev)
    y.backward(cot)
    y_ref.backward(cot.double())
    for a, b in zip(srcs, s64):
        assert relerr(a.grad, b.grad) < TOL
    assert relerr(fusion[0].weight.grad, ref[0].weight.grad) < TOL
    assert relerr(fusion[1].weight.grad, ref[1].weight.grad) < TOL and relerr(fusion[1].bias.grad, ref[1].bias.grad) < TOL
    assert float(fusion[0].bias.grad.abs().max()) == 0.0


@pytest.mark.parametrize("training", [True, False])
def test_dgcnnseg_att_fused_heads_match_the_torch_heads(dev, training, monkeypatch):
    """Pre-training model (reference dgcnn.py:155-202): attention + BaseLearner + segmenter on the HIP point-wise kernels
    against the same module with its heads on torch (same HIP backbone), logits and every parameter gradient."""
    import types
    import model.dgcnn as D
    a = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20,
                              base_widths=[128, 64], output_dim=64)
    torch.manual_seed(1)
    m = D.DGCNNSeg_att(a, num_classes=13).to(dev)
    _randomise_bn(m)
    m.segmenter[6].p = 0.0                                          # the dropout draw is torch's in both paths; keep it out of the comparison
    m.train(training)
    B, N = 2, 256
    if training:
        m.att_learner._forced_keep_mask = torch.rand(B, N, N, device=dev) > 0.1
    x = torch.rand(B, 9, N, device=dev)
    cot = torch.randn(B, 13, N, device=dev)
    sd = copy.deepcopy(m.state_dict())

    def run(fused):
        m.load_state_dict(sd)
        for p in m.parameters():
            p.grad = None
        if not fused:
            monkeypatch.setattr(D, "segmenter_fusable", lambda seg: False)
        else:
            monkeypatch.undo()
        with torch.set_grad_enabled(training):
            logits, l1 = m(x, return_feat=True)
        if training:
            logits.backward(cot)
        return logits.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, copy.deepcopy(m.state_dict())

    y0, g0, s0 = run(False)
    y1, g1, s1 = run(True)
    assert relerr(y1, y0) < TOL
    for k in s0:
        if "running" in k:
            assert maxerr(s1[k], s0[k]) < 1e-5, k
    if training:
        assert set(g0) == set(g1)
        for n in g0:
            if float(g0[n].abs().max()) < 1e-4:      # constants a following BatchNorm removes (conv biases, the base learner's last beta): analytically zero
                assert float(g1[n].abs().max()) < 1e-4, n
            else:
                assert relerr(g1[n], g0[n]) < 5e-4, n


def test_whole_model_train_step_bit_reproducible_in_deterministic_mode(dev):
    """set_deterministic(True) covers the point-wise kernels too: split-K weight gradients are partial tiles added in split
    order (always), BatchNorm column sums are ordered partial rows (in this mode).  Three runs of a full CAPL training forward
    + backward at B = 8 x 2048 - EdgeConv blocks, point-wise stack, BaseLearner, q|k|v GEMM, attention, fusion, word
    assignment, prototype heads - give bit-identical loss, running statistics and gradients, with other work in between."""
    import random
    import types
    from gfs3d import ops
    from gfs3d.synthetic import synthetic_blocks
    from model.capl import mpti_net_Point_GeoAsWeight_v2
    a = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20,
                              base_widths=[128, 64], output_dim=64, eval_weight=1.2)
    torch.manual_seed(3)
    gp = torch.randn(200, 192, generator=torch.Generator().manual_seed(2)).to(dev)
    m = mpti_net_Point_GeoAsWeight_v2(classes=13, criterion=torch.nn.CrossEntropyLoss(ignore_index=255), args=a, base_num=7, gp=gp,
                                      energy=0.9).to(dev).train()
    sd = copy.deepcopy(m.state_dict())
    B, N = 8, 2048
    x = synthetic_blocks(B, N, seed=4).to(dev)
    y = torch.randint(0, 8, (B, N), generator=torch.Generator().manual_seed(6)).to(dev)
    m.att_learner._forced_keep_mask = (torch.rand(B, N, N, generator=torch.Generator().manual_seed(7)) > 0.1).to(dev)
    runs = []
    ops.set_deterministic(True)
    try:
        for it in range(3):
            m.load_state_dict(sd)
            for p in m.parameters():
                p.grad = None
            if it:
                junk = torch.randn(1 << (20 + it), device=dev).sin_().sum()      # disturb the schedule and the allocator
            random.seed(0)
            pred, loss = m(x, y=y)
            loss.backward()
            runs.append([loss.detach().clone(), pred.clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None]
                        + [b.clone() for n, b in m.named_buffers() if "running" in n])
    finally:
        ops.set_deterministic(False)
    assert len(runs[0]) > 40
    for other in runs[1:]:
        assert len(other) == len(runs[0])
        # the scalar loss is torch's nll_loss2d forward, a float-atomic reduction (on PyTorch's own list of non-deterministic
        # CUDA ops): equal to rounding; its gradient is element-wise, so everything downstream is exact
        assert abs(float(other[0]) - float(runs[0][0])) <= 1e-6 * abs(float(runs[0][0]))
        for a_, b_ in zip(runs[0][1:], other[1:]):
            assert torch.equal(a_, b_)


def test_base_learner_two_shards_synchronised_equal_whole_batch(dev):
    """The head options of the K4 stack under synchronised BatchNorm (phased C ABI): a BaseLearner - conv bias, ReLU then no
    activation, one 256-wide input segment, point-major result - run as two block shards whose fp64 column sums are added
    between the phases reproduces the whole-batch run: outputs, running statistics (the bias enters the running mean
    once), input gradients; shard parameter gradients add up."""
    from gfs3d import ops
    from model.dgcnn import BaseLearner
    torch.manual_seed(31)
    bl = BaseLearner(256, [128, 64]).to(dev).train()
    _randomise_bn(bl)
    for blk in bl.convs:
        blk[0].bias.data.normal_(0, 0.5)
    B, N = 4, 150
    x = torch.randn(B, N, 256, device=dev)
    cot = torch.randn(B, N, 64, device=dev)
    whole = copy.deepcopy(bl)
    xw = x.clone().requires_grad_(True)
    yw = whole.forward_pm(xw)
    yw.backward(cot)

    mods = [copy.deepcopy(bl), copy.deepcopy(bl)]
    slopes = [0.0, 1.0]
    runs = []
    for r, m in enumerate(mods):
        layers = [(c.weight.detach(), bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var) for c, bn in m.fused_parts()]
        runs.append(ops.PointwiseRun([x[2 * r:2 * r + 2].contiguous()], layers, True, 1e-5, 0.1, 0.0, private_workspace=True,
                                     biases=[c.bias.detach() for c, _ in m.fused_parts()], slopes=slopes, want_cf=False, want_pm=True))
    row = 2 * 128
    accs = [torch.zeros(2 * 2 * row, dtype=torch.float64, device=dev) for _ in runs]
    count = float(B * N)

    def exchange(l):
        tot = accs[0][l * row:(l + 1) * row] + accs[1][l * row:(l + 1) * row]
        for a_ in accs:
            a_[l * row:(l + 1) * row] = tot
    for ph in range(3):
        for run, a_ in zip(runs, accs):
            run.forward_phase(ph, a_, count)
        if ph < 2:
            exchange(ph)
    assert maxerr(torch.cat([r_.y_pm for r_ in runs], 0), yw) < 1e-5
    gs = [None, None]
    for ph in range(3):
        for i, (run, a_) in enumerate(zip(runs, accs)):
            gs[i] = run.backward_phase(ph, a_, count, None, cot[2 * i:2 * i + 2] if ph == 0 else None)
        if ph < 2:
            exchange(1 - ph)
    assert relerr(torch.cat([gs[0][0][0], gs[1][0][0]], 0), xw.grad) < TOL
    parts = whole.fused_parts()
    for l, (c, bn) in enumerate(parts):
        assert relerr(gs[0][1][l] + gs[1][1][l], c.weight.grad.reshape(gs[0][1][l].shape)) < TOL
        assert relerr(gs[0][2][l] + gs[1][2][l], bn.weight.grad) < TOL
        assert relerr(gs[0][3][l] + gs[1][3][l], bn.bias.grad) < 1e-3 or float(bn.bias.grad.abs().max()) < 1e-4
    for (c0, bn0), (cw, bnw) in zip(mods[0].fused_parts(), parts):
        assert maxerr(bn0.running_mean, bnw.running_mean) < 1e-6 and maxerr(bn0.running_var, bnw.running_var) < 1e-6


# ------------------------------------------------------------------ full-size properties (BASELINE configs[3] shapes)
def test_attention_full_size_properties(dev):
    """B = 32 x 2048, d = 64 (no oracle finishes this in seconds): size-independent properties of softmax attention.
    (i) constant values come back unchanged (the weights of a query sum to 1); (ii) linearity in v; (iii) a joint permutation
    of keys and values changes nothing; (iv) the output of query i only depends on q_i (changing other queries leaves it
    bit-identical)."""
    from gfs3d import ops
    g = torch.Generator().manual_seed(11)
    B, N = 32, 2048
    q, k, v1, v2 = (torch.randn(B, 64, N, generator=g).to(dev) for _ in range(4))
    scale = 0.125
    with torch.no_grad():
        const = torch.linspace(-2, 2, 64, device=dev).view(1, 64, 1).expand(B, 64, N).contiguous()
        y = ops.attention(q, k, const, scale)
        assert maxerr(y, const) < 3e-5                      # |const| <= 2: 2048 fp32 probabilities per query, each summed with rounding
        y1, y2 = ops.attention(q, k, v1, scale), ops.attention(q, k, v2, scale)
        y12 = ops.attention(q, k, 0.5 * v1 - 2.0 * v2, scale)
        assert relerr(y12, 0.5 * y1 - 2.0 * y2) < 1e-5
        perm = torch.randperm(N, generator=g).to(dev)
        yp = ops.attention(q, k[:, :, perm].contiguous(), v1[:, :, perm].contiguous(), scale)
        assert relerr(yp, y1) < 1e-5
        q2 = q.clone()
        q2[:, :, 1::2] = torch.randn(B, 64, N // 2, generator=g).to(dev)
        y1b = ops.attention(q2, k, v1, scale)
        assert torch.equal(y1b[:, :, 0::2], y1[:, :, 0::2])


def test_word_assign_full_size_properties(dev):
    """B = 32 x 2048, C = 192, K_w = 200: soft assignments sum to 1, the one-hot tensor is the arg-max of the soft one and
    agrees with the index output, and permuting the words permutes both."""
    from gfs3d import ops
    g = torch.Generator().manual_seed(12)
    feat = torch.randn(32, 192, 2048, generator=g).to(dev)
    words = torch.randn(200, 192, generator=g).to(dev)
    with torch.no_grad():
        soft, onehot, assign = ops.WordAssignFunction.apply(feat, words, 10.0)
        soft2, assign2 = ops.WordSoftAssignFunction.apply(feat, words, 10.0)
        assert maxerr(soft.sum(1), torch.ones(32, 2048)) < 1e-5
        assert bool((onehot.sum(1) == 1).all()) and torch.equal(onehot.argmax(1).int(), assign)
        assert torch.equal(soft2, soft) and torch.equal(assign2, assign)
        top = soft.max(1)[0]
        assert bool((soft.gather(1, assign.long().unsqueeze(1)).squeeze(1) == top).all())
        perm = torch.randperm(200, generator=g).to(dev)
        softp, _, assignp = ops.WordAssignFunction.apply(feat, words[perm].contiguous(), 10.0)
        assert relerr(softp, soft[:, perm]) < 1e-5
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(200, device=dev)
        clear = (soft.topk(2, dim=1)[0][:, 0] - soft.topk(2, dim=1)[0][:, 1]) > 1e-4      # ties at rounding level may flip
        assert bool((inv[assign.long()] == assignp.long())[clear].all())


def test_pack_unpack_round_trip_full_size(dev):
    """The fusion input of configs[3] at B = 32 x 2048: unpack(pack(x)) == x bit for bit, pad columns zero."""
    from gfs3d import ops
    g = torch.Generator().manual_seed(13)
    B, N = 32, 2048
    srcs = [torch.randn(B, 200, N, generator=g).to(dev).requires_grad_(True), torch.randn(B, N, 64, generator=g).to(dev).requires_grad_(True),
            torch.randn(B, 64, N, generator=g).to(dev).requires_grad_(True), torch.randn(B, N, 64, generator=g).to(dev).requires_grad_(True)]
    packed = ops.pack_rows(srcs, [True, False, True, False], ld=448)
    assert float(packed.detach()[:, :, 392:].abs().max()) == 0.0
    packed.backward(packed.detach())                       # the adjoint applied to the packed matrix itself
    for t in srcs:
        assert torch.equal(t.grad, t.detach())


@pytest.mark.parametrize("M,C,K", [(100003, 192, 200), (777, 64, 12), (5000, 33, 256), (70, 16, 3), (1, 8, 1), (3000, 192, 200)])
def test_kmeans_centroid_sums_are_sorted_segment_sums(dev, M, C, K):
    """get_basis.py:210 (Lloyd step): with the workspace form of the entry point the centroid sums come from a stable
    counting sort of the points by label + fixed-order segment sums - equal to an fp64 index_add within fp32 rounding, member
    counts exact, empty clusters (K close to M) zero, and two runs bit-identical (no float atomics)."""
    from gfs3d import ops
    g = torch.Generator().manual_seed(M + K)
    pts = torch.randn(M, C, generator=g).to(dev)
    cent = torch.randn(K, C, generator=g).to(dev) * 0.5
    lab, sums, cnt, inertia = ops.kmeans_assign(pts, cent)
    ref = torch.zeros(K, C, dtype=torch.float64, device=dev).index_add_(0, lab.long(), pts.double())
    assert maxerr(sums, ref) <= 1e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(cnt.long(), torch.bincount(lab.long(), minlength=K))
    empty = cnt == 0
    assert float(sums[empty].abs().max()) == 0.0 if bool(empty.any()) else True
    d2 = ((pts.double()[:, None, :] - cent.double()[None]) ** 2).sum(-1) if M * K <= 2_000_000 else None
    if d2 is not None:
        assert abs(float(inertia) - float(d2.min(1)[0].sum())) <= 1e-4 * float(d2.min(1)[0].sum())
    lab2, sums2, cnt2, _ = ops.kmeans_assign(pts, cent)
    assert torch.equal(lab2, lab) and torch.equal(sums2, sums) and torch.equal(cnt2, cnt)


def test_fused_adam_equals_torch_adam_over_steps_groups_and_schedules(dev):
    """gfs3d.optim.FusedAdam (gfs3d_adam_multi_f32: every parameter of the model in one launch) against torch.optim.Adam in the
    configuration of the reference's train.py:425-435 - parameter groups with their own learning rates, a StepLR schedule, L2
    weight decay on one group - over 12 steps on the DGCNN's 30 tensors (ragged sizes: 64 ... 131 072 elements): parameters
    and both moment estimates agree to fp32 rounding (5e-6 of the tensor's scale), and the state dict of one loads into the
    other."""
    from gfs3d.optim import FusedAdam
    from model.dgcnn import DGCNN
    torch.manual_seed(4)
    net_a = DGCNN([[64, 64]] * 3, [512, 256], 9, k=20, return_edgeconvs=True).to(dev)
    net_b = copy.deepcopy(net_a)

    def groups(net):
        return [{"params": list(net.edge_convs.parameters()), "lr": 1e-4}, {"params": list(net.conv.parameters()), "weight_decay": 1e-2}]
    oa = FusedAdam(groups(net_a), lr=1e-3)
    ob = torch.optim.Adam(groups(net_b), lr=1e-3)
    sa, sb = (torch.optim.lr_scheduler.StepLR(o, step_size=5, gamma=0.5) for o in (oa, ob))
    g = torch.Generator(device=dev).manual_seed(9)
    for step in range(12):
        for pa, pb in zip(net_a.parameters(), net_b.parameters()):
            gr = torch.randn(pa.shape, device=dev, generator=g) * (10.0 ** (step % 4 - 3))
            pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step(); ob.step(); sa.step(); sb.step()
    for (n, pa), pb in zip(net_a.named_parameters(), net_b.parameters()):
        scale = float(pb.abs().max()) + 1e-12
        assert float((pa - pb).abs().max()) < 5e-6 * scale, n
        assert float((oa.state[pa]["exp_avg"] - ob.state[pb]["exp_avg"]).abs().max()) <= 5e-6 * float(ob.state[pb]["exp_avg"].abs().max()), n
        assert float((oa.state[pa]["exp_avg_sq"] - ob.state[pb]["exp_avg_sq"]).abs().max()) <= 5e-6 * float(ob.state[pb]["exp_avg_sq"].abs().max()), n
    assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"] == 1e-4 * 0.25
    ob.load_state_dict(oa.state_dict())            # the state layouts are interchangeable
    cpu_p = torch.nn.Parameter(torch.zeros(4))
    cpu_p.grad = torch.zeros(4)
    with pytest.raises(RuntimeError):              # no CPU fallback
        FusedAdam([cpu_p]).step()
