"""Stress of the hand-rolled cross-workgroup hand-off in the point-wise stack: `bn_tail`
(csrc/pointwise.hip) lets the LAST workgroup of the kernel that completes a BatchNorm layer's fp64 column sums finalise that
layer, ordered only by `s_waitcnt vmcnt(0)` + a relaxed ticket atomic.  If "vmcnt reached zero" did not mean "the adds are
visible at the device coherence point" on some XCD, the finalised numbers would silently miss a workgroup's share.  The sums the
finalisation must have used stay in the workspace after the call (include/gfs3d.h), so every iteration recomputes the
finalisation from them in fp64 and compares - a missing share shows as ~1/workgroups (1e-3), rounding as 1e-7."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _make_run(dev, B, N, nseg, widths, seed):
    from gfs3d import ops
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(B, N, 64, generator=g).to(dev) for _ in range(nseg)]
    layers, prev = [], 64 * nseg
    for c in widths:
        w = (torch.randn(c, prev, generator=g) / prev ** 0.5).to(dev)
        layers.append((w, (0.5 + torch.rand(c, generator=g)).to(dev), (torch.rand(c, generator=g) - 0.5).to(dev),
                       torch.zeros(c, device=dev), torch.ones(c, device=dev)))
        prev = c
    run = ops.PointwiseRun(xs, layers, True, 1e-5, 0.1, 0.2, private_workspace=True)
    return run, layers


def _sums(run):
    """Per layer l the pair (sum, sum of squares | sum g, sum g h): rows of C doubles at [2 maxw l] and [2 maxw l + C]."""
    L, maxw = len(run.widths), max(run.widths)
    flat = run._private_ws[:L * 2 * maxw * 8].view(torch.float64).clone()
    return [(flat[2 * maxw * l:2 * maxw * l + C], flat[2 * maxw * l + C:2 * maxw * l + 2 * C]) for l, C in enumerate(run.widths)]


def _fwd_dev(run, acc):
    """max over layers of the relative deviation of the finalised (mean, rstd, scale, shift) from the sums in `acc`."""
    count = float(run.B * run.N)
    worst = torch.zeros((), dtype=torch.float64, device=run.stats[0].device)
    for l, C in enumerate(run.widths):
        _w, gam, bet, _rm, _rv = run.layers[l]
        mean = acc[l][0] / count
        var = (acc[l][1] / count - mean * mean).clamp_min(0.0)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        scale = gam.double() * rstd
        ref = torch.stack([scale, bet.double() - mean * scale, mean, rstd])
        got = run.stats[l].double()
        worst = torch.maximum(worst, ((got - ref).abs() / ref.abs().amax(1, keepdim=True)).max())
    return worst


def _bwd_dev(run, acc, dgs, dbs):
    worst = torch.zeros((), dtype=torch.float64, device=run.stats[0].device)
    for l, C in enumerate(run.widths):
        s1, sgh = acc[l]
        mean, rstd = run.stats[l][2].double(), run.stats[l][3].double()
        dgamma = rstd * (sgh - mean * s1)
        for got, ref in ((dgs[l].double(), dgamma), (dbs[l].double(), s1)):
            worst = torch.maximum(worst, ((got - ref).abs() / ref.abs().max()).max())
    return worst


@pytest.mark.parametrize("tail", [1, 0])
def test_bn_tail_finalisation_uses_complete_sums(dev, tail):
    """2000 forward + backward calls of the DGCNN stack (192 -> 512 -> 256) at B = 32 x 2048 (1024 + 512 row-GEMM workgroups per
    layer): both tail kinds (forward finalisation, backward coefficients) against the sums left in the workspace, in every
    iteration.  tail = 0 runs the same check on the separate finalisation kernels (gfs3d_set_bn_tail: the field switch)."""
    from gfs3d import _lib
    L = _lib.lib()
    run, layers = _make_run(dev, 32, 2048, 3, [512, 256], seed=7)
    dy = torch.randn(32, 256, 2048, generator=torch.Generator().manual_seed(8)).to(dev)
    _lib.check(L.gfs3d_set_bn_tail(tail), "gfs3d_set_bn_tail")
    try:
        worst_f = torch.zeros((), dtype=torch.float64, device=dev)
        worst_b = torch.zeros((), dtype=torch.float64, device=dev)
        for it in range(2000):
            run.forward()
            worst_f = torch.maximum(worst_f, _fwd_dev(run, _sums(run)))
            dxs, dws, dgs, dbs = run.backward(dy)
            worst_b = torch.maximum(worst_b, _bwd_dev(run, _sums(run), dgs, dbs))
            if it % 500 == 499:
                torch.cuda.synchronize()
        wf, wb = float(worst_f), float(worst_b)
    finally:
        L.gfs3d_set_bn_tail(1)
    assert wf < 2e-6, f"forward BatchNorm finalisation deviates from the workspace sums by {wf:.3e} (tail={tail})"
    assert wb < 2e-6, f"backward BatchNorm coefficients deviate from the workspace sums by {wb:.3e} (tail={tail})"


def test_bn_tail_small_and_ragged_grids(dev):
    """Few workgroups (the last one is likely also the first) and ragged M: 300 iterations each."""
    for (B, N, nseg, widths) in ((1, 130, 1, [64]), (2, 200, 3, [512, 256]), (3, 1000, 2, [192, 128, 64])):
        run, _ = _make_run(dev, B, N, nseg, widths, seed=B * 10 + nseg)
        dy = torch.randn(B, widths[-1], N, generator=torch.Generator().manual_seed(9)).to(dev)
        wf = torch.zeros((), dtype=torch.float64, device=dev)
        wb = torch.zeros((), dtype=torch.float64, device=dev)
        for _ in range(300):
            run.forward()
            wf = torch.maximum(wf, _fwd_dev(run, _sums(run)))
            _dx, _dw, dgs, dbs = run.backward(dy)
            wb = torch.maximum(wb, _bwd_dev(run, _sums(run), dgs, dbs))
        assert float(wf) < 2e-6 and float(wb) < 2e-6, (B, N, nseg, widths, float(wf), float(wb))
