"""configs[2] (get_basis.py:162-222 of the reference) on the device: k-means++ seeding, per-class subsample, per-cluster
means from the final assignment - each against the host restatement it replaces."""
import types

import numpy as np
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _clustered(M, C, K, seed, spread=0.5):
    g = torch.Generator().manual_seed(seed)
    cent = torch.randn(K, C, generator=g) * 2
    return (cent[torch.randint(0, K, (M,), generator=g)] + spread * torch.randn(M, C, generator=g)).contiguous()


@pytest.mark.parametrize("M,C,K,seed", [(5000, 192, 40, 0), (1030, 64, 33, 1), (70000, 192, 24, 2), (257, 16, 9, 3), (4096, 4, 12, 4),
                                        (3000, 200, 10, 5)])
def test_kmeanspp_device_picks_equal_the_host_form(dev, M, C, K, seed):
    """The device loop draws the same points as the host form (numpy rng.choice on the probabilities copied back per pick)
    from the same RandomState: same first index, then pick t = searchsorted(cdf, u_t) on the same variate.  (The two forms
    round d2 differently - 1e-7 relative - so a variate within that of a cdf step could differ: 1e-5 per pick at these sizes.)"""
    from get_basis import kmeans_plus_plus, kmeans_plus_plus_host
    pts = _clustered(M, C, max(K // 2, 2), seed).to(dev)
    ref = kmeans_plus_plus_host(pts, K, np.random.RandomState(seed))
    cent, picks = kmeans_plus_plus(pts, K, np.random.RandomState(seed), return_picks=True)
    assert torch.equal(cent, pts[picks])
    assert torch.equal(cent, ref)
    assert len(set(picks.tolist())) == K                       # a picked point has d2 = 0 and cannot be drawn again


def test_kmeanspp_pruned_equals_unpruned_and_is_reproducible(dev):
    """The triangle-inequality skip (|c_new - c_a|^2 >= 4 d2: the point cannot move) changes which rows are READ, not the result;
    two runs are identical (fixed summation orders)."""
    from gfs3d import ops
    for M, C, K, spread in ((200000, 192, 60, 0.3), (50000, 64, 200, 0.05), (20000, 192, 30, 2.0)):
        pts = _clustered(M, C, 25, M % 97, spread).to(dev)
        rng = np.random.RandomState(1)
        first, u = int(rng.randint(M)), rng.random_sample(K - 1)
        c1, p1 = ops.kmeanspp_seed(pts, K, first, u, prune=True)
        c0, p0 = ops.kmeanspp_seed(pts, K, first, u, prune=False)
        c2, p2 = ops.kmeanspp_seed(pts, K, first, u, prune=True)
        assert torch.equal(p1, p0) and torch.equal(c1, c0)
        assert torch.equal(p1, p2)


def test_kmeanspp_degenerate_inputs(dev):
    """All points identical (no mass after the first pick), K = M, K = 1."""
    from gfs3d import ops
    same = torch.ones(500, 8, device=dev)
    c, p = ops.kmeanspp_seed(same, 5, 3, np.array([0.1, 0.5, 0.9, 0.999]))
    assert torch.equal(c, same[:5]) and int(p.min()) >= 0 and int(p.max()) < 500
    pts = torch.randn(6, 4, generator=torch.Generator().manual_seed(0)).to(dev)
    c, p = ops.kmeanspp_seed(pts, 6, 2, np.linspace(0.05, 0.95, 5))
    assert sorted(p.tolist()) == list(range(6))
    c, p = ops.kmeanspp_seed(pts, 1, 4, np.zeros(0))
    assert p.tolist() == [4] and torch.equal(c[0], pts[4])
    with pytest.raises(RuntimeError):
        ops.kmeanspp_seed(pts, 7, 0, np.zeros(6))


def test_device_means_equal_kmean2proto(dev):
    """Per-cluster means from the final assignment's segment sums (device) = Kmean2Proto's host loop over masks
    (reference get_basis.py:27-44) on the same labels."""
    from get_basis import Kmean2Proto, device_kmeans
    pts = _clustered(30000, 192, 12, 7).to(dev)
    cent, label, inertia, n_iter, means, counts = device_kmeans(pts, 12, seed=3, max_iter=50, return_means=True)
    ref = Kmean2Proto(pts.cpu().numpy(), label.cpu().numpy(), 12)
    assert np.abs(means.cpu().numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())
    assert np.array_equal(counts.cpu().numpy().astype(np.int64), np.bincount(label.cpu().numpy(), minlength=12))


@pytest.mark.parametrize("sampler", ["device", "numpy"])
def test_subsample_per_class_takes_the_right_rows(dev, sampler):
    from get_basis import subsample_per_class
    g = torch.Generator().manual_seed(0)
    y = torch.randint(0, 5, (7, 400), generator=g)
    y[y == 3] = 2                                              # class 3 absent
    feat = torch.stack([y.reshape(-1).float(), torch.arange(y.numel()).float()], 1).to(dev)      # (class id, point id)
    out = subsample_per_class(feat, y, 5, max_per_class=300, seed=1, sampler=sampler).cpu()
    cls, pid = out[:, 0].long(), out[:, 1].long()
    assert bool((cls[1:] >= cls[:-1]).all()) and set(cls.tolist()) == {1, 2, 4}
    yy = y.reshape(-1)
    for c in (1, 2, 4):
        n = int((yy == c).sum())
        got = pid[cls == c]
        assert len(got) == min(n, 300) and len(set(got.tolist())) == len(got)
        assert bool((yy[got] == c).all())
        if n <= 300:
            assert torch.equal(got, torch.nonzero(yy == c).reshape(-1))       # untouched classes keep point order
    again = subsample_per_class(feat, y, 5, max_per_class=300, seed=1, sampler=sampler).cpu()
    assert torch.equal(out, again)


def test_collect_edgeconv_features_equals_return_feat(dev):
    """The resident point-major feature table = what model(pc, return_feat=True) returns block by block (get_basis.py:170)."""
    from get_basis import DGCNNSeg, collect_edgeconv_features
    a = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20)
    torch.manual_seed(0)
    seg = DGCNNSeg(a, num_classes=4).to(dev).eval()
    blocks = R.synthetic_blocks(5, 256, seed=3)
    with torch.no_grad():
        table = collect_edgeconv_features(seg, blocks, batch=2)
        for b in range(5):
            _, feat = seg(blocks[b:b + 1].to(dev), return_feat=True)
            assert torch.equal(table[b * 256:(b + 1) * 256], feat[0].t())


def test_extract_geometric_words_timings_and_host_share(dev):
    from get_basis import DGCNNSeg, extract_geometric_words
    a = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20)
    torch.manual_seed(0)
    seg = DGCNNSeg(a, num_classes=4).to(dev)
    blocks = R.synthetic_blocks(6, 512, seed=21)
    labels = torch.randint(0, 4, (6, 512), generator=torch.Generator().manual_seed(2))
    tm = {}
    w = extract_geometric_words(seg, blocks, labels, num_classes=4, num_cnt=16, max_per_class=600, batch=4, seed=0, max_iter=20, timings=tm)
    assert w.shape == (16, 192) and np.isfinite(w).all()
    assert {"features", "subsample", "seeding", "lloyd", "means_to_host", "svd_host", "lloyd_iterations", "points"} <= set(tm)
    assert tm["points"] <= 3 * 600


def test_bench_c3_end_to_end_line_small(dev):
    """`extra.c3_end_to_end` of the bench line on a small instance: every stage timed, host share = SVD + read-back."""
    import bench
    r = bench.c3_end_to_end(dev, n_blocks=100, num_cnt=16)
    assert set(r["stage_ms"]) == {"features", "subsample", "seeding", "lloyd", "means_to_host", "svd_host"}
    assert r["words_shape"] == [16, 192] and 0.0 < r["host_share"] < 1.0
    assert abs(sum(r["stage_ms"].values()) - r["total_ms"]) < 0.25 * r["total_ms"]
