"""configs[2] (get_basis.py:162-222 of the reference) on the device: k-means++ seeding, per-class subsample, per-cluster
means from the final assignment - each against the host restatement it replaces."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden
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


def _equal_up_to_a_rounding_tie(picks, ref, margins, tol):
    """picks == ref, or they part at a pick whose recorded margin (distance of the deciding comparison from a tie, relative to
    the potential) is below `tol`: a rounding tie - the two forms round d2 differently (~1e-7 relative per value) - after which
    the sequences legitimately differ.  Returns the number of picks compared."""
    picks, ref = np.asarray(picks), np.asarray(ref)
    diff = np.nonzero(picks != ref)[0]
    if diff.size == 0:
        return len(ref)
    first = int(diff[0])
    assert float(np.min(margins[first])) < tol, f"pick {first}: {picks[first]} != {ref[first]} with margin {margins[first]}"
    return first


def _plain_host_picks_with_margins(pts, K, rng):
    """get_basis.kmeans_plus_plus_host step by step, recording how far each variate was from a cdf step."""
    M = pts.shape[0]
    first = int(rng.randint(M))
    picks, margins = [first], [1.0]
    d2 = ((pts - pts[first]) ** 2).sum(1)
    for _ in range(1, K):
        prob = (d2 / d2.sum()).double().cpu().numpy()
        prob = prob / prob.sum()
        state = rng.get_state()
        nxt = int(rng.choice(M, p=prob))
        probe = np.random.RandomState()
        probe.set_state(state)
        u = probe.random_sample()
        cdf = np.cumsum(prob)
        cdf /= cdf[-1]
        margins.append(float(np.abs(cdf - u).min()))
        picks.append(nxt)
        d2 = torch.minimum(d2, ((pts - pts[nxt]) ** 2).sum(1))
    return np.array(picks), np.array(margins)


@pytest.mark.parametrize("M,C,K,seed", [(5000, 192, 40, 0), (1030, 64, 33, 1), (70000, 192, 24, 2), (257, 16, 9, 3), (4096, 4, 12, 4),
                                        (3000, 200, 10, 5)])
def test_plain_kmeanspp_device_picks_equal_the_host_form(dev, M, C, K, seed):
    """The PLAIN one-draw-per-pick form (n_local_trials=0, gfs3d_kmeanspp_seed_f32) draws the same points as the host form
    (numpy rng.choice on the probabilities copied back per pick) from the same RandomState.  The two forms round d2
    differently (1e-7 relative), so a variate within that of a cdf step may differ: equality is required up to the first
    pick whose variate lay within 1e-6 of a step."""
    from get_basis import kmeans_plus_plus
    pts = _clustered(M, C, max(K // 2, 2), seed).to(dev)
    ref, margins = _plain_host_picks_with_margins(pts, K, np.random.RandomState(seed))
    cent, picks = kmeans_plus_plus(pts, K, np.random.RandomState(seed), return_picks=True, n_local_trials=0)
    assert torch.equal(cent, pts[picks])
    n = _equal_up_to_a_rounding_tie(picks.cpu().numpy(), ref, margins, 1e-6)
    assert n >= 2
    assert len(set(picks.tolist())) == K                       # a picked point has d2 = 0 and cannot be drawn again


def _kpp_case(g, name):
    M, C, K, blobs, data_seed, rs_seed = (int(v) for v in g[name + "_params"])
    return R.clustered_points(M, C, blobs, seed=data_seed), K, rs_seed


@pytest.mark.parametrize("name", ["a", "b", "c", "d", "e", "f"])
def test_greedy_kmeanspp_equals_sklearn_picks(dev, name):
    """gfs3d_kmeanspp_greedy_f32 from sklearn's variates == the indices sklearn.cluster.kmeans_plusplus itself returned
    (tests/golden/kmeanspp.npz: K = 200 over 3 000 - 50 000 x 192 points and two small-K shapes; per case the RandomState of six
    whose picks stay farthest from a tie).  Allowance, documented: the device rounds d2 (fp32 sum of squared differences) and
    sums potentials (fp64, fixed order) differently from sklearn (float64 distances rounded to fp32, float32 BLAS potentials), so
    the sequences may part at a pick whose recorded margin - variate to cdf step, or best to second-best potential, relative
    to the potential - is below 1e-7; nothing else is tolerated."""
    from get_basis import kmeans_plus_plus
    g = load_golden("kmeanspp")
    X, K, s = _kpp_case(g, name)
    pts = torch.from_numpy(X).to(dev)
    cent, picks = kmeans_plus_plus(pts, K, np.random.RandomState(s), return_picks=True)
    assert torch.equal(cent, pts[picks])
    n = _equal_up_to_a_rounding_tie(picks.cpu().numpy(), g[name + "_indices"], g[name + "_margins"], 1e-7)
    print(f"case {name}: {n} of {K} picks compared equal (min margins {g[name + '_margins'][1:].min(0)})")
    # every pick in front of the first recorded rounding tie must be sklearn's - for the tie-free fixtures (a, e, f) that is all K
    ties = np.nonzero(g[name + "_margins"].min(axis=1) < 1e-7)[0]
    must = int(ties[0]) if ties.size else K
    assert n >= must, f"case {name}: the sequences part at pick {n}, in front of the first recorded tie (pick {must})"
    if not ties.size:
        assert n == K


@pytest.mark.parametrize("name", ["a", "b", "c", "f"])
def test_device_kmeans_equals_sklearn_fit(dev, name):
    """get_basis.device_kmeans == `KMeans(n_clusters, init='k-means++', n_init=1, random_state=s).fit(X)` as sklearn ran it
    (get_basis.py:210 of the reference): (i) the Lloyd loop from sklearn's own picks ends after the same number of iterations
    in the same clustering - inertia within 1e-5 relative (case c, 50 000 points: 1e-4 - sklearn's float32 GEMM decides 0.13 % of
    its labels differently from an exact evaluation, measured against the float64 restatement); (ii) where the device's picks
    equal sklearn's, the whole fit from the RandomState alone does too."""
    from get_basis import device_kmeans, kmeans_plus_plus
    g = load_golden("kmeanspp")
    X, K, s = _kpp_case(g, name)
    pts = torch.from_numpy(X).to(dev)
    tol_inertia, min_same = (1e-4, 0.995) if name == "c" else (1e-5, 0.9995)
    ref_inertia, ref_iter = float(g[name + "_fit_inertia"]), int(g[name + "_fit_n_iter"])
    ref_labels = torch.from_numpy(g[name + "_fit_labels"].astype(np.int64)).to(dev)

    def check(res):
        cent, lab, inertia, n_iter = res
        assert abs(inertia - ref_inertia) < tol_inertia * ref_inertia, (inertia, ref_inertia)
        assert abs(n_iter - ref_iter) <= (3 if name == "c" else 0), (n_iter, ref_iter)
        assert float((lab == ref_labels).float().mean()) >= min_same
        if name != "c":
            assert np.abs(cent.cpu().numpy() - g[name + "_fit_centers"]).max() < 1e-3

    idx = torch.from_numpy(g[name + "_fit_indices"]).to(dev)
    check(device_kmeans(pts, K, init=pts[idx]))
    centred = pts - pts.mean(dim=0)
    _, picks = kmeans_plus_plus(centred, K, np.random.RandomState(s), return_picks=True)
    if torch.equal(picks, idx):
        check(device_kmeans(pts, K, rng=np.random.RandomState(s)))
    else:
        print(f"case {name}: the device's picks on the centred data part from sklearn's at pick "
              f"{int(torch.nonzero(picks != idx)[0])} (rounding tie): end-to-end comparison skipped")


def test_device_kmeans_relocates_empty_clusters_like_sklearn(dev):
    """A start with three centres far from every point (fixture case g): sklearn relocates the empty clusters onto the
    farthest points inside the first iteration; the device loop must end in sklearn's clustering."""
    from get_basis import device_kmeans
    g = load_golden("kmeanspp")
    X, K, _ = _kpp_case(g, "g")
    pts = torch.from_numpy(X).to(dev)
    cent, lab, inertia, n_iter = device_kmeans(pts, K, init=torch.from_numpy(g["g_init"]).to(dev))
    assert n_iter == int(g["g_fit_n_iter"])
    assert np.array_equal(lab.cpu().numpy(), g["g_fit_labels"].astype(np.int64))
    assert abs(inertia - float(g["g_fit_inertia"])) < 1e-5 * inertia
    assert np.abs(cent.cpu().numpy() - g["g_fit_centers"]).max() < 1e-4


def test_greedy_kmeanspp_pruned_equals_unpruned_and_is_reproducible(dev):
    """The gain pass's triangle-inequality skip (no candidate can come closer than the nearest centre) changes which rows are
    READ, not the picks; two runs are identical (fixed summation orders)."""
    from gfs3d import ops
    for M, C, K, spread in ((200000, 192, 60, 0.3), (50000, 64, 200, 0.05), (20000, 192, 30, 2.0)):
        pts = _clustered(M, C, 25, M % 97, spread).to(dev)
        rng = np.random.RandomState(1)
        T = 2 + int(np.log(K))
        first, u = int(rng.randint(M)), rng.uniform(size=(K - 1, T))
        c1, p1 = ops.kmeanspp_greedy(pts, K, first, u, T, prune=True)
        c0, p0 = ops.kmeanspp_greedy(pts, K, first, u, T, prune=False)
        c2, p2 = ops.kmeanspp_greedy(pts, K, first, u, T, prune=True)
        assert torch.equal(p1, p0) and torch.equal(c1, c0)
        assert torch.equal(p1, p2)
        assert len(set(p1.tolist())) == K


def test_greedy_kmeanspp_degenerate_inputs(dev):
    """All points identical (no mass after the first pick: sklearn's searchsorted(zeros, 0) = 0 every time), K = M, K = 1,
    one trial, the maximum of eight trials, K > M refused."""
    from gfs3d import ops
    same = torch.ones(500, 8, device=dev)
    c, p = ops.kmeanspp_greedy(same, 5, 3, np.full((4, 3), 0.5), 3)
    assert p.tolist() == [3, 0, 0, 0, 0] and torch.equal(c, same[:5])
    pts = torch.randn(6, 4, generator=torch.Generator().manual_seed(0)).to(dev)
    c, p = ops.kmeanspp_greedy(pts, 6, 2, np.random.RandomState(0).uniform(size=(5, 8)), 8)
    assert sorted(p.tolist()) == list(range(6))
    c, p = ops.kmeanspp_greedy(pts, 1, 4, np.zeros((0, 2)), 2)
    assert p.tolist() == [4] and torch.equal(c[0], pts[4])
    c, p = ops.kmeanspp_greedy(pts, 3, 1, np.array([[0.0], [0.999999]]), 1)
    assert p[0].item() == 1 and len(set(p.tolist())) == 3
    with pytest.raises(RuntimeError):
        ops.kmeanspp_greedy(pts, 7, 0, np.zeros((6, 2)), 2)
    with pytest.raises(RuntimeError):
        ops.kmeanspp_greedy(pts, 3, 0, np.zeros((2, 9)), 9)


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


def test_lloyd_delta_sums_follow_the_labels_exactly(dev):
    """ops.KMeansDelta (gfs3d_kmeans_delta_fix): after loading a full pass's sums, moving the points whose label changed leaves
    exactly the sums / counts a full pass over the new labels gives (to the fixed-point resolution, far below fp32 rounding), counts
    the moved points, and is independent of the order of arrival: two runs are bit-identical."""
    from gfs3d import ops
    M, C, K = 50000, 192, 200
    g = torch.Generator().manual_seed(5)
    pts = (torch.randn(M, C, generator=g) * 0.7 + 0.3).to(dev)
    lab0 = torch.randint(0, K, (M,), generator=g, dtype=torch.int32).to(dev)
    lab1 = lab0.clone()
    moved = torch.randperm(M, generator=g)[:1500].to(dev)
    lab1[moved] = ((lab0[moved].long() + 1 + torch.randint(0, K - 1, (1500,), generator=g).to(dev)) % K).int()

    def full(lab):
        s = torch.zeros(K, C, dtype=torch.float64, device=dev).index_add_(0, lab.long(), pts.double())
        return s, torch.bincount(lab.long(), minlength=K).double()

    s0, c0 = full(lab0)
    s1, c1 = full(lab1)
    outs = []
    for _ in range(2):
        d = ops.KMeansDelta(pts, K)
        d.load(s0.float(), c0.float())
        n = d.move(lab0, lab1)
        assert int(n) == int((lab0 != lab1).sum()) == 1500
        sums, counts = d.sums_counts()
        outs.append((sums, counts, d.sums_fix.clone()))
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][0], outs[1][0])
    sums, counts, _ = outs[0]
    assert torch.equal(counts.double(), c1)
    # the loaded start is s0 rounded to fp32; the MOVE itself is exact: compare the change of the sums
    assert float(((sums.double() - s0.float().double()) - (s1 - s0)).abs().max()) < 1e-4 * float(s1.abs().max()) * 1e-3 + 1e-5


def test_lloyd_fused_tail_equals_the_elementwise_form(dev):
    """ops.KMeansDelta.step (gfs3d_kmeans_fix_step, one launch): sums / counts bit for bit those of sums_counts(), the new centres bit
    for bit `where(counts > 0, sums / counts, old)` - an empty cluster keeps its centre -, and the three scalars: squared centre
    shift (1e-6 relative of the tensor-library sum), labels changed, empty clusters."""
    from gfs3d import ops
    M, C, K = 30000, 192, 200
    g = torch.Generator().manual_seed(9)
    pts = (torch.randn(M, C, generator=g) * 0.7 + 0.3).to(dev)
    lab0 = torch.randint(0, K - 3, (M,), generator=g, dtype=torch.int32).to(dev)          # clusters K-3 .. K-1 stay empty
    lab1 = lab0.clone()
    moved = torch.randperm(M, generator=g)[:700].to(dev)
    lab1[moved] = ((lab0[moved].long() + 1) % (K - 3)).int()
    s0 = torch.zeros(K, C, dtype=torch.float64, device=dev).index_add_(0, lab0.long(), pts.double())
    c0 = torch.bincount(lab0.long(), minlength=K).double()
    old = torch.randn(K, C, generator=g).to(dev)
    d = ops.KMeansDelta(pts, K)
    d.load(s0.float(), c0.float())
    d.move(lab0, lab1)
    sums_ref, counts_ref = d.sums_counts()
    sums, counts, new, state = d.step(old)
    assert torch.equal(sums, sums_ref) and torch.equal(counts, counts_ref)
    new_ref = torch.where(counts_ref[:, None] > 0, sums_ref / counts_ref.clamp_min(1)[:, None], old)
    assert torch.equal(new, new_ref) and torch.equal(new[K - 3:], old[K - 3:])
    shift, changed, empty = state.tolist()
    ref_shift = float(((new_ref.double() - old.double()) ** 2).sum())
    assert abs(shift - ref_shift) <= 1e-6 * ref_shift
    assert changed == float((lab0 != lab1).sum()) == 700.0 and empty == 3.0


def test_device_kmeans_with_delta_updates_equals_full_recomputation(dev, monkeypatch):
    """The Lloyd loop with the sums following the labels (moved share <= 3 %) ends where the loop that rebuilds the sums from all
    points every iteration ends: same iteration count, labels, inertia."""
    import get_basis
    g = load_golden("kmeanspp")
    X, K, s = _kpp_case(g, "c")
    pts = torch.from_numpy(X).to(dev)
    init = pts[torch.from_numpy(g["c_fit_indices"]).to(dev)]
    a = get_basis.device_kmeans(pts, K, init=init)
    monkeypatch.setattr(get_basis, "_DELTA_SHARE", -1.0)            # never: full passes only
    b = get_basis.device_kmeans(pts, K, init=init)
    assert a[3] == b[3] and abs(a[2] - b[2]) < 1e-6 * b[2]
    assert float((a[1] == b[1]).float().mean()) > 0.9999


def test_lloyd_run_ahead_equals_the_blocking_loop(dev, monkeypatch):
    """get_basis._lloyd_run_ahead (iteration n + 1 queued before the scalars of iteration n are read) ends exactly where the loop that
    reads them every iteration ends: same iteration count, bit-identical centres, same labels and inertia."""
    import get_basis
    g = load_golden("kmeanspp")
    for case in ("c", "f"):
        X, K, s = _kpp_case(g, case)
        pts = torch.from_numpy(X).to(dev)
        init = pts[torch.from_numpy(g[case + "_fit_indices"]).to(dev)]
        monkeypatch.setattr(get_basis, "_RUN_AHEAD", True)
        a = get_basis.device_kmeans(pts, K, init=init)
        monkeypatch.setattr(get_basis, "_RUN_AHEAD", False)
        b = get_basis.device_kmeans(pts, K, init=init)
        assert a[3] == b[3] and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]


def test_lloyd_run_ahead_undoes_a_queued_iteration(dev, monkeypatch):
    """When the iteration being judged relocates empty clusters, the iteration queued behind it started from centres that no longer
    hold: its move of the running sums is undone (integer sums: exactly) and it is queued again.  Forced here: the fused tail reports
    an empty cluster at its third call, the relocation is the identity - the run must end exactly like the undisturbed one."""
    import get_basis
    from gfs3d import ops
    g = load_golden("kmeanspp")
    X, K, s = _kpp_case(g, "c")
    pts = torch.from_numpy(X).to(dev)
    init = pts[torch.from_numpy(g["c_fit_indices"]).to(dev)]
    ref = get_basis.device_kmeans(pts, K, init=init)
    calls, undone = [0], [0]
    step0, move0 = ops.KMeansDelta.step, ops.KMeansDelta.move

    def step(self, cent_old):
        sums, counts, new, state = step0(self, cent_old)
        calls[0] += 1
        if calls[0] == 3:
            state = state.clone()
            state[2] = 1.0
        return sums, counts, new, state

    def move(self, a, b):
        out = move0(self, a, b)
        undone[0] += 1
        return out
    monkeypatch.setattr(ops.KMeansDelta, "step", step)
    monkeypatch.setattr(ops.KMeansDelta, "move", move)
    monkeypatch.setattr(get_basis, "_relocate_empty_clusters", lambda p, c, l, sums, counts, sh: (sums, counts))
    got = get_basis.device_kmeans(pts, K, init=init)
    assert calls[0] >= 4                                            # the loop went on after the forced relocation ...
    assert undone[0] == calls[0] + 1                                # ... with one extra move: the undo of the queued iteration
    assert got[3] == ref[3] and torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and got[2] == ref[2]


def test_kmeans_full_size_vs_fp64(dev):
    """configs[2] at its REAL size (7 x 300 000 points x 192, 200 centroids): the screened Lloyd assignment and the
    exact final assignment against a float64 evaluation of EVERY point (chunked on the device): labels identical wherever the two
    best centroids are separated by more than fp32 rounding, counts exact, sums and inertia consistent with the kernel's own labels."""
    from gfs3d import ops
    M, C, K = 7 * 300000, 192, 200
    g = torch.Generator().manual_seed(11)
    cent = (torch.randn(K, C, generator=g) * 0.6).to(dev)
    pts = torch.empty(M, C, device=dev)
    for s in range(0, M, 300000):                                  # clustered points, built on the device chunk by chunk
        n = min(300000, M - s)
        which = torch.randint(0, K, (n,), generator=g).to(dev)
        pts[s:s + n] = cent[which] * 0.9 + 0.45 * torch.randn(n, C, generator=g).to(dev)
    ref_lab = torch.empty(M, dtype=torch.int64, device=dev)
    clear = torch.empty(M, dtype=torch.bool, device=dev)
    ref_best = torch.empty(M, dtype=torch.float64, device=dev)
    c64 = cent.double()
    for s in range(0, M, 131072):
        d = torch.cdist(pts[s:s + 131072].double(), c64) ** 2
        top = torch.topk(d, 2, dim=1, largest=False)
        ref_lab[s:s + 131072] = top.indices[:, 0]
        ref_best[s:s + 131072] = top.values[:, 0]
        clear[s:s + 131072] = (top.values[:, 1] - top.values[:, 0]) > 1e-4 * top.values[:, 1]
    for fast in (True, False):
        label, sums, counts, inertia = ops.kmeans_assign(pts, cent, fast=fast)
        lab = label.long()
        assert bool((lab[clear] == ref_lab[clear]).all())
        assert float((lab == ref_lab).float().mean()) > 0.9999
        assert torch.equal(counts.double(), torch.bincount(lab, minlength=K).double()) and float(counts.sum()) == M
        exp_sums = torch.zeros(K, C, dtype=torch.float64, device=dev).index_add_(0, lab, pts.double())
        assert float((sums.double() - exp_sums).abs().max()) < 1e-4 * float(exp_sums.abs().max())
        if not fast:                                               # the exact form's inertia is the reported one
            own = ((pts.double() - c64[lab]) ** 2).sum(1)
            assert abs(float(inertia) - float(own.sum())) < 1e-5 * float(own.sum())
            assert abs(float(inertia) - float(ref_best.sum())) < 1e-5 * float(ref_best.sum())


def test_subsample_per_class_skips_labels_outside_the_class_range(dev):
    """an ignore index (-100), labels >= num_classes and float labels match no `y == c` mask of the reference
    (get_basis.py:176-181): they are skipped, not an error."""
    from get_basis import subsample_per_class
    y = torch.tensor([[1, -100, 2, 7, 1, 0, 2, 2]], dtype=torch.int64)
    feat = torch.arange(8, dtype=torch.float32)[:, None].repeat(1, 4).to(dev)
    out = subsample_per_class(feat, y, 3, max_per_class=10).cpu()
    assert out[:, 0].tolist() == [0.0, 4.0, 2.0, 6.0, 7.0]                      # class 1 rows, then class 2 rows, point order kept
    out2 = subsample_per_class(feat, y.float(), 3, max_per_class=10).cpu()
    assert torch.equal(out, out2)


def test_greedy_kmeanspp_at_3e5_points_agrees_with_sklearn_up_to_the_recorded_ties(dev):
    """tests/golden/kmeanspp_large.npz (300 000 x 64 points, K = 24, four RandomStates).  At this size sklearn's float32
    potential / cdf sit ~1e-7 of the potential away from an exact evaluation - the distance of the closest variate to a cdf step
    is of the same order - so seed-for-seed equality is NOT the contract: the device's sequence (fp64 block sums, fp32 d2) must
    equal sklearn's up to the first pick whose recorded margin is below 1e-6, and may differ afterwards.  The float64
    restatement agrees with sklearn on three of the four seeds (and parts at pick 5, margin 1.05e-7, on the fourth); the
    device is expected to do about as well: at least two whole sequences."""
    from get_basis import kmeans_plus_plus
    from conftest import load_golden as lg
    g = lg("kmeanspp_large")
    M, C, K, blobs, data_seed = (int(v) for v in g["params"])
    pts = torch.from_numpy(R.clustered_points(M, C, blobs, seed=data_seed)).to(dev)
    whole = 0
    for s in range(4):
        ref, marg, valid = g[f"s{s}_indices"], g[f"s{s}_margins"], int(g[f"s{s}_first_diff"])
        _, picks = kmeans_plus_plus(pts, K, np.random.RandomState(s), return_picks=True)
        diff = np.nonzero(picks.cpu().numpy() != ref)[0]
        first = int(diff[0]) if diff.size else K
        whole += first == K
        print(f"seed {s}: device == sklearn up to pick {first} of {K} (float64 restatement: {valid})")
        if first < min(valid + 1, K):           # the margins are the restatement's: meaningful up to ITS parting pick
            assert float(marg[first].min()) < 1e-6, f"seed {s}: parts at pick {first} with margin {marg[first]}"
    assert whole >= 2


def test_extract_geometric_words_on_one_stream_equals_the_reference_flow(dev):
    """The whole configs[2] flow against the reference's OWN function: tests/golden/basis_flow.npz holds what
    `Get_GlobalProto_GlobalKmeans` (get_basis.py:112-222) produced after one `np.random.seed(s)` (subsample draw of the
    over-full class, sklearn's k-means++ picks on the centred rows, n_iter, inertia, cluster sizes, Kmean2Proto means, SVD
    words; MAX_NUM patched to 30 000 in the code object).  `extract_geometric_words(rng=RandomState(s), sampler="numpy")`
    consumes ONE stream the same way: (i) the subsample rows are identical; (ii) the seeding picks equal sklearn's, with the
    documented allowance (the device's features differ from the CPU reference's by ~1e-6, its d2 / potentials round differently:
    the sequences may part at a pick whose recorded margin is below 1e-6 - smallest recorded: 4.3e-7); (iii) where the picks are
    equal the loop stops within 2 iterations of sklearn's, at its inertia (1e-4), cluster sizes (+-0.2 % of the rows) and the
    words agree to 1e-3."""
    from get_basis import DGCNNSeg, extract_geometric_words
    from conftest import state_dict_from
    g = load_golden("basis_flow")
    NB, N, K, bseed, lseed, s, max_num = (int(v) for v in g["params"])
    blocks, labels = R.basis_flow_inputs(NB, N, bseed, lseed)
    a = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20)
    seg = DGCNNSeg(a, num_classes=8)
    seg.load_state_dict(state_dict_from(load_golden("dgcnnseg")))
    seg = seg.to(dev).eval()
    d = {}
    words = extract_geometric_words(seg, blocks.to(dev), labels.to(dev), 8, K, max_per_class=max_num, batch=8, rng=np.random.RandomState(s),
                                    sampler="numpy", details=d)
    assert list(d["sub_idx"]) == [1] and np.array_equal(d["sub_idx"][1], g["sub_idx"])          # (i) the reference's own draw
    assert d["rows"] == int(g["rows"])
    n = _equal_up_to_a_rounding_tie(d["picks"].cpu().numpy(), g["picks"], g["margins"], 1e-6)    # (ii)
    print(f"basis flow: {n} of {K} picks equal sklearn's; n_iter {d['n_iter']} (sklearn {int(g['n_iter'])}), inertia {d['inertia']:.4f} "
          f"(sklearn {float(g['inertia']):.4f})")
    ref_inertia = float(g["inertia"])
    if n == K:                                                                                   # (iii)
        assert abs(d["n_iter"] - int(g["n_iter"])) <= 2
        assert abs(d["inertia"] - ref_inertia) < 1e-4 * ref_inertia
        sizes = np.sort(d["counts"].cpu().numpy().astype(np.int64))
        assert np.abs(sizes - g["sizes"]).max() <= 0.002 * int(g["rows"])
        assert np.abs(words - g["words"]).max() < 1e-3
    else:               # a legitimate tie: another, equally valid k-means++ start - the fit must still be of the same quality
        assert abs(d["inertia"] - ref_inertia) < 0.02 * ref_inertia
