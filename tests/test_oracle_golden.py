"""Pins the oracle (oracle/ref_ops.py, oracle/knn_oracle.c) against golden
vectors produced by the real reference (oracle/gen_golden.py).
CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, state_dict_from
from oracle import c_oracle, ref_ops as R

TOL = 1e-4   # north_star: features/logits/distances within 1e-4 fp32


def close(a, b, tol=TOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= tol, f"max abs err {err}"


def knn_rows_ok(x, idx_a, idx_ref, k, gap_rel=1e-6):
    """SURVEY 8(c) parity rule: per-row neighbour *sets* equal wherever the fp64
    gap between rank k and rank k+1 exceeds gap_rel * scale."""
    x64 = x.astype(np.float64)
    bad = 0
    for b in range(x.shape[0]):
        d = -((x64[b][:, :, None] - x64[b][:, None, :]) ** 2).sum(0)
        srt = np.sort(d, axis=1)[:, ::-1]
        gap = srt[:, k - 1] - srt[:, k]
        scale = np.abs(srt[:, k - 1]) + (x64[b] ** 2).sum(0).max()
        for i in range(x.shape[2]):
            if set(idx_a[b, i]) != set(int(v) for v in idx_ref[b, i]):
                if gap[i] > gap_rel * scale[i]:
                    bad += 1
    return bad


@pytest.mark.parametrize("key", ["9", "64", "dup"])
def test_knn_c_oracle_vs_reference_sets(key):
    g = load_golden("knn")
    k = int(g["k"])
    x = g["x" + key]
    mine = c_oracle.knn(x, k)
    assert knn_rows_ok(x, mine, g["idx" + key], k) == 0
    # torch restatement is the same op sequence as the reference: identical
    assert (R.knn(torch.from_numpy(x), k).numpy() == g["idx" + key]).all()


def test_knn_c_oracle_full_block():
    g = load_golden("knn")
    k = int(g["k"])
    x = R.synthetic_blocks(1, 2048, seed=int(g["xfull_seed"])).numpy()
    mine = c_oracle.knn(x, k)
    assert knn_rows_ok(x, mine, g["idxfull"], k) == 0
    order_equal = (mine == g["idxfull"]).all(-1).mean()
    assert order_equal > 0.99          # SURVEY: order differs on 0.02-0.13 % of rows


def test_knn_c_equals_numpy_exact_order():
    x = R.synthetic_blocks(1, 300, seed=5, dup_tail=40)
    assert (c_oracle.knn(x.numpy(), 20) == R.knn_exact_order(x, 20).numpy()).all()
    x = torch.randn(1, 64, 200, generator=torch.Generator().manual_seed(3))
    assert (c_oracle.knn(x.numpy(), 7) == R.knn_exact_order(x, 7).numpy()).all()


def test_knn_ties_lowest_index_first():
    x = torch.zeros(1, 3, 16)            # all points identical: every distance ties
    assert (c_oracle.knn(x.numpy(), 5)[0] == np.arange(5)[None, :]).all()


def test_edge_feature():
    g = load_golden("edge_feature")
    x, idx = torch.from_numpy(g["x"]), torch.from_numpy(g["idx"])
    assert (R.edge_feature(x, 6, idx).numpy() == g["out"]).all()
    assert (R.edge_feature(x, 6).numpy() == g["out_auto"]).all()


def test_dgcnn_eval_train_and_grads():
    g = load_golden("dgcnn")
    sd = state_dict_from(g)
    x = torch.from_numpy(g["x"])
    ecs, out = R.dgcnn_forward(sd, x, training=False)
    for i in range(3):
        close(ecs[i], g[f"ec{i}_eval"], 1e-5)
    close(out, g["out_eval"], 1e-5)
    # train mode + grads
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    xg = x.clone().requires_grad_(True)
    stats = {}
    ecs, out = R.dgcnn_forward(p, xg, training=True, new_stats=stats)
    for i in range(3):
        close(ecs[i].detach(), g[f"ec{i}_train"], 1e-5)
    close(out.detach(), g["out_train"], 1e-5)
    loss = sum((torch.from_numpy(g[f"cot{i}"]) * t).sum() for i, t in enumerate(ecs + [out]))
    loss.backward()
    gx = g["grad_x"]
    close(xg.grad, gx, 1e-4 * np.abs(gx).max())
    for k in g.files:
        if k.startswith("grad.") and k != "grad_x":
            ref = g[k]
            close(p[k[5:]].grad, ref, 1e-4 * max(np.abs(ref).max(), 1e-6))
        if k.startswith("after.") and "running" in k:
            close(stats[k[6:]], g[k], 1e-5)


def test_capl_get_features_and_heads():
    g = load_golden("capl")
    sd = state_dict_from(g)
    gp = torch.from_numpy(g["gp"])
    x = torch.from_numpy(g["x"])
    pf, sem, oh = R.get_features(sd, gp, x)
    close(pf, g["gf_point"]); close(sem, g["gf_sem"])
    assert (oh.argmax(1).numpy() == g["gf_word"]).all()
    feat, p2, p3 = (torch.from_numpy(g[k]) for k in ("gp_feat", "gp_p2", "gp_p3"))
    bg = sd["bg_proto"]
    close(R.get_pred(feat, p2), g["gp_2d"], 1e-5); close(R.get_pred(feat, p2, bg), g["gp_2d_bg"], 1e-5)
    close(R.get_pred(feat, p3), g["gp_3d"], 1e-5); close(R.get_pred(feat, p3, bg), g["gp_3d_bg"], 1e-5)
    close(R.post_refine_proto_v2(p2, feat, feat), g["prp"], 1e-5)
    close(R.post_refine_proto_v2(p2, feat, feat, bg), g["prp_bg"], 1e-5)
    coding = torch.from_numpy(g["gw_coding"])
    w, acc, nacc = R.get_gp_weight(coding, oh, 7, 1.2, torch.from_numpy(g["gw_lab13"]))
    close(w, g["gw_test"], 0); close(acc, g["gw_test_acc"], 1e-6); close(nacc, g["gw_test_nacc"], 1e-6)
    w, acc, _ = R.get_gp_weight(coding[:7], oh, 7, 1.5, torch.from_numpy(g["gw_lab8"]), use_bg_weight=True)
    close(w, g["gw_train"], 0); close(acc, g["gw_train_acc"], 1e-6)
    w, _, _ = R.get_gp_weight(coding, oh, 7, 2.0)
    close(w, g["gw_nolabel"], 0)


def test_capl_forward_eval():
    g = load_golden("capl")
    sd = state_dict_from(g)
    coding = torch.from_numpy(g["gw_coding"])
    lg, acc, nacc = R.capl_forward_eval(sd, torch.from_numpy(g["gp"]), torch.from_numpy(g["x"]),
                                        torch.from_numpy(g["gw_lab13"]), torch.from_numpy(g["fe_gened"]).unsqueeze(0),
                                        coding[:7], coding[7:], base_num=7, eval_weight=1.2)
    close(lg, g["fe_logits"]); close(acc, g["fe_acc"], 1e-6); close(nacc, g["fe_nacc"], 1e-6)


def test_capl_generate_fake_proto_and_hard_coding():
    g = load_golden("capl")
    sd = state_dict_from(g)
    pf = torch.from_numpy(g["gf_point"]); y = torch.from_numpy(g["gfp_y"])
    fp, _ = R.generate_fake_proto(pf[2:], y[2:], sd["main_proto"].clone(), fake_novel=[2, 5])
    close(fp, g["gfp_proto"], 1e-5)
    cv = torch.from_numpy(g["hc_in"])
    assert (R.hard_coding(cv, 0.9).numpy() == g["hc_model_090"]).all()
    assert (R.hard_coding(cv, 0.95).numpy() == g["hc_train_095"]).all()


def test_capl_forward_train_loss_and_grads():
    g = load_golden("capl")
    sd = state_dict_from(g)
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    mask = np.unpackbits(g["ft_mask"])[: int(np.prod(g["ft_mask_shape"]))].reshape(g["ft_mask_shape"])
    dm = torch.from_numpy(mask.astype(np.float32)) / 0.9
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    stats = {}
    pred, loss, _ = R.capl_forward_train(p, torch.from_numpy(g["gp"]), x, torch.from_numpy(g["y"]), base_num=7,
                                         fake_novel=[int(v) for v in g["ft_fake_novel"]], new_stats=stats, dropout_mask=dm)
    assert abs(loss.item() - float(g["ft_loss"])) <= 1e-5 * abs(float(g["ft_loss"]))
    assert (pred.numpy() == g["ft_pred"]).mean() > 0.999
    loss.backward()
    gx = g["ft_grad_x"]
    close(x.grad, gx, 1e-4 * np.abs(gx).max())
    for k in g.files:
        if k.startswith("ft_grad."):
            ref = g[k]
            close(p[k[8:]].grad, ref, 1e-4 * max(np.abs(ref).max(), 1e-7))


def test_capl_scannet_shapes():
    g = load_golden("capl_scannet")
    pf, sem, oh = R.get_features(state_dict_from(g), torch.from_numpy(g["gp"]), torch.from_numpy(g["x"]))
    close(pf, g["gf_point"]); close(sem, g["gf_sem"])
    assert (oh.argmax(1).numpy() == g["gf_word"]).all()


def test_capl_scannet_heads_and_eval_forward():
    """ScanNet class counts (21 + bg, base_num 15, K_w = 180): get_pred with the bg row and the eval forward."""
    g = load_golden("capl_scannet")
    sd = state_dict_from(g)
    pf = torch.from_numpy(g["gf_point"])
    close(R.get_pred(pf, torch.from_numpy(g["gp_p2"]), sd["bg_proto"]), g["gp_2d_bg"], 1e-5)
    coding = torch.from_numpy(g["gw_coding"])
    lg, acc, nacc = R.capl_forward_eval(sd, torch.from_numpy(g["gp"]), torch.from_numpy(g["x"]),
                                        torch.from_numpy(g["gw_lab21"]), torch.from_numpy(g["fe_gened"]).unsqueeze(0),
                                        coding[:15], coding[15:], base_num=15, eval_weight=1.2)
    close(lg, g["fe_logits"]); close(acc, g["fe_acc"], 1e-6); close(nacc, g["fe_nacc"], 1e-6)


def test_capl_get_fg_feat():
    g = load_golden("capl")
    f, w = R.get_fg_feat(state_dict_from(g), torch.from_numpy(g["gp"]), torch.from_numpy(g["x"][1:2]),
                         torch.from_numpy(g["fg_mask"]))
    assert f.shape == g["fg_feat"].shape and int(g["fg_mask"].sum()) == f.shape[0]
    close(f, g["fg_feat"])
    assert (w.argmax(1).numpy() == g["fg_word"]).all() and bool((w.sum(1) == 1).all())


def test_basis_numerics():
    g = load_golden("basis")
    c, lab, inertia = R.lloyd_kmeans(g["pts"], g["init"], int(g["km_iters"]) + 2)
    close(c, g["km_centers"], 1e-4)
    assert (lab == g["km_labels"]).all()
    assert abs(inertia - float(g["km_inertia"])) <= 1e-5 * float(g["km_inertia"])
    close(R.kmean2proto(g["pts"], g["km_labels"], 12), g["proto"], 1e-6)
    close(R.svd_reconstruct(g["proto"]), g["basis"], 1e-5)
    close(R.svd_reconstruct(g["p200"]), g["basis200"], 1e-4)


def test_dgcnnseg_features():
    g = load_golden("dgcnnseg")
    close(R.dgcnnseg_features(state_dict_from(g), torch.from_numpy(g["x"])), g["feat"], 1e-5)


def test_metric():
    g = load_golden("metric")
    r = R.evaluate_metric_gfs(list(g["prs"]), list(g["gts"]), list(range(13)), list(g["novel"]), list(g["order"]))
    close(np.array(r[:4]), g["res"], 1e-12); close(r[4], g["iou"], 1e-12)
    r = R.evaluate_metric_gfs(list(g["prs_sc"]), list(g["gts_sc"]), list(range(21)), list(g["novel_sc"]),
                              list(g["order_sc"]), scannet=True)
    close(np.array(r[:4]), g["res_sc"], 1e-12); close(r[4], g["iou_sc"], 1e-12)


def test_loader_restatement_matches_reference_sample_pointcloud():
    """oracle loader_assemble / loader_sample_indices vs the real dataloaders.loader.sample_pointcloud (golden 16):
    with the recorded numpy draws the restatement reproduces ptcloud, labels and segment labels exactly; with the
    same numpy seed it also reproduces the draws."""
    g = load_golden("loader")
    classes = [int(c) for c in g["sampled_classes"]]
    for i in range(5):
        data = g["data_" + str(g[f"r{i}_scan"])]
        sc, support, rnd, use_all = (int(v) for v in g[f"r{i}_kw"])
        np.random.seed(100 + i)
        inds = R.loader_sample_indices(data[:, 6], 2048, sc, bool(rnd))
        assert np.array_equal(inds, g[f"r{i}_inds"]), i
        pc, gt, seg = R.loader_assemble(data, inds, "xyzrgbXYZ", classes, sc, bool(support), bool(use_all))
        assert np.array_equal(pc, g[f"r{i}_pc"]) and pc.dtype == g[f"r{i}_pc"].dtype, i
        assert np.array_equal(gt, g[f"r{i}_gt"]), i
        assert np.array_equal(seg, g[f"r{i}_seg"]), i


# ---------------------------------------------------------------- sklearn's KMeans(init='k-means++') (get_basis.py:210)
def _kpp_case(g, name):
    M, C, K, blobs, data_seed, rs_seed = (int(v) for v in g[name + "_params"])
    return R.clustered_points(M, C, blobs, seed=data_seed), K, rs_seed


@pytest.mark.parametrize("name", ["a", "b", "d", "e", "f"])
def test_greedy_kmeanspp_restatement_equals_sklearn_picks(name):
    """oracle.ref_ops.greedy_kmeans_plusplus == what sklearn.cluster.kmeans_plusplus returned (tests/golden/kmeanspp.npz) from the
    same RandomState; the recorded margins are the restatement's own."""
    g = load_golden("kmeanspp")
    X, K, s = _kpp_case(g, name)
    idx, marg = R.greedy_kmeans_plusplus(X, K, np.random.RandomState(s), with_margins=True)
    assert np.array_equal(idx, g[name + "_indices"])
    assert np.allclose(marg, g[name + "_margins"], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("name", ["a", "f"])
def test_lloyd_restatement_equals_sklearn_fit(name):
    """oracle.ref_ops.sklearn_lloyd from sklearn's own picks == KMeans(init='k-means++', n_init=1, random_state=s).fit: the same
    iteration count, labels, centres and inertia (sklearn works on the centred data in float32 GEMM chunks)."""
    g = load_golden("kmeanspp")
    X, K, s = _kpp_case(g, name)
    mean = X.mean(axis=0)
    Xc = X - mean
    cent, lab, inertia, n_iter = R.sklearn_lloyd(Xc, Xc[g[name + "_fit_indices"]])
    assert n_iter == int(g[name + "_fit_n_iter"])
    assert np.array_equal(lab, g[name + "_fit_labels"].astype(np.int64))
    assert abs(inertia - float(g[name + "_fit_inertia"])) < 1e-6 * inertia
    close(cent + mean, g[name + "_fit_centers"], 1e-5)
    c1, _, _, _ = R.sklearn_lloyd(Xc, Xc[g[name + "_fit_indices"]], max_iter=1)
    close(c1 + mean, g[name + "_fit1_centers"], 1e-5)


def test_lloyd_restatement_relocates_empty_clusters_like_sklearn():
    """Three start centres far from every point: sklearn moves each empty cluster onto one of the points farthest from
    their centres (fixture case g); the restatement takes the same points and ends in the same clustering."""
    g = load_golden("kmeanspp")
    X, K, _ = _kpp_case(g, "g")
    mean = X.mean(axis=0)
    cent, lab, inertia, n_iter = R.sklearn_lloyd(X - mean, g["g_init"] - mean)
    assert n_iter == int(g["g_fit_n_iter"])
    assert np.array_equal(lab, g["g_fit_labels"].astype(np.int64))
    assert abs(inertia - float(g["g_fit_inertia"])) < 1e-6 * inertia
    close(cent + mean, g["g_fit_centers"], 1e-5)


def test_kmeanspp_restatement_at_3e5_points_parts_where_recorded():
    """tests/golden/kmeanspp_large.npz: sklearn.cluster.kmeans_plusplus on 300 000 x 64 points (K = 24, four RandomStates).  At this
    size sklearn's float32 potential / cdf deviate from an exact evaluation by ~1e-7 of the potential, which is the order of the
    distance of the closest variate to a cdf step: the float64 restatement reproduces sklearn's sequence entirely on three of the
    four seeds and parts from it on one, at a pick whose margin is 1.05e-7 (no seed-for-seed parity at configs[2]
    size - the same algorithm on the same variates).  Two seeds are re-run here (one of each kind)."""
    g = load_golden("kmeanspp_large")
    M, C, K, blobs, data_seed = (int(v) for v in g["params"])
    X = R.clustered_points(M, C, blobs, seed=data_seed)
    firsts = [int(g[f"s{s}_first_diff"]) for s in range(4)]
    assert sum(f == K for f in firsts) == 3
    for s in (0, 1):
        idx, marg = R.greedy_kmeans_plusplus(X, K, np.random.RandomState(s), with_margins=True)
        first = firsts[s]
        assert np.array_equal(idx[:first], g[f"s{s}_indices"][:first])
        if first < K:
            assert idx[first] != g[f"s{s}_indices"][first] and marg[first].min() < 1e-6


def test_basis_flow_restatement_equals_the_reference_function():
    """tests/golden/basis_flow.npz = the reference's OWN `Get_GlobalProto_GlobalKmeans` (get_basis.py:112-222) run end to end after
    one `np.random.seed(s)` on stubbed loaders (oracle/gen_golden.py: g_basis_flow).  oracle.ref_ops.geometric_words_flow on ONE
    RandomState(s) must make the same subsample draw, the same k-means++ picks (sklearn seeds from the same stream, after the
    draw), stop after the same number of Lloyd iterations and arrive at the same words."""
    g = load_golden("basis_flow")
    NB, N, K, bseed, lseed, s, max_num = (int(v) for v in g["params"])
    blocks, labels = R.basis_flow_inputs(NB, N, bseed, lseed)
    sd = {k: v for k, v in state_dict_from(load_golden("dgcnnseg")).items()}
    out = R.geometric_words_flow(sd, blocks, labels, 8, K, max_num, np.random.RandomState(s))
    assert list(out["sub_idx"]) == [1] and np.array_equal(out["sub_idx"][1], g["sub_idx"])
    assert out["X"].shape[0] == int(g["rows"])
    assert np.allclose(out["X"].astype(np.float64).sum(0), g["row_checksum"], rtol=1e-5, atol=1e-3)
    assert np.array_equal(out["picks"], g["picks"])
    assert np.allclose(out["margins"], g["margins"], rtol=1e-3, atol=1e-9)
    assert abs(out["n_iter"] - int(g["n_iter"])) <= 1
    assert abs(out["inertia"] - float(g["inertia"])) < 1e-5 * float(g["inertia"])
    assert np.array_equal(np.sort(np.bincount(out["labels"], minlength=K)), g["sizes"]) or \
        np.abs(np.sort(np.bincount(out["labels"], minlength=K)) - g["sizes"]).max() <= 3
    close(out["proto"], g["proto"], 1e-4)
    close(out["words"], g["words"], 1e-4)
