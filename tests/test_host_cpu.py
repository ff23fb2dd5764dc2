"""CPU-side checks of the product package: module surface / state_dict layout,
host logic vs golden vectors, the C ABI loads and exports every declared symbol,
and the hot path refuses CPU tensors (no fallback)."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, state_dict_from


def _args():
    return types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20,
                                 base_widths=[128, 64], output_dim=64, eval_weight=1.2)


def test_library_exports_every_declared_symbol():
    from gfs3d import _lib, build_library
    so = build_library()
    L = ctypes.CDLL(so)
    header = open(os.path.join(ROOT, "include", "gfs3d.h")).read()
    declared = set(re.findall(r"\b(gfs3d_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for name in declared:
        assert hasattr(L, name), name
    L.gfs3d_version.restype = ctypes.c_int
    assert L.gfs3d_version() >= 100


def _header_fields(header, name):
    body = header.split("typedef struct %s {" % name)[1].split("} %s;" % name)[0]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            tok = re.sub(r"[\s\*]", " ", part).split()[-1]
            names.append(re.sub(r"\[.*\]", "", tok))
    return names


def test_args_structs_match_header_field_order_and_size():
    """The ctypes mirrors follow include/gfs3d.h field by field (nested struct `f` first in the backward args), and
    a C compiler agrees on their size."""
    import ctypes, subprocess, tempfile
    from gfs3d import _lib
    header = open(os.path.join(ROOT, "include", "gfs3d.h")).read()
    pairs = [("gfs3d_edgeconv_args", _lib.EdgeConvArgs), ("gfs3d_edgeconv_bwd_args", _lib.EdgeConvBwdArgs),
             ("gfs3d_pointwise_args", _lib.PointwiseArgs), ("gfs3d_pointwise_bwd_args", _lib.PointwiseBwdArgs),
             ("gfs3d_pack_args", _lib.PackArgs), ("gfs3d_capl_head_args", _lib.CaplHeadArgs),
             ("gfs3d_capl_eval_args", _lib.CaplEvalArgs)]
    for cname, cls in pairs:
        assert _header_fields(header, cname) == [f[0] for f in cls._fields_], cname
    src = '#include <stdio.h>\n#include "gfs3d.h"\nint main(void){printf("' + "%zu " * len(pairs) + '\\n",' + ",".join(
        "sizeof(%s)" % c for c, _ in pairs) + ");return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).split()]
    assert sizes == [ctypes.sizeof(cls) for _, cls in pairs]


def test_entry_points_reject_bad_arguments_before_touching_a_device():
    """Argument validation of the entry points happens on the host: wrong shapes / NULL pointers come back as an error code
    with a message (no GPU needed, nothing is launched), and the workspace queries return 0 for unsupported shapes.  The
    calls live in tests/host_cases.py: test_host_sanitizer.py replays them against an AddressSanitizer / UBSan build of the
    launch code."""
    import host_cases
    from gfs3d import _lib, ops
    L = _lib.lib()
    host_cases.round2_cases(L, _lib)
    host_cases.round3_cases(L, _lib)
    host_cases.round5_cases(L, _lib)
    host_cases.round6_cases(L, _lib)
    host_cases.workspace_sweep(L)
    assert not ops.capl_head_supported(torch.zeros(2, 64, 16), 13, 2)       # (CPU tensor / wrong width: the model keeps its tensor-library head)


def test_dgcnn_state_dict_and_seeded_init_match_reference():
    from model.dgcnn import DGCNN
    g = load_golden("dgcnn")
    torch.manual_seed(0)
    net = DGCNN([[64, 64]] * 3, [512, 256], 9, k=20, return_edgeconvs=True)
    sd = net.state_dict()
    assert sorted(sd) == sorted(k[3:] for k in g.files if k.startswith("sd."))
    assert np.array_equal(sd["edge_convs.0.layer.0.weight"].numpy(), g["init.edge_convs.0.layer.0.weight"])
    net.load_state_dict(state_dict_from(g))        # strict


def test_capl_state_dict_and_seeded_init_match_reference():
    from model.capl import mpti_net_Point_GeoAsWeight_v2
    g = load_golden("capl")
    torch.manual_seed(0)
    m = mpti_net_Point_GeoAsWeight_v2(classes=13, args=_args(), base_num=7, gp=torch.from_numpy(g["gp"]), energy=0.9)
    assert sorted(m.state_dict()) == sorted(str(k) for k in g["keys"])
    assert np.array_equal(m.main_proto.detach().numpy(), g["init_main_proto"])
    m.load_state_dict(state_dict_from(g))
    for attr in ("encoder", "base_learner", "att_learner", "main_proto", "bg_proto", "fusion", "gp"):
        assert hasattr(m, attr)


def test_hot_path_refuses_cpu_tensors():
    from gfs3d import ops
    from model.dgcnn import DGCNN, knn
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        knn(torch.randn(1, 3, 32), 4)
    net = DGCNN([[64, 64]] * 3, [512, 256], 9, k=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.randn(1, 9, 32))
    with pytest.raises(NotImplementedError):
        DGCNN([[32, 32]], [64], 9, k=4).edge_convs[0].fused_parts()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.proto_cosine_raw(torch.randn(1, 8, 16), torch.randn(3, 8))


def test_hard_coding_matches_reference():
    from model.capl import hard_coding_mask
    g = load_golden("capl")
    cv = torch.from_numpy(g["hc_in"])
    assert (hard_coding_mask(cv, 0.9).numpy() == g["hc_model_090"]).all()
    assert (hard_coding_mask(cv, 0.95).numpy() == g["hc_train_095"]).all()


def test_get_gp_weight_matches_reference():
    from model.capl import mpti_net_Point_GeoAsWeight_v2
    import torch.nn.functional as F
    g = load_golden("capl")
    m = mpti_net_Point_GeoAsWeight_v2(classes=13, args=_args(), base_num=7, gp=torch.from_numpy(g["gp"]), energy=0.9)
    oh = F.one_hot(torch.from_numpy(g["gf_word"].astype(np.int64)), 200).transpose(2, 1).float()
    coding = torch.from_numpy(g["gw_coding"])
    w, acc, nacc = m.get_gp_weight(coding, oh, gt_label=torch.from_numpy(g["gw_lab13"]), th=1.2)
    assert np.array_equal(w.numpy(), g["gw_test"])
    assert abs(float(acc) - float(g["gw_test_acc"])) < 1e-6 and abs(float(nacc) - float(g["gw_test_nacc"])) < 1e-6
    w, acc, _ = m.get_gp_weight(coding[:7], oh, use_bg_weight=True, gt_label=torch.from_numpy(g["gw_lab8"]), th=1.5)
    assert np.array_equal(w.numpy(), g["gw_train"]) and abs(float(acc) - float(g["gw_train_acc"])) < 1e-6
    w, _, _ = m.get_gp_weight(coding, oh, th=2.0)
    assert np.array_equal(w.numpy(), g["gw_nolabel"])


def test_generate_fake_proto_matches_reference():
    from model.capl import mpti_net_Point_GeoAsWeight_v2
    g = load_golden("capl")
    m = mpti_net_Point_GeoAsWeight_v2(classes=13, args=_args(), base_num=7, gp=torch.from_numpy(g["gp"]), energy=0.9)
    m.load_state_dict(state_dict_from(g))
    pf, y = torch.from_numpy(g["gf_point"]), torch.from_numpy(g["gfp_y"])
    fp, _ = m.generate_fake_proto(pf[2:], y[2:], m.main_proto.detach().clone(), fake_novel=[2, 5])
    assert np.abs(fp.numpy() - g["gfp_proto"]).max() < 1e-5
    import random
    random.seed(0)
    _, chosen = m.generate_fake_proto(pf[2:], y[2:], m.main_proto.detach().clone())
    assert sorted(chosen) == sorted(int(v) for v in g["ft_fake_novel"])


class _Log:
    def cprint(self, *a):
        pass


def test_metric_matches_reference():
    from runs.eval import evaluate_metric_GFS
    g = load_golden("metric")
    r = evaluate_metric_GFS(_Log(), [torch.from_numpy(p) for p in g["prs"]], list(g["gts"]), list(range(13)),
                            list(g["novel"]), list(g["order"]))
    assert np.allclose(r[:4], g["res"], rtol=0, atol=1e-12) and np.allclose(r[4], g["iou"], atol=1e-12)
    r = evaluate_metric_GFS(_Log(), list(g["prs_sc"]), list(g["gts_sc"]), list(range(21)), list(g["novel_sc"]),
                            list(g["order_sc"]), scannet=True)
    assert np.allclose(r[:4], g["res_sc"], atol=1e-12) and np.allclose(r[4], g["iou_sc"], atol=1e-12)


def test_basis_host_numerics_match_reference():
    from get_basis import Kmean2Proto, compute_svd
    g = load_golden("basis")
    assert np.abs(Kmean2Proto(g["pts"], g["km_labels"], 12) - g["proto"]).max() < 1e-6
    assert np.abs(compute_svd(g["proto"]) - g["basis"]).max() < 1e-5
    assert np.abs(compute_svd(g["p200"]) - g["basis200"]).max() < 1e-4


def test_bench_roofline_block_reports_both_fractions():
    """bench.py's roofline block: role flops / time against the fp32 peak AND the pipe the kernel executes on, per launch for
    the single-kernel ids, per step for the ids that bracket whole passes (several launches per step)."""
    import bench
    cfg = dict(bench.WORKLOADS["c2"])
    B = cfg["B"]
    # 3 steps: the heavy backward 9 launches of 0.2 ms, the point-wise backward one 0.6-ms pass per step
    prof = {2: (9 * 0.2, 9), 7: (3 * 0.6, 3), 0: (3 * 0.4, 9)}
    r = bench.roofline_of(prof, 3, B, cfg, tag="r04")
    assert r["kernel"].startswith("pointwise_stack_bwd") or r["kernel"].startswith("ec_main_bwd_pair_kernel")
    pk = r["per_kernel"]
    conv2 = 32 * 2048 * 20 * 2 * (2 * 64 * 64)                               # one layer's dgrad + wgrad
    want = 3 * conv2 / (0.6e-3) / 1e12                                        # 3 launches in 0.6 ms per step
    assert abs(pk["ec_main_bwd_pair_kernel"]["algorithmic_TFLOPps"] - round(want, 2)) < 0.02
    assert abs(pk["ec_main_bwd_pair_kernel"]["frac_fp32"] - want / 157.3) < 1e-3
    assert abs(pk["ec_main_bwd_pair_kernel"]["frac_executed_pipe"] - want / (2500.0 / 6)) < 1e-3
    assert pk["knn_topk_kernel"]["frac_executed_pipe"] is None                # the bit-exact fp32 chain: priced at the fp32 peak only
    c5 = dict(bench.WORKLOADS["c5"])
    r5 = bench.roofline_of({2: (9 * 0.3, 9)}, 3, c5["B"], c5)
    assert r5["kernel"] == "ec_main_bwd_pair_kernel<1>" and r5["peak_executed_pipe"] == 2500.0
    assert abs(r5["frac_executed_pipe"] - r5["achieved"] / 2500.0) < 1e-4


# ---------------------------------------------------------------- the seeding / Lloyd host logic (get_basis.py:210)
def _kpp_case(g, name):
    from oracle import ref_ops as R
    M, C, K, blobs, data_seed, rs_seed = (int(v) for v in g[name + "_params"])
    return R.clustered_points(M, C, blobs, seed=data_seed), K, rs_seed


@pytest.mark.parametrize("name", ["a", "e"])
def test_greedy_seeding_host_form_equals_sklearn_picks(name):
    """get_basis.kmeans_plus_plus on CPU tensors (the tensor-op form of the greedy seeding the gloo logic tests use) draws
    sklearn's variates in sklearn's order: the picks are those sklearn.cluster.kmeans_plusplus returned."""
    from get_basis import kmeans_plus_plus
    g = load_golden("kmeanspp")
    X, K, s = _kpp_case(g, name)
    cent, picks = kmeans_plus_plus(torch.from_numpy(X), K, np.random.RandomState(s), return_picks=True)
    assert np.array_equal(picks.numpy(), g[name + "_indices"])
    assert np.array_equal(cent.numpy(), X[g[name + "_indices"]])


def _cpu_assign(pts, cent):
    d = torch.cdist(pts.double(), cent.double()) ** 2
    lab = d.argmin(1)
    sums = torch.zeros_like(cent).index_add_(0, lab, pts)
    counts = torch.bincount(lab, minlength=cent.shape[0]).to(pts.dtype)
    return lab, sums, counts, d.gather(1, lab[:, None]).sum().reshape(1)


def test_lloyd_host_logic_equals_sklearn_fit_incl_relocation():
    """get_basis.device_kmeans' host logic (centring, strict-convergence / shift stop, empty-cluster relocation) with a CPU
    stand-in for the assignment kernel == sklearn's fit: case a from k-means++ (same RandomState -> same picks), case g from a
    start with three empty clusters."""
    from get_basis import device_kmeans
    g = load_golden("kmeanspp")
    X, K, s = _kpp_case(g, "a")
    cent, lab, inertia, n_iter = device_kmeans(torch.from_numpy(X), K, rng=np.random.RandomState(s), assign_fn=_cpu_assign)
    assert n_iter == int(g["a_fit_n_iter"])
    assert np.array_equal(lab.numpy(), g["a_fit_labels"].astype(np.int64))
    assert abs(inertia - float(g["a_fit_inertia"])) < 1e-5 * inertia
    assert np.abs(cent.numpy() - g["a_fit_centers"]).max() < 1e-4
    X, K, _ = _kpp_case(g, "g")
    cent, lab, inertia, n_iter = device_kmeans(torch.from_numpy(X), K, init=g["g_init"], assign_fn=_cpu_assign)
    assert n_iter == int(g["g_fit_n_iter"])
    assert np.array_equal(lab.numpy(), g["g_fit_labels"].astype(np.int64))
    assert abs(inertia - float(g["g_fit_inertia"])) < 1e-5 * inertia
    assert np.abs(cent.numpy() - g["g_fit_centers"]).max() < 1e-4


# ---------------------------------------------------------------- batched block sampler (loader.py:39-151)
def _toy_blocks(sizes, n_classes=6, seed=0, cols=7):
    g = torch.Generator().manual_seed(seed)
    out = []
    for m in sizes:
        b = torch.rand(m, cols, generator=g)
        b[:, 6] = torch.randint(0, n_classes, (m,), generator=g).float()
        out.append(b)
    return out


def test_batched_sampler_has_the_reference_draw_distributions():
    """gfs3d.pipeline.BlockSampler.sample_batch_indices (one set of batch-wide tensor ops, no per-block Python) against
    the rules of sample_pointcloud (loader.py:64-82): indices inside the block; without replacement when the block has
    >= num_point points, with replacement otherwise; the class share int(n_class / M * num_point) (all class points for a
    small block) drawn from the class without replacement; every point equally likely."""
    from gfs3d import pipeline
    sizes = [500, 3000, 9000, 2048, 1500, 12000]
    blocks = _toy_blocks(sizes)
    smp = pipeline.BlockSampler(blocks, "cpu")
    ids = [5, 0, 1, 2, 3, 4, 1]
    idx = smp.sample_batch_indices(ids, 2048, None, torch.Generator().manual_seed(1))
    assert tuple(idx.shape) == (len(ids), 2048)
    for r, b in enumerate(ids):
        loc = idx[r] - smp.offsets[b]
        assert int(loc.min()) >= 0 and int(loc.max()) < sizes[b]
        if sizes[b] >= 2048:
            assert loc.unique().numel() == 2048
    for cls in (3, 0):
        idx = smp.sample_batch_indices(ids, 2048, cls, torch.Generator().manual_seed(2))
        for r, b in enumerate(ids):
            loc = idx[r] - smp.offsets[b]
            lab = blocks[b][:, 6]
            n_cls = int((lab == cls).sum())
            want = n_cls if sizes[b] < 2048 else int(n_cls / float(sizes[b]) * 2048)
            assert int(loc.min()) >= 0 and int(loc.max()) < sizes[b]
            assert int((lab[loc[:want]] == cls).sum()) == want and loc[:want].unique().numel() == want
            if sizes[b] >= 2048:
                assert loc[want:].unique().numel() == 2048 - want
    # uniformity: inclusion frequency of every point of a 3000-point block over 300 draws of 2048 (expected 0.683 +- 0.027)
    gen = torch.Generator().manual_seed(3)
    hits = torch.zeros(3000)
    for _ in range(300):
        loc = smp.sample_batch_indices([1], 2048, None, gen)[0] - smp.offsets[1]
        hits[loc] += 1
    freq = hits / 300
    assert abs(float(freq.mean()) - 2048 / 3000) < 1e-6 and float((freq - 2048 / 3000).abs().max()) < 0.14
    # same generator state -> same batch; the x / y / seg triple has the Dataset's shapes
    a = smp.batch([0, 5, 2], 2048, sampled_classes=[1, 2, 3], generator=torch.Generator().manual_seed(4))
    b = smp.batch([0, 5, 2], 2048, sampled_classes=[1, 2, 3], generator=torch.Generator().manual_seed(4))
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    assert tuple(a[0].shape) == (3, 9, 2048) and tuple(a[1].shape) == (3, 2048) and a[1].dtype == torch.int64
    assert float(a[0][:, 6:9].min()) == 0.0 and float(a[0][:, 6:9].max()) == 1.0 and int(a[1].max()) <= 3


def test_batched_augmentation_matrices_compose_like_the_reference():
    """augment_draws_batch: M = mirror_y mirror_x R_z zoom (loader.py:134-145) from four uniforms per block."""
    from gfs3d import pipeline
    from oracle import ref_ops as R
    cfg = {"scale": 1.3, "rot": 1, "mirror_prob": 0.8, "jitter": 1}
    m, jit = pipeline.augment_draws_batch(cfg, 64, 100, "cpu", torch.Generator().manual_seed(5))
    u = torch.rand((64, 4), dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    seen = set()
    for b in range(64):
        s = 1.0 / 1.3 + float(u[b, 0]) * (1.3 - 1.0 / 1.3)
        mx, my = bool(u[b, 2] < 0.4), bool(u[b, 3] < 0.4)
        seen.add((mx, my))
        ref = R.loader_augment_matrix(s, float(u[b, 1]) * 2.0 * np.pi, mx, my)
        assert np.abs(m[b].numpy() - ref).max() < 1e-12
    assert len(seen) == 4
    assert tuple(jit.shape) == (64, 100, 3) and float(jit.abs().max()) <= 0.05 and 0.005 < float(jit.std()) < 0.012
